// group.cpp — see group.hpp.
#include "group.hpp"
#include <algorithm>
#include <cstring>
#include <thread>

namespace mi355rt {

std::unique_ptr<DeviceGroup> DeviceGroup::create(const SceneData& scene, const mi355rt_config& cfg_in, std::string& err, int& code)
{
    std::unique_ptr<DeviceGroup> g(new DeviceGroup());
    mi355rt_config cfg = cfg_in;
    const uint32_t n = cfg.device_count <= 1 ? 1u : cfg.device_count;
    code = MI355RT_E_INVALID;
    if (n > 64) { err = "device_count > 64"; return nullptr; }
    if (n > 1 && cfg.stripe_world > 1) { err = "a device group deals the rows itself: stripe_world must be <= 1 when device_count > 1"; return nullptr; }
    for (uint32_t i = 0; i < n; ++i) {
        mi355rt_config c = cfg;
        c.device_count = 1;
        if (n > 1) {
            c.device = (cfg.flags & MI355RT_FLAG_GROUP_SHARES_DEVICE) ? cfg.device : cfg.device + (int32_t)i;
            if (c.stripe_rows == 0) c.stripe_rows = MI355RT_DEFAULT_STRIPE_ROWS;
            c.stripe_rank = i; c.stripe_world = n;
        }
        std::unique_ptr<Renderer> r = Renderer::create(scene, c, err, code);
        if (!r) { if (n > 1) err = "device " + std::to_string(c.device) + ": " + err; return nullptr; }
        g->devs_.push_back(std::move(r));
    }
    if (n > 1) {
        // peer access for the stripe copies (a failure is not fatal: hipMemcpyPeerAsync then stages through the host)
        for (uint32_t i = 1; i < n; ++i) {
            const int a = g->devs_[0]->cfg.device, b = g->devs_[i]->cfg.device;
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, b, a) == hipSuccess && can && hipSetDevice(b) == hipSuccess) {
                hipError_t e = hipDeviceEnablePeerAccess(a, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
        for (uint32_t i = 0; i < n; ++i)
            if (!g->devs_[i]->gather_prepare(i == 0)) { err = g->devs_[i]->last_error; code = MI355RT_E_HIP; return nullptr; }
    }
    code = MI355RT_OK;
    return g;
}

uint32_t DeviceGroup::trace_frame_additive()
{
    error_.clear();
    uint32_t ret = 0;
    // the 50-row frame is asynchronous on every device (one fused launch each): the devices run side by side
    for (size_t i = 0; i < devs_.size(); ++i) {
        ret = devs_[i]->trace_frame_additive();
        if (ret == 0) { fail_from(i); return 0; }
    }
    counts_from_render_ = false;
    return ret;
}

bool DeviceGroup::render(uint32_t spp, bool wait)
{
    error_.clear();
    if (devs_.size() == 1) { counts_from_render_ = false; return primary()->render(spp, wait); }     // (a group of several devices always waits: its members render on host threads)
    std::vector<char> ok(devs_.size(), 0);
    std::vector<std::thread> th;
    for (size_t i = 0; i < devs_.size(); ++i) th.emplace_back([&, i] { ok[i] = devs_[i]->render(spp) ? 1 : 0; });    // render() waits for its device
    for (auto& t : th) t.join();
    counts_ = mi355rt_ray_counts{};
    for (size_t i = 0; i < devs_.size(); ++i) {
        if (!ok[i]) return fail_from(i);
        const mi355rt_ray_counts& c = devs_[i]->counts;
        counts_.primary += c.primary; counts_.bounce += c.bounce; counts_.shadow += c.shadow; counts_.shadow_skipped += c.shadow_skipped; counts_.primary_hits += c.primary_hits;
        counts_.primary_culled += c.primary_culled; counts_.nodes_visited += c.nodes_visited; counts_.tris_tested += c.tris_tested;
        counts_.trace_launches += c.trace_launches; counts_.inner_execs += c.inner_execs; counts_.leaf_execs += c.leaf_execs;
        counts_.trace_ms += c.trace_ms; counts_.total_ms = std::max(counts_.total_ms, c.total_ms);
        counts_.trace_secondary_ms += c.trace_secondary_ms; counts_.trace_secondary_launches += c.trace_secondary_launches; counts_.shader_clock_mhz = std::max(counts_.shader_clock_mhz, c.shader_clock_mhz);
    }
    counts_from_render_ = true;
    return true;
}

bool DeviceGroup::last_counts(mi355rt_ray_counts& out)
{
    error_.clear();
    if (devs_.size() == 1) return primary()->last_counts(out);
    if (counts_from_render_) { out = counts_; return true; }
    out = mi355rt_ray_counts{};
    for (size_t i = 0; i < devs_.size(); ++i) {
        mi355rt_ray_counts c{};
        if (!devs_[i]->last_counts(c)) return fail_from(i);
        out.primary += c.primary; out.bounce += c.bounce; out.shadow += c.shadow; out.shadow_skipped += c.shadow_skipped; out.primary_hits += c.primary_hits;
        out.primary_culled += c.primary_culled; out.nodes_visited += c.nodes_visited; out.tris_tested += c.tris_tested;
        out.trace_launches += c.trace_launches; out.inner_execs += c.inner_execs; out.leaf_execs += c.leaf_execs;
        out.trace_ms += c.trace_ms; out.total_ms = std::max(out.total_ms, c.total_ms);
        out.trace_secondary_ms += c.trace_secondary_ms; out.trace_secondary_launches += c.trace_secondary_launches; out.shader_clock_mhz = std::max(out.shader_clock_mhz, c.shader_clock_mhz);
    }
    return true;
}

// get_tonemapped_pixels of the group: every device maps ITS rows (packed), devices 1.. copy them into their slot on
// device 0 (peer copy on their own stream, then an event), device 0 waits for the events, places all slots into the
// frame and copies it to the host.
bool DeviceGroup::get_tonemapped(uint32_t* out, size_t n)
{
    error_.clear();
    if (devs_.size() == 1) return primary()->get_tonemapped(out, n);
    Renderer* root = primary();
    for (size_t i = 1; i < devs_.size(); ++i) {
        Renderer* d = devs_[i].get();
        if (!d->tonemap_to_gather_slot()) return fail_from(i);
        const size_t bytes = d->owned_rows.size() * (size_t)d->cfg.width * 4;
        hipError_t e = bytes ? hipMemcpyPeerAsync(root->gather_slot((uint32_t)i), root->cfg.device, d->gather_slot((uint32_t)i), d->cfg.device, bytes, d->stream()) : hipSuccess;
        if (e == hipSuccess) e = hipEventRecord(d->gather_event(), d->stream());
        if (e != hipSuccess) { error_ = std::string("stripe copy to device 0: ") + hipGetErrorString(e); return false; }
    }
    if (!root->tonemap_to_gather_slot()) return fail_from(0);
    for (size_t i = 1; i < devs_.size(); ++i) {
        hipError_t e = hipStreamWaitEvent(root->stream(), devs_[i]->gather_event(), 0);
        if (e != hipSuccess) { error_ = std::string("hipStreamWaitEvent: ") + hipGetErrorString(e); return false; }
    }
    if (!root->finish_gather(out, n)) return fail_from(0);
    return true;
}

bool DeviceGroup::film_get(float* sum, float* sumsq, uint32_t* n)
{
    error_.clear();
    if (devs_.size() == 1) return primary()->film_get(sum, sumsq, n);
    const uint32_t w = primary()->cfg.width; const size_t npix = (size_t)w * primary()->cfg.height;
    std::vector<float> ts(sum ? npix * 3 : 0), tq(sumsq ? npix * 3 : 0); std::vector<uint32_t> tn(n ? npix : 0);
    for (size_t i = 0; i < devs_.size(); ++i) {
        if (!devs_[i]->film_get(sum ? ts.data() : nullptr, sumsq ? tq.data() : nullptr, n ? tn.data() : nullptr)) return fail_from(i);
        for (uint32_t r : devs_[i]->owned_rows) {
            if (sum) std::memcpy(sum + (size_t)r * w * 3, ts.data() + (size_t)r * w * 3, (size_t)w * 12);
            if (sumsq) std::memcpy(sumsq + (size_t)r * w * 3, tq.data() + (size_t)r * w * 3, (size_t)w * 12);
            if (n) std::memcpy(n + (size_t)r * w, tn.data() + (size_t)r * w, (size_t)w * 4);
        }
    }
    return true;
}

bool DeviceGroup::film_stat(bool variances, float* rgb)
{
    error_.clear();
    if (devs_.size() == 1) return primary()->film_stat(variances, rgb);
    const uint32_t w = primary()->cfg.width; const size_t npix = (size_t)w * primary()->cfg.height;
    std::vector<float> t(npix * 3);
    for (size_t i = 0; i < devs_.size(); ++i) {
        if (!devs_[i]->film_stat(variances, t.data())) return fail_from(i);
        for (uint32_t r : devs_[i]->owned_rows) std::memcpy(rgb + (size_t)r * w * 3, t.data() + (size_t)r * w * 3, (size_t)w * 12);
    }
    return true;
}

bool DeviceGroup::film_clear()
{
    error_.clear();
    for (size_t i = 0; i < devs_.size(); ++i) if (!devs_[i]->film_clear()) return fail_from(i);
    return true;
}
void DeviceGroup::camera_move_rel(float x, float y, float z) { for (auto& d : devs_) d->camera.move_rel(x, y, z); }
void DeviceGroup::camera_add_x_angle(float r) { for (auto& d : devs_) d->camera.add_x_angle(r); }
void DeviceGroup::camera_add_y_angle(float r) { for (auto& d : devs_) d->camera.add_y_angle(r); }
bool DeviceGroup::set_seed(uint64_t seed)
{
    error_.clear();
    for (size_t i = 0; i < devs_.size(); ++i) if (!devs_[i]->set_seed(seed)) return fail_from(i);
    return true;
}
bool DeviceGroup::set_flags(uint32_t flags)
{
    error_.clear();
    for (size_t i = 0; i < devs_.size(); ++i) if (!devs_[i]->set_flags(flags)) return fail_from(i);
    return true;
}
void DeviceGroup::set_slices(uint32_t slices) { for (auto& d : devs_) { d->slices = slices; d->slices_explicit = true; } }
bool DeviceGroup::synchronize()
{
    error_.clear();
    for (size_t i = 0; i < devs_.size(); ++i) if (!devs_[i]->synchronize()) return fail_from(i);
    return true;
}

}  // namespace mi355rt
