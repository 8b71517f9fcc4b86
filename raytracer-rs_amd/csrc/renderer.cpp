// renderer.cpp — see renderer.hpp.
#include "renderer.hpp"
#include "lightmap.hpp"
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "kernels.hpp"
#include "octree.hpp"

namespace mi355rt {

namespace {
// counter RNG, identical to the device copy in kernels.hip (pcg4d, Jarzynski & Olano 2020)
void pcg4d(uint32_t v[4])
{
    for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
}
float u01(uint32_t bits) { return (float)(bits >> 9) * (1.0f / 8388608.0f); }
}  // namespace

bool Renderer::fail(hipError_t e, const char* what)
{
    last_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define HIP_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail(e__, #expr); } while (0)

bool Renderer::bind()
{
    HIP_TRY(hipSetDevice(cfg.device));
    return true;
}

template <class T> bool Renderer::upload(T*& dptr, const void* src, size_t bytes)
{
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    allocs_.push_back(p);
    alloc_bytes_ += bytes ? bytes : 16;
    if (bytes && src) HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    else if (bytes) HIP_TRY(hipMemset(p, 0, bytes));
    dptr = static_cast<T*>(p);
    return true;
}

std::unique_ptr<Renderer> Renderer::create(const SceneData& scene, const mi355rt_config& cfg, std::string& err, int& code)
{
    std::unique_ptr<Renderer> r(new Renderer());
    r->cfg = cfg;
    if (!r->init(scene, err, code)) return nullptr;
    return r;
}

// SampleGenerator::new, sample_generator.rs:15-24 + 36-52, seeded: 65 536 unit vectors by rejection in the unit ball
void Renderer::build_sample_table(std::vector<float>& table4)
{
    table.resize((size_t)kNumSamples * 3);
    table4.assign((size_t)kNumSamples * 4, 0.0f);
    for (uint32_t i = 0; i < kNumSamples; ++i) {
        for (uint32_t attempt = 0;; ++attempt) {
            uint32_t h[4] = { i, attempt, 0xFFFFFFFFu, (uint32_t)cfg.seed };
            pcg4d(h);
            Vec3 d(u01(h[0]) * 2.0f + -1.0f, u01(h[1]) * 2.0f + -1.0f, u01(h[2]) * 2.0f + -1.0f);   // random_range(-1.0..1.0)
            if (dot(d, d) < 1.0f) {
                Vec3 nrm = d.normalized();
                table[3 * (size_t)i] = nrm.x; table[3 * (size_t)i + 1] = nrm.y; table[3 * (size_t)i + 2] = nrm.z;
                table4[4 * (size_t)i] = nrm.x; table4[4 * (size_t)i + 1] = nrm.y; table4[4 * (size_t)i + 2] = nrm.z;
                break;
            }
        }
    }
}

// mi355rt_set_seed: the handle afterwards equals one created with this seed (per-sample hash key AND direction table)
bool Renderer::set_seed(uint64_t seed)
{
    if (!bind()) return false;
    if (!settle_speculation()) return false;
    cfg.seed = seed;
    std::vector<float> table4;
    build_sample_table(table4);
    for (Slice& o : slices_) if (o.stream) HIP_TRY(hipStreamSynchronize(o.stream));
    HIP_TRY(hipMemcpy(const_cast<void*>(dscene_.table), table4.data(), table4.size() * sizeof(float), hipMemcpyHostToDevice));
    return true;
}

// mi355rt_set_flags: run-time flags only; a create-time flag (the intersector) cannot be changed on a live handle
bool Renderer::set_flags(uint32_t flags)
{
    constexpr uint32_t kCreateMask = MI355RT_FLAG_OCTREE_SEMANTICS | MI355RT_FLAG_TRUE_CLOSEST_HIT | MI355RT_FLAG_GROUP_SHARES_DEVICE | MI355RT_FLAG_DEVICE_LBVH;
    if ((flags ^ cfg.flags) & kCreateMask) { last_error = "the intersector flags (OCTREE_SEMANTICS, TRUE_CLOSEST_HIT) are create-time flags: they cannot be changed with mi355rt_set_flags"; return false; }
    if (flags != cfg.flags && !settle_speculation()) return false;
    cfg.flags = flags;
    return true;
}

bool Renderer::init(const SceneData& scene, std::string& err, int& code)
{
    code = MI355RT_E_INVALID;
    if (cfg.recursions == 0 && cfg.spread == 0) { cfg.recursions = 2; cfg.spread = 1; }   // RECURSIONS / SUB_SPREAD, mod.rs:81-82
    if (cfg.spread == 0) cfg.spread = 1;
    if (cfg.triangles_per_leaf == 0) cfg.triangles_per_leaf = MI355RT_DEFAULT_TRIANGLES_PER_LEAF;
    if (cfg.width == 0 || cfg.height == 0) { err = "width and height must be non-zero"; return false; }
    if ((uint64_t)cfg.width * cfg.height > 0x7FFFFFFFull / 4) { err = "image too large"; return false; }
    if (cfg.recursions > kMaxRecursions) { err = "recursions > 3 not supported"; return false; }
    if (scene.cameras.empty()) { err = "scene has no camera"; return false; }     // scene.cameras[0] panics in the reference, lib.rs:39
    if (scene.tri_geom.size() * 9 != scene.tri_verts.size()) { err = "tri_verts / tri_geom size mismatch"; return false; }
    for (uint32_t g : scene.tri_geom) if (g >= scene.materials.size()) { err = "tri_geom entry out of range"; return false; }
    for (auto& m : scene.materials) if (m.kind == 1 && m.tex_id >= scene.textures.size()) { err = "material texture id out of range"; return false; }
    if (scene.tri_geom.size() >= (1u << 26)) { err = "too many triangles"; return false; }     // 32-bit byte offsets into the 48-byte triangle array
    {   // radiance tree: level l has prod_{j<l} spread*(recursions-j) nodes
        uint64_t count = 1, first = 0;
        for (uint32_t l = 0; l <= cfg.recursions; ++l) {
            level_first[l] = (uint32_t)first; first += count; count *= (uint64_t)cfg.spread * (cfg.recursions - l);
            if (first > 0xFFFF) { err = "radiance tree too large (spread * recursions)"; return false; }
        }
        for (uint32_t l = cfg.recursions + 1; l <= kMaxLevels; ++l) level_first[l] = (uint32_t)first;
        nodes_per_sample = (uint32_t)first;
    }
    if (cfg.stripe_world <= 1) { cfg.stripe_world = 1; cfg.stripe_rank = 0; }
    if (cfg.stripe_rows == 0) cfg.stripe_rows = MI355RT_DEFAULT_STRIPE_ROWS;
    if (cfg.stripe_rank >= cfg.stripe_world) { err = "stripe_rank >= stripe_world"; return false; }

    code = MI355RT_E_NO_DEVICE;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { err = std::string("no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU fallback"; return false; }
    if (cfg.device < 0 || cfg.device >= ndev) { err = "HIP device ordinal out of range"; return false; }
    auto bail = [&]() { err = last_error; return false; };
    if (!bind()) return bail();
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg.device) != hipSuccess) { err = "hipGetDeviceProperties failed"; return false; }
    num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
    if (hipEventCreate(&ev_begin_) != hipSuccess || hipEventCreate(&ev_end_) != hipSuccess) { err = "hipEventCreate failed"; return false; }

    ntri = scene.ntri();
    nlights_ = (uint32_t)scene.lights.size();
    camera = Camera::from_orientation_matrix(cfg.width, cfg.height, Matrix::from_array(scene.cameras[0].orientation), scene.cameras[0].fov_deg);

    // --- acceleration structure (host build, once) + per-triangle normals (calc_normal, mod.rs:198-205)
    const auto t_bvh = std::chrono::steady_clock::now();
    bool on_device = false;
    if (cfg.flags & MI355RT_FLAG_DEVICE_LBVH) {
        std::string why; double ms[2];
        on_device = build_bvh_device(scene.tri_verts.data(), scene.tri_geom.data(), ntri, bvh, why, ms);
        if (on_device) { lbvh_device_ms_ = ms[0]; }
        else last_error = "device LBVH not used (" + why + "): host SAH build instead";      // not an error: create goes on
    }
    if (!on_device) build_bvh(scene.tri_verts.data(), scene.tri_geom.data(), ntri, bvh);
    bvh_on_device_ = on_device;
    wide_ = false;
    if (kernels_walk_wide_nodes()) {
        build_wide(bvh);
        if (!bvh.nodes4.empty()) wide_ = true;
        else if (bvh.root >= 0) { err = "this build's kernels walk the 4-wide tree, which does not serve this scene"; code = MI355RT_E_INVALID; return false; }
    }
    build_ms_[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bvh).count();
    collect_cull_boxes();
    if (bvh.max_depth > kBvhMaxDepth) { err = "internal: BVH deeper than the traversal stack"; code = MI355RT_E_INVALID; return false; }
    std::vector<float> normals((size_t)std::max(ntri, 1u) * 4, 0.0f);
    for (uint32_t t = 0; t < ntri; ++t) {
        const float* v = &scene.tri_verts[9 * (size_t)t];
        Vec3 v0(v[0], v[1], v[2]), v1(v[3], v[4], v[5]), v2(v[6], v[7], v[8]);
        Vec3 n = cross(v1 - v0, v2 - v0).normalized();
        normals[4 * (size_t)t] = n.x; normals[4 * (size_t)t + 1] = n.y; normals[4 * (size_t)t + 2] = n.z;
        uint32_t g = scene.tri_geom[t];
        std::memcpy(&normals[4 * (size_t)t + 3], &g, 4);
    }
    std::vector<float> table4;
    build_sample_table(table4);
    // --- uploads
    void* d_nodes = nullptr; void* d_tris = nullptr; void* d_normals = nullptr; void* d_table = nullptr;
    DMaterial* d_mats = nullptr; DLight* d_lights = nullptr; DTexture* d_tex = nullptr; float* d_texels = nullptr;
    if (wide_ ? !upload(d_nodes, bvh.nodes4.data(), bvh.nodes4.size() * sizeof(BvhNode4)) : !upload(d_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode))) return bail();
    if (!upload(d_tris, bvh.tris.data(), bvh.tris.size() * sizeof(BvhTri))) return bail();
    if (!upload(d_normals, normals.data(), normals.size() * sizeof(float))) return bail();
    if (!upload(d_table, table4.data(), table4.size() * sizeof(float))) return bail();
    std::vector<DMaterial> mats(std::max<size_t>(scene.materials.size(), 1));
    for (size_t i = 0; i < scene.materials.size(); ++i) {
        const MaterialData& m = scene.materials[i];
        mats[i] = DMaterial{ m.rgb[0], m.rgb[1], m.rgb[2], m.kind == 1 ? (0x80000000u | m.tex_id) : 0u };
    }
    if (!upload(d_mats, mats.data(), mats.size() * sizeof(DMaterial))) return bail();
    std::vector<DLight> lights(std::max<size_t>(scene.lights.size(), 1));
    for (size_t i = 0; i < scene.lights.size(); ++i) {
        const LightData& l = scene.lights[i];
        lights[i] = DLight{ l.pos[0], l.pos[1], l.pos[2], l.color[0], l.color[1], l.color[2], 0.0f, 0.0f };
    }
    // Depth cube maps around the lights (lightmap.hpp): shadow rays they prove free are never made.  Resolution: 512 texels per face edge, halved
    // until the build stays under ~4 M texel updates (a scene of a few huge triangles) and the maps under 64 MiB; padded by ten times the BVH's
    // own padding (bvh.cpp: 2e-5 of the scene diagonal), the margin the culling mask uses.
    float* d_light_maps = nullptr; uint32_t light_map_res = 0;
    if (!scene.lights.empty() && scene.lights.size() <= 8 && ntri > 0 && !getenv("MI355RT_NO_LIGHT_MAP")) {
        const auto t_lm = std::chrono::steady_clock::now();
        double diag2 = 0.0;
        { float mn[3] = { 3e38f, 3e38f, 3e38f }, mx[3] = { -3e38f, -3e38f, -3e38f };
          for (size_t i = 0; i < scene.tri_verts.size(); ++i) { const int a = (int)(i % 3); mn[a] = std::min(mn[a], scene.tri_verts[i]); mx[a] = std::max(mx[a], scene.tri_verts[i]); }
          for (int a = 0; a < 3; ++a) diag2 += ((double)mx[a] - mn[a]) * ((double)mx[a] - mn[a]); }
        const double pad = 2e-4 * std::sqrt(diag2) + 1e-7;
        uint32_t res = 512;
        if (const char* e = getenv("MI355RT_LIGHT_MAP_RES")) { int v = atoi(e); if (v >= 16 && v <= 2048) res = (uint32_t)v; }
        while (res > 16 && (size_t)6 * res * res * 4 * scene.lights.size() > ((size_t)64 << 20)) res /= 2;
        uint64_t work_cap = 4ull << 20;
        if (const char* e = getenv("MI355RT_LIGHT_MAP_WORK")) { long long v = atoll(e); if (v >= 1024) work_cap = (uint64_t)v; }
        for (const LightData& l : scene.lights) while (res > 16 && light_map_work(scene.tri_verts.data(), ntri, l.pos, res) > work_cap) res /= 2;
        std::vector<float> all; all.reserve((size_t)6 * res * res * scene.lights.size());
        bool finite = true;
        for (size_t i = 0; i < scene.lights.size() && finite; ++i) {
            const LightData& l = scene.lights[i];
            if (!std::isfinite(l.pos[0]) || !std::isfinite(l.pos[1]) || !std::isfinite(l.pos[2])) { finite = false; break; }
            LightMap lm; build_light_map(scene.tri_verts.data(), ntri, l.pos, pad, res, lm);
            all.insert(all.end(), lm.dist2.begin(), lm.dist2.end());
            const double tail = std::max(lm.nearest - pad, 0.0);
            lights[i].tail2 = std::isfinite(tail) ? (float)std::min(tail * tail * (1.0 - 1e-6), 3.0e38) : 3.0e38f;
            if ((double)lights[i].tail2 > tail * tail) lights[i].tail2 = std::nextafterf(lights[i].tail2, 0.0f);
        }
        if (finite) {
            if (!upload(d_light_maps, all.data(), all.size() * sizeof(float))) return bail();
            light_map_res = res;
        }
        light_map_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lm).count();
        if (getenv("MI355RT_DEBUG_CULL")) fprintf(stderr, "[mi355rt] light maps: %zu lights, %u texels per face edge, %.1f ms\n", scene.lights.size(), light_map_res, light_map_ms_);
    }
    if (!upload(d_lights, lights.data(), lights.size() * sizeof(DLight))) return bail();
    std::vector<DTexture> tex(std::max<size_t>(scene.textures.size(), 1));
    std::vector<float> texels;
    for (size_t i = 0; i < scene.textures.size(); ++i) {
        const TextureData& t = scene.textures[i];
        if (t.rgb.size() != (size_t)t.width * t.height * 3 || t.rgb.empty()) { err = "texture size mismatch"; code = MI355RT_E_INVALID; return false; }
        tex[i] = DTexture{ t.width, t.height, texels.size() / 3 };
        texels.insert(texels.end(), t.rgb.begin(), t.rgb.end());
    }
    if (!upload(d_tex, tex.data(), tex.size() * sizeof(DTexture))) return bail();
    if (!upload(d_texels, texels.data(), texels.size() * sizeof(float))) return bail();
    dscene_.nodes = d_nodes; dscene_.tris = d_tris; dscene_.normals = d_normals; dscene_.materials = d_mats;
    dscene_.lights = d_lights; dscene_.textures = d_tex; dscene_.texels = d_texels; dscene_.table = d_table;
    dscene_.root = bvh.root; dscene_.nlights = nlights_; dscene_.ntri = ntri;
    dscene_.light_maps = d_light_maps; dscene_.light_map_res = light_map_res;
    dscene_.oct_nodes = nullptr; dscene_.oct_leaf_tris = nullptr; dscene_.prim_tris = nullptr; dscene_.oct_info = nullptr; dscene_.tri_home = nullptr; dscene_.oct_single_leaf = 0u;
    std::memset(dscene_.oct_root, 0, sizeof dscene_.oct_root);
    // Intersector semantics.  Default: the reference's default intersector (OctTreeIntersector), served by
    // the BVH + the octree confirm step.  MI355RT_FLAG_OCTREE_SEMANTICS: the octree walked directly (slow cross-check).
    // MI355RT_FLAG_TRUE_CLOSEST_HIT: BVH only (NoAccelerationIntersector semantics), no octree is built.
    mode_ = (cfg.flags & MI355RT_FLAG_OCTREE_SEMANTICS) ? kModeOctreeWalk : (cfg.flags & MI355RT_FLAG_TRUE_CLOSEST_HIT) ? kModeTrueClosest : kModeConfirm;
    if (mode_ != kModeTrueClosest) {
        // the reference's own structure (OctTreeIntersector::with_triangles_per_leaf, OCT:66-81)
        Octree oct;
        const auto t_oct = std::chrono::steady_clock::now();
        build_octree(scene.tri_verts.data(), ntri, cfg.triangles_per_leaf, oct);
        build_ms_[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_oct).count();
        oct_stats_[0] = (uint32_t)oct.nodes.size(); oct_stats_[1] = oct.inner; oct_stats_[2] = oct.leaves;
        oct_stats_[3] = oct.empty_leaves; oct_stats_[4] = oct.max_depth; oct_stats_[5] = (uint32_t)oct.leaf_tris.size();
        std::vector<BvhTri> prim_tris(std::max(ntri, 1u));
        std::memset(prim_tris.data(), 0, prim_tris.size() * sizeof(BvhTri));
        for (uint32_t t = 0; t < ntri; ++t) {
            const float* v = &scene.tri_verts[9 * (size_t)t];
            for (int a = 0; a < 3; ++a) { prim_tris[t].v0[a] = v[a]; prim_tris[t].e1[a] = v[3 + a] - v[a]; prim_tris[t].e2[a] = v[6 + a] - v[a]; }
            prim_tris[t].prim = t; prim_tris[t].geom = scene.tri_geom[t];
        }
        void* d_oct = nullptr; uint32_t* d_leaf = nullptr; void* d_ptris = nullptr;
        if (!upload(d_oct, oct.nodes.data(), oct.nodes.size() * sizeof(OctNodeFlat))) return bail();
        if (!upload(d_leaf, oct.leaf_tris.data(), oct.leaf_tris.size() * 4)) return bail();
        if (!upload(d_ptris, prim_tris.data(), prim_tris.size() * sizeof(BvhTri))) return bail();
        dscene_.oct_nodes = d_oct; dscene_.oct_leaf_tris = d_leaf; dscene_.prim_tris = d_ptris;
        // compact views for the confirm walk
        std::vector<int2> info(oct.nodes.size());
        std::vector<uint32_t> home(std::max(ntri, 1u), 0xFFFFFFFEu);               // 0xFFFFFFFE: in no leaf (yet)
        for (size_t i = 0; i < oct.nodes.size(); ++i) {
            const OctNodeFlat& f = oct.nodes[i];
            if (f.first_child >= 0) { info[i].x = f.first_child; info[i].y = 0; continue; }
            info[i].x = ~(int32_t)f.tri_first; info[i].y = (int32_t)f.tri_count;
            for (uint32_t k = 0; k < f.tri_count; ++k) {
                uint32_t& h = home[oct.leaf_tris[f.tri_first + k]];
                h = h == 0xFFFFFFFEu ? (uint32_t)i : 0xFFFFFFFFu;
            }
        }
        int2* d_info = nullptr; uint32_t* d_home = nullptr;
        if (!upload(d_info, info.data(), info.size() * sizeof(int2))) return bail();
        if (!upload(d_home, home.data(), home.size() * 4)) return bail();
        dscene_.oct_info = d_info; dscene_.tri_home = d_home;
        dscene_.oct_single_leaf = (mode_ == kModeConfirm && oct.nodes.size() == 1) ? 1u : 0u;
        for (int a = 0; a < 3; ++a) { dscene_.oct_root[a] = oct.nodes[0].cmin[a]; dscene_.oct_root[3 + a] = oct.nodes[0].cmax[a]; }
    }

    // --- film (film.rs:27-35) and row lists
    const size_t npix = (size_t)cfg.width * cfg.height;
    if (!upload(d_film_sum_, nullptr, npix * 12) || !upload(d_film_sumsq_, nullptr, npix * 12) || !upload(d_film_n_, nullptr, npix * 4)) return bail();
    if (!upload(d_ldr_, nullptr, npix * 4)) return bail();
    ldr_dirty_.assign(cfg.height, (uint8_t)1);
    if (hipEventCreateWithFlags(&ev_tonemap_, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_gather_, hipEventDisableTiming) != hipSuccess) { err = "hipEventCreate failed"; return false; }
    if (hipEventCreateWithFlags(&ev_call_done_, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_spec_done_, hipEventDisableTiming) != hipSuccess
        || hipStreamCreateWithFlags(&read_stream_, hipStreamNonBlocking) != hipSuccess) { err = "hipEventCreate / hipStreamCreate failed"; return false; }
    std::vector<uint32_t> all(cfg.height);
    for (uint32_t r = 0; r < cfg.height; ++r) {
        all[r] = r;
        if ((r / cfg.stripe_rows) % cfg.stripe_world == cfg.stripe_rank) owned_rows.push_back(r);
    }
    if (!upload(d_all_rows_, all.data(), all.size() * 4)) return bail();
    if (!upload(d_owned_rows_, owned_rows.data(), owned_rows.size() * 4)) return bail();
    if (!upload(d_tmp_rows_, nullptr, 64 * 4)) return bail();
    if (!upload(d_counters_, nullptr, sizeof(DCounters) * kShards)) return bail();
    if (hipHostMalloc((void**)&h_counters_, sizeof(DCounters) * kShards, hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc failed"; return false; }
    std::memset(h_counters_, 0, sizeof(DCounters) * kShards);
    slices_[0].stream = stream_;
    if (hipStreamCreateWithFlags(&trace_stream_, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
    for (uint32_t i = 0; i < kMaxSlices; ++i) {
        if (i && hipStreamCreateWithFlags(&slices_[i].stream, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
        if (hipEventCreateWithFlags(&slices_[i].done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&slices_[i].ev_ready, hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&slices_[i].ev_traced, hipEventDisableTiming) != hipSuccess) { err = "hipEventCreate failed"; return false; }
        if (!upload(slices_[i].d_ctrl, nullptr, kMaxRounds * kCtrlWordsPerRound * 4)) return bail();
        if (!upload(slices_[i].d_rows, nullptr, (size_t)cfg.height * 4)) return bail();
    }
    if (const char* e = getenv("MI355RT_SLICES")) { int v = atoi(e); if (v >= 1 && v <= (int)kMaxSlices) slices = (uint32_t)v; }
    if (!upload(d_debug_color_, nullptr, 16)) return bail();

    // queue records one sample can put into one round's output queue (shadow rays of level l + rays of level l+1)
    records_per_sample_ = 1;
    for (uint32_t l = 0; l <= cfg.recursions; ++l) {
        uint32_t nl = level_first[l + 1] - level_first[l];
        uint32_t nn = l < cfg.recursions ? level_first[l + 2] - level_first[l + 1] : 0;
        records_per_sample_ = std::max(records_per_sample_, nl * std::max(nlights_, 1u) + nn);
    }
    max_level_nodes_ = 1;
    for (uint32_t l = 0; l <= cfg.recursions; ++l) max_level_nodes_ = std::max(max_level_nodes_, level_first[l + 1] - level_first[l]);
    chunk_ = 256;
    if (const char* e = getenv("MI355RT_LEAF_THRESHOLD")) { int v = atoi(e); if (v >= 1 && v <= 64) leaf_threshold_ = (uint32_t)v; }
    if (const char* e = getenv("MI355RT_CHUNK")) { int v = atoi(e); if (v >= 64 && v <= 65536) chunk_ = (uint32_t)v; }
    code = MI355RT_OK;
    return true;
}

Renderer::~Renderer()
{
    if (hipSetDevice(cfg.device) != hipSuccess) return;
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (read_stream_) (void)hipStreamSynchronize(read_stream_);          // nothing of this handle is in flight when its memory goes
    for (uint32_t i = 0; i < kMaxSlices; ++i) {
        if (i && slices_[i].stream) { (void)hipStreamSynchronize(slices_[i].stream); (void)hipStreamDestroy(slices_[i].stream); }
        if (slices_[i].done) (void)hipEventDestroy(slices_[i].done);
        if (slices_[i].ev_ready) (void)hipEventDestroy(slices_[i].ev_ready);
        if (slices_[i].ev_traced) (void)hipEventDestroy(slices_[i].ev_traced);
    }
    if (trace_stream_) { (void)hipStreamSynchronize(trace_stream_); (void)hipStreamDestroy(trace_stream_); }
    for (void* p : allocs_) (void)hipFree(p);
    if (d_tile_ofs_) (void)hipFree(d_tile_ofs_);
    if (d_tile_entries_) (void)hipFree(d_tile_entries_);
    if (h_ldr_) (void)hipHostFree(h_ldr_);
    if (h_counters_) (void)hipHostFree(h_counters_);
    comm_destroy();
    if (d_gather_) (void)hipFree(d_gather_);
    if (ev_tonemap_) (void)hipEventDestroy(ev_tonemap_);
    if (read_stream_) (void)hipStreamDestroy(read_stream_);
    if (ev_call_done_) (void)hipEventDestroy(ev_call_done_);
    if (ev_spec_done_) (void)hipEventDestroy(ev_spec_done_);
    if (h_counters_spec_) (void)hipHostFree(h_counters_spec_);
    if (ev_gather_) (void)hipEventDestroy(ev_gather_);
    free_pass_buffers();
    for (hipEvent_t e : ev_pool_) (void)hipEventDestroy(e);
    if (ev_begin_) (void)hipEventDestroy(ev_begin_);
    if (ev_end_) (void)hipEventDestroy(ev_end_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

// The (padded, half-precision) boxes of the top BVH subtrees, at most kCullRects of them: the boxes the
// primary-chunk frustum culling tests against.  Breadth-first refinement: always split the largest box.
void Renderer::collect_cull_boxes()
{
    cull_boxes_.clear();
    if (ntri == 0) return;
    auto half_to_float = [](uint32_t h) {
        _Float16 v; uint16_t b = (uint16_t)h; std::memcpy(&v, &b, 2); return (float)v;
    };
    struct Item { std::array<float, 6> box; int32_t node; };
    std::vector<Item> items;
    Item root; root.node = bvh.root;
    for (int a = 0; a < 3; ++a) { root.box[a] = bvh.scene_min[a]; root.box[3 + a] = bvh.scene_max[a]; }
    items.push_back(root);
    for (;;) {
        int best = -1; float best_size = -1.0f;
        for (size_t i = 0; i < items.size(); ++i) {
            if (items[i].node < 0) continue;                        // a leaf cannot be refined
            const auto& b = items[i].box;
            const float size = (b[3] - b[0]) * (b[4] - b[1]) + (b[4] - b[1]) * (b[5] - b[2]) + (b[5] - b[2]) * (b[3] - b[0]);
            if (size > best_size) { best_size = size; best = (int)i; }
        }
        if (best < 0 || items.size() + 1 > kCullRects) break;
        const BvhNode& n = bvh.nodes[items[best].node];
        Item c0, c1;
        for (int a = 0; a < 3; ++a) {
            c0.box[a] = half_to_float(n.h0[a] & 0xFFFFu); c0.box[3 + a] = half_to_float(n.h0[a] >> 16);
            c1.box[a] = half_to_float(n.h1[a] & 0xFFFFu); c1.box[3 + a] = half_to_float(n.h1[a] >> 16);
        }
        c0.node = n.child0; c1.node = n.child1;
        items[best] = c0;
        items.push_back(c1);
    }
    for (const Item& it : items) {
        bool finite = true;
        for (float v : it.box) if (!std::isfinite(v)) finite = false;
        if (!finite) { cull_boxes_.clear(); return; }               // coordinates beyond the half range: no culling
        cull_boxes_.push_back(it.box);
    }
}

// Second culling stage: the coverage mask (device_types.hpp).  Every triangle's three vertices, padded like the boxes of the first stage, are
// projected into the camera's (dir_x, dir_y) plane; the cells their bounding rectangle touches are set.  Any primary ray that hits a triangle has
// its (dir_x, dir_y) inside that triangle's rectangle, so a chunk footprint that touches no set cell holds only misses.  Rebuilt (and uploaded,
// with the renderer's streams idle) only when the camera changed; scenes of more than 2 M triangles keep the first stage only.
// The screen rectangle of every triangle (BVH order) in the camera's (dir_x, dir_y) plane, with every vertex padded by `pad` in world space: what the
// coverage mask and the tile bins are made of.  A point p maps to v = (p - origin) * inv, (dir_x, dir_y) = (v.x, -v.y) / v.z; moving p by at most pad
// per axis moves v.x by at most pad * kx (kx = the absolute column sum of inv), so dir_x by at most pad * (kx + |dir_x| kz) / (v.z - pad kz): the
// rectangle of the three projected vertices, widened by that, holds the projection of the whole padded triangle.  false: something is not in front.
// Cached per camera (both consumers are rebuilt when it changes).
bool Renderer::project_triangles(const DCamera& c, const double inv[3][3], double pad, double zmin)
{
    std::vector<float> key(c.rot, c.rot + 16);
    key.insert(key.end(), c.origin, c.origin + 3);
    if (key == rects_key_ && tri_rects_.size() == bvh.tris.size()) return rects_ok_;
    rects_key_ = key; rects_ok_ = false;
    tri_rects_.resize(bvh.tris.size());
    const double kx = std::fabs(inv[0][0]) + std::fabs(inv[1][0]) + std::fabs(inv[2][0]), ky = std::fabs(inv[0][1]) + std::fabs(inv[1][1]) + std::fabs(inv[2][1]);
    const double kz = std::fabs(inv[0][2]) + std::fabs(inv[1][2]) + std::fabs(inv[2][2]);
    for (size_t i = 0; i < bvh.tris.size(); ++i) {
        const BvhTri& t = bvh.tris[i];
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (int v = 0; v < 3; ++v) {
            double w[3];
            for (int a = 0; a < 3; ++a) w[a] = (double)t.v0[a] + (v == 1 ? (double)t.e1[a] : v == 2 ? (double)t.e2[a] : 0.0) - c.origin[a];
            const double vx = w[0] * inv[0][0] + w[1] * inv[1][0] + w[2] * inv[2][0];
            const double vy = w[0] * inv[0][1] + w[1] * inv[1][1] + w[2] * inv[2][1];
            const double vz = w[0] * inv[0][2] + w[1] * inv[1][2] + w[2] * inv[2][2] - pad * kz;       // the nearest the padded vertex can be
            if (!(vz > zmin)) return false;
            const double dx = vx / (vz + pad * kz), dy = -vy / (vz + pad * kz);
            const double mx = pad * (kx + std::fabs(dx) * kz) / vz * 1.0001 + 4e-16 * (1.0 + std::fabs(dx)), my = pad * (ky + std::fabs(dy) * kz) / vz * 1.0001 + 4e-16 * (1.0 + std::fabs(dy));
            x0 = std::min(x0, dx - mx); x1 = std::max(x1, dx + mx); y0 = std::min(y0, dy - my); y1 = std::max(y1, dy + my);
        }
        tri_rects_[i] = TriRect{ x0, x1, y0, y1 };
    }
    rects_ok_ = true;
    return true;
}

bool Renderer::refresh_cull_mask(DCamera& c, const double inv[3][3], double pad, double zmin, bool build)
{
    c.cull_mask = nullptr; c.mask_x0 = c.mask_y0 = 0.0f; c.mask_inv_cx = c.mask_inv_cy = 0.0f;
    if (c.cull_valid == 0 || bvh.tris.empty() || bvh.tris.size() > (2u << 20) || getenv("MI355RT_NO_CULL_MASK")) return true;
    std::vector<float> key(c.rot, c.rot + 16);
    key.insert(key.end(), c.origin, c.origin + 3); key.push_back(c.max_x); key.push_back(c.max_y);
    if (key == mask_key_) {
        if (mask_valid_) { c.cull_mask = d_cull_mask_; c.mask_x0 = mask_dom_[0]; c.mask_y0 = mask_dom_[1]; c.mask_inv_cx = mask_dom_[2]; c.mask_inv_cy = mask_dom_[3]; }
        return true;
    }
    // A 50-row frame of the drop-in loop (0.3 ms) is not worth a rebuild (1.5 ms) when the camera has just moved — the reference's loop moves it between
    // any two calls while a key is held, main.rs:116-169: such calls cull with the first stage alone until a whole-frame pass builds the mask
    if (!build) return true;
    mask_key_ = key; mask_valid_ = false;
    // domain: the bounding rectangle of the first stage's rectangles (everything outside is empty by the first stage)
    double X0 = 1e300, X1 = -1e300, Y0 = 1e300, Y1 = -1e300;
    for (uint32_t k = 0; k < c.cull_valid; ++k) { X0 = std::min(X0, (double)c.cull_rect[k][0]); X1 = std::max(X1, (double)c.cull_rect[k][1]); Y0 = std::min(Y0, (double)c.cull_rect[k][2]); Y1 = std::max(Y1, (double)c.cull_rect[k][3]); }
    if (!(X1 > X0) || !(Y1 > Y0)) return true;
    const double icx = kCullGrid / (X1 - X0), icy = kCullGrid / (Y1 - Y0);
    constexpr uint32_t wpr = kCullGrid / 32u;
    std::vector<uint32_t> bits((size_t)kCullGrid * wpr, 0u);
    if (!project_triangles(c, inv, pad, zmin)) return true;            // cannot happen behind a valid first stage; no mask then
    for (const TriRect& tr : tri_rects_) {
        const double x0 = tr.x0, x1 = tr.x1, y0 = tr.y0, y1 = tr.y1;
        const double mx = 1e-4 * (x1 - x0) + 1e-6, my = 1e-4 * (y1 - y0) + 1e-6;
        // 1/100 of a cell of slack: the kernel computes its cell indices in f32 (error < 1e-4 cells for |dir| of a few units)
        const long i0 = std::max(0L, (long)std::floor((x0 - mx - X0) * icx - 0.01)), i1 = std::min((long)kCullGrid - 1, (long)std::floor((x1 + mx - X0) * icx + 0.01));
        const long j0 = std::max(0L, (long)std::floor((y0 - my - Y0) * icy - 0.01)), j1 = std::min((long)kCullGrid - 1, (long)std::floor((y1 + my - Y0) * icy + 0.01));
        for (long j = j0; j <= j1; ++j) for (long i = i0; i <= i1; ++i) bits[(size_t)j * wpr + (size_t)(i >> 5)] |= 1u << (i & 31);
    }
    if (!bind()) return false;
    if (!d_cull_mask_) { if (hipMalloc((void**)&d_cull_mask_, bits.size() * 4) != hipSuccess) { d_cull_mask_ = nullptr; return true; } allocs_.push_back(d_cull_mask_); }
    // kernels of earlier calls may still read the old mask: wait for them, then replace it (camera changes are rare and clear the film anyway, main.rs:116-169)
    (void)hipStreamSynchronize(stream_);
    if (trace_stream_) (void)hipStreamSynchronize(trace_stream_);
    for (Slice& sl : slices_) if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (hipMemcpy(d_cull_mask_, bits.data(), bits.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return true;
    mask_dom_[0] = (float)X0; mask_dom_[1] = (float)Y0; mask_dom_[2] = (float)icx; mask_dom_[3] = (float)icy;
    mask_valid_ = true;
    c.cull_mask = d_cull_mask_; c.mask_x0 = mask_dom_[0]; c.mask_y0 = mask_dom_[1]; c.mask_inv_cx = mask_dom_[2]; c.mask_inv_cy = mask_dom_[3];
    if (getenv("MI355RT_DEBUG_CULL")) { size_t n = 0; for (uint32_t w : bits) n += (size_t)__builtin_popcount(w); fprintf(stderr, "[mi355rt] cull mask: %zu of %u cells set, domain x [%g, %g] y [%g, %g]\n", n, kCullGrid * kCullGrid, X0, X1, Y0, Y1); }
    return true;
}

// Screen-space triangle bins for the primary rays (device_types.hpp, kernels.hip raster_kernel).  The 64 consecutive samples a wave takes are
// 64 / sample_group pixels: tile_cols columns x row_group rows of one row group (kernels.hip, pass_column).  When every group of the pass's row list
// is row_group CONSECUTIVE image rows starting at a multiple of row_group (always, for whole images and power-of-two stripes) a tile is a fixed
// block of the image, and every triangle whose padded screen rectangle (the culling mask's) meets the tile's (dir_x, dir_y) footprint — computed
// like chunk_is_culled computes a chunk's — goes on the tile's list, nearest first.  Any primary ray that hits a triangle has its direction inside
// that triangle's rectangle, so the closest hit over the tile's list IS the closest hit over the scene.  Rebuilt with the mask when the camera
// (or the layout) changes; scenes whose lists get long (tiny triangles: > 24 per tile on average) keep the BVH walk.
bool Renderer::refresh_tile_bins(DCamera& c, const double inv[3][3], double pad, double zmin, const DPass& ps, const std::vector<uint32_t>& rows)
{
#define BINS_OUT(k) do { if (getenv("MI355RT_DEBUG_CULL")) fprintf(stderr, "[mi355rt] tile bins: not built (exit %d)\n", k); return true; } while (0)
    c.tile_ofs = nullptr; c.tile_entries = nullptr; c.tile_cols = c.tile_rg = c.tile_nblocks = 0;
    if (c.cull_valid == 0 || bvh.tris.empty() || bvh.tris.size() > (2u << 20) || getenv("MI355RT_NO_RASTER")) BINS_OUT(1);
    const uint32_t W = cfg.width, H = cfg.height, rg = ps.row_group, G = ps.sample_group;
    if (ps.use_explicit || G == 0 || 64u % G || (64u / G) % rg || ps.chunk % 64u) BINS_OUT(2);
    const uint32_t cols = 64u / G / rg;
    if (W % cols || rows.size() % rg || rows.empty() || ps.npix != rows.size() * (size_t)W) BINS_OUT(3);
    for (size_t i = 0; i < rows.size(); i += rg) {                      // every group: rg consecutive image rows from a multiple of rg
        if (rows[i] % rg) BINS_OUT(4);
        for (uint32_t k = 1; k < rg; ++k) if (rows[i + k] != rows[i] + k) BINS_OUT(5);
    }
    const bool fix_row = (ps.flags & 1u) != 0;
    std::vector<float> key(c.rot, c.rot + 16);
    key.insert(key.end(), c.origin, c.origin + 3); key.push_back(c.max_x); key.push_back(c.max_y);
    key.push_back((float)cols); key.push_back((float)rg); key.push_back(fix_row ? 1.0f : 0.0f);
    const uint32_t nblocks = W / cols, ngroups = (H + rg - 1) / rg;
    auto publish = [&]() { c.tile_ofs = d_tile_ofs_; c.tile_entries = d_tile_entries_; c.tile_cols = cols; c.tile_rg = rg; c.tile_nblocks = nblocks; };
    if (key == bins_key_) { if (bins_valid_) publish(); return true; }
    bins_key_ = key; bins_valid_ = false;
    const auto t_begin = std::chrono::steady_clock::now();
    const double mx = c.max_x, my = c.max_y;
    const uint64_t vmax = fix_row ? (uint64_t)H - 1 : ((uint64_t)W * H - 1) / H;
    struct Ref { uint32_t tile; float dmin; uint32_t tri; };
    std::vector<Ref> refs;
    refs.reserve(bvh.tris.size() * 6);
    const double org[3] = { c.origin[0], c.origin[1], c.origin[2] };
    if (!project_triangles(c, inv, pad, zmin)) BINS_OUT(7);
    for (size_t k = 0; k < bvh.tris.size(); ++k) {
        const BvhTri& t = bvh.tris[k];
        float v9[9];
        for (int a = 0; a < 3; ++a) { v9[a] = t.v0[a]; v9[3 + a] = t.v0[a] + t.e1[a]; v9[6 + a] = t.v0[a] + t.e2[a]; }
        const double x0 = tri_rects_[k].x0, x1 = tri_rects_[k].x1, y0 = tri_rects_[k].y0, y1 = tri_rects_[k].y1;     // the padded screen rectangle, as in the culling mask
        const double ex = 1e-4 * (x1 - x0) + 1e-6, ey = 1e-4 * (y1 - y0) + 1e-6;
        // columns cu and row indices cv (kernels.hip, primary_sample) whose jitter range [cu, cu + 1) / W, [cv, cv + 1) / H can reach the rectangle;
        // 1/100 of a column of slack covers the f32 rounding of the kernel's own expressions
        const double fx0 = (x0 - ex + mx) / (2.0 * mx) * W - 0.01, fx1 = (x1 + ex + mx) / (2.0 * mx) * W + 0.01;
        const double fy0 = (y0 - ey + my) / (2.0 * my) * H - 0.01, fy1 = (y1 + ey + my) / (2.0 * my) * H + 0.01;
        if (fx1 < 0.0 || fx0 >= (double)W || fy1 < 0.0 || fy0 > (double)vmax + 1.0) continue;
        const uint64_t c0 = (uint64_t)std::max(0.0, std::floor(fx0)), c1 = (uint64_t)std::min((double)W - 1.0, std::floor(fx1));
        const uint64_t v0 = (uint64_t)std::max(0.0, std::floor(fy0)), v1 = (uint64_t)std::min((double)vmax, std::floor(fy1));
        if (c1 < c0 || v1 < v0) continue;
        // image rows whose pixels of columns c0..c1 have cv in [v0, v1]: cv = (row * W + x) / H, or the row itself with the fixed row index
        uint64_t r0, r1;
        if (fix_row) { r0 = v0; r1 = std::min<uint64_t>(v1, H - 1); }
        else {
            const uint64_t lo = v0 * H, hi = (v1 + 1) * H - 1;                           // row * W + x in [lo, hi] for some x in [c0, c1]
            r0 = lo > c1 ? (lo - c1 + W - 1) / W : 0; r1 = hi >= c0 ? (hi - c0) / W : 0;
            if (hi < c0) continue;
            r1 = std::min<uint64_t>(r1, H - 1);
        }
        if (r1 < r0) continue;
        const double org_d = point_triangle_distance_lower(org, v9);
        const float dmin = (float)std::max((org_d - pad) * (1.0 - 1e-6), 0.0);
        const float dmin_lo = (double)dmin > std::max((org_d - pad) * (1.0 - 1e-6), 0.0) ? std::nextafterf(dmin, 0.0f) : dmin;
        for (uint64_t g = r0 / rg; g <= r1 / rg && g < ngroups; ++g) {
            const uint64_t ra = g * rg, rb = std::min<uint64_t>(ra + rg - 1, H - 1);
            for (uint64_t b = c0 / cols; b <= c1 / cols; ++b) {
                const uint64_t xa = b * cols, xb = xa + cols - 1;
                // the tile's own cv range (every pixel of it lies between its first and its last): must meet [v0, v1]
                const uint64_t ta = fix_row ? ra : (ra * W + xa) / H, tb = fix_row ? rb : (rb * W + xb) / H;
                if (tb < v0 || ta > v1) continue;
                refs.push_back(Ref{ (uint32_t)(g * nblocks + b), dmin_lo, (uint32_t)k });
            }
        }
    }
    const size_t ntiles = (size_t)nblocks * ngroups;
    std::vector<uint2> ofs(ntiles, make_uint2(0u, 0u));
    for (const Ref& r : refs) ++ofs[r.tile].y;
    size_t live = 0; uint32_t run = 0;
    for (uint2& o : ofs) { o.x = run; run += o.y; live += o.y != 0; o.y = 0; }
    if (refs.size() > ((size_t)64 << 20) || (live && refs.size() > 24 * live)) BINS_OUT(8);     // long lists: the BVH walk is the better search
    std::vector<uint2> entries(std::max<size_t>(refs.size(), 1));
    for (const Ref& r : refs) { uint2& o = ofs[r.tile]; uint32_t bits; std::memcpy(&bits, &r.dmin, 4); entries[o.x + o.y++] = make_uint2(r.tri, bits); }
    for (const uint2& o : ofs) if (o.y > 1)
        std::sort(entries.begin() + o.x, entries.begin() + o.x + o.y, [](const uint2& a, const uint2& b) { return a.y != b.y ? a.y < b.y : a.x < b.x; });   // non-negative floats order like their bits
    if (!bind()) return false;
    // kernels of earlier calls may still read the old bins (as with the mask): wait, then replace
    (void)hipStreamSynchronize(stream_);
    if (trace_stream_) (void)hipStreamSynchronize(trace_stream_);
    for (Slice& sl : slices_) if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (ofs.size() > tile_ofs_cap_) { if (d_tile_ofs_) (void)hipFree(d_tile_ofs_); d_tile_ofs_ = nullptr; tile_ofs_cap_ = 0; if (hipMalloc((void**)&d_tile_ofs_, ofs.size() * sizeof(uint2)) != hipSuccess) { d_tile_ofs_ = nullptr; BINS_OUT(9); } tile_ofs_cap_ = ofs.size(); }
    if (entries.size() > tile_entries_cap_) { if (d_tile_entries_) (void)hipFree(d_tile_entries_); d_tile_entries_ = nullptr; tile_entries_cap_ = 0; const size_t cap = entries.size() + entries.size() / 4; if (hipMalloc((void**)&d_tile_entries_, cap * sizeof(uint2)) != hipSuccess) { d_tile_entries_ = nullptr; BINS_OUT(10); } tile_entries_cap_ = cap; }
    if (hipMemcpy(d_tile_ofs_, ofs.data(), ofs.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess) BINS_OUT(11);
    if (hipMemcpy(d_tile_entries_, entries.data(), entries.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess) BINS_OUT(12);
    bins_valid_ = true; bins_entries_ = refs.size();
    bins_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    publish();
    if (getenv("MI355RT_DEBUG_CULL")) fprintf(stderr, "[mi355rt] tile bins: %u x %u tiles of %u columns x %u rows, %zu live, %zu entries (%.1f per live tile), %.1f ms\n", nblocks, ngroups, cols, rg, live, refs.size(), live ? (double)refs.size() / live : 0.0, bins_ms_);
    return true;
}
#undef BINS_OUT

DCamera Renderer::device_camera(const DPass* layout, const std::vector<uint32_t>* rows)
{
    DCamera c;
    std::memcpy(c.rot, camera.rotation().e, sizeof c.rot);
    Vec4 pos = camera.orientation() * Vec4(0.0f, 0.0f, 0.0f, 1.0f);      // camera.rs:88
    c.origin[0] = pos.x; c.origin[1] = pos.y; c.origin[2] = pos.z;
    c.max_x = camera.max_x(); c.max_y = camera.max_y();
    c.width = cfg.width; c.height = cfg.height;
    // Screen-space bounds of the padded scene box for the primary-chunk frustum culling (kernels.hip,
    // chunk_is_culled).  dir = (dir_x, -dir_y, 1) * R3x3  =>  (dir_x, -dir_y, 1) ~ (p - origin) * R3x3^-1.
    // Valid only if every corner of the box is in front of the camera; computed in double, widened.
    c.cull_valid = 0;
    std::memset(c.cull_rect, 0, sizeof c.cull_rect);
    c.cull_mask = nullptr; c.mask_x0 = c.mask_y0 = 0.0f; c.mask_inv_cx = c.mask_inv_cy = 0.0f;
    c.tile_ofs = nullptr; c.tile_entries = nullptr; c.tile_cols = c.tile_rg = c.tile_nblocks = 0;
    if (!cull_boxes_.empty() && mode_ != kModeOctreeWalk && !getenv("MI355RT_NO_CULL")) {
        const float* e = c.rot;
        const double m[3][3] = { { e[0], e[1], e[2] }, { e[4], e[5], e[6] }, { e[8], e[9], e[10] } };
        const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                         + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        if (std::fabs(det) > 1e-12) {
            double inv[3][3];
            inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det; inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
            inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det; inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
            inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det; inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
            double diag = 0.0;
            for (int a = 0; a < 3; ++a) diag += (double)(bvh.scene_max[a] - bvh.scene_min[a]) * (bvh.scene_max[a] - bvh.scene_min[a]);
            const double pad = 1e-3 * std::sqrt(diag) + 1e-6;            // on top of the (already padded) BVH boxes
            bool front = true;
            uint32_t n = 0;
            for (const auto& bx : cull_boxes_) {
                double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
                for (int k = 0; k < 8 && front; ++k) {
                    double w[3];
                    for (int a = 0; a < 3; ++a) w[a] = ((k >> a) & 1 ? bx[3 + a] + pad : bx[a] - pad) - c.origin[a];
                    const double vx = w[0] * inv[0][0] + w[1] * inv[1][0] + w[2] * inv[2][0];
                    const double vy = w[0] * inv[0][1] + w[1] * inv[1][1] + w[2] * inv[2][1];
                    const double vz = w[0] * inv[0][2] + w[1] * inv[1][2] + w[2] * inv[2][2];
                    if (!(vz > 1e-6 * std::sqrt(diag))) { front = false; break; }
                    const double dx = vx / vz, dy = -vy / vz;
                    x0 = std::min(x0, dx); x1 = std::max(x1, dx); y0 = std::min(y0, dy); y1 = std::max(y1, dy);
                }
                if (!front) break;
                const double mx = 1e-4 * (x1 - x0) + 1e-6, my = 1e-4 * (y1 - y0) + 1e-6;
                c.cull_rect[n][0] = (float)(x0 - mx); c.cull_rect[n][1] = (float)(x1 + mx);
                c.cull_rect[n][2] = (float)(y0 - my); c.cull_rect[n][3] = (float)(y1 + my);
                ++n;
            }
            if (front) c.cull_valid = n;         // a box behind / around the camera: no culling at all
            // the mask pads every vertex by ten times what the BVH pads its boxes with (bvh.cpp: 2e-5 of the diagonal) — the same assumption about
            // the triangle test's rounding that the traversal itself rests on, with a wider margin
            if (front) (void)refresh_cull_mask(c, inv, 2e-4 * std::sqrt(diag) + 1e-7, 1e-6 * std::sqrt(diag), layout != nullptr);
            if (front && layout && rows) (void)refresh_tile_bins(c, inv, 2e-4 * std::sqrt(diag) + 1e-7, 1e-6 * std::sqrt(diag), *layout, *rows);
            if (getenv("MI355RT_DEBUG_CULL")) { fprintf(stderr, "[mi355rt] cull rects %u front %d max_x %g\n", n, (int)front, c.max_x); for (uint32_t k = 0; k < n; ++k) fprintf(stderr, "   x [%g, %g] y [%g, %g]\n", c.cull_rect[k][0], c.cull_rect[k][1], c.cull_rect[k][2], c.cull_rect[k][3]); }
        }
    }
    return c;
}

#define HIP_ALLOC(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { alloc_failed_ = e__ == hipErrorOutOfMemory; free_pass_buffers(); return fail(e__, #expr); } } while (0)

void Renderer::free_pass_buffers()
{
    for (Slice& sl : slices_) {
        for (int i = 0; i < 2; ++i) {
            if (sl.d_queue[i]) { (void)hipFree(sl.d_queue[i]); sl.d_queue[i] = nullptr; }
            if (sl.d_chunk_counts[i]) { (void)hipFree(sl.d_chunk_counts[i]); sl.d_chunk_counts[i] = nullptr; }
        }
        if (sl.d_slot_L) { (void)hipFree(sl.d_slot_L); sl.d_slot_L = nullptr; }
        if (sl.d_sample_slot) { (void)hipFree(sl.d_sample_slot); sl.d_sample_slot = nullptr; }
        if (sl.d_live) { (void)hipFree(sl.d_live); sl.d_live = nullptr; }
        if (sl.d_block_culled) { (void)hipFree(sl.d_block_culled); sl.d_block_culled = nullptr; sl.block_culled_cap = 0; sl.cull_key.clear(); }
        if (sl.d_slot_ps) { (void)hipFree(sl.d_slot_ps); sl.d_slot_ps = nullptr; }
        if (sl.d_hits) { (void)hipFree(sl.d_hits); sl.d_hits = nullptr; }
        if (sl.d_hit_prim) { (void)hipFree(sl.d_hit_prim); sl.d_hit_prim = nullptr; }
        sl.capacity = 0; sl.bytes = 0; sl.guards.clear();
    }
}

bool Renderer::ensure_pass_capacity(Slice& sl, size_t nsamples)
{
    alloc_failed_ = false;
    if (nsamples <= sl.capacity) return true;
    for (Slice& o : slices_) if (o.stream) HIP_TRY(hipStreamSynchronize(o.stream));
    if (sl.capacity) {          // grow: release this slice's buffers only
        for (int i = 0; i < 2; ++i) { (void)hipFree(sl.d_queue[i]); sl.d_queue[i] = nullptr; (void)hipFree(sl.d_chunk_counts[i]); sl.d_chunk_counts[i] = nullptr; }
        (void)hipFree(sl.d_slot_L); sl.d_slot_L = nullptr; (void)hipFree(sl.d_sample_slot); sl.d_sample_slot = nullptr;
        (void)hipFree(sl.d_slot_ps); sl.d_slot_ps = nullptr; (void)hipFree(sl.d_live); sl.d_live = nullptr;
        (void)hipFree(sl.d_hits); sl.d_hits = nullptr; (void)hipFree(sl.d_hit_prim); sl.d_hit_prim = nullptr;
        sl.capacity = 0; sl.bytes = 0; sl.guards.clear();
    }
    const size_t nchunks = (nsamples + chunk_ - 1) / chunk_;
    const size_t records = nchunks * chunk_ * records_per_sample_;
    // the kernels index records, light-term floats and hit flags with 32 bits (byte offsets are 64-bit)
    if (nsamples > 0x7FFFFFFFull || records > 0x3FFFFFFFull || nchunks * chunk_ * nodes_per_sample * std::max(nlights_, 1u) * 3ull > 0xFFFFFFFFull) {
        last_error = "pass too large"; alloc_failed_ = true; return false;     // the caller retries with a smaller pass
    }
    // (n_radiance, n_shadow) per chunk: the fused 50-row launch cuts the same samples into chunks as small as kMinChunk
    const size_t count_entries = nchunks * chunk_ / std::min(chunk_, kMinChunk);
    // MI355RT_DEBUG_GUARD: every pass buffer gets a 256-byte tail filled with 0xA5 that check_guards() reads back
    // (tests/test_gpu_dropin.py: no launch may write past the sizes computed here)
    const size_t guard = getenv("MI355RT_DEBUG_GUARD") ? kGuardBytes : 0;
    sl.guards.clear();
    auto pass_alloc = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes + guard);
        if (e == hipSuccess && guard) { e = hipMemset((char*)*p + bytes, 0xA5, guard); sl.guards.push_back((const uint8_t*)*p + bytes); }
        return e;
    };
    for (int i = 0; i < 2; ++i) {
        HIP_ALLOC(pass_alloc(&sl.d_queue[i], records * kRayRecordBytes));
        HIP_ALLOC(pass_alloc((void**)&sl.d_chunk_counts[i], count_entries * 8));
    }
    HIP_ALLOC(pass_alloc(&sl.d_hits, records * 16));
    HIP_ALLOC(pass_alloc((void**)&sl.d_hit_prim, records * 4));
    HIP_ALLOC(pass_alloc((void**)&sl.d_slot_L, nchunks * chunk_ * nodes_per_sample * std::max(nlights_, 1u) * 12));
    HIP_ALLOC(pass_alloc((void**)&sl.d_sample_slot, nchunks * chunk_ * 4));
    HIP_ALLOC(pass_alloc((void**)&sl.d_slot_ps, nchunks * chunk_ * 8));
    HIP_ALLOC(pass_alloc((void**)&sl.d_live, (nchunks + kMaxCursors) * 4));
    sl.capacity = nsamples;
    sl.queue_records = records;
    sl.count_entries = count_entries;
    sl.bytes = 2 * (records * kRayRecordBytes + count_entries * 8) + records * 20 + nchunks * chunk_ * ((size_t)nodes_per_sample * std::max(nlights_, 1u) * 12 + 12);
    return true;
}

// Split the owned rows into `nslices` interleaved shares (blocks of stripe_rows rows, round-robin) and
// upload each slice's row list.
bool Renderer::assign_slice_rows(uint32_t nslices)
{
    if (rows_assigned_for_ == nslices) return true;
    for (Slice& o : slices_) if (o.stream) HIP_TRY(hipStreamSynchronize(o.stream));
    for (Slice& o : slices_) { o.rows.clear(); o.cull_key.clear(); }       // other rows: other pixel blocks, other verdicts
    for (size_t i = 0; i < owned_rows.size(); ++i) slices_[(i / cfg.stripe_rows) % nslices].rows.push_back(owned_rows[i]);
    for (uint32_t s = 0; s < nslices; ++s)
        if (!slices_[s].rows.empty()) HIP_TRY(hipMemcpy(slices_[s].d_rows, slices_[s].rows.data(), slices_[s].rows.size() * 4, hipMemcpyHostToDevice));
    rows_assigned_for_ = nslices;
    return true;
}

// One wavefront pass: `nrows` rows starting at d_rows[row0], spp samples per pixel.
// Fill the pass descriptor shared by the wavefront kernels and the fused kernel.
void Renderer::describe_pass(DPass& ps, const Slice& sl, const uint32_t* d_rows, uint32_t row0, uint32_t row_wrap, uint32_t npix, size_t nsamples, uint32_t chunk,
                             bool explicit_sample, uint32_t epixel, uint32_t esample) const
{
    ps = DPass{};
    ps.rows = d_rows; ps.row0 = row0; ps.row_wrap = row_wrap; ps.npix = npix; ps.nsamples = (uint32_t)nsamples;
    ps.spp = npix ? (uint32_t)(nsamples / npix) : 1u;
    // samples of a pixel kept together in the pass order (kernels.hip, sample_of): 2 — thai2 frame 21.1 / 20.7 / 20.7 / 20.9 / 21.5 / 21.6 ms at
    // 1 / 2 / 4 / 8 / 16 / 64, one rank's share of eight 3.27 / 3.24 / 3.30 / 3.34 / 3.40 / 3.63 ms (profiles/r03_notes.md); the largest divisor of spp <= the wish.
    // With the primary rays going through the tile bins (a wave's tile is 64 / group pixels: shorter lists for smaller tiles) and most shadow rays
    // never made: 15.5 / 15.4 / 15.1 / 15.5 ms at 2 / 4 / 8 / 16 — 8 since.
    uint32_t group = 8u;
    if (const char* e = getenv("MI355RT_SAMPLE_GROUP")) { int v = atoi(e); if (v >= 1) group = (uint32_t)v; }
    group = std::max(1u, std::min(group, ps.spp));
    while (ps.spp % group) --group;
    // a power of two by preference (12 spp: 4, not 6): the tile bins and the cached culling verdicts need the group to divide a wave's 64 samples
    if (64u % group) { uint32_t p2 = 8u; while (p2 > 1u && (p2 > group || ps.spp % p2)) p2 >>= 1; group = p2; }
    ps.sample_group = group;
    // tile groups of 8 rows, or of the stripe height when rows are dealt (to ranks and to slices) in blocks of fewer rows: a group
    // that spans two blocks lying far apart in the image makes loose culling rectangles and incoherent tiles
    ps.row_group_shift = 3; while (ps.row_group_shift > 0 && (1u << ps.row_group_shift) > cfg.stripe_rows) --ps.row_group_shift;
    // ... and of a height that divides the pass's rows when there is one (1050 rows: groups of 2, not 4 with a ragged last group): whole groups everywhere is what
    // the tile bins of the primary rays need, and costs nothing
    if (!explicit_sample && cfg.width) { const uint32_t prow = npix / cfg.width; while (ps.row_group_shift > 0 && prow % (1u << ps.row_group_shift)) --ps.row_group_shift; }
    ps.row_group = 1u << ps.row_group_shift;
    ps.seed = (uint32_t)cfg.seed; ps.flags = cfg.flags; ps.recursions = cfg.recursions; ps.spread = cfg.spread;
    ps.nodes_per_sample = nodes_per_sample;
    ps.nslots = (uint32_t)((nsamples + chunk - 1) / chunk) * chunk;
    std::memcpy(ps.level_first, level_first, sizeof ps.level_first);
    ps.use_explicit = explicit_sample ? 1u : 0u; ps.explicit_pixel = epixel; ps.explicit_sampleno = esample;
    ps.chunk = chunk; ps.nchunks = (uint32_t)((nsamples + chunk - 1) / chunk); ps.region = chunk * records_per_sample_;
    ps.hit_prim = sl.d_hit_prim; ps.qstride = sl.queue_records; ps.slot_ps = (uint2*)sl.d_slot_ps;
    ps.stack_depth = traversal_rows(); ps.list_cap = chunk * max_level_nodes_;
    ps.leaf_threshold = leaf_threshold_;
    ps.refill_threshold = 24; if (const char* e = getenv("MI355RT_REFILL")) { int v = atoi(e); if (v >= 1 && v <= 64) ps.refill_threshold = (uint32_t)v; }
    // the primary launch refills later: its rays are neighbours on the screen, and the more of them start together the more lanes of a quad
    // share the lines they fetch (cache-line accesses of the launch at 8 / 24 / 48 / 64 idle lanes: 1.48 / 1.21 / 1.03 / 0.94e10 per 7 frames,
    // profiles/r03_notes.md); secondary rays do not gain from it (5.8e10 at 24 and at 48)
    refill_primary_ = 48; if (const char* e = getenv("MI355RT_REFILL_PRIMARY")) { int v = atoi(e); if (v >= 1 && v <= 64) refill_primary_ = (uint32_t)v; }
    ps.pull_mode = 4u;                  // 64 interleaved cursors (see pull_chunk in kernels.hip)
    if (const char* e = getenv("MI355RT_PULL")) ps.pull_mode = (uint32_t)atoi(e);
    ps.pull_group = 1; if (const char* e = getenv("MI355RT_GROUP")) { int v = atoi(e); if (v >= 1 && v <= 64) ps.pull_group = (uint32_t)v; }
    ps.tail_chunks = 2; if (const char* e = getenv("MI355RT_TAIL_CHUNKS")) { int v = atoi(e); if (v >= 0 && v <= 64) ps.tail_chunks = (uint32_t)v; }
    ps.tail_split_shift = 2; if (const char* e = getenv("MI355RT_TAIL_SPLIT")) { int v = atoi(e); if (v >= 0 && v <= 4) ps.tail_split_shift = (uint32_t)v; }
    ps.ncursors = 64; if (const char* e = getenv("MI355RT_CURSORS")) { int v = atoi(e); if (v >= 1 && v <= (int)kMaxCursors) ps.ncursors = (uint32_t)v; }
    // live-chunk lists (device_types.hpp): with the cursor scheme the wavefront launches use by default, and the chunk size the buffers were sized for.
    // Not with the direct octree walk: its trace launch strides over ALL chunks, and the ray counts of the chunks the shade launches no longer visit
    // are whatever an earlier pass left there.
    ps.live = nullptr; ps.live_count = nullptr; ps.live_cap = 0;
    if (ps.pull_mode == 4u && ps.pull_group == 1u && chunk == chunk_ && sl.d_live != nullptr && !explicit_sample && mode_ != kModeOctreeWalk && !getenv("MI355RT_NO_LIVE")) {
        ps.live = sl.d_live; ps.live_count = sl.d_ctrl; ps.live_cap = (ps.nchunks + ps.ncursors - 1u) / ps.ncursors;
    }
}

// A wavefront pass in three steps, so that a frame can interleave the rounds of its slices (render()):
// pass_begin: buffers, descriptor, cursors; pass_round(r): trace (+ confirm) (+ shade) of round r; pass_end: resolve.
// The trace launch of a round goes to `trace_stream` when one is given — ordered after the slice's own stream and before
// what the slice queues next — everything else to the slice's stream.
bool Renderer::pass_begin(PassRun& run, Slice& sl, const uint32_t* d_rows, uint32_t row0, uint32_t nrows, uint32_t spp, bool explicit_sample, uint32_t epixel, uint32_t esample, uint32_t row_wrap)
{
    run.sl = &sl; run.live = false;
    const uint32_t npix = explicit_sample ? 1u : nrows * cfg.width;
    const size_t nsamples = (size_t)npix * spp;
    if (nsamples == 0) return true;
    if (!ensure_pass_capacity(sl, nsamples)) return false;
    describe_pass(run.ps, sl, d_rows, row0, row_wrap, npix, nsamples, chunk_, explicit_sample, epixel, esample);
    // the tile bins of the primary rays need a pass that walks the slice's whole row list (render(); not the odd row windows of the other callers)
    const bool whole = !explicit_sample && d_rows == sl.d_rows && row0 == 0 && nrows == sl.rows.size() && row_wrap == 0xFFFFFFFFu;
    run.cam = device_camera(whole ? &run.ps : nullptr, whole ? &sl.rows : nullptr);
    // culling verdicts per pixel block, computed once per camera and layout (DPass::block_culled) instead of per launch and chunk
    run.ps.block_culled = nullptr; run.ps.cull_blocks = 0;
    // (needs chunks that hold whole pixels — the sample group divides the chunk — and sample groups that are whole numbers of chunks: then every sample of a pixel
    // lies in chunks of ONE block)
    if (whole && run.cam.cull_valid != 0 && mode_ != kModeOctreeWalk && run.ps.chunk % run.ps.sample_group == 0 && ((size_t)run.ps.npix * run.ps.sample_group) % run.ps.chunk == 0 && !getenv("MI355RT_NO_CULL_CACHE")) {
        const size_t nblocks = (size_t)run.ps.npix * run.ps.sample_group / run.ps.chunk;
        std::vector<float> key(run.cam.rot, run.cam.rot + 16);
        key.insert(key.end(), run.cam.origin, run.cam.origin + 3); key.push_back(run.cam.max_x); key.push_back(run.cam.max_y);
        key.push_back((float)nrows); key.push_back((float)run.ps.chunk); key.push_back((float)run.ps.sample_group); key.push_back((float)run.ps.row_group);
        key.push_back((float)(run.ps.flags & 1u)); key.push_back(run.cam.cull_mask ? 1.0f : 0.0f); key.push_back((float)nblocks);
        if (nblocks > sl.block_culled_cap) {
            HIP_TRY(hipStreamSynchronize(sl.stream));
            if (sl.d_block_culled) { (void)hipFree(sl.d_block_culled); sl.d_block_culled = nullptr; sl.block_culled_cap = 0; }
            if (hipMalloc((void**)&sl.d_block_culled, nblocks * 4) == hipSuccess) sl.block_culled_cap = nblocks; else { (void)hipGetLastError(); sl.d_block_culled = nullptr; }
            sl.cull_key.clear();
        }
        if (sl.d_block_culled) {
            if (key != sl.cull_key) {          // stream-ordered behind the launches that read the old verdicts
                HIP_TRY(launch_cull_blocks(sl.stream, run.cam, run.ps, (uint32_t)nblocks, sl.d_block_culled));
                sl.cull_key = key;
            }
            run.ps.block_culled = sl.d_block_culled; run.ps.cull_blocks = (uint32_t)nblocks;
        }
    }
    run.rounds = cfg.recursions + 2;
    // the work cursors: zeroed at creation and again by the resolve kernel of every pass that ran to its end (pass_end)
    if (!sl.ctrl_clean) HIP_TRY(hipMemsetAsync(sl.d_ctrl, 0, kMaxRounds * kCtrlWordsPerRound * 4, sl.stream));
    sl.ctrl_clean = false;
    run.live = true;
    return true;
}

// round r: trace the rays of level r (+ the shadow rays emitted by level r-1), then shade level r
bool Renderer::pass_round(PassRun& run, uint32_t r, hipStream_t trace_stream, int trace_blocks_per_cu)
{
    if (!run.live || r >= run.rounds) return true;
    Slice& sl = *run.sl;
    const DPass& ps = run.ps;
    const DCamera& cam = run.cam;
    hipStream_t st = sl.stream;             // confirm / shade launches of this pass go to the slice's stream
    hipStream_t tst = trace_stream ? trace_stream : st;
    const bool count = (cfg.flags & MI355RT_FLAG_COUNT_STEPS) != 0;
    const bool timed = (cfg.flags & MI355RT_FLAG_TIME_KERNELS) != 0;
    // Primary round with tile bins: the shade launch finds the closest hits itself (kernels.hip, shade_chunk<RASTER>) — no trace launch at all —
    // whenever that launch also does what else the round needs (the octree confirm step of its hits, or nothing to confirm).
    const bool confirm_round = mode_ == kModeConfirm && !dscene_.oct_single_leaf;
    const char* sw_env = getenv("MI355RT_SHADE_WALK");
    const bool fuse_primary = r == 0 && cam.tile_ofs != nullptr && mode_ != kModeOctreeWalk && (!confirm_round || !sw_env || atoi(sw_env) >= 1) && !getenv("MI355RT_NO_FUSE_PRIMARY");
    const void* in_q = r == 0 ? nullptr : sl.d_queue[(r - 1) & 1];
    const void* in_c = r == 0 ? nullptr : sl.d_chunk_counts[(r - 1) & 1];
    const bool balance_dbg = !fuse_primary && count && getenv("MI355RT_DEBUG_UTIL") && mode_ != kModeOctreeWalk;
    if (!fuse_primary) {
    if (tst != st) { HIP_TRY(hipEventRecord(sl.ev_ready, st)); HIP_TRY(hipStreamWaitEvent(tst, sl.ev_ready, 0)); }
    if (timed) {
        if (ev_used_ + 2 > ev_pool_.size()) {
            for (int k = 0; k < 2; ++k) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); ev_pool_.push_back(ev); }
        }
        HIP_TRY(hipEventRecord(ev_pool_[ev_used_], tst));
        ev_secondary_.resize(ev_pool_.size() / 2); ev_secondary_[ev_used_ / 2] = r > 0;
    }
    if (balance_dbg) {       // load-balance diagnostics of this launch (debug only: synchronises)
        DCounters init{};
        HIP_TRY(hipMemcpy(&init, d_counters_, sizeof init, hipMemcpyDeviceToHost));
        init.t_first_end = ~0ull; init.t_start = ~0ull; init.t_last_end = 0; init.t_sum_end = 0; init.n_waves = 0;
        HIP_TRY(hipMemcpy(d_counters_, &init, sizeof init, hipMemcpyHostToDevice));
    }
    if (mode_ == kModeOctreeWalk)
        HIP_TRY(launch_trace_octree(tst, num_cus_, r == 0, dscene_, cam, ps, in_q, in_c, sl.d_hits, sl.d_slot_L, d_film_n_));
    else {
        DPass pr = ps;
        if (r == 0) pr.refill_threshold = refill_primary_;             // primary rays: see describe_pass
        if (r == 0 && cam.tile_ofs != nullptr)                         // the primary rays' closest hits from the screen-space triangle bins instead of the tree
            HIP_TRY(launch_raster(tst, num_cus_, count, mode_ == kModeConfirm, dscene_, cam, ps, sl.d_hits, sl.d_ctrl + r * kCtrlWordsPerRound, d_film_n_, d_counters_));
        else
        HIP_TRY(launch_trace(tst, num_cus_, trace_blocks_per_cu, r == 0, count, mode_ == kModeConfirm, dscene_, cam, pr, in_q, in_c, sl.d_hits, sl.d_ctrl + r * kCtrlWordsPerRound, sl.d_slot_L, d_film_n_, d_counters_));
    }
    if (timed) { HIP_TRY(hipEventRecord(ev_pool_[ev_used_ + 1], tst)); ev_used_ += 2; }
    ++launches_;
    if (tst != st) { HIP_TRY(hipEventRecord(sl.ev_traced, tst)); HIP_TRY(hipStreamWaitEvent(st, sl.ev_traced, 0)); }
    }
    // true closest hits -> the reference intersector's answers; settles the shadow rays of this round.  One-leaf octrees: done in the
    // trace kernel.  Radiance hits: confirmed by the round's SHADE kernel, which loads the ray and the hit record anyway (primary
    // round: no confirm launch at all; secondary rounds: the confirm launch handles the shadow records only).  24.3 -> 23.3 ms per frame.
    const bool confirm_here = mode_ == kModeConfirm && !dscene_.oct_single_leaf;
    const char* sw = getenv("MI355RT_SHADE_WALK");                 // experiment knob: 0 = confirm launches only, 1 = the primary round's shade kernel walks, 2 (default) = every shade kernel walks its radiance hits
    const int sw_mode = sw ? atoi(sw) : 2;
    const bool shade_walks = confirm_here && r <= cfg.recursions && (r == 0 ? sw_mode >= 1 : sw_mode >= 2);
    if (confirm_here && !(shade_walks && r == 0))
        HIP_TRY(launch_confirm(st, num_cus_, r == 0, shade_walks, dscene_, cam, ps, in_q, in_c, sl.d_hits, sl.d_ctrl + r * kCtrlWordsPerRound + kConfirmCursorOffset, sl.d_slot_L, d_film_n_));
    if (balance_dbg) {
        DCounters c0{};
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(&c0, d_counters_, sizeof c0, hipMemcpyDeviceToHost));
        {   // lane census of the launch (summed over the counter shards; cumulative over the call's launches: print differences)
            DCounters sh[kShards], t{};
            HIP_TRY(hipMemcpy(sh, d_counters_, sizeof sh, hipMemcpyDeviceToHost));
            for (const DCounters& x : sh) { t.lanes_inner += x.lanes_inner; t.lanes_leaf += x.lanes_leaf; t.lanes_done += x.lanes_done; t.lane_samples += x.lane_samples;
                                            t.refills += x.refills; t.refill_passes += x.refill_passes; t.refill_rays += x.refill_rays; t.inner_execs += x.inner_execs; t.leaf_execs += x.leaf_execs; }
            static DCounters prev{};
            if (t.lane_samples < prev.lane_samples) prev = DCounters{};
            const double it = (double)(t.lane_samples - prev.lane_samples);
            if (it > 0)
                fprintf(stderr, "[mi355rt] trace round %u census: iterations %.0f, lanes at inner %.1f / at leaf %.1f / idle-or-finished %.1f per iteration; inner execs %.2f, leaf execs %.2f per iteration; refills %llu, passes %llu, rays per refill %.1f\n",
                        r, it, (t.lanes_inner - prev.lanes_inner) / it, (t.lanes_leaf - prev.lanes_leaf) / it, (t.lanes_done - prev.lanes_done) / it,
                        (t.inner_execs - prev.inner_execs) / it, (t.leaf_execs - prev.leaf_execs) / it,
                        t.refills - prev.refills, t.refill_passes - prev.refill_passes, (double)(t.refill_rays - prev.refill_rays) / std::max<double>(1.0, (double)(t.refills - prev.refills)));
            prev = t;
        }
        if (c0.n_waves)
            fprintf(stderr, "[mi355rt] trace round %u: %llu waves, mean wave busy %.1f us, first wave out of work at %.1f us, last at %.1f us\n", r, c0.n_waves,
                    (double)c0.t_sum_end / c0.n_waves / 100.0, (double)(c0.t_first_end - c0.t_start) / 100.0, (double)(c0.t_last_end - c0.t_start) / 100.0);
    }
    if (r <= cfg.recursions)
        HIP_TRY(launch_shade(st, num_cus_, r == 0, shade_walks, dscene_, cam, ps, r, in_q, in_c, sl.d_hits, sl.d_queue[r & 1], sl.d_chunk_counts[r & 1], sl.d_ctrl + r * kCtrlWordsPerRound + kShadeCursorOffset, sl.d_slot_L, sl.d_sample_slot, d_film_n_, d_counters_, fuse_primary));
    return true;
}

bool Renderer::pass_end(PassRun& run)
{
    if (!run.live) return true;
    Slice& sl = *run.sl;
    HIP_TRY(launch_resolve(sl.stream, run.ps, cfg.width, nlights_, sl.d_slot_L, sl.d_sample_slot, d_film_sum_, d_film_sumsq_, d_film_n_, d_debug_color_, sl.d_ctrl));
    sl.ctrl_clean = true;
    run.live = false;
    return true;
}

bool Renderer::run_pass(Slice& sl, const uint32_t* d_rows, uint32_t row0, uint32_t nrows, uint32_t spp, bool explicit_sample, uint32_t epixel, uint32_t esample, uint32_t row_wrap)
{
    PassRun run;
    if (!pass_begin(run, sl, d_rows, row0, nrows, spp, explicit_sample, epixel, esample, row_wrap)) return false;
    for (uint32_t r = 0; r < run.rounds; ++r) if (!pass_round(run, r, nullptr, 0)) return false;
    return pass_end(run);
}

bool Renderer::begin_call()
{
    if (!bind()) return false;
    if (!settle_speculation()) return false;
    call_done_valid_ = false;
    ev_used_ = 0; launches_ = 0;
    counts_pending_ = false;
    HIP_TRY(hipMemsetAsync(d_counters_, 0, sizeof(DCounters) * kShards, stream_));
    HIP_TRY(hipEventRecord(ev_begin_, stream_));
    active_slices_ = 1;
    return true;
}

// Download the device counters of the call that just ended (the stream must be idle).
bool Renderer::fetch_counts(uint64_t primary, bool timed_call)
{
    DCounters c{};
    const DCounters* shard = h_counters_;          // pinned; filled by the asynchronous copy queue_counts_copy() put behind the call's kernels
    for (uint32_t i = 0; i < kShards; ++i) {
        const DCounters& s = shard[i];
        c.bounce += s.bounce; c.shadow += s.shadow; c.primary_hits += s.primary_hits;
        c.nodes_visited += s.nodes_visited; c.tris_tested += s.tris_tested; c.overflow |= s.overflow;
        c.inner_execs += s.inner_execs; c.leaf_execs += s.leaf_execs; c.primary_culled += s.primary_culled; c.shadow_skipped += s.shadow_skipped;
        c.t_sum_cycles += s.t_sum_cycles; c.t_sum_real += s.t_sum_real;
        for (int k = 0; k < 6; ++k) c.visits_below[k] += s.visits_below[k];
    }
    counts = mi355rt_ray_counts{};
    counts.primary = primary; counts.bounce = c.bounce; counts.shadow = c.shadow + c.shadow_skipped; counts.primary_hits = c.primary_hits;
    counts.shadow_skipped = c.shadow_skipped;     // shadow rays of mod.rs:226 that were never made: the light's depth map proved them free
    counts.nodes_visited = c.nodes_visited; counts.tris_tested = c.tris_tested; counts.trace_launches = launches_;
    counts.inner_execs = c.inner_execs; counts.leaf_execs = c.leaf_execs; counts.primary_culled = c.primary_culled;
    if (getenv("MI355RT_DEBUG_UTIL") && c.nodes_visited)
        fprintf(stderr, "[mi355rt] inner-node visits with node index < 64 / 128 / 256 / 512 / 1024 / 2048: %.3f / %.3f / %.3f / %.3f / %.3f / %.3f of %llu\n", (double)c.visits_below[0] / c.nodes_visited, (double)c.visits_below[1] / c.nodes_visited,
                (double)c.visits_below[2] / c.nodes_visited, (double)c.visits_below[3] / c.nodes_visited, (double)c.visits_below[4] / c.nodes_visited, (double)c.visits_below[5] / c.nodes_visited, c.nodes_visited);
    if (getenv("MI355RT_DEBUG_UTIL")) fprintf(stderr, "[mi355rt] inner execs %llu (lane util %.3f) leaf execs %llu (lane util %.3f)\n", c.inner_execs, c.inner_execs ? (double)c.nodes_visited / (64.0 * c.inner_execs) : 0.0, c.leaf_execs, c.leaf_execs ? (double)c.tris_tested / (64.0 * c.leaf_execs) : 0.0);
    if (timed_call) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev_begin_, ev_end_));
        counts.total_ms = ms;
    }
    double tms = 0.0, sms = 0.0; uint64_t nsec = 0;
    for (size_t i = 0; i + 1 < ev_used_; i += 2) {           // MI355RT_FLAG_TIME_KERNELS: events around every trace / fused launch
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, ev_pool_[i], ev_pool_[i + 1]));
        tms += t;
        if (i / 2 < ev_secondary_.size() && ev_secondary_[i / 2]) { sms += t; ++nsec; }
    }
    counts.trace_ms = tms; counts.trace_secondary_ms = sms; counts.trace_secondary_launches = nsec;
    // s_memtime ticks are shader cycles, s_memrealtime ticks 10 ns: the clock the trace waves ran at (COUNT builds stamp both)
    counts.shader_clock_mhz = c.t_sum_real ? (double)c.t_sum_cycles / (double)c.t_sum_real * 100.0 : 0.0;
    if (c.overflow) { last_error = (c.overflow & 2u) ? "internal: traversal stack overflow (instrumented build: a ray held more deferred nodes than the stack has rows)" : "internal: ray queue overflow"; return false; }
    return true;
}

// the call's device counters -> the pinned host mirror, stream-ordered behind the call's kernels (one wait serves both)
bool Renderer::queue_counts_copy()
{
    HIP_TRY(hipMemcpyAsync(h_counters_, d_counters_, sizeof(DCounters) * kShards, hipMemcpyDeviceToHost, stream_));
    return true;
}

// wait == false (mi355rt_render_async): everything is queued — the slices joined on the main stream, the counters' copy behind them —
// and the call returns; mi355rt_last_counts / mi355rt_synchronize wait.  Consecutive frames then run back to back on the device
// instead of one host round trip apart (a few % of a 3 ms frame: one rank's share of a strong-scaled frame).
bool Renderer::end_call(uint64_t primary, bool wait)
{
    for (uint32_t i = 1; i < active_slices_; ++i) {          // join the slices on the main stream
        HIP_TRY(hipEventRecord(slices_[i].done, slices_[i].stream));
        HIP_TRY(hipStreamWaitEvent(stream_, slices_[i].done, 0));
    }
    active_slices_ = 1;
    HIP_TRY(hipEventRecord(ev_end_, stream_));
    if (!queue_counts_copy()) return false;
    if (!wait) { counts_pending_ = true; pending_primary_ = primary; pending_timed_ = true; return true; }
    HIP_TRY(hipStreamSynchronize(stream_));
    return fetch_counts(primary, true);
}

// Counters of the last call.  A 50-row frame (trace_frame_additive) returns without waiting for the device;
// its counters are fetched when somebody asks for them.
bool Renderer::last_counts(mi355rt_ray_counts& out)
{
    if (counts_pending_) {
        if (!bind()) return false;
        if (call_done_valid_) HIP_TRY(hipEventSynchronize(ev_call_done_));       // a 50-row frame: its kernel and its counters' copy, not what was queued behind them
        else HIP_TRY(hipStreamSynchronize(stream_));
        counts_pending_ = false;
        const bool timed = pending_timed_; pending_timed_ = false;
        if (!fetch_counts(pending_primary_, timed)) return false;
    }
    out = counts;
    return true;
}

// The 50-row window of one trace_frame_additive call (mod.rs:87,114) as runs of the device-resident owned-row
// list: entry i of run k is owned_rows[(first + i) % owned], and no row occurs twice inside a run (a row can
// occur more than once per call only when height < 50; each occurrence must see the previous one's film update).
struct FrameWindow { uint32_t first = 0, total = 0, next_row = 0; };
static FrameWindow frame_window(uint32_t current_row, uint32_t height, uint32_t stripe_rows, uint32_t world, uint32_t rank, const std::vector<uint32_t>& owned)
{
    FrameWindow w;
    uint32_t row = current_row;
    bool have_first = false;
    for (int k = 0; k < 50; ++k) {
        if ((row / stripe_rows) % world == rank) {
            if (!have_first) { w.first = (uint32_t)(std::lower_bound(owned.begin(), owned.end(), row) - owned.begin()); have_first = true; }
            ++w.total;
        }
        row = (row + 1) % height;
    }
    w.next_row = row;
    return w;
}

void Renderer::mark_dirty_window(uint32_t first, uint32_t total)
{
    const uint32_t nown = (uint32_t)owned_rows.size();
    for (uint32_t i = 0; i < total && i < nown; ++i) ldr_dirty_[owned_rows[(first + i) % nown]] = 1;
}

uint32_t Renderer::trace_frame_additive()
{
    const uint32_t nown = (uint32_t)owned_rows.size();
    const FrameWindow win = frame_window(current_row, cfg.height, cfg.stripe_rows, cfg.stripe_world, cfg.stripe_rank, owned_rows);
    // instrumented / timed / reference-exact-octree calls go through the wavefront rounds (several launches, waits for the device)
    const bool wavefront = (cfg.flags & MI355RT_FLAG_COUNT_STEPS) != 0 || mode_ == kModeOctreeWalk
                           || fused_pass_lds_rows(traversal_rows(), max_level_nodes_, records_per_sample_) > 60u || getenv("MI355RT_NO_FUSED");
    if (wavefront) {
        if (!begin_call()) return 0;
        for (uint32_t done = 0; done < win.total; done += nown) {
            const uint32_t n = std::min(nown, win.total - done);
            if (!run_pass(slices_[0], d_owned_rows_, (win.first + done) % nown, n, 1, false, 0, 0, nown)) return 0;
        }
        current_row = win.next_row;
        mark_dirty_window(win.first, win.total);
        if (!end_call((uint64_t)win.total * cfg.width)) return 0;
        return 50u * cfg.width;
    }
    // fast path: one launch per run, nothing uploaded, nothing waited for
    if (!bind()) return 0;
    ev_used_ = 0; launches_ = 0;
    const bool timed = (cfg.flags & MI355RT_FLAG_TIME_KERNELS) != 0;
    const DCamera cam = device_camera();
    std::vector<float> cam_key(cam.rot, cam.rot + 16);
    cam_key.insert(cam_key.end(), cam.origin, cam.origin + 3); cam_key.push_back(cam.max_x); cam_key.push_back(cam.max_y);
    // Was this frame launched already, speculatively, behind the previous one?  Then it is done or under way: take it over.
    bool adopted = false;
    if (spec_.valid) {
        if (spec_.row == current_row && spec_.first == win.first && spec_.total == win.total && spec_.seed == cfg.seed && spec_.flags == cfg.flags && spec_.cam_key == cam_key && !timed) {
            adopted = true; spec_.valid = false; ++spec_adopted_;
            std::swap(d_counters_, d_counters_spec_); std::swap(h_counters_, h_counters_spec_); std::swap(ev_call_done_, ev_spec_done_);
            launches_ = 1;
        } else if (!settle_speculation()) return 0;
    }
    if (!adopted) {
        if (hipMemsetAsync(d_counters_, 0, sizeof(DCounters) * kShards, stream_) != hipSuccess) { last_error = "hipMemsetAsync failed"; return 0; }
        for (uint32_t done = 0; done < win.total; done += nown) {
            const uint32_t n = std::min(nown, win.total - done);
            if (timed) {
                while (ev_used_ + 2 > ev_pool_.size()) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) { last_error = "hipEventCreate failed"; return 0; } ev_pool_.push_back(ev); }
                (void)hipEventRecord(ev_pool_[ev_used_], stream_);
                ev_secondary_.resize(ev_pool_.size() / 2); ev_secondary_[ev_used_ / 2] = false;
            }
            if (!launch_fused_window((win.first + done) % nown, n, cam, d_counters_)) return 0;
            if (timed) { (void)hipEventRecord(ev_pool_[ev_used_ + 1], stream_); ev_used_ += 2; }
            ++launches_;
        }
        if (!queue_counts_copy()) return 0;
        if (ev_call_done_ && hipEventRecord(ev_call_done_, stream_) != hipSuccess) { last_error = "hipEventRecord failed"; return 0; }
    }
    call_done_valid_ = ev_call_done_ != nullptr;
    current_row = win.next_row;
    mark_dirty_window(win.first, win.total);
    counts_pending_ = true; pending_primary_ = (uint64_t)win.total * cfg.width; pending_timed_ = false;
    // The next frame, speculatively (see renderer.hpp): only in the plain case — the whole image on one device (striped handles and device-group members
    // read their pixels out through paths that would give every speculation up), one launch per frame, two consecutive windows that share no row,
    // no instrumentation.
    const bool no_spec = getenv("MI355RT_NO_SPECULATE") != nullptr;
    const FrameWindow nxt = frame_window(current_row, cfg.height, cfg.stripe_rows, cfg.stripe_world, cfg.stripe_rank, owned_rows);
    if (!no_spec && !timed && cfg.stripe_world <= 1 && ev_call_done_ && read_stream_ && win.total <= nown && nxt.total != 0 && win.total + nxt.total <= nown) {
        const size_t bk = (size_t)50 * cfg.width;
        if (!d_bk_sum_) {
            if (hipMalloc((void**)&d_bk_sum_, bk * 12) != hipSuccess || hipMalloc((void**)&d_bk_sumsq_, bk * 12) != hipSuccess || hipMalloc((void**)&d_bk_n_, bk * 4) != hipSuccess
                || hipMalloc((void**)&d_counters_spec_, sizeof(DCounters) * kShards) != hipSuccess || hipHostMalloc((void**)&h_counters_spec_, sizeof(DCounters) * kShards, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError(); return 50u * cfg.width;            // no room to speculate: the frame asked for is queued all the same
            }
            allocs_.push_back(d_bk_sum_); allocs_.push_back(d_bk_sumsq_); allocs_.push_back(d_bk_n_); allocs_.push_back(d_counters_spec_);
        }
        bool ok = launch_film_rows_copy(stream_, d_owned_rows_, nxt.first, nxt.total, nown, cfg.width, d_film_sum_, d_film_sumsq_, d_film_n_, d_bk_sum_, d_bk_sumsq_, d_bk_n_, false) == hipSuccess
                  && hipMemsetAsync(d_counters_spec_, 0, sizeof(DCounters) * kShards, stream_) == hipSuccess
                  && launch_fused_window(nxt.first, nxt.total, cam, d_counters_spec_)
                  && hipMemcpyAsync(h_counters_spec_, d_counters_spec_, sizeof(DCounters) * kShards, hipMemcpyDeviceToHost, stream_) == hipSuccess
                  && hipEventRecord(ev_spec_done_, stream_) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); HIP_TRY(hipStreamSynchronize(stream_));        // what was queued of it ran; put the rows back and go on without
                   (void)launch_film_rows_copy(stream_, d_owned_rows_, nxt.first, nxt.total, nown, cfg.width, d_film_sum_, d_film_sumsq_, d_film_n_, d_bk_sum_, d_bk_sumsq_, d_bk_n_, true); }
        else { spec_.valid = true; spec_.row = current_row; spec_.first = nxt.first; spec_.total = nxt.total; spec_.next_row = nxt.next_row; spec_.cam_key = cam_key; spec_.seed = cfg.seed; spec_.flags = cfg.flags; ++spec_launched_; }
    }
    return 50u * cfg.width;
}

// one launch of the fused kernel over `total` rows of the owned-row list from entry `first` (cyclic), 1 sample per pixel
bool Renderer::launch_fused_window(uint32_t first, uint32_t total, const DCamera& cam, DCounters* dcounters)
{
    const uint32_t nown = (uint32_t)owned_rows.size();
    Slice& sl = slices_[0];
    const size_t nsamples = (size_t)total * cfg.width;
    if (!ensure_pass_capacity(sl, nsamples)) return false;
    DPass ps;
    uint32_t fchunk = 32u;                                            // samples per wave (a 4x8 pixel tile): 64 / 32 / 16 measure 0.303 / 0.266 / 0.277 ms per launch
    if (const char* e = getenv("MI355RT_FUSED_CHUNK")) { int v = atoi(e); if (v == 16 || v == 32 || v == 64) fchunk = (uint32_t)v; }   // <= 64: the LDS lists hold one row per record of a sample; >= kMinChunk: the counts arrays
    describe_pass(ps, sl, d_owned_rows_, first, nown, (uint32_t)nsamples, nsamples, fchunk, false, 0, 0);
    if (ps.nchunks > sl.count_entries || (size_t)ps.nchunks * ps.region > sl.queue_records) { last_error = "internal: fused pass does not fit the pass buffers"; return false; }
    hipError_t e = launch_fused_pass(stream_, num_cus_, mode_ == kModeConfirm, dscene_, cam, ps, max_level_nodes_, records_per_sample_, sl.d_queue[0], sl.d_queue[1],
                                     sl.d_hits, sl.d_slot_L, sl.d_sample_slot, d_film_sum_, d_film_sumsq_, d_film_n_, dcounters);
    if (e != hipSuccess) return fail(e, "fused pass launch");
    return true;
}

// A speculative 50-row frame is out and the caller did something else: its rows go back to what they were (stream-ordered behind it).
bool Renderer::settle_speculation()
{
    if (!spec_.valid) return true;
    spec_.valid = false;
    if (!bind()) return false;
    HIP_TRY(launch_film_rows_copy(stream_, d_owned_rows_, spec_.first, spec_.total, (uint32_t)owned_rows.size(), cfg.width, d_film_sum_, d_film_sumsq_, d_film_n_, d_bk_sum_, d_bk_sumsq_, d_bk_n_, true));
    return true;
}

bool Renderer::render(uint32_t spp, bool wait)
{
    if (!begin_call()) return false;
    const uint32_t nrows = (uint32_t)owned_rows.size();
    if (nrows && spp) {
        // concurrent frame slices: worth their extra launches from ~32 Mi samples per slice on (measured on one rank's share of a
        // strong-scaled 1080p x 64 frame, tools/strong_slices_probe.py: 17 / 35 / 71 / 133 M samples are fastest with 1 / 1-2 / 2 / 2-3 slices)
        uint32_t nsl = std::max(1u, std::min(slices, kMaxSlices));
        if (!slices_explicit) nsl = (uint32_t)std::min<uint64_t>(nsl, std::max<uint64_t>(1, (uint64_t)nrows * cfg.width * spp >> 25));
        nsl = std::min(nsl, (nrows + cfg.stripe_rows - 1) / cfg.stripe_rows);
        if (!assign_slice_rows(nsl)) return false;
        // samples per pass of one slice: 144 Mi — the whole 1080p x 64 spp frame in ONE pass (55 GB of pass buffers).  Every pass costs
        // its launches' ramps and tails: 3 passes of 44 M samples 22.9 ms, 2 passes 22.2-22.6 ms, 1 pass 21.6 ms (profiles/r03_notes.md,
        // with the HBM each takes: 18.9 / 27.4 / 54.5 GB); MI355RT_PASS_SAMPLES or config.samples_per_pass trade the memory back.
        size_t target = (size_t)144 << 20;
        if (const char* e = getenv("MI355RT_PASS_SAMPLES")) { long v = atol(e); if (v >= 1024) target = (size_t)v; }
        // The pass buffers (two ray queues, hit records, light terms: ~410 B per sample) are sized for the
        // largest pass.  If the device cannot hold them (another tenant, a 16 GB part), halve the pass and
        // try again; results do not depend on how samples are batched into passes.
        struct PassDesc { uint32_t r0, nr, kk; };
        std::vector<PassDesc> plan[kMaxSlices];
        for (;;) {
            bool ok = true;
            for (uint32_t s = 0; s < nsl && ok; ++s) {
                Slice& sl = slices_[s];
                plan[s].clear();
                const uint32_t snrows = (uint32_t)sl.rows.size();
                if (!snrows) continue;
                const size_t starget = cfg.samples_per_pass ? (size_t)cfg.samples_per_pass * cfg.width * snrows : target;
                const uint32_t rows_per_pass = (uint32_t)std::min<size_t>(snrows, std::max<size_t>(1, starget / cfg.width));
                uint32_t k = (uint32_t)std::max<size_t>(1, std::min<size_t>(spp, starget / ((size_t)rows_per_pass * cfg.width)));
                if (cfg.samples_per_pass) k = std::min(spp, cfg.samples_per_pass);
                else {
                    const uint32_t np = (spp + k - 1) / k; k = (spp + np - 1) / np;      // equal passes: 48+16 -> 32+32
                    // ... of a multiple of 8 samples per pixel when there is room for it: 8 samples of a pixel sit together in the pass order (describe_pass), and the
                    // tile bins of the primary rays and the cached culling verdicts need that group to divide the 64 samples of a wave (C5: 256 spp in passes of 18 had neither)
                    if (k > 8u && k % 8u) { const uint32_t k8 = k / 8u * 8u; if ((size_t)rows_per_pass * cfg.width * (k8 + 8u) <= starget) k = k8 + 8u; else k = k8; }
                }
                ok = ensure_pass_capacity(sl, (size_t)rows_per_pass * cfg.width * std::min(k, spp));
                for (uint32_t done = 0; ok && done < spp; done += k)
                    for (uint32_t r0 = 0; r0 < snrows; r0 += rows_per_pass)
                        plan[s].push_back(PassDesc{ r0, std::min(rows_per_pass, snrows - r0), std::min(k, spp - done) });
            }
            if (ok) break;
            if (!alloc_failed_ || cfg.samples_per_pass || target <= ((size_t)64 << 10)) return false;
            (void)hipGetLastError();
            target /= 2;
        }
        // fork: the other slices start after everything already queued on the main stream
        for (uint32_t s = 1; s < nsl; ++s) HIP_TRY(hipStreamWaitEvent(slices_[s].stream, ev_begin_, 0));
        active_slices_ = nsl;
        // Enqueue the passes round-robin so that no stream waits for the host: every slice's whole pass on its own stream.
        // Measured alternative (MI355RT_PIPELINE=1, profiles/r02_notes.md): all trace launches on ONE stream, round by round,
        // so that a slice's confirm / shade / resolve launches (memory latency) run beside the NEXT slice's trace (VALU issue).
        // The schedule comes out as planned, but a trace kernel with a streaming kernel beside it runs 20-25 % longer and the
        // frame loses 2 ms against the slices left to themselves (which fall into lockstep: 3 traces, then 3 shades).
        const bool pipelined = nsl > 1 && getenv("MI355RT_PIPELINE");
        int tcap = 0;
        if (const char* e = getenv("MI355RT_PIPE_BLOCKS")) { int v = atoi(e); if (v >= 0) tcap = v; }
        if (pipelined) HIP_TRY(hipStreamWaitEvent(trace_stream_, ev_begin_, 0));
        for (size_t p = 0;; ++p) {
            bool any = false;
            PassRun run[kMaxSlices];
            uint32_t rounds = 0;
            for (uint32_t s = 0; s < nsl; ++s) {
                if (p >= plan[s].size()) continue;
                any = true;
                const PassDesc& d = plan[s][p];
                if (!pipelined) { if (!run_pass(slices_[s], slices_[s].d_rows, d.r0, d.nr, d.kk, false, 0, 0)) return false; continue; }
                if (!pass_begin(run[s], slices_[s], slices_[s].d_rows, d.r0, d.nr, d.kk, false, 0, 0, 0xFFFFFFFFu)) return false;
                rounds = std::max(rounds, run[s].rounds);
            }
            if (!any) break;
            if (!pipelined) continue;
            for (uint32_t r = 0; r < rounds; ++r)
                for (uint32_t s = 0; s < nsl; ++s)
                    if (!pass_round(run[s], r, trace_stream_, tcap)) return false;
            for (uint32_t s = 0; s < nsl; ++s) if (!pass_end(run[s])) return false;
        }
    }
    for (uint32_t r : owned_rows) ldr_dirty_[r] = 1;
    return end_call((uint64_t)nrows * cfg.width * spp, wait);
}

// get_tonemapped_pixels, mod.rs:120-128.  The reference maps the whole film on every call although one
// trace_frame_additive changes 50 rows; here only the rows written since the last read-out are mapped again and
// copied (into a pinned host mirror of the frame), then the caller gets its own copy of the whole frame.
bool Renderer::get_tonemapped(uint32_t* out, size_t n)
{
    if (!bind()) return false;
    const size_t npix = (size_t)cfg.width * cfg.height;
    if (n < npix || !out) { last_error = "output buffer too small"; return false; }
    if (!h_ldr_) HIP_TRY(hipHostMalloc((void**)&h_ldr_, npix * 4, hipHostMallocDefault));
    // A speculative frame may be changing rows right now (trace_frame_additive).  Rows it touches that have to be read (everything is dirty after a
    // clear) would show samples the caller has not asked for yet: then the speculation is given up.  Otherwise the read-out runs beside it, on its own
    // stream, behind the frame the caller did ask for.
    if (spec_.valid) {
        bool clash = false;
        const uint32_t nown = (uint32_t)owned_rows.size();
        for (uint32_t i = 0; i < spec_.total && !clash; ++i) clash = ldr_dirty_[owned_rows[(spec_.first + i) % nown]] != 0;
        if (clash && !settle_speculation()) return false;
    }
    hipStream_t rs = stream_;
    if (spec_.valid && call_done_valid_ && read_stream_) { HIP_TRY(hipStreamWaitEvent(read_stream_, ev_call_done_, 0)); rs = read_stream_; }
    for (uint32_t r = 0; r < cfg.height;) {
        if (!ldr_dirty_[r]) { ++r; continue; }
        uint32_t e = r;
        while (e < cfg.height && ldr_dirty_[e]) ldr_dirty_[e++] = 0;
        HIP_TRY(launch_tonemap(rs, nullptr, r, e - r, cfg.width, false, d_film_sum_, d_film_n_, d_ldr_));
        HIP_TRY(hipMemcpyAsync(h_ldr_ + (size_t)r * cfg.width, d_ldr_ + (size_t)r * cfg.width, (size_t)(e - r) * cfg.width * 4, hipMemcpyDeviceToHost, rs));
        r = e;
    }
    HIP_TRY(hipStreamSynchronize(rs));
    std::memcpy(out, h_ldr_, npix * 4);
    return true;
}

// Owned rows, packed, into caller-owned DEVICE memory.  caller_stream == null: runs on the handle's stream and
// returns when done.  Otherwise the kernel is launched on the caller's stream (after everything this handle has
// queued) and the call returns at once: the write is ordered with whatever the caller does on that stream
// before and after (its collective, its buffer initialisation).
bool Renderer::tonemap_owned_rows_device(uint32_t* device_out, size_t n, hipStream_t caller_stream)
{
    if (!bind()) return false;
    if (!settle_speculation()) return false;
    if (n < owned_rows.size() * (size_t)cfg.width || !device_out) { last_error = "output buffer too small"; return false; }
    if (caller_stream) {
        HIP_TRY(hipEventRecord(slices_[0].done, stream_));
        HIP_TRY(hipStreamWaitEvent(caller_stream, slices_[0].done, 0));
        HIP_TRY(launch_tonemap(caller_stream, d_owned_rows_, 0, (uint32_t)owned_rows.size(), cfg.width, true, d_film_sum_, d_film_n_, device_out));
        // later work of this handle (film.clear, the next frame) must not overtake the read of the film
        HIP_TRY(hipEventRecord(ev_tonemap_, caller_stream));
        HIP_TRY(hipStreamWaitEvent(stream_, ev_tonemap_, 0));
        return true;
    }
    HIP_TRY(launch_tonemap(stream_, d_owned_rows_, 0, (uint32_t)owned_rows.size(), cfg.width, true, d_film_sum_, d_film_n_, device_out));
    HIP_TRY(hipStreamSynchronize(stream_));
    return true;
}

bool Renderer::synchronize()
{
    if (!bind()) return false;
    for (Slice& o : slices_) if (o.stream) HIP_TRY(hipStreamSynchronize(o.stream));
    if (read_stream_) HIP_TRY(hipStreamSynchronize(read_stream_));
    return true;
}

// MI355RT_DEBUG_GUARD builds of the pass buffers: number of guard bytes that no longer hold the fill pattern (-1: error)
long Renderer::check_guards()
{
    if (!bind()) return -1;
    long bad = 0;
    std::vector<uint8_t> h(kGuardBytes);
    for (Slice& sl : slices_) {
        if (sl.stream && hipStreamSynchronize(sl.stream) != hipSuccess) return -1;
        for (const uint8_t* g : sl.guards) {
            if (hipMemcpy(h.data(), g, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            for (uint8_t b : h) bad += b != 0xA5;
        }
    }
    return bad;
}

size_t Renderer::hbm_allocated_bytes() const
{
    size_t b = alloc_bytes_;
    for (const Slice& sl : slices_) b += sl.bytes;
    if (d_gather_) b += (size_t)slot_rows() * cfg.width * 4 * (gather_is_root_ ? cfg.stripe_world : 1);
    return b;
}

// ---- multi-GPU gather ------------------------------------------------------------------------------------------
uint32_t Renderer::slot_rows() const
{
    const uint32_t stripes = (cfg.height + cfg.stripe_rows - 1) / cfg.stripe_rows;
    return ((stripes + cfg.stripe_world - 1) / cfg.stripe_world) * cfg.stripe_rows;
}
uint32_t Renderer::rows_of_rank(uint32_t rank) const
{
    uint32_t n = 0;
    for (uint32_t s = rank; (uint64_t)s * cfg.stripe_rows < cfg.height; s += cfg.stripe_world) n += std::min(cfg.stripe_rows, cfg.height - s * cfg.stripe_rows);
    return n;
}
bool Renderer::gather_prepare(bool root)
{
    if (!bind()) return false;
    if (d_gather_ && gather_is_root_ == root) return true;
    if (d_gather_) { HIP_TRY(hipStreamSynchronize(stream_)); (void)hipFree(d_gather_); d_gather_ = nullptr; }
    const size_t slot = (size_t)slot_rows() * cfg.width * 4;
    HIP_TRY(hipMalloc((void**)&d_gather_, slot * (root ? cfg.stripe_world : 1)));
    gather_is_root_ = root;
    return true;
}
bool Renderer::tonemap_to_gather_slot()
{
    if (!bind()) return false;
    if (!settle_speculation()) return false;
    HIP_TRY(launch_tonemap(stream_, d_owned_rows_, 0, (uint32_t)owned_rows.size(), cfg.width, true, d_film_sum_, d_film_n_, gather_slot(cfg.stripe_rank)));
    return true;
}
bool Renderer::finish_gather(uint32_t* host_out, size_t n)
{
    if (!bind()) return false;
    const size_t npix = (size_t)cfg.width * cfg.height;
    if (host_out && n < npix) { last_error = "output buffer too small"; return false; }
    HIP_TRY(launch_place_stripes(stream_, d_gather_, d_ldr_, cfg.width, cfg.height, cfg.stripe_rows, cfg.stripe_world, slot_rows()));
    if (host_out) {
        if (!h_ldr_) HIP_TRY(hipHostMalloc((void**)&h_ldr_, npix * 4, hipHostMallocDefault));
        HIP_TRY(hipMemcpyAsync(h_ldr_, d_ldr_, npix * 4, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        std::memcpy(host_out, h_ldr_, npix * 4);
    }
    return true;
}

bool Renderer::film_get(float* sum, float* sumsq, uint32_t* n)
{
    if (!bind()) return false;
    if (!settle_speculation()) return false;
    const size_t npix = (size_t)cfg.width * cfg.height;
    HIP_TRY(hipStreamSynchronize(stream_));
    if (sum) HIP_TRY(hipMemcpy(sum, d_film_sum_, npix * 12, hipMemcpyDeviceToHost));
    if (sumsq) HIP_TRY(hipMemcpy(sumsq, d_film_sumsq_, npix * 12, hipMemcpyDeviceToHost));
    if (n) HIP_TRY(hipMemcpy(n, d_film_n_, npix * 4, hipMemcpyDeviceToHost));
    return true;
}

bool Renderer::film_clear()
{
    if (!bind()) return false;
    if (!settle_speculation()) return false;
    const size_t npix = (size_t)cfg.width * cfg.height;
    std::fill(ldr_dirty_.begin(), ldr_dirty_.end(), (uint8_t)1);     // unsampled rows read back white (NaN -> 255)
    if (cfg.stripe_world > 1) {
        // a striped handle only ever writes its own rows (the others stay as created: zero): one launch over them instead of three
        // whole-film memsets — 20 us of a 3.3 ms frame on one rank of eight
        HIP_TRY(launch_film_clear_rows(stream_, d_owned_rows_, (uint32_t)owned_rows.size(), cfg.width, d_film_sum_, d_film_sumsq_, d_film_n_));
        return true;
    }
    HIP_TRY(hipMemsetAsync(d_film_sum_, 0, npix * 12, stream_));
    HIP_TRY(hipMemsetAsync(d_film_sumsq_, 0, npix * 12, stream_));
    HIP_TRY(hipMemsetAsync(d_film_n_, 0, npix * 4, stream_));
    return true;            // stream-ordered: every later call on this handle starts on the same stream
}

bool Renderer::intersect(const float* rays6, size_t n, float* tuv, uint32_t* prim, uint8_t* blocked)
{
    if (!bind()) return false;
    if (n == 0) return true;
    if (n > 0x7FFFFFFFull / 24) { last_error = "too many rays in one batch"; return false; }
    const bool shadow = blocked != nullptr;
    float* d_rays = nullptr; float* d_tuv = nullptr; uint32_t* d_prim = nullptr; uint8_t* d_blocked = nullptr;
    bool ok = true;
    auto chk = [&](hipError_t e, const char* w) { if (ok && e != hipSuccess) { ok = fail(e, w); } };
    chk(hipMalloc((void**)&d_rays, n * 24), "hipMalloc rays");
    chk(hipMalloc((void**)&d_tuv, n * 12), "hipMalloc tuv");
    chk(hipMalloc((void**)&d_prim, n * 4), "hipMalloc prim");
    chk(hipMalloc((void**)&d_blocked, n), "hipMalloc blocked");
    if (ok) chk(hipMemcpyAsync(d_rays, rays6, n * 24, hipMemcpyHostToDevice, stream_), "upload rays");
    if (ok && !shadow) chk(hipMemcpyAsync(d_tuv, tuv, n * 12, hipMemcpyHostToDevice, stream_), "upload tuv");   // misses stay untouched
    if (ok) chk(launch_intersect(stream_, dscene_, traversal_rows(), mode_ == kModeConfirm ? 0 : mode_ == kModeOctreeWalk ? 1 : 2, d_rays, (uint32_t)n, shadow, d_tuv, d_prim, d_blocked), "intersect kernel");
    if (ok && shadow) chk(hipMemcpyAsync(blocked, d_blocked, n, hipMemcpyDeviceToHost, stream_), "download blocked");
    if (ok && !shadow) {
        chk(hipMemcpyAsync(tuv, d_tuv, n * 12, hipMemcpyDeviceToHost, stream_), "download tuv");
        chk(hipMemcpyAsync(prim, d_prim, n * 4, hipMemcpyDeviceToHost, stream_), "download prim");
    }
    if (ok) chk(hipStreamSynchronize(stream_), "sync");
    (void)hipFree(d_rays); (void)hipFree(d_tuv); (void)hipFree(d_prim); (void)hipFree(d_blocked);
    return ok;
}

bool Renderer::debug_numerics(const float* a, const float* b, size_t n, float* q, float* r, float* p)
{
    if (!bind()) return false;
    if (n == 0) return true;
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, n * 4 * 5));
    bool ok = true;
    auto chk = [&](hipError_t e, const char* w) { if (ok && e != hipSuccess) ok = fail(e, w); };
    chk(hipMemcpyAsync(d, a, n * 4, hipMemcpyHostToDevice, stream_), "upload a");
    chk(hipMemcpyAsync(d + n, b, n * 4, hipMemcpyHostToDevice, stream_), "upload b");
    if (ok) chk(launch_numerics(stream_, d, d + n, (uint32_t)n, d + 2 * n, d + 3 * n, d + 4 * n), "numerics kernel");
    chk(hipMemcpyAsync(q, d + 2 * n, n * 4, hipMemcpyDeviceToHost, stream_), "download q");
    chk(hipMemcpyAsync(r, d + 3 * n, n * 4, hipMemcpyDeviceToHost, stream_), "download r");
    chk(hipMemcpyAsync(p, d + 4 * n, n * 4, hipMemcpyDeviceToHost, stream_), "download p");
    chk(hipStreamSynchronize(stream_), "sync");
    (void)hipFree(d);
    return ok;
}

// mi355rt_debug_gather_rate: the divergent-gather rate of this device's vector memory pipe (kernels.hip, gather_rate_kernel).
// out[0] = cache-line accesses per second (64 lanes x 2 loads per wave-step, each lane on its own line), out[1] = kernel ms,
// out[2] = 32-byte node fetches per second.
bool Renderer::debug_gather_rate(uint32_t table_nodes, uint32_t steps, double out[3])
{
    if (!bind()) return false;
    if (table_nodes < 1024 || steps == 0) { last_error = "bad debug_gather_rate arguments"; return false; }
    std::vector<uint32_t> host((size_t)table_nodes * 8);
    uint32_t x = 12345u;
    for (uint32_t& v : host) { x = x * 1664525u + 1013904223u; v = x >> 8; }
    void* d = nullptr; uint32_t* sink = nullptr;
    HIP_TRY(hipMalloc(&d, host.size() * 4));
    bool ok = true;
    auto chk = [&](hipError_t e, const char* w) { if (ok && e != hipSuccess) ok = fail(e, w); };
    chk(hipMalloc((void**)&sink, 4), "hipMalloc");
    chk(hipMemcpy(d, host.data(), host.size() * 4, hipMemcpyHostToDevice), "upload table");
    float best = 0.0f;
    for (int rep = 0; ok && rep < 3; ++rep) {                          // first launch warms the caches and the clocks
        chk(hipEventRecord(ev_begin_, stream_), "event");
        chk(launch_gather_rate(stream_, num_cus_, d, table_nodes, steps, sink), "gather kernel");
        chk(hipEventRecord(ev_end_, stream_), "event");
        chk(hipStreamSynchronize(stream_), "sync");
        float ms = 0.0f;
        if (ok) chk(hipEventElapsedTime(&ms, ev_begin_, ev_end_), "elapsed");
        if (rep > 0 && (best == 0.0f || ms < best)) best = ms;
    }
    (void)hipFree(d); if (sink) (void)hipFree(sink);
    if (!ok || best <= 0.0f) return false;
    const double wave_steps = (double)num_cus_ * 8.0 * 4.0 * steps;
    out[0] = wave_steps * 128.0 / (best * 1e-3); out[1] = best; out[2] = wave_steps * 64.0 / (best * 1e-3);
    return true;
}

bool Renderer::debug_slab(const float* inv_rays6, const float* cubes6, size_t n, uint8_t* hit, float* tmin)
{
    if (!bind()) return false;
    if (n == 0) return true;
    if (n > (1u << 24)) { last_error = "too many slab tests in one batch"; return false; }
    float* d = nullptr; uint8_t* dh = nullptr;
    HIP_TRY(hipMalloc((void**)&d, n * 4 * 13));
    bool ok = true;
    auto chk = [&](hipError_t e, const char* w) { if (ok && e != hipSuccess) ok = fail(e, w); };
    chk(hipMalloc((void**)&dh, n), "hipMalloc");
    chk(hipMemcpyAsync(d, inv_rays6, n * 24, hipMemcpyHostToDevice, stream_), "upload rays");
    chk(hipMemcpyAsync(d + 6 * n, cubes6, n * 24, hipMemcpyHostToDevice, stream_), "upload cubes");
    if (ok) chk(launch_slab(stream_, d, d + 6 * n, (uint32_t)n, dh, d + 12 * n), "slab kernel");
    chk(hipMemcpyAsync(hit, dh, n, hipMemcpyDeviceToHost, stream_), "download hit");
    chk(hipMemcpyAsync(tmin, d + 12 * n, n * 4, hipMemcpyDeviceToHost, stream_), "download tmin");
    chk(hipStreamSynchronize(stream_), "sync");
    (void)hipFree(d); if (dh) (void)hipFree(dh);
    return ok;
}

// Film::get_pixels / get_estimated_variances computed on the device, downloaded as width*height*3 floats
bool Renderer::film_stat(bool variances, float* rgb)
{
    if (!bind()) return false;
    if (!settle_speculation()) return false;
    const size_t npix = (size_t)cfg.width * cfg.height;
    float* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, npix * 12));
    bool ok = true;
    auto chk = [&](hipError_t e, const char* w) { if (ok && e != hipSuccess) ok = fail(e, w); };
    chk(launch_film_stat(stream_, variances, npix, d_film_sum_, d_film_sumsq_, d_film_n_, d), "film_stat kernel");
    chk(hipMemcpyAsync(rgb, d, npix * 12, hipMemcpyDeviceToHost, stream_), "download");
    chk(hipStreamSynchronize(stream_), "sync");
    (void)hipFree(d);
    return ok;
}

bool Renderer::debug_sample(uint32_t pixel, uint32_t sampleno, float* color3, float* node_L, size_t nodes)
{
    if (!bind()) return false;
    if (nodes < nodes_per_sample || pixel >= cfg.width * cfg.height) { last_error = "bad debug_sample arguments"; return false; }
    if (!settle_speculation()) return false;
    call_done_valid_ = false;
    HIP_TRY(hipMemsetAsync(d_counters_, 0, sizeof(DCounters) * kShards, stream_));
    ev_used_ = 0; counts_pending_ = false;
    if (!run_pass(slices_[0], nullptr, 0, 1, 1, true, pixel, sampleno)) return false;
    HIP_TRY(hipStreamSynchronize(stream_));
    const uint32_t nl = std::max(nlights_, 1u);
    std::vector<float> raw((size_t)nodes_per_sample * nl * 3);
    uint32_t sl = 0xFFFFFFFFu;
    HIP_TRY(hipMemcpy(&sl, slices_[0].d_sample_slot, 4, hipMemcpyDeviceToHost));
    std::fill(raw.begin(), raw.end(), 0.0f);
    if (sl != 0xFFFFFFFFu)         // slot_L is node-major: plane q = node * nlights + light, chunk_ slots in this one-sample pass
        for (size_t q = 0; q < (size_t)nodes_per_sample * nl; ++q)
            HIP_TRY(hipMemcpy(raw.data() + 3 * q, slices_[0].d_slot_L + 3 * (q * chunk_ + sl), 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(color3, d_debug_color_, 12, hipMemcpyDeviceToHost));
    for (uint32_t nd = 0; nd < nodes_per_sample; ++nd) {
        float acc[3] = { 0.0f, 0.0f, 0.0f };
        for (uint32_t li = 0; li < nlights_; ++li)
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + raw[3 * ((size_t)nd * nl + li) + c];
        node_L[3 * nd] = acc[0]; node_L[3 * nd + 1] = acc[1]; node_L[3 * nd + 2] = acc[2];
    }
    return true;
}

}  // namespace mi355rt
