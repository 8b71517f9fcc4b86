// group.hpp — what a mi355rt_handle is: one Renderer per HIP device of ONE process.
// With mi355rt_config.device_count <= 1 it is a single renderer and every call forwards to it.  With N devices the
// rows of the frame are dealt to the devices in stripes (the same decomposition the one-process-per-GPU setup uses,
// DESIGN.md §7): every device renders its stripes into its own film; get_tonemapped_pixels moves the packed u32
// stripes to device 0 with hipMemcpyPeerAsync over xGMI and places them there.  The reference's callers
// (raytracer/src/main.rs:183-216) see one RayTracer either way.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "renderer.hpp"

namespace mi355rt {

class DeviceGroup {
public:
    static std::unique_ptr<DeviceGroup> create(const SceneData& scene, const mi355rt_config& cfg, std::string& err, int& code);

    Renderer* primary() const { return devs_[0].get(); }
    size_t size() const { return devs_.size(); }
    Renderer* device(size_t i) const { return devs_[i].get(); }

    uint32_t trace_frame_additive();
    bool render(uint32_t spp, bool wait = true);
    bool last_counts(mi355rt_ray_counts& out);
    bool get_tonemapped(uint32_t* out, size_t n);
    bool film_get(float* sum, float* sumsq, uint32_t* n);
    bool film_stat(bool variances, float* rgb);
    bool film_clear();
    void camera_move_rel(float x, float y, float z);
    void camera_add_x_angle(float r);
    void camera_add_y_angle(float r);
    bool set_seed(uint64_t seed);
    bool set_flags(uint32_t flags);
    void set_slices(uint32_t slices);
    bool synchronize();
    const std::string& last_error() const { return error_.empty() ? primary()->last_error : error_; }

private:
    bool fail_from(size_t i) { error_ = "device " + std::to_string(devs_[i]->cfg.device) + ": " + devs_[i]->last_error; return false; }
    std::vector<std::unique_ptr<Renderer>> devs_;
    mi355rt_ray_counts counts_{};
    bool counts_from_render_ = false;
    std::string error_;
};

}  // namespace mi355rt
