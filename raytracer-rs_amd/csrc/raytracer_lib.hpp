// raytracer_lib.hpp — C++ mirror of the reference crate's public surface (raytracer_lib/src/lib.rs)
// over the C ABI of libmi355rt.so.  Same names, argument order and error behaviour:
//   create_raytracer(collada_doc, triangles_per_leaf, width, height) -> Result<RayTracer, String>   lib.rs:15-20
//   create_raytracer_from_file(collada_filename, ...)                                               lib.rs:22-27
//   RayTracer::trace_frame_additive() -> u32, get_tonemapped_pixels() -> Vec<u32>                   raytracer/mod.rs:80,120
//   RayTracer::camera.{move_rel, add_x_angle, add_y_angle}, RayTracer::film.clear()                 camera.rs:63-78, film.rs:37
//   stats::Stats::{new, stats, mean_stats}                                                          stats.rs:11-39
//   DEFAULT_TRIANGLES_PER_LEAF                                                                      lib.rs:7
// Err(String) becomes a thrown std::runtime_error carrying the same text.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mi355rt.h"

namespace raytracer_lib {

constexpr size_t DEFAULT_TRIANGLES_PER_LEAF = MI355RT_DEFAULT_TRIANGLES_PER_LEAF;

class RayTracer;

class Camera {            // scene/camera.rs (pub methods used by raytracer/src/main.rs:125-161)
public:
    void move_rel(float x, float y, float z) { mi355rt_camera_move_rel(h_, x, y, z); }
    void add_x_angle(float radians) { mi355rt_camera_add_x_angle(h_, radians); }
    void add_y_angle(float radians) { mi355rt_camera_add_y_angle(h_, radians); }
private:
    friend class RayTracer;
    mi355rt_handle* h_ = nullptr;
};

class Film {              // raytracer/film.rs
public:
    void clear() { mi355rt_film_clear(h_); }
    std::vector<float> get_pixels() const
    {
        std::vector<float> out((size_t)mi355rt_width(h_) * mi355rt_height(h_) * 3);
        mi355rt_film_get_pixels(h_, out.data());
        return out;
    }
private:
    friend class RayTracer;
    mi355rt_handle* h_ = nullptr;
};

class RayTracer {         // raytracer/mod.rs:32-47
public:
    Camera camera;
    Film film;
    explicit RayTracer(mi355rt_handle* h) : h_(h) { camera.h_ = h; film.h_ = h; }
    RayTracer(RayTracer&& o) noexcept : camera(o.camera), film(o.film), h_(o.h_) { o.h_ = nullptr; }
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;
    ~RayTracer() { mi355rt_destroy(h_); }

    uint32_t trace_frame_additive()
    {
        uint32_t n = mi355rt_trace_frame_additive(h_);
        if (n == 0) throw std::runtime_error(mi355rt_last_error(h_));
        return n;
    }
    std::vector<uint32_t> get_tonemapped_pixels() const
    {
        std::vector<uint32_t> out((size_t)mi355rt_width(h_) * mi355rt_height(h_));
        if (mi355rt_get_tonemapped_pixels(h_, out.data(), out.size()) != MI355RT_OK) throw std::runtime_error(mi355rt_last_error(h_));
        return out;
    }
    // additions without a reference counterpart
    mi355rt_ray_counts render(uint32_t spp)
    {
        mi355rt_ray_counts c{};
        if (mi355rt_render(h_, spp, &c) != MI355RT_OK) throw std::runtime_error(mi355rt_last_error(h_));
        return c;
    }
    mi355rt_handle* handle() const { return h_; }
private:
    mi355rt_handle* h_;
};

inline mi355rt_config make_config(size_t triangles_per_leaf, size_t width, size_t height)
{
    mi355rt_config cfg;
    mi355rt_default_config(&cfg);
    cfg.triangles_per_leaf = (uint32_t)triangles_per_leaf; cfg.width = (uint32_t)width; cfg.height = (uint32_t)height;
    return cfg;
}
inline RayTracer create_raytracer(const std::string& collada_doc, size_t triangles_per_leaf, size_t width, size_t height,
                                  const mi355rt_config* cfg_override = nullptr)
{
    mi355rt_config cfg = cfg_override ? *cfg_override : make_config(triangles_per_leaf, width, height);
    mi355rt_handle* h = nullptr;
    if (mi355rt_create_from_collada_str(collada_doc.data(), collada_doc.size(), nullptr, &cfg, &h) != MI355RT_OK)
        throw std::runtime_error(mi355rt_last_error(nullptr));
    return RayTracer(h);
}
inline RayTracer create_raytracer_from_file(const std::string& collada_filename, size_t triangles_per_leaf, size_t width, size_t height,
                                            const mi355rt_config* cfg_override = nullptr)
{
    mi355rt_config cfg = cfg_override ? *cfg_override : make_config(triangles_per_leaf, width, height);
    mi355rt_handle* h = nullptr;
    const bool scene_file = collada_filename.size() > 6 && collada_filename.compare(collada_filename.size() - 6, 6, ".scene") == 0;
    int rc = scene_file ? mi355rt_create_from_scene_file(collada_filename.c_str(), &cfg, &h)
                        : mi355rt_create_from_collada_file(collada_filename.c_str(), &cfg, &h);
    if (rc != MI355RT_OK) throw std::runtime_error(mi355rt_last_error(nullptr));
    return RayTracer(h);
}

namespace stats {
class Stats {             // stats.rs:3-40
public:
    Stats() : last_iteration_(std::chrono::steady_clock::now()) {}
    std::string stats(uint32_t num_primary_rays)
    {
        auto now = std::chrono::steady_clock::now();
        float secs = std::chrono::duration<float>(now - last_iteration_).count();
        last_iteration_ = now;
        float fps = 1.0f / secs;
        fps_sum_ += fps;
        float prs = (float)num_primary_rays / secs;
        primrays_per_sec_sum_ += prs;
        num_measurements_ += 1;
        char buf[128];
        std::snprintf(buf, sizeof buf, "fps: %g  primary rays/s: %u", fps, (unsigned)prs);
        return buf;
    }
    std::string mean_stats() const
    {
        char buf[128];
        std::snprintf(buf, sizeof buf, "mean fps: %g  mean primary rays/s: %g", fps_sum_ / (float)num_measurements_,
                      primrays_per_sec_sum_ / (float)num_measurements_);
        return buf;
    }
private:
    std::chrono::steady_clock::time_point last_iteration_;
    float fps_sum_ = 0.0f, primrays_per_sec_sum_ = 0.0f;
    uint32_t num_measurements_ = 0;
};
}  // namespace stats

}  // namespace raytracer_lib
