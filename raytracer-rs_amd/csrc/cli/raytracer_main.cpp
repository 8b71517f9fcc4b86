// raytracer — headless counterpart of the reference's `raytracer` binary (raytracer/src/main.rs).
// Same flags and defaults (main.rs:13-15, 26-99: -f/--file, -m/--max_triangles, -i/--frame_iterations,
// --width, --height; unparsable numbers silently fall back to the defaults) and the same loop
// (main.rs:197-216: trace_frame_additive -> get_tonemapped_pixels -> print stats), with the window
// replaced by an optional image file.  Additions: --spp N (whole frames of N samples per pixel through
// mi355rt_render instead of the 50-row calls), --seed S, --out file.ppm, --fix-row-index.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "../raytracer_lib.hpp"

static bool parse_usize(const char* s, size_t& out)
{
    if (!s || !*s) return false;
    char* end = nullptr;
    unsigned long long v = std::strtoull(s, &end, 10);
    if (*end != '\0' || s[0] == '-') return false;
    out = (size_t)v;
    return true;
}

int main(int argc, char** argv)
{
    const size_t DEFAULT_WIDTH = 1024, DEFAULT_HEIGHT = 768;          // main.rs:13-14
    std::string file = "./data/thai2.dae";                           // main.rs:15
    size_t max_triangles = raytracer_lib::DEFAULT_TRIANGLES_PER_LEAF, width = DEFAULT_WIDTH, height = DEFAULT_HEIGHT;
    size_t frame_iterations = 0, spp = 0, seed = 1;
    bool have_iterations = false, fix_row = false;
    std::string out;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        const char* v = i + 1 < argc ? argv[i + 1] : nullptr;
        auto take = [&]() { ++i; return v; };
        if (a == "-f" || a == "--file") { if (v) file = take(); }
        else if (a == "-m" || a == "--max_triangles") { size_t t; if (parse_usize(take(), t)) max_triangles = t; }
        else if (a == "-i" || a == "--frame_iterations") { size_t t; if (parse_usize(take(), t)) { frame_iterations = t; have_iterations = true; } }
        else if (a == "--width") { size_t t; if (parse_usize(take(), t)) width = t; }
        else if (a == "--height") { size_t t; if (parse_usize(take(), t)) height = t; }
        else if (a == "--spp") { size_t t; if (parse_usize(take(), t)) spp = t; }
        else if (a == "--seed") { size_t t; if (parse_usize(take(), t)) seed = t; }
        else if (a == "--out") { if (v) out = take(); }
        else if (a == "--fix-row-index") fix_row = true;
        else if (a == "-h" || a == "--help") {
            std::printf("raytracer-rs (MI355X) 0.1.0\nusage: raytracer [-f COLLADA_FILENAME] [-m MAX_TRIS] [-i FRAME_ITERATIONS] [--width W] [--height H]\n"
                        "                 [--spp N] [--seed S] [--out image.ppm] [--fix-row-index]\n");
            return 0;
        }
    }
    std::printf("max triangles per leaf: %zu\n", max_triangles);      // main.rs:66
    if (have_iterations) std::printf("will quit after %zu frame iterations\n", frame_iterations);   // main.rs:73
    if (!have_iterations) { frame_iterations = spp ? 1 : (height + 49) / 50; }   // headless: one sweep of the frame

    try {
        mi355rt_config cfg = raytracer_lib::make_config(max_triangles, width, height);
        cfg.seed = seed;
        if (fix_row) cfg.flags |= MI355RT_FLAG_FIX_ROW_INDEX;
        raytracer_lib::RayTracer rt = raytracer_lib::create_raytracer_from_file(file, max_triangles, width, height, &cfg);
        std::printf("number of triangles: %u\n", mi355rt_triangle_count(rt.handle()));   // colladaloader.rs:265
        raytracer_lib::stats::Stats stats;
        std::vector<uint32_t> ldr;
        for (size_t it = 0; it < frame_iterations; ++it) {
            uint32_t num_primary_rays;
            if (spp) {
                mi355rt_ray_counts c = rt.render((uint32_t)spp);
                num_primary_rays = (uint32_t)c.primary;
                std::printf("frame: %.3f ms  rays: %llu primary %llu bounce %llu shadow -> %.1f Mrays/s\n", c.total_ms,
                            (unsigned long long)c.primary, (unsigned long long)c.bounce, (unsigned long long)c.shadow,
                            (double)(c.primary + c.bounce + c.shadow) / c.total_ms / 1e3);
            } else {
                num_primary_rays = rt.trace_frame_additive();
            }
            ldr = rt.get_tonemapped_pixels();
            std::printf("%s\n", stats.stats(num_primary_rays).c_str());  // main.rs:213
        }
        std::printf("%s\n\n\n", stats.mean_stats().c_str());             // main.rs:216
        if (!out.empty()) {
            FILE* f = std::fopen(out.c_str(), "wb");
            if (!f) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
            std::fprintf(f, "P6\n%zu %zu\n255\n", width, height);
            for (uint32_t p : ldr) { unsigned char rgb[3] = { (unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p }; std::fwrite(rgb, 1, 3, f); }
            std::fclose(f);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());                   // main() -> Result<(), String>
        return 1;
    }
    return 0;
}
