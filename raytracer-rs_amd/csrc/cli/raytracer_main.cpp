// raytracer — headless counterpart of the reference's `raytracer` binary (raytracer/src/main.rs).
// Same flags and defaults (main.rs:13-15, 26-99: -f/--file, -m/--max_triangles, -i/--frame_iterations,
// --width, --height; unparsable numbers silently fall back to the defaults) and the same loop
// (main.rs:197-216: trace_frame_additive -> get_tonemapped_pixels -> print stats), with the window
// replaced by an optional image file.  Additions: --spp N (whole frames of N samples per pixel through
// mi355rt_render instead of the 50-row calls), --seed S, --gpus N (a device group: the N GPUs of this process share the
// rows, mi355rt_config.device_count), --out file.ppm | file.png, --fix-row-index.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <zlib.h>
#include "../raytracer_lib.hpp"

// 8-bit RGB PNG: one IDAT chunk, filter type 0 on every scanline
static bool write_png(const std::string& path, const std::vector<uint32_t>& argb, size_t width, size_t height)
{
    std::vector<unsigned char> raw;
    raw.reserve((width * 3 + 1) * height);
    for (size_t y = 0; y < height; ++y) {
        raw.push_back(0);
        for (size_t x = 0; x < width; ++x) { uint32_t p = argb[y * width + x]; raw.push_back((unsigned char)(p >> 16)); raw.push_back((unsigned char)(p >> 8)); raw.push_back((unsigned char)p); }
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto be32 = [](unsigned char* b, uint32_t v) { b[0] = (unsigned char)(v >> 24); b[1] = (unsigned char)(v >> 16); b[2] = (unsigned char)(v >> 8); b[3] = (unsigned char)v; };
    auto chunk = [&](const char* type, const unsigned char* data, uint32_t len) {
        unsigned char hdr[8]; be32(hdr, len); std::memcpy(hdr + 4, type, 4);
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(data, 1, len, f);
        uLong crc = crc32(0L, (const Bytef*)type, 4);
        if (len) crc = crc32(crc, data, len);
        unsigned char c[4]; be32(c, (uint32_t)crc); std::fwrite(c, 1, 4, f);
    };
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    std::fwrite(sig, 1, 8, f);
    unsigned char ihdr[13]; be32(ihdr, (uint32_t)width); be32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;          // 8 bit, colour type 2 (RGB)
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    return std::fclose(f) == 0;
}

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
    size_t frame_iterations = 0, spp = 0, seed = 1, gpus = 1;
    bool have_iterations = false, fix_row = false, share_device = false, device_lbvh = false;
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
        else if (a == "--gpus") { size_t t; if (parse_usize(take(), t) && t >= 1) gpus = t; }
        else if (a == "--share-device") share_device = true;
        else if (a == "--out") { if (v) out = take(); }
        else if (a == "--fix-row-index") fix_row = true;
        else if (a == "--device-lbvh") device_lbvh = true;
        else if (a == "-h" || a == "--help") {
            std::printf("raytracer-rs (MI355X) 0.1.0\nusage: raytracer [-f COLLADA_FILENAME] [-m MAX_TRIS] [-i FRAME_ITERATIONS] [--width W] [--height H]\n"
                        "                 [--spp N] [--seed S] [--gpus N] [--out image.ppm|image.png] [--fix-row-index] [--device-lbvh]\n");
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
        if (device_lbvh) cfg.flags |= MI355RT_FLAG_DEVICE_LBVH;             // BVH built on the GPU (Morton order) instead of the host's SAH build
        cfg.device_count = (uint32_t)gpus;
        if (share_device) cfg.flags |= MI355RT_FLAG_GROUP_SHARES_DEVICE;      // testing: the whole group on one GPU
        if (gpus > 1) std::printf("rendering on %zu GPUs (rows dealt in stripes of %u)\n", gpus, cfg.stripe_rows);
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
            const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0;
            if (png) {
                if (!write_png(out, ldr, width, height)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
            } else {
                FILE* f = std::fopen(out.c_str(), "wb");
                if (!f) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
                std::fprintf(f, "P6\n%zu %zu\n255\n", width, height);
                for (uint32_t p : ldr) { unsigned char rgb[3] = { (unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p }; std::fwrite(rgb, 1, 3, f); }
                std::fclose(f);
            }
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());                   // main() -> Result<(), String>
        return 1;
    }
    return 0;
}
