// bvh_sim — CPU census of the BVH traversal the trace kernels run (tools only; not part of the library).
// Builds the library's own host BVH (csrc/bvh.cpp) for a .scene file, shoots primary rays from the file camera, reflection rays from
// their hit points (uniform hemisphere about the geometric normal, mod.rs:178-196) and counts, per ray kind, where the node visits go:
// visits with no child hit, pops that a stored entry distance would have culled, visits of the first descent, ...
// build: g++ -O2 -std=c++17 -I raytracer-rs_amd/csrc -o /tmp/bvh_sim tools/micro/bvh_sim.cpp raytracer-rs_amd/csrc/bvh.cpp raytracer-rs_amd/csrc/collada.cpp
//        raytracer-rs_amd/csrc/xml_mini.cpp raytracer-rs_amd/csrc/png_decode.cpp -lz
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "bvh.hpp"
#include "camera.hpp"
#include "scene.hpp"
using namespace mi355rt;

static float half_to_float(uint16_t h)
{
    const uint32_t s = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 1024u), (int)e - 25);
    return s ? -v : v;
}
struct Box6 { float mn[3], mx[3]; };
static Box6 child_box(const uint32_t h[3])
{
    Box6 b;
    for (int a = 0; a < 3; ++a) { b.mn[a] = half_to_float((uint16_t)(h[a] & 0xFFFFu)); b.mx[a] = half_to_float((uint16_t)(h[a] >> 16)); }
    return b;
}
struct Stats {
    double rays = 0, visits = 0, tris = 0, none = 0, one = 0, both = 0, pops = 0, pops_culled = 0, first_descent = 0, hits = 0, visits_after_hit = 0, maxsp = 0;
    double culled_q[4] = { 0, 0, 0, 0 };
    void print(const char* name) const
    {
        std::printf("%-10s rays %.0f hit %.3f | visits/ray %.2f (none %.2f one %.2f both %.2f) tris/ray %.2f | first descent %.2f | pops %.2f of which a stored entry distance culls %.2f"
                    " (quantised 8/10/12/16 bits: %.2f %.2f %.2f %.2f) | visits after the final hit %.2f | max sp %.0f\n",
                    name, rays, hits / rays, visits / rays, none / rays, one / rays, both / rays, tris / rays, first_descent / rays, pops / rays, pops_culled / rays,
                    culled_q[0] / rays, culled_q[1] / rays, culled_q[2] / rays, culled_q[3] / rays, visits_after_hit / rays, maxsp);
    }
};
static bool slab(const Box6& b, const float o[3], const float id[3], float tlimit, float& tn)
{
    float t0 = 0.0f, t1 = tlimit;
    for (int a = 0; a < 3; ++a) {
        float x0 = (b.mn[a] - o[a]) * id[a], x1 = (b.mx[a] - o[a]) * id[a];
        if (x0 > x1) std::swap(x0, x1);
        t0 = std::max(t0, x0); t1 = std::min(t1, x1);
    }
    tn = t0;
    return t0 <= t1;
}
// quantise a non-negative entry distance DOWN to `bits` bits (sign dropped, exponent + top mantissa bits kept): what a packed stack entry could carry
static float quant_down(float t, int bits)
{
    uint32_t u; std::memcpy(&u, &t, 4);
    u &= 0x7FFFFFFFu;
    const int drop = 31 - bits;
    u = (u >> drop) << drop;
    float r; std::memcpy(&r, &u, 4);
    return r;
}
struct HitRec { float t = INFINITY, u = 0, v = 0; uint32_t prim = 0xFFFFFFFFu; uint32_t bvh_tri = 0; };
struct Paths { std::vector<int32_t> parent; std::vector<int32_t> leaf_of_tri; std::vector<int32_t> leaf_parent; std::vector<uint8_t> leaf_side; };
static Paths g_paths;
static void build_paths(const Bvh& bvh)
{
    g_paths.parent.assign(bvh.nodes.size(), -1); g_paths.leaf_of_tri.assign(bvh.tris.size(), 0); g_paths.leaf_parent.assign(bvh.tris.size(), -1); g_paths.leaf_side.assign(bvh.tris.size(), 0);
    for (size_t i = 0; i < bvh.nodes.size(); ++i) for (int c = 0; c < 2; ++c) {
        const int32_t l = c ? bvh.nodes[i].child1 : bvh.nodes[i].child0;
        if (l >= 0) g_paths.parent[l] = (int32_t)i;
        else { const uint32_t code = ~(uint32_t)l, first = code >> 3, cnt = (code & 7u) + 1u; for (uint32_t k = 0; k < cnt; ++k) { g_paths.leaf_of_tri[first + k] = l; g_paths.leaf_parent[first + k] = (int32_t)i; g_paths.leaf_side[first + k] = (uint8_t)c; } }
    }
}
static HitRec trace(const Bvh& bvh, const float o[3], const float d[3], float tlimit0, Stats& st, int start_tri = -1, Stats* pst = nullptr);
// path start: the ray's origin lies on BVH triangle start_tri.  Slots = the sibling of every node on the way from the root to its leaf.
static double g_slot_loads = 0, g_path_rays = 0, g_slot_hits = 0;
static HitRec trace(const Bvh& bvh, const float o[3], const float d[3], float tlimit0, Stats& st, int start_tri, Stats* pst)
{
    HitRec h;
    float id[3];
    for (int a = 0; a < 3; ++a) id[a] = 1.0f / (std::fabs(d[a]) < 1e-20f ? std::copysign(1e-20f, d[a]) : d[a]);
    (void)pst;
    struct Entry { int32_t node; float tn; };
    Entry stack[64]; int sp = 0;
    int32_t node = bvh.root;
    float tlimit = tlimit0;
    if (start_tri >= 0) {
        // siblings from the root down: collect the chain bottom-up, push top-down
        int32_t chain_node[64]; uint8_t chain_side[64]; int n = 0;
        int32_t p = g_paths.leaf_parent[start_tri]; uint8_t side = g_paths.leaf_side[start_tri];
        while (p >= 0) { chain_node[n] = p; chain_side[n] = side; ++n; const int32_t pp = g_paths.parent[p]; if (pp >= 0) side = bvh.nodes[pp].child1 == p ? 1 : 0; p = pp; }
        g_path_rays += 1; g_slot_loads += n + 1;
        for (int k = n - 1; k >= 0; --k) {
            const BvhNode& nd = bvh.nodes[(size_t)chain_node[k]];
            float tn; const bool other = !chain_side[k];
            if (slab(child_box(other ? nd.h1 : nd.h0), o, id, tlimit, tn)) { stack[sp++] = Entry{ other ? nd.child1 : nd.child0, tn }; g_slot_hits += 1; }
        }
        node = g_paths.leaf_of_tri[start_tri];
    }
    bool in_first_descent = true;
    double visits_at_last_hit = 0, v0 = st.visits;
    st.rays += 1;
    for (;;) {
        if (node >= 0) {
            st.visits += 1; if (in_first_descent) st.first_descent += 1;
            const BvhNode& n = bvh.nodes[(size_t)node];
            float tn0, tn1;
            const bool h0 = slab(child_box(n.h0), o, id, tlimit, tn0), h1 = slab(child_box(n.h1), o, id, tlimit, tn1);
            if (h0 && h1) {
                st.both += 1;
                const bool sw = tn1 < tn0;
                stack[sp++] = Entry{ sw ? n.child0 : n.child1, sw ? tn0 : tn1 };
                st.maxsp = std::max(st.maxsp, (double)sp);
                node = sw ? n.child1 : n.child0;
                continue;
            }
            if (h0) { st.one += 1; node = n.child0; continue; }
            if (h1) { st.one += 1; node = n.child1; continue; }
            st.none += 1;
        } else {
            in_first_descent = false;
            const uint32_t code = ~(uint32_t)node, first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                st.tris += 1;
                const BvhTri& t = bvh.tris[first + k];
                const float* v0v1 = t.e1; const float* v0v2 = t.e2;
                const float p[3] = { d[1] * v0v2[2] - d[2] * v0v2[1], d[2] * v0v2[0] - d[0] * v0v2[2], d[0] * v0v2[1] - d[1] * v0v2[0] };
                const float det = v0v1[0] * p[0] + v0v1[1] * p[1] + v0v1[2] * p[2];
                if (std::fabs(det) < 1.1920929e-7f) continue;
                const float inv = 1.0f / det;
                const float tv[3] = { o[0] - t.v0[0], o[1] - t.v0[1], o[2] - t.v0[2] };
                const float u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv;
                const float q[3] = { tv[1] * v0v1[2] - tv[2] * v0v1[1], tv[2] * v0v1[0] - tv[0] * v0v1[2], tv[0] * v0v1[1] - tv[1] * v0v1[0] };
                const float v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
                const float tt = (v0v2[0] * q[0] + v0v2[1] * q[1] + v0v2[2] * q[2]) * inv;
                if (u < 0 || u > 1 || v < 0 || u + v > 1 || tt < 0) continue;
                if (tt <= tlimit && (h.prim == 0xFFFFFFFFu || tt < h.t || (tt == h.t && t.prim < h.prim))) {
                    h.t = tt; h.u = u; h.v = v; h.prim = t.prim; h.bvh_tri = first + k; tlimit = tt; visits_at_last_hit = st.visits;
                }
            }
        }
        if (sp == 0) break;
        const Entry e = stack[--sp];
        st.pops += 1;
        if (e.node >= 0) {           // a stored distance can only save the fetch of an INNER node (a leaf's triangles would be skipped as well, counted too)
            if (e.tn > tlimit) st.pops_culled += 1;
            const int qb[4] = { 8, 10, 12, 16 };
            for (int k = 0; k < 4; ++k) if (quant_down(e.tn, qb[k]) > tlimit) st.culled_q[k] += 1;
        }
        node = e.node;
    }
    if (h.prim != 0xFFFFFFFFu) { st.hits += 1; st.visits_after_hit += st.visits - visits_at_last_hit; }
    (void)v0;
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s file.scene [width height spp]\n", argv[0]); return 2; }
    SceneData s; std::string err;
    if (!read_scene_file(argv[1], s, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const uint32_t W = argc > 2 ? atoi(argv[2]) : 480, H = argc > 3 ? atoi(argv[3]) : 270, SPP = argc > 4 ? atoi(argv[4]) : 1;
    Bvh bvh; build_bvh(s.tri_verts.data(), s.tri_geom.data(), s.ntri(), bvh);
    std::printf("%u triangles, %zu nodes, depth %u\n", s.ntri(), bvh.nodes.size(), bvh.max_depth);
    Matrix m; std::memcpy(m.e, s.cameras[0].orientation, 64);
    Camera cam = Camera::from_orientation_matrix(W, H, m, s.cameras[0].fov_deg);
    std::mt19937 rng(1); std::uniform_real_distribution<float> U(0.0f, 1.0f), S(-1.0f, 1.0f);
    Stats prim, refl1, refl2, shadow, p_refl1, p_refl2, p_shadow, c_refl1, c_refl2;
    build_paths(bvh);
    // upper bound of what an "escape" table could buy (profiles/r03_notes.md): a reflection ray that misses everything is traced again with its search cut
    // at the point where it leaves the 3 x 3 x 3 block of grid cells around its origin (a table that KNEW it escapes would allow exactly that)
    const int GRID = getenv("SIM_GRID") ? atoi(getenv("SIM_GRID")) : 32;
    auto block_exit = [&](const float o[3], const float d[3]) {
        float t_exit = INFINITY;
        for (int a = 0; a < 3; ++a) {
            const float lo = bvh.scene_min[a], hi = bvh.scene_max[a], cs = (hi - lo) / GRID;
            int c = (int)std::floor((o[a] - lo) / cs); c = std::max(0, std::min(GRID - 1, c));
            const float b0 = lo + (c - 1) * cs, b1 = lo + (c + 2) * cs;
            if (d[a] > 0) t_exit = std::min(t_exit, (b1 - o[a]) / d[a]);
            else if (d[a] < 0) t_exit = std::min(t_exit, (b0 - o[a]) / d[a]);
        }
        return t_exit;
    };
    auto reflect = [&](const float o[3], const float d[3], const HitRec& h, float ro[3], float rd[3]) {
        const BvhTri& t = bvh.tris[h.bvh_tri];
        float n[3] = { t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1], t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2], t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0] };
        for (;;) {
            float x = S(rng), y = S(rng), z = S(rng); const float l2 = x * x + y * y + z * z;
            if (!(l2 < 1.0f) || l2 == 0.0f) continue;
            const float l = std::sqrt(l2); x /= l; y /= l; z /= l;
            if (x * n[0] + y * n[1] + z * n[2] <= 0.0f) continue;
            rd[0] = x; rd[1] = y; rd[2] = z; break;
        }
        for (int a = 0; a < 3; ++a) ro[a] = (o[a] + d[a] * h.t) + 1e-5f * rd[a];
    };
    auto shadow_ray = [&](const float o[3], const float d[3], const HitRec& h) {
        float hp[3], l[3], so[3];
        for (int a = 0; a < 3; ++a) { hp[a] = o[a] + d[a] * h.t; l[a] = s.lights[0].pos[a] - hp[a]; so[a] = hp[a] + 0.01f * l[a]; }
        const BvhTri& t = bvh.tris[h.bvh_tri];
        float n[3] = { t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1], t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2], t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0] };
        if (n[0] * l[0] + n[1] * l[1] + n[2] * l[2] < 0.0f) return;
        HitRec a = trace(bvh, so, l, 0x1.fffffep-1f, shadow);
        HitRec b = trace(bvh, so, l, 0x1.fffffep-1f, p_shadow, (int)h.bvh_tri);
        if (a.prim != b.prim || a.t != b.t) std::printf("MISMATCH shadow\n");
    };
    for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) for (uint32_t k = 0; k < SPP; ++k) {
        Ray r = cam.get_ray(x, y * W / H /* the reference's idx / height quirk stretches v over 0..w */, U(rng), U(rng));
        const float o[3] = { r.pos.x, r.pos.y, r.pos.z }, d[3] = { r.dir.x, r.dir.y, r.dir.z };
        HitRec h = trace(bvh, o, d, INFINITY, prim);
        if (h.prim == 0xFFFFFFFFu) continue;
        shadow_ray(o, d, h);
        for (int c = 0; c < 2; ++c) {
            float ro[3], rd[3]; reflect(o, d, h, ro, rd);
            HitRec h1 = trace(bvh, ro, rd, INFINITY, refl1);
            trace(bvh, ro, rd, h1.prim == 0xFFFFFFFFu ? block_exit(ro, rd) : INFINITY, c_refl1);
            { HitRec b = trace(bvh, ro, rd, INFINITY, p_refl1, (int)h.bvh_tri); if (b.prim != h1.prim || b.t != h1.t) std::printf("MISMATCH r1\n"); }
            if (h1.prim == 0xFFFFFFFFu) continue;
            shadow_ray(ro, rd, h1);
            float ro2[3], rd2[3]; reflect(ro, rd, h1, ro2, rd2);
            HitRec h2 = trace(bvh, ro2, rd2, INFINITY, refl2);
            trace(bvh, ro2, rd2, h2.prim == 0xFFFFFFFFu ? block_exit(ro2, rd2) : INFINITY, c_refl2);
            { HitRec b = trace(bvh, ro2, rd2, INFINITY, p_refl2, (int)h1.bvh_tri); if (b.prim != h2.prim || b.t != h2.t) std::printf("MISMATCH r2\n"); }
            if (h2.prim != 0xFFFFFFFFu) shadow_ray(ro2, rd2, h2);
        }
    }
    prim.print("primary"); refl1.print("reflect 1"); refl2.print("reflect 2"); shadow.print("shadow");
    c_refl1.print("C reflect1"); c_refl2.print("C reflect2");
    p_refl1.print("P reflect1"); p_refl2.print("P reflect2"); p_shadow.print("P shadow");
    std::printf("path rays %.0f: slot loads per ray %.2f, of which hit (pushed) %.2f\n", g_path_rays, g_slot_loads / g_path_rays, g_slot_hits / g_path_rays);
    auto loads = [](const Stats& a) { return (2.0 * a.visits + 3.0 * a.tris) / a.rays; };
    std::printf("16-byte loads per ray: reflect1 %.1f -> %.1f + slots, reflect2 %.1f -> %.1f + slots, shadow %.1f -> %.1f + slots\n", loads(refl1), loads(p_refl1), loads(refl2), loads(p_refl2), loads(shadow), loads(p_shadow));
    return 0;
}
