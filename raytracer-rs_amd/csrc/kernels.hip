// kernels.hip — gfx950 (CDNA4, wave64) kernels of the render path.
//
// Wavefront structure of one pass over N primary samples:
//   round 0      : ray-gen + closest hit of the primary rays                       (mod.rs:93-98)
//   round r >= 1 : closest hit of level-r reflection rays and of the shadow rays
//                  emitted by level r-1                                             (mod.rs:158, 226)
//   every round  : on a radiance hit, shade() set-up — normal, Phong terms, shadow ray
//                  (mod.rs:198-257) — and the level r+1 reflection rays (mod.rs:178-196) are
//                  appended to the next queue with a wave64 ballot + prefix-popcount and ONE
//                  atomic per wave; a finished unblocked shadow ray stores its light term.
//   resolve      : per pixel, combine the per-node light terms in the reference's own
//                  summation order (mod.rs:154-175) and add the samples to the film in sample
//                  order (film.rs:20-24).
//
// Numerics: this file is compiled with -ffp-contract=off; everything that decides a result
// (Moller-Trumbore, shading, camera, film, tonemap) is written in the reference's operation
// order with IEEE f32 +,-,*,/ and sqrt, so results are bit-comparable with unfused CPU code.
// Only the BVH box tests are free-form: boxes are padded (bvh.cpp) and the test is conservative.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_types.hpp"
#include "kernels.hpp"

namespace mi355rt {

constexpr int kBlock = 256;
constexpr int kStackDepth = 32;          // >= kBvhMaxDepth + 1

struct f3 { float x, y, z; };
// IEEE correctly rounded f32 divide / sqrt: plain `/` and sqrtf under
// -fhip-fp32-correctly-rounded-divide-sqrt (HIP's __fdiv_rn/__fsqrt_rn are NOT: __fsqrt_rn is the
// native approximation).  tests/test_gpu_numerics.py checks both against IEEE on the device.
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn(float a) { return __builtin_sqrtf(a); }
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 vscale(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }    // Vec3 * f32
__device__ __forceinline__ f3 sscale(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }    // f32 * Vec3
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vecmath.rs:74-76
__device__ __forceinline__ f3 cross3(f3 a, f3 b)                                                  // vecmath.rs:79-85
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalized3(f3 a)                                                   // vecmath.rs:23-26
{
    float len = sqrt_rn(a.x * a.x + a.y * a.y + a.z * a.z);
    return mk3(div_rn(a.x, len), div_rn(a.y, len), div_rn(a.z, len));
}

// counter RNG (replaces the reference's OS-entropy StdRng): pcg4d, Jarzynski & Olano 2020
__device__ __forceinline__ void pcg4d(uint32_t& x, uint32_t& y, uint32_t& z, uint32_t& w)
{
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
}
__device__ __forceinline__ float u01(uint32_t bits) { return (float)(bits >> 9) * (1.0f / 8388608.0f); }

// x.powf(32.0), mod.rs:255: five squarings in f64, rounded once
__device__ __forceinline__ float pow32(float x)
{
    double d = (double)x;
    d = d * d; d = d * d; d = d * d; d = d * d; d = d * d;
    return (float)d;
}

struct Hit {
    float t, u, v;
    uint32_t prim;      // 0xFFFFFFFF = miss
    int occ;            // shadow rays: 0 nothing in [0,1), 1 blocked (closest hit in (0.01,1)), 2 a hit at t <= 0.01
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Closest-hit traversal of the BVH2.  Radiance rays: true closest hit (lowest t, ties to the
// lowest triangle index == first in the reference's list order, no_acceleration_intersector.rs).
// Shadow rays: only the predicate of mod.rs:226-229 is needed — "the CLOSEST hit has
// 0.01 < t < 1.0" — so the search interval shrinks to [0, 0.01] after the first hit inside
// (0.01, 1) and stops at the first hit with t <= 0.01.
template <bool COUNT>
__device__ __forceinline__ void traverse(const DScene& sc, f3 o, f3 d, bool shadow, int* stack, Hit& r,
                                          uint32_t& n_nodes, uint32_t& n_tris)
{
    const float4* __restrict__ nodes = (const float4*)sc.nodes;
    const float4* __restrict__ tris = (const float4*)sc.tris;
    const float idx = __builtin_amdgcn_rcpf(d.x), idy = __builtin_amdgcn_rcpf(d.y), idz = __builtin_amdgcn_rcpf(d.z);
    float tlimit = shadow ? 0x1.fffffep-1f : __builtin_inff();     // shadow: t < 1.0
    r.t = __builtin_inff(); r.u = 0.0f; r.v = 0.0f; r.prim = 0xFFFFFFFFu; r.occ = 0;
    int node = sc.root;
    int sp = 0;
    for (;;) {
        if (node >= 0) {
            if (COUNT) ++n_nodes;
            const float4 q0 = nodes[4 * node], q1 = nodes[4 * node + 1], q2 = nodes[4 * node + 2];
            const float4 q3 = nodes[4 * node + 3];
            float a1 = (q0.x - o.x) * idx, a2 = (q0.y - o.x) * idx;
            float b1 = (q0.z - o.y) * idy, b2 = (q0.w - o.y) * idy;
            float c1 = (q2.x - o.z) * idz, c2 = (q2.y - o.z) * idz;
            float tn0 = fmaxf(fmaxf(fminf(a1, a2), fminf(b1, b2)), fmaxf(fminf(c1, c2), 0.0f));
            float tf0 = fminf(fminf(fmaxf(a1, a2), fmaxf(b1, b2)), fminf(fmaxf(c1, c2), tlimit));
            a1 = (q1.x - o.x) * idx; a2 = (q1.y - o.x) * idx;
            b1 = (q1.z - o.y) * idy; b2 = (q1.w - o.y) * idy;
            c1 = (q2.z - o.z) * idz; c2 = (q2.w - o.z) * idz;
            float tn1 = fmaxf(fmaxf(fminf(a1, a2), fminf(b1, b2)), fmaxf(fminf(c1, c2), 0.0f));
            float tf1 = fminf(fminf(fmaxf(a1, a2), fmaxf(b1, b2)), fminf(fmaxf(c1, c2), tlimit));
            const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
            const int c0i = __float_as_int(q3.x), c1i = __float_as_int(q3.y);
            if (h0 && h1) {
                const bool sw = tn1 < tn0;
                stack[sp * kBlock] = sw ? c0i : c1i;
                ++sp;
                node = sw ? c1i : c0i;
                continue;
            }
            if (h0) { node = c0i; continue; }
            if (h1) { node = c1i; continue; }
        } else {
            const uint32_t code = ~(uint32_t)node;
            const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
            bool done = false;
            for (uint32_t i = 0; i < cnt; ++i) {
                if (COUNT) ++n_tris;
                const float4 t0 = tris[3 * (first + i)], t1 = tris[3 * (first + i) + 1], t2 = tris[3 * (first + i) + 2];
                // Moller-Trumbore "late out", intersect.rs:62-98, same operation order
                const f3 v0 = mk3(t0.x, t0.y, t0.z), v0v1 = mk3(t1.x, t1.y, t1.z), v0v2 = mk3(t2.x, t2.y, t2.z);
                const f3 pvec = cross3(d, v0v2);
                const float det = dot3(v0v1, pvec);
                if (fabsf(det) < 1.1920929e-7f) continue;
                const float inv_det = div_rn(1.0f, det);
                const f3 tvec = sub3(o, v0);
                const float u = dot3(tvec, pvec) * inv_det;
                const f3 qvec = cross3(tvec, v0v1);
                const float v = dot3(d, qvec) * inv_det;
                const float t = dot3(v0v2, qvec) * inv_det;
                if (u < 0.0f || u > 1.0f) continue;
                if (v < 0.0f || u + v > 1.0f) continue;
                if (t < 0.0f) continue;
                const uint32_t prim = __float_as_uint(t0.w);
                if (!shadow) {
                    if (r.prim == 0xFFFFFFFFu || t < r.t || (t == r.t && prim < r.prim)) {
                        r.t = t; r.u = u; r.v = v; r.prim = prim; tlimit = t;
                    }
                } else if (t <= tlimit) {
                    if (t > 0.01f) { r.occ = 1; tlimit = 0.01f; }
                    else { r.occ = 2; done = true; break; }
                }
            }
            if (done) break;
        }
        if (sp == 0) break;
        --sp;
        node = stack[sp * kBlock];
    }
}

// wave64 compaction: every lane of the wave calls this; lanes with want == true get consecutive
// indices of the output queue; one atomic per wave.
__device__ __forceinline__ uint32_t wave_append(bool want, uint32_t* counter, uint32_t& n_out)
{
    const unsigned long long mask = __ballot(want);
    n_out = (uint32_t)__popcll(mask);
    if (mask == 0ull) return 0u;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0u;
    if (lane_id() == leader) base = atomicAdd(counter, n_out);
    base = (uint32_t)__shfl((int)base, leader, 64);
    return base + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
}

__device__ __forceinline__ f3 fetch_texel(const DScene& sc, uint32_t tex, float u, float v)     // texture.rs:21-27
{
    const DTexture t = sc.textures[tex];
    // `as usize`: truncation, saturating, NaN -> 0.  An index past the end panics in the reference;
    // here it is clamped to the last texel.
    const float fx = u * (float)t.width, fy = v * (float)t.height;
    unsigned long long x = fx > 0.0f ? (fx >= 1.8446744e19f ? ~0ull : (unsigned long long)fx) : 0ull;
    unsigned long long y = fy > 0.0f ? (fy >= 1.8446744e19f ? ~0ull : (unsigned long long)fy) : 0ull;
    const unsigned long long n = (unsigned long long)t.width * t.height;
    unsigned long long i = (y > n ? n : y) * t.width + (x > n ? n : x);
    if (i >= n) i = n - 1;
    const float* p = sc.texels + 3ull * (t.offset + i);
    return mk3(p[0], p[1], p[2]);
}

template <bool PRIMARY, bool COUNT>
__global__ __launch_bounds__(kBlock) void trace_round_kernel(DScene sc, DCamera cam, DPass ps, uint32_t level,
                                                            const float4* __restrict__ in_q, const uint32_t* __restrict__ in_count,
                                                            float4* __restrict__ out_q, uint32_t* out_count, uint32_t* cursor,
                                                            float* __restrict__ slot_L, const uint32_t* __restrict__ film_n,
                                                            DCounters* counters)
{
    __shared__ int s_stack[kStackDepth * kBlock];
    int* stack = &s_stack[threadIdx.x];
    const uint32_t total = PRIMARY ? ps.nsamples : *in_count;
    const int lane = lane_id();
    uint32_t acc_nodes = 0, acc_tris = 0, acc_bounce = 0, acc_shadow = 0, acc_phits = 0;

    for (;;) {
        uint32_t base = 0u;
        if (lane == 0) base = atomicAdd(cursor, 64u);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= total) break;
        const uint32_t i = base + (uint32_t)lane;
        const bool active = i < total;

        f3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(0.0f, 0.0f, 1.0f);
        uint32_t slot = 0u, meta = 0u, pixel = 0u, sampleno = 0u;
        f3 L = mk3(0.0f, 0.0f, 0.0f);
        if (active) {
            if (PRIMARY) {
                // pixel -> ray, mod.rs:93-96 + camera.rs:80-90
                const uint32_t s = i / ps.npix, p = i - s * ps.npix;
                if (ps.use_explicit) pixel = ps.explicit_pixel;
                else pixel = ps.rows[ps.row0 + p / cam.width] * cam.width + p % cam.width;
                sampleno = ps.use_explicit ? ps.explicit_sampleno : film_n[pixel] + s;
                slot = i;
                uint32_t h0 = pixel, h1 = sampleno, h2 = 0u, h3 = ps.seed;
                pcg4d(h0, h1, h2, h3);
                const uint32_t cu = pixel % cam.width;
                const uint32_t cv = (ps.flags & 1u) ? pixel / cam.width : pixel / cam.height;   // reference: idx / height
                const float dir_x = -cam.max_x + 2.0f * cam.max_x * div_rn((float)cu + u01(h0), (float)cam.width);
                const float dir_y = -cam.max_y + 2.0f * cam.max_y * div_rn((float)cv + u01(h1), (float)cam.height);
                const float vx = dir_x, vy = -dir_y, vz = 1.0f, vw = 1.0f;
                d.x = vx * cam.rot[0] + vy * cam.rot[4] + vz * cam.rot[8] + vw * cam.rot[12];
                d.y = vx * cam.rot[1] + vy * cam.rot[5] + vz * cam.rot[9] + vw * cam.rot[13];
                d.z = vx * cam.rot[2] + vy * cam.rot[6] + vz * cam.rot[10] + vw * cam.rot[14];
                o = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
                meta = 0u;
            } else {
                const float4 r0 = in_q[3 * (size_t)i], r1 = in_q[3 * (size_t)i + 1], r2 = in_q[3 * (size_t)i + 2];
                o = mk3(r0.x, r0.y, r0.z); d = mk3(r0.w, r1.x, r1.y);
                slot = __float_as_uint(r1.z); meta = __float_as_uint(r1.w);
                if (meta & 1u) L = mk3(r2.x, r2.y, r2.z);
                else { pixel = __float_as_uint(r2.x); sampleno = __float_as_uint(r2.y); }
            }
        }
        const bool shadow = (meta & 1u) != 0u;
        Hit hit;
        hit.prim = 0xFFFFFFFFu; hit.occ = 0; hit.t = 0.0f; hit.u = 0.0f; hit.v = 0.0f;
        if (active) traverse<COUNT>(sc, o, d, shadow, stack, hit, acc_nodes, acc_tris);

        const uint32_t node = (meta >> 8) & 0xFFFFu;
        if (active && shadow) {
            if (hit.occ != 1) {                                    // not blocked, mod.rs:232
                const uint32_t light = meta >> 24;
                float* dst = slot_L + 3ull * (((size_t)slot * ps.nodes_per_sample + node) * sc.nlights + light);
                dst[0] = L.x; dst[1] = L.y; dst[2] = L.z;
            }
        }
        const bool rad_hit = active && !shadow && hit.prim != 0xFFFFFFFFu;
        if (PRIMARY) acc_phits += rad_hit ? 1u : 0u;
        if (__ballot(rad_hit) == 0ull) continue;

        // ---- shade() set-up, mod.rs:207-257 (every lane walks the loops; only hit lanes emit)
        f3 hp = mk3(0, 0, 0), n = mk3(0, 0, 1);
        uint32_t geom = 0u;
        if (rad_hit) {
            hp = add3(o, sscale(hit.t, d));                                          // mod.rs:212
            const float4 nn = ((const float4*)sc.normals)[hit.prim];                 // calc_normal, mod.rs:198-205 (precomputed)
            n = mk3(nn.x, nn.y, nn.z);
            geom = __float_as_uint(nn.w);
        }
        for (uint32_t li = 0; li < sc.nlights; ++li) {
            bool want = false;
            f3 so = mk3(0, 0, 0), sd = mk3(0, 0, 1), c = mk3(0, 0, 0);
            if (rad_hit) {
                const DLight lt = sc.lights[li];
                const f3 l = sub3(mk3(lt.px, lt.py, lt.pz), hp);                       // mod.rs:215
                const f3 ln = normalized3(l);
                const float ndl = dot3(n, ln);                                         // mod.rs:216
                if (!(ndl < 0.0f)) {                                                   // mod.rs:218
                    want = true;
                    so = add3(hp, vscale(l, 0.01f)); sd = l;                           // mod.rs:224-225
                    const DMaterial m = sc.materials[geom];
                    f3 diffuse = mk3(m.r, m.g, m.b);
                    if (m.kind_tex & 0x80000000u) diffuse = fetch_texel(sc, m.kind_tex & 0x7FFFFFFFu, hit.u, hit.v);
                    const f3 view = normalized3(d);                                    // mod.rs:251
                    const f3 refl = sub3(sscale(2.0f * ndl, n), ln);                   // mod.rs:252-253
                    const float spec = pow32(dot3(view, refl));                        // mod.rs:255, SPECULAR = white
                    c = mk3((diffuse.x * ndl + spec) * lt.cr, (diffuse.y * ndl + spec) * lt.cg, (diffuse.z * ndl + spec) * lt.cb);
                }
            }
            uint32_t n_out;
            const uint32_t oi = wave_append(want, out_count, n_out);
            if (lane == 0) acc_shadow += n_out;
            if (want) {
                if (oi < ps.out_capacity) {
                    out_q[3 * (size_t)oi] = make_float4(so.x, so.y, so.z, sd.x);
                    out_q[3 * (size_t)oi + 1] = make_float4(sd.y, sd.z, __uint_as_float(slot), __uint_as_float(1u | (level << 4) | (node << 8) | (li << 24)));
                    out_q[3 * (size_t)oi + 2] = make_float4(c.x, c.y, c.z, 0.0f);
                } else counters->overflow = 1u;
            }
        }
        // ---- reflection rays, mod.rs:146-158 + 178-196
        if (level < ps.recursions) {
            const uint32_t k = ps.spread * (ps.recursions - level);                    // num_sub_rays, mod.rs:150
            const uint32_t index_in_level = node - ps.level_first[level];
            for (uint32_t ci = 0; ci < k; ++ci) {
                const uint32_t child_node = ps.level_first[level + 1] + index_in_level * k + ci;
                f3 bo = mk3(0, 0, 0), bd = mk3(0, 0, 1);
                if (rad_hit) {
                    uint32_t h0 = pixel, h1 = sampleno, h2 = 1u + child_node, h3 = ps.seed;
                    pcg4d(h0, h1, h2, h3);
                    uint32_t j = __umulhi(h0, 65535u);                                 // uniform in [0, 65534], sample_generator.rs:32
                    const float4* __restrict__ table = (const float4*)sc.table;
                    float4 tv = table[j];
                    uint32_t guard = 0u;
                    while (tv.x * n.x + tv.y * n.y + tv.z * n.z <= 0.0f && guard < kNumSamples) {   // mod.rs:187-189
                        j = (j + 1u) % kSampleMax;                                     // sample_generator.rs:27
                        tv = table[j];
                        ++guard;
                    }
                    bd = mk3(tv.x, tv.y, tv.z);
                    bo = add3(hp, sscale(0.00001f, bd));                               // mod.rs:192-193
                }
                uint32_t n_out;
                const uint32_t oi = wave_append(rad_hit, out_count, n_out);
                if (lane == 0) acc_bounce += n_out;
                if (rad_hit) {
                    if (oi < ps.out_capacity) {
                        out_q[3 * (size_t)oi] = make_float4(bo.x, bo.y, bo.z, bd.x);
                        out_q[3 * (size_t)oi + 1] = make_float4(bd.y, bd.z, __uint_as_float(slot), __uint_as_float(((level + 1u) << 4) | (child_node << 8)));
                        out_q[3 * (size_t)oi + 2] = make_float4(__uint_as_float(pixel), __uint_as_float(sampleno), 0.0f, 0.0f);
                    } else counters->overflow = 1u;
                }
            }
        }
    }
    // per-wave counter flush
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { acc_nodes += __shfl_down((int)acc_nodes, off, 64); acc_tris += __shfl_down((int)acc_tris, off, 64); }
        if (lane == 0) { atomicAdd(&counters->nodes_visited, (unsigned long long)acc_nodes); atomicAdd(&counters->tris_tested, (unsigned long long)acc_tris); }
    }
    if (PRIMARY) {
        for (int off = 32; off > 0; off >>= 1) acc_phits += __shfl_down((int)acc_phits, off, 64);
        if (lane == 0 && acc_phits) atomicAdd(&counters->primary_hits, (unsigned long long)acc_phits);
    }
    if (lane == 0) {
        if (acc_bounce) atomicAdd(&counters->bounce, (unsigned long long)acc_bounce);
        if (acc_shadow) atomicAdd(&counters->shadow, (unsigned long long)acc_shadow);
    }
}

// ---- resolve: radiance tree -> sample colour -> film ------------------------------------------
template <int REC>
__device__ f3 node_radiance(const float* __restrict__ L, const DPass& ps, uint32_t nlights, uint32_t level, uint32_t index)
{
    const uint32_t node = ps.level_first[level] + index;
    f3 rad = mk3(0.0f, 0.0f, 0.0f);                                    // accum_color, mod.rs:211
    for (uint32_t li = 0; li < nlights; ++li) {
        const float* p = L + 3ull * ((size_t)node * nlights + li);
        rad = add3(rad, mk3(p[0], p[1], p[2]));                         // mod.rs:254
    }
    if constexpr (REC == 0) {
        return rad;                                                     // mod.rs:146-148
    } else {
        const uint32_t k = ps.spread * (uint32_t)REC;                   // mod.rs:150
        f3 sum = mk3(0.0f, 0.0f, 0.0f);
        for (uint32_t c = 0; c < k; ++c) sum = add3(sum, node_radiance<REC - 1>(L, ps, nlights, level + 1, index * k + c));   // fold, mod.rs:173
        const f3 sub = vscale(sum, div_rn(1.0f, (float)k));          // mod.rs:174
        return add3(rad, sub);                                          // mod.rs:175
    }
}

__global__ __launch_bounds__(256) void resolve_kernel(DPass ps, uint32_t width, uint32_t nlights, const float* __restrict__ slot_L,
                                                     float* film_sum, float* film_sumsq, uint32_t* film_n, float* debug_color)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ps.npix) return;
    const uint32_t spp = ps.nsamples / ps.npix;
    const uint32_t pixel = ps.use_explicit ? ps.explicit_pixel : ps.rows[ps.row0 + p / width] * width + p % width;
    f3 sum = mk3(0, 0, 0), sumsq = mk3(0, 0, 0);
    uint32_t n = 0;
    if (!ps.use_explicit) {
        sum = mk3(film_sum[3ull * pixel], film_sum[3ull * pixel + 1], film_sum[3ull * pixel + 2]);
        sumsq = mk3(film_sumsq[3ull * pixel], film_sumsq[3ull * pixel + 1], film_sumsq[3ull * pixel + 2]);
        n = film_n[pixel];
    }
    for (uint32_t s = 0; s < spp; ++s) {
        const float* L = slot_L + 3ull * ((size_t)(s * ps.npix + p) * ps.nodes_per_sample * nlights);
        f3 c;
        switch (ps.recursions) {
            case 0: c = node_radiance<0>(L, ps, nlights, 0, 0); break;
            case 1: c = node_radiance<1>(L, ps, nlights, 0, 0); break;
            case 2: c = node_radiance<2>(L, ps, nlights, 0, 0); break;
            default: c = node_radiance<3>(L, ps, nlights, 0, 0); break;
        }
        // PixelData::add_sample, film.rs:20-24
        sum = add3(sum, c);
        sumsq = add3(sumsq, mk3(c.x * c.x, c.y * c.y, c.z * c.z));
        n += 1u;
        if (ps.use_explicit) { debug_color[0] = c.x; debug_color[1] = c.y; debug_color[2] = c.z; }
    }
    if (!ps.use_explicit) {
        film_sum[3ull * pixel] = sum.x; film_sum[3ull * pixel + 1] = sum.y; film_sum[3ull * pixel + 2] = sum.z;
        film_sumsq[3ull * pixel] = sumsq.x; film_sumsq[3ull * pixel + 1] = sumsq.y; film_sumsq[3ull * pixel + 2] = sumsq.z;
        film_n[pixel] = n;
    }
}

// ---- get_tonemapped_pixels, mod.rs:120-128 = film.rs:43-47 + tonemap.rs:4-10 + color.rs:85-95 --
__device__ __forceinline__ uint32_t to_u8(float c)
{
    // Rust f32::min/max return the non-NaN operand (so NaN -> 1.0 -> 255); `as u8` truncates
    const float m = fmaxf(fminf(c, 1.0f), 0.0f) * 255.0f;
    return (uint32_t)m & 0xFFu;
}
__global__ __launch_bounds__(256) void tonemap_kernel(const uint32_t* __restrict__ rows, uint32_t nrows, uint32_t width, int packed,
                                                     const float* __restrict__ film_sum, const uint32_t* __restrict__ film_n, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nrows * width) return;
    const uint32_t r = (uint32_t)(i / width), x = (uint32_t)(i % width);
    const size_t pixel = (size_t)rows[r] * width + x;
    const float inv = div_rn(1.0f, (float)film_n[pixel]);                     // film.rs:46
    const float cr = film_sum[3 * pixel] * inv, cg = film_sum[3 * pixel + 1] * inv, cb = film_sum[3 * pixel + 2] * inv;
    const uint32_t R = to_u8(div_rn(cr, 1.0f + cr)), G = to_u8(div_rn(cg, 1.0f + cg)), B = to_u8(div_rn(cb, 1.0f + cb));
    out[packed ? i : pixel] = B | (G << 8) | (R << 16) | (255u << 24);           // color.rs:94
}

// ---- batched Intersector seam (accel_intersect.rs:10-13) --------------------------------------
__global__ __launch_bounds__(kBlock) void intersect_kernel(DScene sc, const float* __restrict__ rays6, uint32_t n, int shadow_mode,
                                                          float* tuv, uint32_t* prim, uint8_t* blocked)
{
    __shared__ int s_stack[kStackDepth * kBlock];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 o = mk3(rays6[6ull * i], rays6[6ull * i + 1], rays6[6ull * i + 2]);
    const f3 d = mk3(rays6[6ull * i + 3], rays6[6ull * i + 4], rays6[6ull * i + 5]);
    Hit h;
    uint32_t a = 0, b = 0;
    traverse<false>(sc, o, d, shadow_mode != 0, &s_stack[threadIdx.x], h, a, b);
    if (shadow_mode) blocked[i] = h.occ == 1 ? 1 : 0;
    else {
        prim[i] = h.prim;
        if (h.prim != 0xFFFFFFFFu) { tuv[3ull * i] = h.t; tuv[3ull * i + 1] = h.u; tuv[3ull * i + 2] = h.v; }
    }
}

// ---- device arithmetic self-check: a/b, sqrt(a), a^32 as the kernels compute them ----------------
__global__ void numerics_kernel(const float* __restrict__ a, const float* __restrict__ b, uint32_t n, float* q, float* r, float* p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q[i] = div_rn(a[i], b[i]);
    r[i] = sqrt_rn(a[i]);
    p[i] = pow32(a[i]);
}
hipError_t launch_numerics(hipStream_t stream, const float* a, const float* b, uint32_t n, float* q, float* r, float* p)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(numerics_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, n, q, r, p);
    return hipGetLastError();
}

// ---- launchers --------------------------------------------------------------------------------
static int g_trace_blocks_per_cu[4] = { 0, 0, 0, 0 };

template <bool P, bool C>
static int blocks_per_cu()
{
    int& cache = g_trace_blocks_per_cu[(P ? 2 : 0) + (C ? 1 : 0)];
    if (cache == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_round_kernel<P, C>, kBlock, 0) != hipSuccess || nb < 1) nb = 1;
        cache = nb;
    }
    return cache;
}

hipError_t launch_trace_round(hipStream_t stream, int num_cus, bool primary, bool count, const DScene& sc, const DCamera& cam, const DPass& ps,
                              uint32_t level, const void* in_q, const uint32_t* in_count, void* out_q, uint32_t* out_count, uint32_t* cursor,
                              float* slot_L, const uint32_t* film_n, DCounters* counters)
{
    // persistent grid: as many blocks as the chip holds; waves pull 64-ray chunks from `cursor`
    int per_cu = primary ? (count ? blocks_per_cu<true, true>() : blocks_per_cu<true, false>())
                         : (count ? blocks_per_cu<false, true>() : blocks_per_cu<false, false>());
    dim3 grid((unsigned)(num_cus * per_cu)), block(kBlock);
    const float4* iq = (const float4*)in_q; float4* oq = (float4*)out_q;
    if (primary) {
        if (count) hipLaunchKernelGGL((trace_round_kernel<true, true>), grid, block, 0, stream, sc, cam, ps, level, iq, in_count, oq, out_count, cursor, slot_L, film_n, counters);
        else hipLaunchKernelGGL((trace_round_kernel<true, false>), grid, block, 0, stream, sc, cam, ps, level, iq, in_count, oq, out_count, cursor, slot_L, film_n, counters);
    } else {
        if (count) hipLaunchKernelGGL((trace_round_kernel<false, true>), grid, block, 0, stream, sc, cam, ps, level, iq, in_count, oq, out_count, cursor, slot_L, film_n, counters);
        else hipLaunchKernelGGL((trace_round_kernel<false, false>), grid, block, 0, stream, sc, cam, ps, level, iq, in_count, oq, out_count, cursor, slot_L, film_n, counters);
    }
    return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t stream, const DPass& ps, uint32_t width, uint32_t nlights, const float* slot_L,
                          float* film_sum, float* film_sumsq, uint32_t* film_n, float* debug_color)
{
    dim3 block(256), grid((ps.npix + 255) / 256);
    hipLaunchKernelGGL(resolve_kernel, grid, block, 0, stream, ps, width, nlights, slot_L, film_sum, film_sumsq, film_n, debug_color);
    return hipGetLastError();
}

hipError_t launch_tonemap(hipStream_t stream, const uint32_t* rows, uint32_t nrows, uint32_t width, bool packed,
                          const float* film_sum, const uint32_t* film_n, uint32_t* out)
{
    const size_t n = (size_t)nrows * width;
    if (n == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(tonemap_kernel, grid, block, 0, stream, rows, nrows, width, packed ? 1 : 0, film_sum, film_n, out);
    return hipGetLastError();
}

hipError_t launch_intersect(hipStream_t stream, const DScene& sc, const float* rays6, uint32_t n, bool shadow_mode,
                            float* tuv, uint32_t* prim, uint8_t* blocked)
{
    if (n == 0) return hipSuccess;
    dim3 block(kBlock), grid((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(intersect_kernel, grid, block, 0, stream, sc, rays6, n, shadow_mode ? 1 : 0, tuv, prim, blocked);
    return hipGetLastError();
}

}  // namespace mi355rt
