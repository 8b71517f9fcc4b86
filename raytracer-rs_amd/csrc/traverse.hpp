// traverse.hpp — per-lane BVH2 traversal state machine (included by kernels.hip only).
//
// Result semantics (what must match the reference):
//   radiance rays: the TRUE closest hit — lowest t, ties to the lowest triangle index, i.e. the
//     first in the reference's list order (no_acceleration_intersector.rs:13-41); the triangle test
//     is Moller-Trumbore "late out" (intersect.rs:62-98) in the reference's operation order.
//   shadow rays: only the predicate of mod.rs:226-229 is needed — "the CLOSEST hit has
//     0.01 < t < 1.0".  Equivalent search: look in [0, 1); a hit in (0.01, 1) marks the ray blocked
//     and shrinks the interval to [0, 0.01]; a hit with t <= 0.01 un-blocks it and ends the search.
// The box tests are conservative (boxes are padded in bvh.cpp), so they never change a result.
#pragma once
#include "device_math.hpp"
#include "device_types.hpp"

#ifndef MI355RT_WIDE
#define MI355RT_WIDE 0                       // 1: the kernels walk the 4-wide tree of 48-byte nodes (bvh.hpp, BvhNode4) instead of the binary one
#endif

namespace mi355rt {

struct RayState {
    f3 o, d;
    float idx, idy, idz;   // 1/d' (approximate reciprocal; d' = d with tiny components pushed away from 0: box tests only)
    float oidx, oidy, oidz; // -o/d'
    uint32_t selx, sely, selz; // v_perm selectors: identity, or swap the two halves when d' < 0
    float tlimit;          // only hits with t <= tlimit can still change the result
    float t, u, v;
    uint32_t prim;         // 0xFFFFFFFF = no hit yet
    int node;              // >= 0 inner node, < 0 leaf code
    uint32_t tri;          // next triangle of the current leaf
    int sp;
    int occ;               // shadow rays: 0 nothing, 1 blocked, 2 unblocked by a hit at t <= 0.01; radiance rays: -1
    bool shadow;
};

__device__ __forceinline__ void ray_init(RayState& s, f3 o, f3 d, bool shadow, int root)
{
    s.o = o; s.d = d;
    // direction used by the BOX tests only: components below 1e-12 of the largest one are pushed to
    // +-1e-12 of it (the ray tilts by < 1e-12 of its length, far inside the box padding), so that 1/d is
    // finite.  A null direction gives 1/d = 0: every box passes, no triangle can be hit (det = 0).
    const float dmax = fmaxf(fabsf(d.x), fmaxf(fabsf(d.y), fabsf(d.z)));
    const float deps = dmax * 1e-12f;
    const float bx = fabsf(d.x) < deps ? __builtin_copysignf(deps, d.x) : d.x;
    const float by = fabsf(d.y) < deps ? __builtin_copysignf(deps, d.y) : d.y;
    const float bz = fabsf(d.z) < deps ? __builtin_copysignf(deps, d.z) : d.z;
    const bool null_dir = !(dmax > 0.0f);
    s.idx = null_dir ? 0.0f : __builtin_amdgcn_rcpf(bx); s.idy = null_dir ? 0.0f : __builtin_amdgcn_rcpf(by); s.idz = null_dir ? 0.0f : __builtin_amdgcn_rcpf(bz);
    s.oidx = -(o.x * s.idx); s.oidy = -(o.y * s.idy); s.oidz = -(o.z * s.idz);
#if MI355RT_WIDE
    // v_perm(hi, lo, sel): the dword of the four children's NEAR planes on this axis (lo bytes, or hi bytes when d' < 0); sel ^ 0x04040404 takes the far ones
    s.selx = __builtin_signbitf(bx) ? 0x07060504u : 0x03020100u;
    s.sely = __builtin_signbitf(by) ? 0x07060504u : 0x03020100u;
    s.selz = __builtin_signbitf(bz) ? 0x07060504u : 0x03020100u;
#else
    s.selx = __builtin_signbitf(bx) ? 0x01000302u : 0x03020100u;
    s.sely = __builtin_signbitf(by) ? 0x01000302u : 0x03020100u;
    s.selz = __builtin_signbitf(bz) ? 0x01000302u : 0x03020100u;
#endif
    s.tlimit = shadow ? 0x1.fffffep-1f : __builtin_inff();     // shadow: t < 1.0
    s.t = __builtin_inff(); s.u = 0.0f; s.v = 0.0f; s.prim = 0xFFFFFFFFu;
    s.node = root; s.tri = 0u; s.sp = 0; s.occ = shadow ? 0 : -1; s.shadow = shadow;
}

// Slab test of one child box of a 32-byte node (bvh.hpp).  Per axis the two half-precision bounds sit
// in one 32-bit word (min | max << 16).  A per-ray byte-permute selector swaps the halves when the
// direction component is negative, so the low half is always the NEAR plane and the high half the FAR
// plane: no per-axis min/max.  Each plane distance is one mixed-precision fma (v_fma_mix_f32 reads the
// half directly): t = b * (1/d) + (-o/d).  This is not the reference's (b - o) * inv and need not be:
// boxes only steer the search; they are padded by 2e-5 of the scene diagonal and rounded outward, far
// above the rounding difference (direction components are kept away from zero in ray_init, so 1/d is
// finite and the fma form cannot produce inf - inf).
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void slab_child(const uint4 q, const RayState& s, float& tn, float& tf)
{
    const half2_t hx = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(q.x, q.x, s.selx));
    const half2_t hy = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(q.y, q.y, s.sely));
    const half2_t hz = __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(q.z, q.z, s.selz));
    const float nx = __builtin_fmaf((float)hx.x, s.idx, s.oidx), fx = __builtin_fmaf((float)hx.y, s.idx, s.oidx);
    const float ny = __builtin_fmaf((float)hy.x, s.idy, s.oidy), fy = __builtin_fmaf((float)hy.y, s.idy, s.oidy);
    const float nz = __builtin_fmaf((float)hz.x, s.idz, s.oidz), fz = __builtin_fmaf((float)hz.y, s.idz, s.oidz);
    tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    tf = fminf(fminf(fx, fy), fminf(fz, s.tlimit));
}

#if MI355RT_WIDE
// The four children of a 48-byte node (bvh.hpp, BvhNode4): slab distances from the 8-bit planes, t = q * (2^e / d) + (org - o) / d — like slab_child
// not the reference's expression and not required to be (the boxes are padded and rounded outward on the host) —, and the children's references
// in the binary tree's own encoding (>= 0 node index, < 0 leaf code).
__device__ __forceinline__ float wide_byte(uint32_t w, int i) { return (float)((w >> (8 * i)) & 0xFFu); }        // v_cvt_f32_ubyte<i>
__device__ __forceinline__ void wide_children(const uint4 q0, const uint4 q1, const uint4 q2, const RayState& s, float tn[4], bool h[4], int ref[4])
{
    const float sx = __uint_as_float((q0.w & 0xFFu) << 23) * s.idx, sy = __uint_as_float(((q0.w >> 8) & 0xFFu) << 23) * s.idy, sz = __uint_as_float(((q0.w >> 16) & 0xFFu) << 23) * s.idz;
    const float bx = __builtin_fmaf(__uint_as_float(q0.x), s.idx, s.oidx), by = __builtin_fmaf(__uint_as_float(q0.y), s.idy, s.oidy), bz = __builtin_fmaf(__uint_as_float(q0.z), s.idz, s.oidz);
    const uint32_t nxw = __builtin_amdgcn_perm(q1.y, q1.x, s.selx), fxw = __builtin_amdgcn_perm(q1.y, q1.x, s.selx ^ 0x04040404u);
    const uint32_t nyw = __builtin_amdgcn_perm(q1.w, q1.z, s.sely), fyw = __builtin_amdgcn_perm(q1.w, q1.z, s.sely ^ 0x04040404u);
    const uint32_t nzw = __builtin_amdgcn_perm(q2.y, q2.x, s.selz), fzw = __builtin_amdgcn_perm(q2.y, q2.x, s.selz ^ 0x04040404u);
    const uint32_t cbm = (q2.w & 0xFFFFFu) - 128u;                                  // inner child: child_base + (byte - 0x80)
    const uint32_t nb = ~(((q2.w >> 20) | ((q0.w >> 24) << 12)) << 3);             // leaf child: ~((tri_base << 3) + byte) = nb - byte
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float nx = __builtin_fmaf(wide_byte(nxw, i), sx, bx), fx = __builtin_fmaf(wide_byte(fxw, i), sx, bx);
        const float ny = __builtin_fmaf(wide_byte(nyw, i), sy, by), fy = __builtin_fmaf(wide_byte(fyw, i), sy, by);
        const float nz = __builtin_fmaf(wide_byte(nzw, i), sz, bz), fz = __builtin_fmaf(wide_byte(fzw, i), sz, bz);
        tn[i] = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
        const float tf = fminf(fminf(fx, fy), fminf(fz, s.tlimit));
        h[i] = tn[i] <= tf;
        const uint32_t t = (q2.z >> (8 * i)) & 0xFFu;
        ref[i] = (t & 0x80u) ? (int)(cbm + t) : (int)(nb - t);
    }
}
// the nearest of the children that were hit: slot and reference (anything when none was)
__device__ __forceinline__ void wide_nearest(const float tn[4], const bool h[4], const int ref[4], int& near, int& near_ref)
{
    const float inf = __builtin_inff();
    const float k0 = h[0] ? tn[0] : inf, k1 = h[1] ? tn[1] : inf, k2 = h[2] ? tn[2] : inf, k3 = h[3] ? tn[3] : inf;
    const bool b1 = k1 < k0, b3 = k3 < k2;
    const float m01 = b1 ? k1 : k0, m23 = b3 ? k3 : k2;
    const bool bb = m23 < m01;
    near = bb ? (b3 ? 3 : 2) : (b1 ? 1 : 0);
    near_ref = bb ? (b3 ? ref[3] : ref[2]) : (b1 ? ref[1] : ref[0]);
}
#endif

// Pop the next deferred node; returns true when the stack is empty (ray finished).
__device__ __forceinline__ bool ray_pop(RayState& s, const int* stack, int stride)
{
    if (s.sp == 0) return true;
    --s.sp;
    s.node = stack[s.sp * stride];
    s.tri = 0u;
    return false;
}

// One inner-node step (s.node >= 0): test both child boxes, descend into the nearer hit child and
// defer the other.  `stack` points at this lane's column of an LDS array with row stride `stride`
// ints.  Returns true when the ray is finished.
template <bool COUNT>
__device__ __forceinline__ bool inner_step(const DScene& sc, RayState& s, int* stack, int stride, uint32_t& n_nodes)
{
    const uint4* __restrict__ nodes = (const uint4*)sc.nodes;
    if (COUNT) ++n_nodes;
#if MI355RT_WIDE
    {
        const uint4 w0 = nodes[4 * s.node], w1 = nodes[4 * s.node + 1], w2 = nodes[4 * s.node + 2];
        float tn[4]; bool h[4]; int ref[4]; int near, near_ref;
        wide_children(w0, w1, w2, s, tn, h, ref);
        wide_nearest(tn, h, ref, near, near_ref);
        s.tri = 0u;
        if (!(h[0] | h[1] | h[2] | h[3])) return ray_pop(s, stack, stride);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (h[i] && i != near) { stack[s.sp * stride] = ref[i]; ++s.sp; }
        s.node = near_ref;
        return false;
    }
#endif
    const uint4 q0 = nodes[2 * s.node], q1 = nodes[2 * s.node + 1];
    float tn0, tf0, tn1, tf1;
    slab_child(q0, s, tn0, tf0);
    slab_child(q1, s, tn1, tf1);
    const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
    const int c0i = (int)q0.w, c1i = (int)q1.w;
    s.tri = 0u;
    if (h0 && h1) {
        const bool sw = tn1 < tn0;
        stack[s.sp * stride] = sw ? c0i : c1i;
        ++s.sp;
        s.node = sw ? c1i : c0i;
        return false;
    }
    if (h0) { s.node = c0i; return false; }
    if (h1) { s.node = c1i; return false; }
    return ray_pop(s, stack, stride);
}

// One leaf step (s.node < 0): test triangle s.tri of the leaf, then advance; after the last triangle
// pop the next deferred node.  Returns true when the ray is finished.
template <bool COUNT>
__device__ __forceinline__ bool leaf_step(const DScene& sc, RayState& s, const int* stack, int stride, uint32_t& n_tris)
{
    const float4* __restrict__ tris = (const float4*)sc.tris;
    const uint32_t code = ~(uint32_t)s.node;
    const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
    if (COUNT) ++n_tris;
    const uint32_t ti = first + s.tri;
    const float4 t0 = tris[3 * ti], t1 = tris[3 * ti + 1], t2 = tris[3 * ti + 2];
    ++s.tri;
    // Moller-Trumbore "late out", intersect.rs:62-98, same operation order
    const f3 v0 = mk3(t0.x, t0.y, t0.z), v0v1 = mk3(t1.x, t1.y, t1.z), v0v2 = mk3(t2.x, t2.y, t2.z);
    const f3 pvec = cross3(s.d, v0v2);
    const float det = dot3(v0v1, pvec);
    if (!(fabsf(det) < 1.1920929e-7f)) {                       // f32::EPSILON
        const float inv_det = div_rn(1.0f, det);
        const f3 tvec = sub3(s.o, v0);
        const float u = dot3(tvec, pvec) * inv_det;
        const f3 qvec = cross3(tvec, v0v1);
        const float v = dot3(s.d, qvec) * inv_det;
        const float t = dot3(v0v2, qvec) * inv_det;
        if (!(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && !(t < 0.0f)) {
            const uint32_t prim = __float_as_uint(t0.w);
            if (!s.shadow) {
                if (t <= s.tlimit && (s.prim == 0xFFFFFFFFu || t < s.t || (t == s.t && prim < s.prim))) {    // tlimit: inf, the closest hit so far, or (closest-hit shadow rays) just below 1
                    s.t = t; s.u = u; s.v = v; s.prim = prim; s.tlimit = t;
                }
            } else if (t <= s.tlimit) {
                if (t > 0.01f) { s.occ = 1; s.tlimit = 0.01f; }
                else { s.occ = 2; return true; }
            }
        }
    }
    if (s.tri < cnt) return false;
    return ray_pop(s, stack, stride);
}

// whole-ray traversal (used by the batched Intersector seam)
template <bool COUNT>
__device__ __forceinline__ void ray_run(const DScene& sc, RayState& s, int* stack, int stride, uint32_t& n_nodes, uint32_t& n_tris)
{
    for (;;) {
        const bool done = s.node >= 0 ? inner_step<COUNT>(sc, s, stack, stride, n_nodes) : leaf_step<COUNT>(sc, s, stack, stride, n_tris);
        if (done) break;
    }
}

// ---- predicated steps for the persistent trace kernel ------------------------------------------------------
// Same arithmetic as inner_step / leaf_step, written WITHOUT per-lane branches: every lane of the wave
// runs the code, state changes are selects, LDS accesses of lanes that neither push nor pop go to a trash
// row.  Divergent branches cost this kernel more scalar exec-mask bookkeeping than the arithmetic they
// skip (profiles/r01_notes.md).
//
// The whole lane state machine lives in s.node, so that every predicate is ONE compare on a VGPR (a lane
// mask in SGPRs) instead of a boolean carried through the loop in a VGPR:
//   node >= 0                   at an inner node
//   kNodeFin < node < 0         at a leaf: ~node = first_triangle << 3 | (triangles left - 1); stepping to
//                               the next triangle of the leaf is code += 7, i.e. node -= 7
//   node == kNodeFin            the ray is finished, its result is not written yet
//   node == kNodeIdle           the lane holds no ray
// (triangle counts are < 2^26, so no leaf code collides with the two sentinels.)  s.occ < 0 marks a
// radiance ray; s.tri and s.shadow are not used on this path.
constexpr int kNodeIdle = (int)0x80000000u, kNodeFin = (int)0x80000001u;
__device__ __forceinline__ bool lane_at_inner(const RayState& s) { return s.node >= 0; }
__device__ __forceinline__ bool lane_at_leaf(const RayState& s) { return (uint32_t)s.node > 0x80000001u; }

// Stack layout of the persistent loop (lane-major LDS column, row stride `stride`): row 0 holds the SENTINEL kNodeFin,
// deferred node k sits in row k + 1, s.sp = number of deferred nodes.  A pop reads row sp — the top entry, or the
// sentinel when nothing is deferred — so "stack empty => ray finished" needs no compare and no select; lanes that do not
// pop read the same (valid) row and drop the value.  A push writes row sp + 1; lanes that do not push write there too:
// above the top is free space (one spare row is allocated), so that needs no select either.
__device__ __forceinline__ int pop_or_finish(RayState& s, bool want, const int* stack, int stride)
{
    const int popped = stack[s.sp * stride];
    s.sp = max(s.sp - (want ? 1 : 0), 0);
    return popped;
}

template <bool COUNT>
__device__ __forceinline__ void inner_pred(const DScene& sc, RayState& s, int* stack, int stride, uint32_t& n_nodes, uint32_t* below = nullptr)
{
    const bool pred = s.node >= 0;
    if (COUNT) {
        n_nodes += pred ? 1u : 0u;
        if (below) for (int k = 0; k < 6; ++k) below[k] += (pred && s.node < (64 << k)) ? 1u : 0u;      // what an LDS copy of the first 64 .. 2048 nodes would serve (profiles/r03_notes.md)
    }
#if MI355RT_WIDE
    {
        const uint32_t woff = pred ? (uint32_t)s.node << 6 : 0u;      // 64-byte slots, 48 bytes read
        const char* __restrict__ wbase = (const char*)sc.nodes;
        const uint4 w0 = *(const uint4*)(wbase + woff), w1 = *(const uint4*)(wbase + woff + 16u), w2 = *(const uint4*)(wbase + woff + 32u);
        float tn[4]; bool h[4]; int ref[4]; int near, near_ref;
        wide_children(w0, w1, w2, s, tn, h, ref);
        wide_nearest(tn, h, ref, near, near_ref);
        const bool none = pred & !(h[0] | h[1] | h[2] | h[3]);
        // the other hit children, pushed in slot order; a lane that does not push writes the row above its top (free space: one spare row)
        int sp = s.sp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            stack[(sp + 1) * stride] = ref[i];
            sp += (pred & h[i] & (near != i)) ? 1 : 0;
        }
        s.sp = sp;
        const int next = pop_or_finish(s, none, stack, stride);
        s.node = pred ? (none ? next : near_ref) : s.node;
        return;
    }
#endif
    const uint32_t off = pred ? (uint32_t)s.node << 5 : 0u;           // 32-byte nodes, unsigned 32-bit byte offset
    const char* __restrict__ base = (const char*)sc.nodes;
    const uint4 q0 = *(const uint4*)(base + off), q1 = *(const uint4*)(base + off + 16u);
#ifdef MI355RT_EXP_EXTRALOAD     // timing experiment (same results): one more divergent 16-byte load per inner step, its value unused — is the step bound by the vector memory pipe?  (yes: +22 % trace time, profiles/r03_notes.md)
    { const uint4 qx = *(const uint4*)((const char*)sc.tris + off); asm volatile("" : : "v"(qx.x), "v"(qx.y), "v"(qx.z), "v"(qx.w)); }
#endif
    float tn0, tf0, tn1, tf1;
    slab_child(q0, s, tn0, tf0);
    slab_child(q1, s, tn1, tf1);
    // bitwise & | on bools throughout: && || would compile to short-circuit branches
    const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
    const int c0i = (int)q0.w, c1i = (int)q1.w;
    const bool sw = tn1 < tn0;                       // child 1 is nearer
    const bool take1 = h1 & (!h0 | sw);
    const int near_c = take1 ? c1i : c0i, far_c = take1 ? c0i : c1i;
    const bool push = pred & h0 & h1;
    const bool none = pred & !(h0 | h1);
    stack[(s.sp + 1) * stride] = far_c;              // row above the top: kept only if sp is incremented
    s.sp += push ? 1 : 0;
    const int next = pop_or_finish(s, none, stack, stride);
    s.node = pred ? (none ? next : near_c) : s.node;
}

// CLOSEST_ONLY (reference-default semantics): every ray wants its closest hit with t <= tlimit — radiance
// rays start with tlimit = inf, shadow rays with the largest float below 1 (mod.rs:226-229 needs the CLOSEST hit of the
// reference's intersector, which the octree confirm step derives from the true closest hit).  No occlusion state.
template <bool COUNT, bool CLOSEST_ONLY>
__device__ __forceinline__ void leaf_pred(const DScene& sc, RayState& s, const int* stack, int stride, uint32_t& n_tris)
{
    const bool pred = lane_at_leaf(s);
    const uint32_t code = ~(uint32_t)s.node;
    if (COUNT) n_tris += pred ? 1u : 0u;
    const uint32_t off = pred ? (code >> 3) * 48u : 0u;              // 48-byte triangles, unsigned 32-bit byte offset
    const char* __restrict__ base = (const char*)sc.tris;
    const float4 t0 = *(const float4*)(base + off), t1 = *(const float4*)(base + off + 16u), t2 = *(const float4*)(base + off + 32u);
    // Moller-Trumbore "late out", intersect.rs:62-98, same operation order
    const f3 v0 = mk3(t0.x, t0.y, t0.z), v0v1 = mk3(t1.x, t1.y, t1.z), v0v2 = mk3(t2.x, t2.y, t2.z);
    const f3 pvec = cross3(s.d, v0v2);
    const float det = dot3(v0v1, pvec);
    const float inv_det = div_rn(1.0f, det);
    const f3 tvec = sub3(s.o, v0);
    const float u = dot3(tvec, pvec) * inv_det;
    const f3 qvec = cross3(tvec, v0v1);
    const float v = dot3(s.d, qvec) * inv_det;
    const float t = dot3(v0v2, qvec) * inv_det;
    const bool ok = pred & !(fabsf(det) < 1.1920929e-7f) & !((u < 0.0f) | (u > 1.0f)) & !((v < 0.0f) | (u + v > 1.0f)) & !(t < 0.0f);
    const uint32_t prim = __float_as_uint(t0.w);
    const bool last = (code & 7u) == 0u;
    if (CLOSEST_ONLY) {
        const bool better = ok & (t <= s.tlimit) & ((s.prim == 0xFFFFFFFFu) | (t < s.t) | ((t == s.t) & (prim < s.prim)));
        s.t = better ? t : s.t; s.u = better ? u : s.u; s.v = better ? v : s.v; s.prim = better ? prim : s.prim;
        s.tlimit = better ? t : s.tlimit;
        const int next = pop_or_finish(s, pred & last, stack, stride);
        s.node = pred ? (last ? next : s.node - 7) : s.node;
    } else {
        const bool is_shadow = s.occ >= 0;
        const bool better = ok & !is_shadow & ((s.prim == 0xFFFFFFFFu) | (t < s.t) | ((t == s.t) & (prim < s.prim)));
        s.t = better ? t : s.t; s.u = better ? u : s.u; s.v = better ? v : s.v; s.prim = better ? prim : s.prim;
        const bool sh = ok & is_shadow & (t <= s.tlimit);
        const bool sh_far = sh & (t > 0.01f), sh_near = sh & !(t > 0.01f);
        s.tlimit = better ? t : (sh_far ? 0.01f : s.tlimit);
        s.occ = sh_near ? 2 : (sh_far ? 1 : s.occ);
        const int next = pop_or_finish(s, pred & last & !sh_near, stack, stride);
        s.node = pred ? (sh_near ? kNodeFin : (last ? next : s.node - 7)) : s.node;
    }
}

// ---- reference-exact intersector (MI355RT_FLAG_OCTREE_SEMANTICS) ---------------------------------------
// OctTreeIntersector::intersect_ray / intersect_node, oct_tree_intersector.rs:148-206, 240-272, 348-372,
// bug for bug: children slab-tested with the precomputed IEEE inverse direction, kept hits sorted by
// tmin with a STABLE sort, visited front to back, FIRST leaf that yields a hit wins; a leaf yields its
// closest triangle only if the hit point lies inside the leaf cube (inclusive).  The root cube is never
// slab-tested.  A slow path by design (<= 70 triangles per leaf): it exists for parity, not for speed.
// intersect_cube_inverse_ray, OCT:348-372: slab test with the precomputed inverse direction; returns the entry
// distance tmin (negative when the origin is inside).  fminf/fmaxf ignore a NaN operand like Rust's f32::min/max,
// so 0 * inf lanes behave as in the reference.
__device__ __forceinline__ bool cube_slab(f3 cmin, f3 cmax, f3 o, f3 inv, float& tmin_out)
{
    const float tx1 = (cmin.x - o.x) * inv.x, tx2 = (cmax.x - o.x) * inv.x;
    float tmin = fminf(tx1, tx2), tmax = fmaxf(tx1, tx2);
    const float ty1 = (cmin.y - o.y) * inv.y, ty2 = (cmax.y - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(ty1, ty2)); tmax = fminf(tmax, fmaxf(ty1, ty2));
    const float tz1 = (cmin.z - o.z) * inv.z, tz2 = (cmax.z - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(tz1, tz2)); tmax = fminf(tmax, fmaxf(tz1, tz2));
    tmin_out = tmin;
    return tmax >= tmin && tmax > 0.0f;
}

__device__ inline void octree_intersect(const DScene& sc, f3 o, f3 d, float& out_t, float& out_u, float& out_v, uint32_t& out_prim)
{
    const float4* __restrict__ nodes = (const float4*)sc.oct_nodes;
    const float4* __restrict__ ptris = (const float4*)sc.prim_tris;
    const f3 inv = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));     // OCT:241-244
    out_prim = 0xFFFFFFFFu; out_t = 0.0f; out_u = 0.0f; out_v = 0.0f;
    int stack[80];                                   // depth <= 9 levels x 8 children
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const int node = stack[--sp];
        const float4 n0 = nodes[3 * node], n1 = nodes[3 * node + 1], n2 = nodes[3 * node + 2];
        const int first_child = __float_as_int(n0.w);
        if (first_child < 0) {
            // intersect_leaf_triangles, OCT:249-272: strict `<` keeps the first of equal t
            const uint32_t tri_first = __float_as_uint(n1.w), tri_count = __float_as_uint(n2.x);
            bool have = false; float bt = 0.0f, bu = 0.0f, bv = 0.0f; uint32_t bp = 0u;
            for (uint32_t k = 0; k < tri_count; ++k) {
                const uint32_t prim = sc.oct_leaf_tris[tri_first + k];
                const float4 t0 = ptris[3 * prim], t1 = ptris[3 * prim + 1], t2 = ptris[3 * prim + 2];
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
                if (!have || t < bt) { have = true; bt = t; bu = u; bv = v; bp = prim; }
            }
            if (have) {
                const f3 hp = add3(o, vscale(d, bt));                          // OCT:164
                const bool inside = !(hp.x < n0.x || hp.x > n1.x || hp.y < n0.y || hp.y > n1.y || hp.z < n0.z || hp.z > n1.z);   // OCT:34-45
                if (inside) { out_t = bt; out_u = bu; out_v = bv; out_prim = bp; return; }
            }
            continue;
        }
        int idx[8]; float dist[8]; int n = 0;
        for (int i = 0; i < 8; ++i) {
            const int c = first_child + i;
            const float4 c0 = nodes[3 * c], c1 = nodes[3 * c + 1];
            float tmin;
            if (cube_slab(mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), o, inv, tmin)) {
                // stable insertion by tmin ascending (OCT:183); a NaN distance compares as "not less"
                int j = n - 1;
                while (j >= 0 && tmin < dist[j]) { idx[j + 1] = idx[j]; dist[j + 1] = dist[j]; --j; }
                idx[j + 1] = c; dist[j + 1] = tmin;
                ++n;
            }
        }
        for (int i = n - 1; i >= 0; --i) stack[sp++] = idx[i];                 // nearest child is popped first
    }
}

// ---- reference-default semantics at BVH speed: the octree CONFIRM walk -------------------------------------------
// Input: the TRUE closest hit H* = (t*, u*, v*, prim*) of a ray (BVH traversal; lowest t, ties to the lowest triangle
// index).  Output: what OctTreeIntersector::intersect_ray (OCT:148-206, 240-272) returns for that ray — exact, with no
// tolerance anywhere.  Why that is possible without repeating the reference's work:
//   * A ray the BVH finds no hit for hits no triangle at all, so the octree returns None: no walk needed.
//   * Every triangle hit of the ray has t >= t* (same Moller-Trumbore arithmetic, so the same floats).  A leaf L accepts
//     its closest hit h only if fl(o + d*t_h) lies in cube(L) (OCT:160-169).  fl(o_k + fl(d_k*t)) is MONOTONE in t, so
//     if the point at t* is already past the far face of a cube C on some axis in the ray's direction of travel
//     (d_k > 0 and hp_k(t*) > C.max_k, or d_k < 0 and hp_k(t*) < C.min_k), the point at any t_h >= t* is past it too:
//     no leaf inside C can accept anything.  Such cubes are skipped, subtree and all.
//   * A leaf whose list contains prim* has h = H* (H* is the minimum over ALL triangles, ties included: the lists are in
//     ascending triangle order, OCT:94-146): its answer is the contains test on fl(o + d*t*), nothing else to compute.
//   * Any other leaf that is reached and not skipped (the hit point sits on or within rounding of a cube boundary: rare)
//     is scanned like the reference scans it (<= triangles_per_leaf exact tests).
// The visiting order is the reference's: children that pass the slab test, by ascending (tmin, child index) — the stable
// sort of OCT:176-185 — first accepting leaf wins.  Child cubes are not loaded: they are the parent's min / mid / max per
// axis with mid = 0.5 * (max + min), the builder's own expression (OCT:274-313), so the floats are identical.
__device__ __forceinline__ bool cube_contains(f3 mn, f3 mx, f3 p)        // Cube::contains, OCT:34-45 (inclusive)
{
    return !(p.x < mn.x || p.x > mx.x || p.y < mn.y || p.y > mx.y || p.z < mn.z || p.z > mx.z);
}

__device__ inline void confirm_walk(const DScene& sc, f3 o, f3 d, float& t, float& u, float& v, uint32_t& prim)
{
    const float4* __restrict__ nodes = (const float4*)sc.oct_nodes;
    const f3 inv = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));     // OCT:241-244
    const f3 hp = add3(o, vscale(d, t));                                               // OCT:164 for H*
    uint32_t node = 0u;
    const float4 root0 = nodes[0], root1 = nodes[1];
    f3 mn = mk3(root0.x, root0.y, root0.z), mx = mk3(root1.x, root1.y, root1.z);
    int2 info = sc.oct_info[0];                               // x: first child (>= 0) or ~tri_first (leaf); y: leaf triangle count
    const uint32_t home = sc.tri_home[prim];                  // the only leaf that lists prim*, or 0xFFFFFFFF (several)
    bool resuming = false; float r_t = 0.0f; int r_i = 0;      // after a child returned None: only children with a larger (tmin, index) key
    // (plane - o) * inv can only be NaN when an inverse component is not finite (0 * inf) or the ray itself holds a NaN:
    // such rays always take the general child scan
    const bool special = !(fabsf(inv.x) < __builtin_inff()) || !(fabsf(inv.y) < __builtin_inff()) || !(fabsf(inv.z) < __builtin_inff())
                         || o.x != o.x || o.y != o.y || o.z != o.z;
    // Mirrored frame for the fast descent: every axis on which the ray travels in the negative direction is negated
    // (cube, origin, hit point, inverse direction), so that the ray travels in the positive direction on all three.
    // Negation is exact and (-p - -o) * -inv == (p - o) * inv bit for bit, so every plane distance and every comparison
    // below is the reference's own; 0.5 * (-mn + -mx) == -(0.5 * (mx + mn)) likewise.
    const bool px = !(d.x < 0.0f), py = !(d.y < 0.0f), pz = !(d.z < 0.0f);
    const f3 om = mk3(px ? o.x : -o.x, py ? o.y : -o.y, pz ? o.z : -o.z);
    const f3 hpm = mk3(px ? hp.x : -hp.x, py ? hp.y : -hp.y, pz ? hp.z : -hp.z);
    const f3 invm = mk3(px ? inv.x : -inv.x, py ? inv.y : -inv.y, pz ? inv.z : -inv.z);
    for (;;) {
        bool descend = false;
        if (!special && !resuming && info.x >= 0) {
            // ---- fast descent.  In the mirrored frame the near half of every axis is the low one.  Per axis: the low half is
            // skippable iff the point at t* lies beyond the mid plane; if it lies beyond the far face the whole node is.  The
            // child c0 of the nearest allowed halves has the smallest sort key of all candidates; another candidate ties with
            // it iff, on an axis where both halves are allowed, the mid-plane distance (= c0's exit distance on that axis)
            // does not exceed c0's key.  No tie and c0 passes the slab test: c0 is the reference's next child.  Everything
            // else (tie, failing slab test, node wholly passed) leaves the loop for the general code below.
            f3 a = mk3(px ? mn.x : -mx.x, py ? mn.y : -mx.y, pz ? mn.z : -mx.z);
            f3 b = mk3(px ? mx.x : -mn.x, py ? mx.y : -mn.y, pz ? mx.z : -mn.z);
            for (;;) {
                const f3 md = mk3(0.5f * (b.x + a.x), 0.5f * (b.y + a.y), 0.5f * (b.z + a.z));
                const bool lx = hpm.x > md.x, ly = hpm.y > md.y, lz = hpm.z > md.z;          // low half skippable
                const bool gone = (hpm.x > b.x) | (hpm.y > b.y) | (hpm.z > b.z);              // every child skippable
                const f3 nr = mk3(lx ? md.x : a.x, ly ? md.y : a.y, lz ? md.z : a.z);          // chosen half: near plane ...
                const f3 fr = mk3(lx ? b.x : md.x, ly ? b.y : md.y, lz ? b.z : md.z);          // ... and far plane
                const float tnx = (nr.x - om.x) * invm.x, tny = (nr.y - om.y) * invm.y, tnz = (nr.z - om.z) * invm.z;
                const float tfx = (fr.x - om.x) * invm.x, tfy = (fr.y - om.y) * invm.y, tfz = (fr.z - om.z) * invm.z;
                const float key0 = fmaxf(fmaxf(tnx, tny), tnz), tmax0 = fminf(fminf(tfx, tfy), tfz);
                const bool tie = (!lx & (tfx <= key0)) | (!ly & (tfy <= key0)) | (!lz & (tfz <= key0));
                if (gone | tie | !(tmax0 >= key0) | !(tmax0 > 0.0f)) break;
                const uint32_t ci = (uint32_t)(lx == px) | ((uint32_t)(ly == py) << 1) | ((uint32_t)(lz == pz) << 2);   // real child index: high half iff (mirrored high) == (not mirrored)
                node = (uint32_t)info.x + ci;
                a = nr; b = fr;
                info = sc.oct_info[node];
                if (info.x < 0) break;
            }
            mn = mk3(px ? a.x : -b.x, py ? a.y : -b.y, pz ? a.z : -b.z);
            mx = mk3(px ? b.x : -a.x, py ? b.y : -a.y, pz ? b.z : -a.z);
#ifdef MI355RT_EXP_FASTONLY      // timing experiment (wrong results): what the walk costs when nothing ever leaves the fast descent
            return;
#endif
        }
        if (info.x < 0) {
            // ---- leaf (reached => not skippable): what does intersect_leaf_triangles + contains give?
            const uint32_t tri_first = (uint32_t)~info.x, tri_count = (uint32_t)info.y;
            // is prim* in this leaf's (ascending) list?
            bool member = home == node;
            if (home == 0xFFFFFFFFu) {
                uint32_t lo = 0u, hi = tri_count;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sc.oct_leaf_tris[tri_first + mid] < prim) lo = mid + 1u; else hi = mid; }
                member = lo < tri_count && sc.oct_leaf_tris[tri_first + lo] == prim;
            }
            if (member) {
                if (cube_contains(mn, mx, hp)) return;                      // the leaf's closest hit is H* and it is accepted
            } else if (tri_count != 0u) {
                // boundary case: scan the list as the reference does (OCT:249-272, strict `<` keeps the first of equal t)
                const float4* __restrict__ ptris = (const float4*)sc.prim_tris;
                bool have = false; float bt = 0.0f, bu = 0.0f, bv = 0.0f; uint32_t bp = 0u;
                for (uint32_t k = 0; k < tri_count; ++k) {
                    const uint32_t p = sc.oct_leaf_tris[tri_first + k];
                    const float4 t0 = ptris[3 * p], t1 = ptris[3 * p + 1], t2 = ptris[3 * p + 2];
                    const f3 v0 = mk3(t0.x, t0.y, t0.z), v0v1 = mk3(t1.x, t1.y, t1.z), v0v2 = mk3(t2.x, t2.y, t2.z);
                    const f3 pvec = cross3(d, v0v2);
                    const float det = dot3(v0v1, pvec);
                    if (fabsf(det) < 1.1920929e-7f) continue;
                    const float inv_det = div_rn(1.0f, det);
                    const f3 tvec = sub3(o, v0);
                    const float uu = dot3(tvec, pvec) * inv_det;
                    const f3 qvec = cross3(tvec, v0v1);
                    const float vv = dot3(d, qvec) * inv_det;
                    const float tt = dot3(v0v2, qvec) * inv_det;
                    if (uu < 0.0f || uu > 1.0f) continue;
                    if (vv < 0.0f || uu + vv > 1.0f) continue;
                    if (tt < 0.0f) continue;
                    if (!have || tt < bt) { have = true; bt = tt; bu = uu; bv = vv; bp = p; }
                }
                if (have && cube_contains(mn, mx, add3(o, vscale(d, bt)))) { t = bt; u = bu; v = bv; prim = bp; return; }
            }
        } else {
            // ---- inner node: first child, in the reference's order, that passes the slab test, is not skippable and
            // comes after the resume key.  Planes per axis: min, mid, max.
            const f3 md = mk3(0.5f * (mx.x + mn.x), 0.5f * (mx.y + mn.y), 0.5f * (mx.z + mn.z));
            // intersect_cube_inverse_ray (OCT:348-372) per half: (plane - origin) * inverse direction
            const float tx0 = (mn.x - o.x) * inv.x, tx1 = (md.x - o.x) * inv.x, tx2 = (mx.x - o.x) * inv.x;
            const float ty0 = (mn.y - o.y) * inv.y, ty1 = (md.y - o.y) * inv.y, ty2 = (mx.y - o.y) * inv.y;
            const float tz0 = (mn.z - o.z) * inv.z, tz1 = (md.z - o.z) * inv.z, tz2 = (mx.z - o.z) * inv.z;
            const float lox[2] = { fminf(tx0, tx1), fminf(tx1, tx2) }, hix[2] = { fmaxf(tx0, tx1), fmaxf(tx1, tx2) };
            const float loy[2] = { fminf(ty0, ty1), fminf(ty1, ty2) }, hiy[2] = { fmaxf(ty0, ty1), fmaxf(ty1, ty2) };
            const float loz[2] = { fminf(tz0, tz1), fminf(tz1, tz2) }, hiz[2] = { fmaxf(tz0, tz1), fmaxf(tz1, tz2) };
            // skippable halves: the point at t* is past the half's far face in the direction of travel
            const bool sx[2] = { (d.x > 0.0f && hp.x > md.x) || (d.x < 0.0f && hp.x < mn.x), (d.x > 0.0f && hp.x > mx.x) || (d.x < 0.0f && hp.x < md.x) };
            const bool sy[2] = { (d.y > 0.0f && hp.y > md.y) || (d.y < 0.0f && hp.y < mn.y), (d.y > 0.0f && hp.y > mx.y) || (d.y < 0.0f && hp.y < md.y) };
            const bool sz[2] = { (d.z > 0.0f && hp.z > md.z) || (d.z < 0.0f && hp.z < mn.z), (d.z > 0.0f && hp.z > mx.z) || (d.z < 0.0f && hp.z < md.z) };
            // General decision (a tie, a failing slab test, a node the point at t* has wholly passed, a resume after a None, an
            // inf / NaN inverse direction): scan all eight children like the reference's sort would order them.
            const bool none_allowed = (sx[0] & sx[1]) | (sy[0] & sy[1]) | (sz[0] & sz[1]);       // the point at t* is past this whole cube
            int best = -1;
            if (!none_allowed) {
                float best_t = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int hx = i & 1, hy = (i >> 1) & 1, hz = (i >> 2) & 1;      // child order of generate_child_cubes, OCT:279-312
                    float tmin = fmaxf(lox[hx], loy[hy]), tmax = fminf(hix[hx], hiy[hy]);
                    tmin = fmaxf(tmin, loz[hz]); tmax = fminf(tmax, hiz[hz]);
                    bool cand = (tmax >= tmin) & (tmax > 0.0f) & !(sx[hx] | sy[hy] | sz[hz]);
                    cand &= !resuming | (tmin > r_t) | ((tmin == r_t) & (i > r_i));
                    if (cand & ((best < 0) | (tmin < best_t))) { best = i; best_t = tmin; }   // ascending i: ties keep the lower index (stable sort)
                }
            }
            if (best >= 0) {
                node = (uint32_t)info.x + (uint32_t)best;
                mn = mk3(best & 1 ? md.x : mn.x, best & 2 ? md.y : mn.y, best & 4 ? md.z : mn.z);
                mx = mk3(best & 1 ? mx.x : md.x, best & 2 ? mx.y : md.y, best & 4 ? mx.z : md.z);
                info = sc.oct_info[node];
                resuming = false;
                descend = true;
            }
        }
        if (descend) continue;
        // ---- this node yields None: back to the parent, resume after this child
        if (node == 0u) { prim = 0xFFFFFFFFu; return; }
        float my_tmin;
        (void)cube_slab(mn, mx, o, inv, my_tmin);                          // this child's sort key, recomputed (same expression => same float)
        const uint32_t parent = __float_as_uint(nodes[3u * node + 2u].y);
        const float4 p0 = nodes[3u * parent], p1 = nodes[3u * parent + 1u];
        info = sc.oct_info[parent];
        r_t = my_tmin; r_i = (int)(node - (uint32_t)info.x); resuming = true;
        node = parent;
        mn = mk3(p0.x, p0.y, p0.z); mx = mk3(p1.x, p1.y, p1.z);
    }
}

}  // namespace mi355rt
