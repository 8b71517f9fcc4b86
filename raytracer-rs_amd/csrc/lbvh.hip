// lbvh.hip — BVH2 build ON THE DEVICE (MI355RT_FLAG_DEVICE_LBVH): Morton codes, radix sort, Karras' parallel hierarchy
// (T. Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012), bottom-up refit,
// subtrees of at most kBvhMaxLeaf triangles collapsed into leaves, nodes emitted in the 32-byte half-precision format of
// bvh.hpp.  The replacement BASELINE.json's north_star names for the reference's octree build
// (oct_tree_intersector.rs:66-146).  Any conservative tree gives the same hits (the exact f32 triangle test decides, ties
// go to the lowest triangle index), so the renderer's results do not depend on which builder ran; what differs is the
// build time and the number of nodes a ray visits (profiles/r02_notes.md).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "bvh.hpp"

namespace mi355rt {
namespace {

constexpr int kLbvhBlock = 256;
constexpr uint32_t kLeafBit = 0x80000000u;      // child reference: leaf (sorted triangle position) or internal node index

struct Box6 { float mn[3], mx[3]; };

__device__ __forceinline__ unsigned long long expand21(unsigned long long v)     // 21 bits -> every third bit of 63
{
    v &= 0x1FFFFFull;
    v = (v | v << 32) & 0x1F00000000FFFFull;
    v = (v | v << 16) & 0x1F0000FF0000FFull;
    v = (v | v << 8) & 0x100F00F00F00F00Full;
    v = (v | v << 4) & 0x10C30C30C30C30C3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// per triangle: its box and the 63-bit Morton code of the box centre (the host builder's centroid: 0.5 * (min + max))
__global__ __launch_bounds__(kLbvhBlock) void lbvh_prim_kernel(const float* __restrict__ verts, uint32_t n, float3 smin, float3 sinv,
                                                              Box6* __restrict__ boxes, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t t = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (t >= n) return;
    const float* v = verts + 9ull * t;
    Box6 b;
    for (int a = 0; a < 3; ++a) {
        b.mn[a] = fminf(fminf(v[a], v[3 + a]), v[6 + a]);
        b.mx[a] = fmaxf(fmaxf(v[a], v[3 + a]), v[6 + a]);
    }
    boxes[t] = b;
    const float cx = (0.5f * (b.mn[0] + b.mx[0]) - smin.x) * sinv.x, cy = (0.5f * (b.mn[1] + b.mx[1]) - smin.y) * sinv.y, cz = (0.5f * (b.mn[2] + b.mx[2]) - smin.z) * sinv.z;
    auto q = [](float c) { c = c * 2097152.0f; c = c > 0.0f ? c : 0.0f; return (unsigned long long)(c < 2097151.0f ? c : 2097151.0f); };   // NaN -> 0
    keys[t] = expand21(q(cx)) << 2 | expand21(q(cy)) << 1 | expand21(q(cz));
    vals[t] = t;
}

// Karras' delta: length of the common prefix of the keys at sorted positions i and j (-1 when j is out of range);
// equal keys are told apart by the position, so the hierarchy is a proper binary tree also with duplicates
__device__ __forceinline__ int lbvh_delta(const unsigned long long* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    return a == b ? 64 + __clz((unsigned)(i ^ j)) : __clzll((long long)(a ^ b));
}

// one thread per internal node (n - 1 of them; node 0 is the root): its key range and its two children
__global__ __launch_bounds__(kLbvhBlock) void lbvh_hierarchy_kernel(const unsigned long long* __restrict__ keys, int n, uint2* __restrict__ child, uint2* __restrict__ range,
                                                                   uint32_t* __restrict__ parent_inner, uint32_t* __restrict__ parent_leaf)
{
    const int i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i >= n - 1) return;
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const bool left_leaf = lo == gamma, right_leaf = hi == gamma + 1;
    child[i] = make_uint2((uint32_t)gamma | (left_leaf ? kLeafBit : 0u), (uint32_t)(gamma + 1) | (right_leaf ? kLeafBit : 0u));
    range[i] = make_uint2((uint32_t)lo, (uint32_t)hi);
    if (left_leaf) parent_leaf[gamma] = (uint32_t)i; else parent_inner[gamma] = (uint32_t)i;
    if (right_leaf) parent_leaf[gamma + 1] = (uint32_t)i; else parent_inner[gamma + 1] = (uint32_t)i;
}

__device__ __forceinline__ Box6 box_union(const Box6& a, const Box6& b)
{
    Box6 r;
    for (int k = 0; k < 3; ++k) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    return r;
}

// bottom-up: one thread per leaf climbs; the second thread to arrive at a node computes its box (and the height of the
// part of the tree that survives the leaf collapse: what the traversal stack has to hold)
__global__ __launch_bounds__(kLbvhBlock) void lbvh_refit_kernel(int n, uint32_t max_leaf, const uint32_t* __restrict__ order, const Box6* __restrict__ prim_boxes,
                                                               const uint2* __restrict__ child, const uint2* __restrict__ range,
                                                               const uint32_t* __restrict__ parent_inner, const uint32_t* __restrict__ parent_leaf,
                                                               Box6* leaf_boxes, Box6* node_boxes, uint32_t* height, uint32_t* flags)      // written and read by different threads: no __restrict__
{
    const int j = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (j >= n) return;
    leaf_boxes[j] = prim_boxes[order[j]];
    __threadfence();
    uint32_t cur = parent_leaf[j];
    for (;;) {
        if (atomicAdd(&flags[cur], 1u) == 0u) return;             // the sibling subtree is not done yet: its thread goes on
        __threadfence();
        const uint2 c = child[cur];
        const Box6 b0 = (c.x & kLeafBit) ? leaf_boxes[c.x & ~kLeafBit] : node_boxes[c.x];
        const Box6 b1 = (c.y & kLeafBit) ? leaf_boxes[c.y & ~kLeafBit] : node_boxes[c.y];
        node_boxes[cur] = box_union(b0, b1);
        const uint2 r = range[cur];
        const uint32_t h0 = (c.x & kLeafBit) ? 0u : height[c.x], h1 = (c.y & kLeafBit) ? 0u : height[c.y];
        height[cur] = r.y - r.x + 1u > max_leaf ? 1u + (h0 > h1 ? h0 : h1) : 0u;      // a subtree of <= max_leaf triangles becomes a leaf
        __threadfence();
        if (cur == 0u) return;
        cur = parent_inner[cur];
    }
}

__global__ __launch_bounds__(kLbvhBlock) void lbvh_survive_kernel(int n_inner, uint32_t max_leaf, const uint2* __restrict__ range, uint32_t* __restrict__ survive)
{
    const int i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i < n_inner) survive[i] = range[i].y - range[i].x + 1u > max_leaf ? 1u : 0u;
}

// bound -> binary16 bits, rounded toward -inf / +inf (bvh.cpp: half_bits_directed)
__device__ __forceinline__ uint32_t half_down(float x) { if (x != x) return 0xFC00u; return (uint32_t)__half_as_ushort(__float2half_rd(x)); }
__device__ __forceinline__ uint32_t half_up(float x) { if (x != x) return 0x7C00u; return (uint32_t)__half_as_ushort(__float2half_ru(x)); }
__device__ __forceinline__ void pack_box_dev(const Box6& b, float pad, uint32_t out[3])
{
    for (int a = 0; a < 3; ++a) out[a] = half_down(b.mn[a] - pad) | half_up(b.mx[a] + pad) << 16;
}

// one thread per surviving internal node: the 32-byte node with both (padded, outward-rounded) child boxes
__global__ __launch_bounds__(kLbvhBlock) void lbvh_emit_nodes_kernel(int n_inner, float pad, const uint2* __restrict__ child, const uint2* __restrict__ range,
                                                                    const uint32_t* __restrict__ survive, const uint32_t* __restrict__ newidx,
                                                                    const Box6* __restrict__ leaf_boxes, const Box6* __restrict__ node_boxes, BvhNode* __restrict__ nodes)
{
    const int i = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (i >= n_inner || !survive[i]) return;
    const uint2 c = child[i];
    BvhNode o;
    const uint32_t refs[2] = { c.x, c.y };
    for (int k = 0; k < 2; ++k) {
        const uint32_t r = refs[k];
        int32_t link;
        const Box6* b;
        if (r & kLeafBit) { const uint32_t p = r & ~kLeafBit; b = &leaf_boxes[p]; link = ~(int32_t)(p << 3); }                       // one triangle: count - 1 == 0
        else if (!survive[r]) { const uint2 rr = range[r]; b = &node_boxes[r]; link = ~(int32_t)(rr.x << 3 | (rr.y - rr.x)); }        // collapsed subtree: its key range, in sorted order
        else { b = &node_boxes[r]; link = (int32_t)newidx[r]; }
        if (k == 0) { pack_box_dev(*b, pad, o.h0); o.child0 = link; } else { pack_box_dev(*b, pad, o.h1); o.child1 = link; }
    }
    nodes[newidx[i]] = o;
}

// triangles in sorted (= leaf) order, with the two edges the reference computes per test (bvh.cpp: emit_leaf)
__global__ __launch_bounds__(kLbvhBlock) void lbvh_emit_tris_kernel(int n, const uint32_t* __restrict__ order, const float* __restrict__ verts, const uint32_t* __restrict__ geom,
                                                                   BvhTri* __restrict__ tris)
{
    const int j = blockIdx.x * kLbvhBlock + threadIdx.x;
    if (j >= n) return;
    const uint32_t t = order[j];
    const float* v = verts + 9ull * t;
    BvhTri r;
    for (int a = 0; a < 3; ++a) { r.v0[a] = v[a]; r.e1[a] = v[3 + a] - v[a]; r.e2[a] = v[6 + a] - v[a]; }
    r.prim = t; r.geom = geom[t]; r.pad = 0u;
    tris[j] = r;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
    template <class T> T* as() const { return (T*)p; }
};

}  // namespace

// Builds `out` on the current device.  Returns false (with `why`) when the device path cannot serve the scene — fewer
// triangles than one leaf holds, a tree deeper than the traversal stack, a HIP error — and the caller builds on the host.
// ms[0]: device time of the build kernels + sort (HIP events), ms[1]: wall time including the uploads and the read-back.
bool build_bvh_device(const float* tri_verts, const uint32_t* tri_geom, uint32_t ntri, Bvh& out, std::string& why, double ms[2])
{
    const auto t_wall = std::chrono::steady_clock::now();
    ms[0] = ms[1] = 0.0;
    uint32_t max_leaf = 1;        // measured (profiles/r02_notes.md): Morton-order subtrees make poor leaves — thai2 28.0 / 28.2 / 29.0 / 30.4 ms per frame at 1 / 2 / 3 / 4
    if (const char* e = std::getenv("MI355RT_MAX_LEAF")) { int v = std::atoi(e); if (v >= 1 && v <= 8) max_leaf = (uint32_t)v; }     // the host builder's knob
    if (ntri <= kBvhMaxLeaf) { why = "scene fits one leaf"; return false; }
    if (ntri >= (1u << 26)) { why = "too many triangles"; return false; }
    out = Bvh();
    // scene bounds on the host (one pass over the vertices; also what the culling and the padding need)
    float smin[3] = { INFINITY, INFINITY, INFINITY }, smax[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < (size_t)ntri * 3; ++i)
        for (int a = 0; a < 3; ++a) { const float x = tri_verts[3 * i + a]; smin[a] = std::fmin(smin[a], x); smax[a] = std::fmax(smax[a], x); }
    std::memcpy(out.scene_min, smin, 12); std::memcpy(out.scene_max, smax, 12);
    const float diag = std::sqrt((smax[0] - smin[0]) * (smax[0] - smin[0]) + (smax[1] - smin[1]) * (smax[1] - smin[1]) + (smax[2] - smin[2]) * (smax[2] - smin[2]));
    const float pad = std::max(diag * 2e-5f, 1e-6f);                       // bvh.cpp: the same conservative padding
    float3 dmin = make_float3(smin[0], smin[1], smin[2]), dinv;
    dinv.x = smax[0] > smin[0] ? 1.0f / (smax[0] - smin[0]) : 0.0f; dinv.y = smax[1] > smin[1] ? 1.0f / (smax[1] - smin[1]) : 0.0f; dinv.z = smax[2] > smin[2] ? 1.0f / (smax[2] - smin[2]) : 0.0f;

    const int n = (int)ntri, n_inner = n - 1;
    const unsigned grid_n = (unsigned)((n + kLbvhBlock - 1) / kLbvhBlock), grid_i = (unsigned)((n_inner + kLbvhBlock - 1) / kLbvhBlock);
    DevBuf d_verts, d_geom, d_boxes, d_keys0, d_keys1, d_vals0, d_order, d_child, d_range, d_pin, d_pleaf, d_lboxes, d_nboxes, d_height, d_flags, d_survive, d_newidx, d_nodes, d_tris, d_tmp;
    if (!d_verts.alloc((size_t)n * 36) || !d_geom.alloc((size_t)n * 4) || !d_boxes.alloc((size_t)n * sizeof(Box6)) || !d_keys0.alloc((size_t)n * 8) || !d_keys1.alloc((size_t)n * 8)
        || !d_vals0.alloc((size_t)n * 4) || !d_order.alloc((size_t)n * 4) || !d_child.alloc((size_t)n_inner * 8) || !d_range.alloc((size_t)n_inner * 8) || !d_pin.alloc((size_t)n_inner * 4)
        || !d_pleaf.alloc((size_t)n * 4) || !d_lboxes.alloc((size_t)n * sizeof(Box6)) || !d_nboxes.alloc((size_t)n_inner * sizeof(Box6)) || !d_height.alloc((size_t)n_inner * 4)
        || !d_flags.alloc((size_t)n_inner * 4) || !d_survive.alloc((size_t)n_inner * 4) || !d_newidx.alloc((size_t)n_inner * 4) || !d_nodes.alloc((size_t)n_inner * sizeof(BvhNode))
        || !d_tris.alloc((size_t)n * sizeof(BvhTri))) { why = "hipMalloc failed"; (void)hipGetLastError(); return false; }
    size_t sort_bytes = 0, scan_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, sort_bytes, d_keys0.as<unsigned long long>(), d_keys1.as<unsigned long long>(), d_vals0.as<uint32_t>(), d_order.as<uint32_t>(), (size_t)n, 0, 63, 0) != hipSuccess
        || rocprim::exclusive_scan(nullptr, scan_bytes, d_survive.as<uint32_t>(), d_newidx.as<uint32_t>(), 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), 0) != hipSuccess
        || !d_tmp.alloc(std::max(sort_bytes, scan_bytes))) { why = "rocprim temporary storage"; (void)hipGetLastError(); return false; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { why = "hipEventCreate failed"; return false; }
    auto fail = [&](const char* what) { why = std::string(what) + ": " + hipGetErrorString(hipGetLastError()); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return false; };
    if (hipMemcpy(d_verts.p, tri_verts, (size_t)n * 36, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_geom.p, tri_geom, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(lbvh_prim_kernel, dim3(grid_n), dim3(kLbvhBlock), 0, 0, d_verts.as<float>(), (uint32_t)n, dmin, dinv, d_boxes.as<Box6>(), d_keys0.as<unsigned long long>(), d_vals0.as<uint32_t>());
    if (rocprim::radix_sort_pairs(d_tmp.p, sort_bytes, d_keys0.as<unsigned long long>(), d_keys1.as<unsigned long long>(), d_vals0.as<uint32_t>(), d_order.as<uint32_t>(), (size_t)n, 0, 63, 0) != hipSuccess) return fail("radix sort");
    hipLaunchKernelGGL(lbvh_hierarchy_kernel, dim3(grid_i), dim3(kLbvhBlock), 0, 0, d_keys1.as<unsigned long long>(), n, d_child.as<uint2>(), d_range.as<uint2>(), d_pin.as<uint32_t>(), d_pleaf.as<uint32_t>());
    (void)hipMemsetAsync(d_flags.p, 0, (size_t)n_inner * 4, 0);
    hipLaunchKernelGGL(lbvh_refit_kernel, dim3(grid_n), dim3(kLbvhBlock), 0, 0, n, max_leaf, d_order.as<uint32_t>(), d_boxes.as<Box6>(), d_child.as<uint2>(), d_range.as<uint2>(),
                       d_pin.as<uint32_t>(), d_pleaf.as<uint32_t>(), d_lboxes.as<Box6>(), d_nboxes.as<Box6>(), d_height.as<uint32_t>(), d_flags.as<uint32_t>());
    hipLaunchKernelGGL(lbvh_survive_kernel, dim3(grid_i), dim3(kLbvhBlock), 0, 0, n_inner, max_leaf, d_range.as<uint2>(), d_survive.as<uint32_t>());
    if (rocprim::exclusive_scan(d_tmp.p, scan_bytes, d_survive.as<uint32_t>(), d_newidx.as<uint32_t>(), 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), 0) != hipSuccess) return fail("scan");
    hipLaunchKernelGGL(lbvh_emit_nodes_kernel, dim3(grid_i), dim3(kLbvhBlock), 0, 0, n_inner, pad, d_child.as<uint2>(), d_range.as<uint2>(), d_survive.as<uint32_t>(), d_newidx.as<uint32_t>(),
                       d_lboxes.as<Box6>(), d_nboxes.as<Box6>(), d_nodes.as<BvhNode>());
    hipLaunchKernelGGL(lbvh_emit_tris_kernel, dim3(grid_n), dim3(kLbvhBlock), 0, 0, n, d_order.as<uint32_t>(), d_verts.as<float>(), d_geom.as<uint32_t>(), d_tris.as<BvhTri>());
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) return fail("build kernels");
    float dev_ms = 0.0f;
    (void)hipEventElapsedTime(&dev_ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    // read back: how many nodes survived, the root's height, then the arrays (the renderer keeps a host copy for the culling
    // boxes and the statistics, like the host build's)
    uint32_t last_idx = 0, last_flag = 0, root_height = 0;
    if (hipMemcpy(&last_idx, d_newidx.as<uint32_t>() + (n_inner - 1), 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&last_flag, d_survive.as<uint32_t>() + (n_inner - 1), 4, hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(&root_height, d_height.p, 4, hipMemcpyDeviceToHost) != hipSuccess) { why = "read-back failed"; return false; }
    const uint32_t nnodes = last_idx + last_flag;
    if (root_height > kBvhMaxDepth) { why = "Morton-order tree deeper than the traversal stack (" + std::to_string(root_height) + " levels)"; return false; }
    if (nnodes == 0) { why = "internal: no surviving node"; return false; }
    out.nodes.resize(nnodes); out.tris.resize((size_t)n);
    if (hipMemcpy(out.nodes.data(), d_nodes.p, (size_t)nnodes * sizeof(BvhNode), hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(out.tris.data(), d_tris.p, (size_t)n * sizeof(BvhTri), hipMemcpyDeviceToHost) != hipSuccess) { why = "read-back failed"; return false; }
    out.root = 0; out.max_depth = root_height;
    for (const BvhNode& nd : out.nodes)
        for (int32_t c : { nd.child0, nd.child1 })
            if (c < 0) { out.leaves++; out.max_leaf = std::max(out.max_leaf, ((uint32_t)~c & 7u) + 1u); }
    ms[0] = dev_ms;
    ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wall).count();
    return true;
}

}  // namespace mi355rt
