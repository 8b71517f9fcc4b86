// bvh.hpp — host-side BVH2 build (binned SAH) and GPU-friendly flattening.
// Replaces the reference's octree build (oct_tree_intersector.rs:66-146) — setup, runs once.
// The traversal over this structure returns the TRUE closest hit, i.e. the semantics of the
// reference's NoAccelerationIntersector (no_acceleration_intersector.rs:13-41): the octree's
// "hit point must lie in the leaf cube" rule (OCT:160-169) is a property of that structure,
// not of the scene, and is not reproduced here ("Octree vs BVH").
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mi355rt {

// 32-byte node holding BOTH child boxes as IEEE half floats (one 2 x 16 B fetch decides both children).
//   h[c] = { min.x | max.x << 16, min.y | max.y << 16, min.z | max.z << 16 } of child c (binary16 bit patterns),
//   child[c] as int32: >= 0 inner node index, < 0 leaf: ~v = first<<3 | count-1
// The boxes only steer the search (the exact f32 triangle test decides every hit), so they may be
// coarse as long as they are conservative: minima are rounded DOWN and maxima UP to half precision
// (after the padding), coordinates beyond the half range become +-inf.  Halving the node halves the
// number of divergent 16-byte fetches per step, which is what bounds the trace kernel.
struct alignas(16) BvhNode { uint32_t h0[3]; int32_t child0; uint32_t h1[3]; int32_t child1; };
static_assert(sizeof(BvhNode) == 32, "node must be 32 bytes");

// 48-byte triangle in leaf order: v0 + the two edges the reference computes per test
// (intersect.rs:65-66: v0v1 = v1 - v0, v0v2 = v2 - v0 — same f32 subtractions, done once).
struct alignas(16) BvhTri { float v0[3]; uint32_t prim; float e1[3]; uint32_t geom; float e2[3]; uint32_t pad; };
static_assert(sizeof(BvhTri) == 48, "triangle must be 48 bytes");

// 4-wide node in 48 bytes (three 16-byte loads; the slot is 64 bytes so that no node straddles a 128-byte line).  The gather microbenchmark
// (tools/micro/gather_bench.hip) prices three loads of a 64-byte slot at 1.23 binary-node visits, and a 4-wide tree needs ~0.55 of the visits.
// The four child boxes are 8-bit offsets in a per-node frame (Ylitie, Karras, Laine 2017): plane = org[a] + q * 2^(e[a] - 127), minima rounded
// down and maxima up from the (already padded, already outward-rounded) half-precision boxes of the binary tree they are collapsed from.
//   q0 = { org.x, org.y, org.z, ex | ey << 8 | ez << 16 | tri_base[19:12] << 24 }
//   q1 = { lo.x[4], hi.x[4], lo.y[4], hi.y[4] }          (byte i = child i)
//   q2 = { lo.z[4], hi.z[4], meta[4], child_base | tri_base[11:0] << 20 }
//   meta byte of an inner child: 0x80 | its place among the node's inner children — they sit at child_base, child_base + 1, ...;
//   of a leaf child: (first triangle - tri_base) << 3 | count - 1 — the node's leaf triangles sit together from tri_base on (leaves of <= 4);
//   an unused slot: an inverted box (lo 255, hi 0) and a leaf byte of 0 (a ray with a null direction passes every box and can hit no triangle).
struct alignas(16) BvhNode4 { float org[3]; uint32_t ew; uint32_t lox, hix, loy, hiy; uint32_t loz, hiz, meta, bases; uint32_t pad[4]; };
static_assert(sizeof(BvhNode4) == 64, "wide node slot must be 64 bytes");

struct Bvh {
    std::vector<BvhNode> nodes;      // breadth-first: the top of the tree has the lowest indices
    std::vector<BvhNode4> nodes4;    // the same tree collapsed to 4-wide nodes (build_wide); empty when it does not apply
    uint32_t stack_need4 = 0;        // most deferred children any root-to-leaf walk of the wide tree can hold
    std::vector<BvhTri> tris;        // leaf order
    int32_t root = 0;                // node index, or a leaf code when the scene is a single leaf
    uint32_t leaves = 0, max_depth = 0, max_leaf = 0;
    float scene_min[3] = { 0, 0, 0 }, scene_max[3] = { 0, 0, 0 };
};

constexpr uint32_t kBvhMaxDepth = 31;     // traversal stack holds this many deferred children
constexpr uint32_t kBvhMaxLeaf = 4;

// tri_verts: ntri*9 world-space floats, tri_geom: ntri geometry indices.
void build_bvh(const float* tri_verts, const uint32_t* tri_geom, uint32_t ntri, Bvh& out);

// Collapses b.nodes into b.nodes4 (re-orders b.tris so that the leaves under one wide node lie together, and re-points the binary leaves).
// Leaves nodes4 empty when the format does not apply (leaves of more than 4 triangles, 2^20 nodes or triangles and beyond, a single-leaf scene).
void build_wide(Bvh& b);

// The same structure built on the current HIP device (lbvh.hip: Morton order, Karras hierarchy, refit; MI355RT_FLAG_DEVICE_LBVH).
// false + `why`: the device path does not serve this scene (the caller builds on the host).  ms[0] device time, ms[1] wall time.
bool build_bvh_device(const float* tri_verts, const uint32_t* tri_geom, uint32_t ntri, Bvh& out, std::string& why, double ms[2]);

}  // namespace mi355rt
