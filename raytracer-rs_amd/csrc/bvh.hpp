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

struct Bvh {
    std::vector<BvhNode> nodes;      // breadth-first: the top of the tree has the lowest indices
    std::vector<BvhTri> tris;        // leaf order
    int32_t root = 0;                // node index, or a leaf code when the scene is a single leaf
    uint32_t leaves = 0, max_depth = 0, max_leaf = 0;
    float scene_min[3] = { 0, 0, 0 }, scene_max[3] = { 0, 0, 0 };
};

constexpr uint32_t kBvhMaxDepth = 31;     // traversal stack holds this many deferred children
constexpr uint32_t kBvhMaxLeaf = 4;

// tri_verts: ntri*9 world-space floats, tri_geom: ntri geometry indices.
void build_bvh(const float* tri_verts, const uint32_t* tri_geom, uint32_t ntri, Bvh& out);

// The same structure built on the current HIP device (lbvh.hip: Morton order, Karras hierarchy, refit; MI355RT_FLAG_DEVICE_LBVH).
// false + `why`: the device path does not serve this scene (the caller builds on the host).  ms[0] device time, ms[1] wall time.
bool build_bvh_device(const float* tri_verts, const uint32_t* tri_geom, uint32_t ntri, Bvh& out, std::string& why, double ms[2]);

}  // namespace mi355rt
