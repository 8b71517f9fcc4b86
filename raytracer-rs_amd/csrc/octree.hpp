// octree.hpp — the reference's acceleration structure, for the optional "reference-exact" intersector
// (MI355RT_FLAG_OCTREE_SEMANTICS).  Host build following oct_tree_intersector.rs:66-146 (split rule,
// child cube order, triangle/cube SAT test :374-469, extents :315-330) so that the node numbering, the
// leaf triangle lists and therefore every traversal decision are the reference's own; flattened for
// the device.  The fast path of this library does not use it (bvh.hpp).
#pragma once
#include <cstdint>
#include <vector>

namespace mi355rt {

// 48-byte device node: cube + either the index of the first of 8 consecutive children or a triangle list
struct alignas(16) OctNodeFlat {
    float cmin[3]; int32_t first_child;     // >= 0: inner node (children first_child .. first_child+7); -1: leaf
    float cmax[3]; uint32_t tri_first;      // leaf: range in Octree::leaf_tris
    uint32_t tri_count; uint32_t parent;    // parent node index (root: 0) — lets the confirm walk climb without a stack
    uint32_t pad[2];
};
static_assert(sizeof(OctNodeFlat) == 48, "octree node must be 48 bytes");

struct Octree {
    std::vector<OctNodeFlat> nodes;          // node index == cube index, as in the reference (OCT:126-135)
    std::vector<uint32_t> leaf_tris;         // global triangle ids, reference list order
    uint32_t inner = 0, leaves = 0, empty_leaves = 0, max_depth = 0;
};

void build_octree(const float* tri_verts, uint32_t ntri, uint32_t tris_per_leaf, Octree& out);

}  // namespace mi355rt
