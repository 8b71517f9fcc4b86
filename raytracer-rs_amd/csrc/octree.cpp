// octree.cpp — see octree.hpp.  f32 arithmetic in the reference's operation order (built with
// -ffp-contract=off): the SAT decisions define which leaf holds which triangle.
#include "octree.hpp"
#include <cmath>
#include <cstring>
#include "vecmath.hpp"

namespace mi355rt {
namespace {

struct Cube { Vec3 mn, mx; };
struct BuildNode { bool leaf = true; uint32_t child[8] = { 0 }; std::vector<uint32_t> tris; };

void project(const Vec3* pts, int n, const Vec3& axis, float& lo, float& hi)     // OCT:460-469
{
    lo = 3.40282347e+38f; hi = -3.40282347e+38f;
    for (int i = 0; i < n; ++i) {
        float val = dot(axis, pts[i]);
        lo = std::fmin(lo, val);
        hi = std::fmax(hi, val);
    }
}

bool triangle_cube_intersection(const Cube& c, const Vec3* tv)                 // OCT:393-458
{
    const Vec3 xa(1, 0, 0), ya(0, 1, 0), za(0, 0, 1);
    float tmin, tmax, cmin, cmax;
    project(tv, 3, xa, tmin, tmax);
    if (tmax < c.mn.x || tmin > c.mx.x) return false;
    project(tv, 3, ya, tmin, tmax);
    if (tmax < c.mn.y || tmin > c.mx.y) return false;
    project(tv, 3, za, tmin, tmax);
    if (tmax < c.mn.z || tmin > c.mx.z) return false;
    const Vec3 cv[8] = { c.mn, Vec3(c.mx.x, c.mn.y, c.mn.z), Vec3(c.mn.x, c.mx.y, c.mn.z), Vec3(c.mn.x, c.mn.y, c.mx.z),
                         Vec3(c.mn.x, c.mx.y, c.mx.z), Vec3(c.mx.x, c.mn.y, c.mx.z), Vec3(c.mx.x, c.mx.y, c.mn.z), c.mx };
    const Vec3 e1 = tv[0] - tv[1], e2 = tv[1] - tv[2];
    const Vec3 n = cross(e1, e2);
    const float off = dot(n, tv[0]);
    project(cv, 8, n, cmin, cmax);
    if (cmax < off || cmin > off) return false;
    const Vec3 e3 = tv[2] - tv[0];
    const Vec3 axes[9] = { cross(e1, xa), cross(e1, ya), cross(e1, za), cross(e2, xa), cross(e2, ya), cross(e2, za),
                           cross(e3, xa), cross(e3, ya), cross(e3, za) };
    for (const Vec3& a : axes) {
        project(cv, 8, a, cmin, cmax);
        project(tv, 3, a, tmin, tmax);
        if (cmax < tmin || cmin > tmax) return false;
    }
    return true;
}

struct Builder {
    const Vec3* verts;
    uint32_t tris_per_leaf;
    std::vector<Cube> cubes;
    std::vector<BuildNode> nodes;

    void split(uint32_t idx, uint32_t level)                                    // OCT:94-146
    {
        if (!nodes[idx].leaf) return;
        if (nodes[idx].tris.size() <= tris_per_leaf || level > 8) return;
        const Cube c = cubes[idx];
        const Vec3 mid = 0.5f * (c.mx + c.mn);                                 // OCT:275
        const Vec3 mn = c.mn, mx = c.mx;
        const Cube kids[8] = {                                                 // OCT:279-312
            { Vec3(mn.x, mn.y, mn.z), Vec3(mid.x, mid.y, mid.z) }, { Vec3(mid.x, mn.y, mn.z), Vec3(mx.x, mid.y, mid.z) },
            { Vec3(mn.x, mid.y, mn.z), Vec3(mid.x, mx.y, mid.z) }, { Vec3(mid.x, mid.y, mn.z), Vec3(mx.x, mx.y, mid.z) },
            { Vec3(mn.x, mn.y, mid.z), Vec3(mid.x, mid.y, mx.z) }, { Vec3(mid.x, mn.y, mid.z), Vec3(mx.x, mid.y, mx.z) },
            { Vec3(mn.x, mid.y, mid.z), Vec3(mid.x, mx.y, mx.z) }, { Vec3(mid.x, mid.y, mid.z), Vec3(mx.x, mx.y, mx.z) },
        };
        std::vector<uint32_t> parent;
        parent.swap(nodes[idx].tris);
        uint32_t first = (uint32_t)nodes.size();
        for (int i = 0; i < 8; ++i) {
            BuildNode child;
            for (uint32_t t : parent)
                if (triangle_cube_intersection(kids[i], &verts[3 * (size_t)t])) child.tris.push_back(t);
            cubes.push_back(kids[i]);
            nodes.push_back(std::move(child));
        }
        nodes[idx].leaf = false;
        for (int i = 0; i < 8; ++i) nodes[idx].child[i] = first + i;
        for (int i = 0; i < 8; ++i) split(first + i, level + 1);
    }
};

void depth_rec(const std::vector<BuildNode>& nodes, uint32_t idx, uint32_t depth, Octree& out)
{
    if (nodes[idx].leaf) {
        out.leaves++;
        if (nodes[idx].tris.empty()) out.empty_leaves++;
        if (depth > out.max_depth) out.max_depth = depth;
        return;
    }
    out.inner++;
    for (int i = 0; i < 8; ++i) depth_rec(nodes, nodes[idx].child[i], depth + 1, out);
}

}  // namespace

void build_octree(const float* tri_verts, uint32_t ntri, uint32_t tris_per_leaf, Octree& out)
{
    out = Octree();
    Builder b;
    b.verts = reinterpret_cast<const Vec3*>(tri_verts);
    b.tris_per_leaf = tris_per_leaf;
    Cube trunk{ Vec3(3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f), Vec3(-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f) };
    for (size_t i = 0; i < (size_t)ntri * 3; ++i) {                             // calc_extents, OCT:315-330
        const Vec3& p = b.verts[i];
        trunk.mn.x = std::fmin(trunk.mn.x, p.x); trunk.mn.y = std::fmin(trunk.mn.y, p.y); trunk.mn.z = std::fmin(trunk.mn.z, p.z);
        trunk.mx.x = std::fmax(trunk.mx.x, p.x); trunk.mx.y = std::fmax(trunk.mx.y, p.y); trunk.mx.z = std::fmax(trunk.mx.z, p.z);
    }
    BuildNode root;
    root.tris.resize(ntri);
    for (uint32_t i = 0; i < ntri; ++i) root.tris[i] = i;
    b.cubes.push_back(trunk);
    b.nodes.push_back(std::move(root));
    b.split(0, 0);
    out.nodes.resize(b.nodes.size());
    for (size_t i = 0; i < b.nodes.size(); ++i) {
        OctNodeFlat& f = out.nodes[i];
        std::memset(&f, 0, sizeof f);
        f.cmin[0] = b.cubes[i].mn.x; f.cmin[1] = b.cubes[i].mn.y; f.cmin[2] = b.cubes[i].mn.z;
        f.cmax[0] = b.cubes[i].mx.x; f.cmax[1] = b.cubes[i].mx.y; f.cmax[2] = b.cubes[i].mx.z;
        if (b.nodes[i].leaf) {
            f.first_child = -1;
            f.tri_first = (uint32_t)out.leaf_tris.size();
            f.tri_count = (uint32_t)b.nodes[i].tris.size();
            out.leaf_tris.insert(out.leaf_tris.end(), b.nodes[i].tris.begin(), b.nodes[i].tris.end());
        } else {
            f.first_child = (int32_t)b.nodes[i].child[0];
        }
    }
    // parents are written in a second pass (a child's record is cleared when its own turn comes in the loop above)
    for (size_t i = 0; i < b.nodes.size(); ++i)
        if (!b.nodes[i].leaf) for (int k = 0; k < 8; ++k) out.nodes[b.nodes[i].child[k]].parent = (uint32_t)i;
    depth_rec(b.nodes, 0, 0, out);
}

}  // namespace mi355rt
