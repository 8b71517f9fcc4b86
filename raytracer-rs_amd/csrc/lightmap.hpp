// lightmap.hpp — conservative depth cube maps around the point lights (round 3).
// Every point of a shadow ray (mod.rs:224-225: from the shaded point towards the light) lies in ONE direction from the light.  The map of a
// light holds, per direction texel of a cube map around it, a LOWER bound of the squared distance from the light to any triangle seen in that
// direction.  A shading point whose own distance, shortened by the ray's 1 % offset, is below that bound cannot have anything between it and the
// light: its shadow ray hits nothing and need not be traced (kernels.hip, light_proves_unoccluded).  Like the BVH's boxes and the culling mask
// the map only ever removes work whose outcome is known; it is padded like they are.
#pragma once
#include <cstdint>
#include <vector>

namespace mi355rt {

struct LightMap {
    uint32_t res = 0;                 // texels per face edge
    std::vector<float> dist2;         // [6][res][res] (face, i over axis a, j over axis b): lower bound of the squared distance, +inf where nothing is seen
    double nearest = 0.0;             // distance from the light to the nearest triangle (unpadded)
};

// tri_verts: ntri * 9 world-space floats; pad: world-space padding of every triangle (what the triangle test's rounding may add to it)
void build_light_map(const float* tri_verts, uint32_t ntri, const float light[3], double pad, uint32_t res, LightMap& out);
// estimate of the texel updates build_light_map makes at resolution `res` (to pick a resolution the build can afford)
// distance from point p to the triangle (v0, v1, v2) given as 9 floats, in double; a degenerate triangle gets a LOWER bound
double point_triangle_distance_lower(const double p[3], const float* tri9);
uint64_t light_map_work(const float* tri_verts, uint32_t ntri, const float light[3], uint32_t res);

}  // namespace mi355rt
