// dae2scene — converts a COLLADA file to the flat scene container (scene.hpp) and prints
// the scene facts used by the tests.  Usage: dae2scene in.dae out.scene
#include <cstdio>
#include "../scene.hpp"
using namespace mi355rt;
int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s in.dae out.scene\n", argv[0]); return 2; }
    SceneData s; std::string err;
    if (!load_collada_file(argv[1], s, err)) { std::fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
    std::printf("number of triangles: %u\n", s.ntri());
    float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
    for (size_t i = 0; i < s.tri_verts.size(); ++i) { int a = i % 3; if (s.tri_verts[i] < mn[a]) mn[a] = s.tri_verts[i]; if (s.tri_verts[i] > mx[a]) mx[a] = s.tri_verts[i]; }
    std::printf("bbox (%.4f,%.4f,%.4f) -> (%.4f,%.4f,%.4f)\n", mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]);
    std::printf("geometries %zu lights %zu cameras %zu textures %zu\n", s.materials.size(), s.lights.size(), s.cameras.size(), s.textures.size());
    for (auto& l : s.lights) std::printf("light pos (%.4f,%.4f,%.4f) color (%g,%g,%g)\n", l.pos[0], l.pos[1], l.pos[2], l.color[0], l.color[1], l.color[2]);
    for (auto& c : s.cameras) std::printf("camera pos (%.4f,%.4f,%.4f) fov %g\n", c.orientation[12], c.orientation[13], c.orientation[14], c.fov_deg);
    if (!write_scene_file(argv[2], s, err)) { std::fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
    return 0;
}
