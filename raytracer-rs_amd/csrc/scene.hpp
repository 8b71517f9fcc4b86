// scene.hpp — flat host-side scene, the array layout that crosses the C ABI.
// Data model of raytracer_lib/src/scene/mod.rs:12-69 flattened: one triangle soup
// (geometry order = visual-scene node order), one material per geometry.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "vecmath.hpp"

namespace mi355rt {

struct MaterialData {          // scene/mod.rs:63-69; only `diffuse` is read by shading
    uint32_t kind = 0;         // 0 = Diffuse::Color, 1 = Diffuse::TextureId (color.rs:98-108)
    float rgb[3] = { 1000.0f, 0.0f, 1000.0f };   // RGB::default(), color.rs:37-41
    uint32_t tex_id = 0;
    float emissive[3] = { 1000.0f, 0.0f, 1000.0f };
    bool has_specular = false;
    float specular = 0.0f;
    float index_of_refraction = 0.0f;
};
struct LightData { float pos[3]; float color[3]; };            // scene/mod.rs:12-16
struct TextureData { uint32_t width = 0, height = 0; std::vector<float> rgb; };   // texture.rs:6-10, texels = byte/256
struct CameraData { float orientation[16]; float fov_deg; };   // args of Camera::from_orientation_matrix, camera.rs:22-27

struct SceneData {
    std::vector<float> tri_verts;        // ntri*9, world space
    std::vector<uint32_t> tri_geom;      // ntri, geometry (= material) index
    std::vector<MaterialData> materials; // one per geometry
    std::vector<LightData> lights;
    std::vector<TextureData> textures;
    std::vector<CameraData> cameras;
    uint32_t ntri() const { return (uint32_t)tri_geom.size(); }
};

// COLLADA ingest (scene/loaders/colladaloader.rs).  Returns false and fills `err` with the
// message the reference's Result<_, String> would carry (SceneLoadError::to_string()).
bool load_collada_str(const std::string& doc, const char* data_dir, SceneData& out, std::string& err);
bool load_collada_file(const std::string& path, SceneData& out, std::string& err);
// texture.rs:35-49 (PNG -> RGB f32 = byte/256.0)
bool load_png_rgb(const std::string& path, TextureData& out, std::string& err);

// Flat binary scene container used by tests/bench on machines without the .dae files.
bool write_scene_file(const std::string& path, const SceneData& s, std::string& err);
bool read_scene_file(const std::string& path, SceneData& s, std::string& err);

Matrix collada_matrix_to_vecmath(const float* collada16);      // collada_types.rs:76-90

}  // namespace mi355rt
