// collada.cpp — COLLADA (.dae) ingest to the flat SceneData.
// Behaviour follows raytracer_lib/src/scene/loaders/colladaloader.rs:
//   Collada::parse :59-135 (strict order of the nine top-level sections),
//   to_scene_flatten :137-273 (per visual-scene node: camera / light / geometry, de-indexing,
//   node matrix baked into vertices, material -> effect resolution),
//   to_cameras :276-321, to_lights :323-350, to_effects :352-466, to_images :468-485,
//   to_materials :487-505, to_visual_scenes :507-551, to_geometries/convert_geometry :553-601,
//   ColladaMatrix::to_vecmath_matrix collada_types.rs:76-90.
// Error strings follow ColladaError's Display (:631-674) with our own detail text, because
// the detail text of the reference comes from the un-vendored `parseval` crate.
#include <cstdlib>
#include <cstdio>
#include <fstream>
#include <sstream>
#include "scene.hpp"
#include "xml_mini.hpp"

namespace mi355rt {
namespace {

struct LoadError { std::string msg; };

[[noreturn]] void fail(const std::string& m) { throw LoadError{ m }; }

const XmlElement& child(const XmlElement& e, const char* name)
{
    const XmlElement* c = e.child_by_name(name);
    if (!c) fail(std::string("ElementError error; no child '") + name + "' in element '" + e.name + "'");
    return *c;
}
const XmlElement& child_attr(const XmlElement& e, const char* key, const std::string& value)
{
    const XmlElement* c = e.child_by_attrib(key, value);
    if (!c) fail(std::string("ElementError error; no child with ") + key + "='" + value + "' in element '" + e.name + "'");
    return *c;
}
const std::string& attr(const XmlElement& e, const char* key)
{
    const std::string* v = e.attrib(key);
    if (!v) fail(std::string("ElementError error; no attribute '") + key + "' in element '" + e.name + "'");
    return *v;
}
const std::string& data_of(const XmlElement& e)
{
    if (!e.has_data) fail("ElementError error; element '" + e.name + "' holds no data");
    return e.data;
}

std::vector<float> array_f32(const std::string& s)
{
    std::vector<float> out;
    const char* p = s.c_str();
    for (;;) {
        char* end = nullptr;
        float v = std::strtof(p, &end);
        if (end == p) break;
        out.push_back(v);
        p = end;
    }
    while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') ++p;
    if (*p != '\0') fail("ParseError error; unexpected character in float array");
    return out;
}
std::vector<uint32_t> array_u32(const std::string& s)
{
    std::vector<uint32_t> out;
    const char* p = s.c_str();
    for (;;) {
        while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') ++p;
        if (*p < '0' || *p > '9') break;
        char* end = nullptr;
        unsigned long v = std::strtoul(p, &end, 10);
        out.push_back((uint32_t)v);
        p = end;
    }
    if (*p != '\0') fail("ParseError error; unexpected character in index array");
    return out;
}
float first_f32(const XmlElement& e, const char* what)
{
    std::vector<float> a = array_f32(data_of(e));
    if (a.empty()) fail(std::string("ParseError error; empty ") + what);
    return a[0];
}

struct CCamera { std::string id; float fov; };
struct CLight { std::string id; float color[3]; };
struct CEffect {
    std::string id;
    float emission[4];
    bool is_tex = false;
    float diffuse[4] = { 0, 0, 0, 0 };
    std::string image_id;
    bool has_specular = false;
    float specular = 0;
    float ior = 0;
};
struct CImage { std::string id, filename; };
struct CMaterial { std::string id, effect_url; };
struct CGeometry { std::vector<float> vertices; std::vector<uint32_t> triangles; std::string id, material_id; };
struct CNode { std::string id; float matrix[16]; };

std::string strip_hash(const std::string& url)   // `[1..]` in the reference
{
    if (url.empty()) fail("ElementError error; empty url");
    return url.substr(1);
}

std::vector<CCamera> to_cameras(const XmlElement& lib)
{
    if (lib.has_data) fail("CamerasConversion error; cant convert cameras");
    std::vector<CCamera> out;
    for (auto& c : lib.children) {
        CCamera cam;
        cam.id = attr(*c, "id");
        const XmlElement& persp = child(child(child(*c, "optics"), "technique_common"), "perspective");
        const XmlElement& fov = child(persp, "xfov");
        const XmlElement& aspect = child(persp, "aspect_ratio");
        if (!fov.has_data) fail("CamerasConversion error; cant read fov");
        cam.fov = first_f32(fov, "fov");
        if (!aspect.has_data) fail("CamerasConversion error; cant read aspect_ratio");
        (void)first_f32(aspect, "aspect_ratio");     // parsed and discarded, camera.rs:41-44
        out.push_back(cam);
    }
    return out;
}
std::vector<CLight> to_lights(const XmlElement& lib)
{
    if (lib.has_data) fail("LightsConversion error; cant convert lights");
    std::vector<CLight> out;
    for (auto& l : lib.children) {
        CLight li;
        li.id = attr(*l, "id");
        const XmlElement& col = child(child(child(*l, "technique_common"), "point"), "color");
        if (!col.has_data) fail("LightsConversion error; cant get color");
        std::vector<float> a = array_f32(col.data);
        if (a.size() < 3) fail("LightsConversion error; cant get color");
        li.color[0] = a[0]; li.color[1] = a[1]; li.color[2] = a[2];
        out.push_back(li);
    }
    return out;
}
std::vector<CEffect> to_effects(const XmlElement& lib)
{
    if (lib.has_data) fail("EffectsConversion error; can't convert effects");
    std::vector<CEffect> out;
    for (auto& e : lib.children) {
        CEffect ef;
        ef.id = attr(*e, "id");
        const XmlElement& profile = child(*e, "profile_COMMON");
        const XmlElement& lambert = child(child(profile, "technique"), "lambert");
        {
            const XmlElement& col = child(child(lambert, "emission"), "color");
            if (!col.has_data) fail("EffectsConversion error; Can't get emission color");
            std::vector<float> a = array_f32(col.data);
            if (a.size() < 4) fail("EffectsConversion error; Can't get emission color");
            for (int i = 0; i < 4; ++i) ef.emission[i] = a[i];
        }
        const XmlElement& diffuse = child(lambert, "diffuse");
        if (const XmlElement* col = diffuse.child_by_name("color")) {
            if (!col->has_data) fail("EffectsConversion error; Cant get diffuse color");
            std::vector<float> a = array_f32(col->data);
            if (a.size() < 4) fail("EffectsConversion error; Cant get diffuse color");
            for (int i = 0; i < 4; ++i) ef.diffuse[i] = a[i];
        } else {
            // texture (sampler name) -> sampler -> surface -> image id
            const XmlElement& tex = child(diffuse, "texture");
            (void)attr(tex, "texcoord");
            std::string sampler = attr(tex, "texture");
            const XmlElement& src = child(child(child_attr(profile, "sid", sampler), "sampler2D"), "source");
            if (!src.has_data) fail("EffectsConversion error; Cant get sampler");
            std::string surface = src.data;
            const XmlElement& init = child(child(child_attr(profile, "sid", surface), "surface"), "init_from");
            if (!init.has_data) fail("EffectsConversion error; Cant get surface");
            ef.is_tex = true;
            ef.image_id = init.data;
        }
        {
            const XmlElement& ior = child_attr(child(lambert, "index_of_refraction"), "sid", "ior");
            if (!ior.has_data) fail("EffectsConversion error; Can't get index of refraction");
            ef.ior = first_f32(ior, "index of refraction");
        }
        if (const XmlElement* refl = lambert.child_by_name("reflectivity")) {
            const XmlElement& sp = child_attr(*refl, "sid", "specular");
            if (!sp.has_data) fail("EffectsConversion error; Can't get specular");
            ef.has_specular = true;
            ef.specular = first_f32(sp, "specular");
        }
        out.push_back(ef);
    }
    return out;
}
std::vector<CImage> to_images(const XmlElement& lib)
{
    if (lib.has_data) fail("ImagesConversion error; can't convert images");
    std::vector<CImage> out;
    for (auto& i : lib.children) {
        CImage im;
        im.id = attr(*i, "id");
        im.filename = data_of(child(*i, "init_from"));
        out.push_back(im);
    }
    return out;
}
std::vector<CMaterial> to_materials(const XmlElement& lib)
{
    if (lib.has_data) fail("MaterialsConversion error; can't convert materials");
    std::vector<CMaterial> out;
    for (auto& m : lib.children) {
        CMaterial ma;
        ma.id = attr(*m, "id");
        ma.effect_url = strip_hash(attr(child(*m, "instance_effect"), "url"));
        out.push_back(ma);
    }
    return out;
}
CGeometry convert_geometry(const XmlElement& g)
{
    CGeometry out;
    out.id = attr(g, "id");
    const XmlElement& mesh = child(g, "mesh");
    const XmlElement& arr = child_attr(child_attr(mesh, "id", out.id + "-positions"), "id", out.id + "-positions-array");
    out.vertices = array_f32(data_of(arr));
    const XmlElement& tris = child(mesh, "triangles");
    out.material_id = attr(tris, "material");
    std::vector<uint32_t> idx = array_u32(data_of(child(tris, "p")));
    // (position, normal, texcoord) index triples: keep every third index (assumes exactly 3 inputs)
    if (idx.size() % 3 != 0) fail("GeometryConversion error");
    for (size_t i = 0; i < idx.size(); i += 3) out.triangles.push_back(idx[i]);
    return out;
}
std::vector<CGeometry> to_geometries(const XmlElement& lib)
{
    if (lib.has_data) fail("GeometryConversion error");
    std::vector<CGeometry> out;
    for (auto& g : lib.children) out.push_back(convert_geometry(*g));
    return out;
}
std::vector<CNode> to_visual_scenes(const XmlElement& lib)
{
    if (lib.has_data) fail("VisualSceneConversion error; No scene element(s)");
    std::vector<CNode> nodes;      // every <visual_scene> is merged into one node list
    for (auto& scene : lib.children) {
        if (scene->has_data) continue;
        for (auto& n : scene->children) {
            const XmlElement* il = n->child_by_name("instance_light");
            const XmlElement* ig = n->child_by_name("instance_geometry");
            const XmlElement* ic = n->child_by_name("instance_camera");
            const XmlElement* inst = il ? il : (ig ? ig : ic);      // precedence light > geometry > camera
            if (!inst) fail("VisualSceneConversion error; unsupported node type");
            CNode node;
            node.id = strip_hash(attr(*inst, "url"));
            const XmlElement& m = child(*n, "matrix");
            if (!m.has_data) continue;
            std::vector<float> a = array_f32(m.data);
            if (a.size() < 16) fail("VisualSceneConversion error; cant create array");
            for (int i = 0; i < 16; ++i) node.matrix[i] = a[i];
            nodes.push_back(node);
        }
    }
    return nodes;
}

const XmlElement& section(const XmlElement& root, size_t pos, const char* name, const char* errname)
{
    if (pos >= root.children.size() || root.children[pos]->name != name)
        fail(std::string(errname) + " error; expected element '" + name + "'");
    return *root.children[pos];
}

void load_impl(const std::string& doc, const char* data_dir, SceneData& out)
{
    XmlDoc xml;
    std::string perr;
    if (!xml_parse(doc, xml, perr)) {
        if (!xml.has_definition && doc.find("<?xml") == std::string::npos) fail("XmlDefinition error; " + perr);
        fail("ColladaElement error; " + perr);
    }
    if (!xml.has_definition) fail("XmlDefinition error; missing <?xml ... ?>");
    const XmlElement& root = *xml.root;
    if (root.name != "COLLADA") fail("Not a collada doc");
    const XmlElement& cams_e = (section(root, 0, "asset", "AssetParsing"), section(root, 1, "library_cameras", "LibraryCamerasParsing"));
    const XmlElement& lights_e = section(root, 2, "library_lights", "LibraryLightsParsing");
    const XmlElement& effects_e = section(root, 3, "library_effects", "LibraryEffectsParsing");
    const XmlElement& images_e = section(root, 4, "library_images", "LibraryImagesParsing");
    const XmlElement& materials_e = section(root, 5, "library_materials", "LibraryMaterialsParsing");
    const XmlElement& geoms_e = section(root, 6, "library_geometries", "LibraryGeometriesParsing");
    const XmlElement& vscenes_e = section(root, 7, "library_visual_scenes", "LibraryVisualScenesParsing");
    section(root, 8, "scene", "LibrarySceneParsing");
    if (root.children.size() > 9) fail("ColladaElement error; expected closing element 'COLLADA'");
    if (!xml.remaining.empty()) fail("RemainingData error; " + xml.remaining);

    std::vector<CCamera> cameras = to_cameras(cams_e);
    std::vector<CLight> lights = to_lights(lights_e);
    std::vector<CEffect> effects = to_effects(effects_e);
    std::vector<CImage> images = to_images(images_e);
    std::vector<CMaterial> materials = to_materials(materials_e);
    std::vector<CGeometry> geometries = to_geometries(geoms_e);
    std::vector<CNode> nodes = to_visual_scenes(vscenes_e);

    out = SceneData();
    for (auto& im : images) {
        std::string path = data_dir && *data_dir ? std::string(data_dir) + "/" + im.filename : im.filename;
        TextureData t;
        std::string terr;
        if (!load_png_rgb(path, t, terr)) fail(terr);
        out.textures.push_back(std::move(t));
    }
    for (auto& node : nodes) {
        Matrix m = collada_matrix_to_vecmath(node.matrix);
        for (auto& c : cameras) {
            if (c.id != node.id) continue;
            CameraData cd;
            std::memcpy(cd.orientation, m.e, sizeof cd.orientation);
            cd.fov_deg = c.fov;
            out.cameras.push_back(cd);
            break;
        }
        for (auto& l : lights) {
            if (l.id != node.id) continue;
            Vec3 p = (m * Vec4::from_vec3(Vec3(0.0f, 0.0f, 0.0f))).xyz();
            LightData ld = { { p.x, p.y, p.z }, { l.color[0], l.color[1], l.color[2] } };
            out.lights.push_back(ld);
            break;
        }
        for (auto& g : geometries) {
            if (g.id != node.id) continue;
            uint32_t geom_index = (uint32_t)out.materials.size();
            for (uint32_t vi : g.triangles) {
                if ((size_t)3 * vi + 2 >= g.vertices.size()) fail("GeometryConversion error");
                Vec3 v(g.vertices[3 * vi], g.vertices[3 * vi + 1], g.vertices[3 * vi + 2]);
                Vec3 w = (m * Vec4::from_vec3(v)).xyz();
                out.tri_verts.push_back(w.x); out.tri_verts.push_back(w.y); out.tri_verts.push_back(w.z);
            }
            size_t ntri = g.triangles.size() / 3;      // chunks(3): a ragged tail would panic in the reference
            if (g.triangles.size() % 3 != 0) fail("GeometryConversion error");
            for (size_t t = 0; t < ntri; ++t) out.tri_geom.push_back(geom_index);

            MaterialData md;                            // Material::default()
            for (auto& ma : materials) {
                if (ma.id != g.material_id) continue;
                for (auto& ef : effects) {
                    if (ef.id != ma.effect_url) continue;
                    if (ef.is_tex) {
                        size_t pos = images.size();
                        for (size_t i = 0; i < images.size(); ++i) if (images[i].id == ef.image_id) { pos = i; break; }
                        if (pos == images.size()) fail("MaterialsConversion error; can't find texture name");
                        md.kind = 1; md.tex_id = (uint32_t)pos;
                    } else {
                        md.kind = 0;
                        md.rgb[0] = ef.diffuse[0]; md.rgb[1] = ef.diffuse[1]; md.rgb[2] = ef.diffuse[2];
                    }
                    md.emissive[0] = ef.emission[0]; md.emissive[1] = ef.emission[1]; md.emissive[2] = ef.emission[2];
                    md.has_specular = ef.has_specular; md.specular = ef.specular;
                    md.index_of_refraction = ef.ior;
                    break;
                }
                break;
            }
            out.materials.push_back(md);
            break;
        }
    }
}

}  // namespace

Matrix collada_matrix_to_vecmath(const float* c)
{
    // COLLADA: right handed, Z up, column-major-in-rows (translation in the 4th column).
    Matrix row_major = Matrix::from_array(c).transpose();
    static const float swap_yz[16] = { 1, 0, 0, 0,  0, 0, 1, 0,  0, 1, 0, 0,  0, 0, 0, 1 };
    static const float reflect_z[16] = { 1, 0, 0, 0,  0, 1, 0, 0,  0, 0, -1, 0,  0, 0, 0, 1 };
    return Matrix::from_array(reflect_z) * row_major * Matrix::from_array(swap_yz);
}

bool load_collada_str(const std::string& doc, const char* data_dir, SceneData& out, std::string& err)
{
    try {
        load_impl(doc, data_dir, out);
        return true;
    } catch (const LoadError& e) {
        err = e.msg;
        return false;
    }
}

bool load_collada_file(const std::string& path, SceneData& out, std::string& err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "No such file or directory (os error 2)"; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string dir;
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) dir = path.substr(0, slash);
    return load_collada_str(ss.str(), dir.c_str(), out, err);
}

// ---------------------------------------------------------------- flat scene container
namespace {
const char kMagic[8] = { 'M', '3', '5', '5', 'S', 'C', 'N', '1' };
template <class T> void put(std::ofstream& f, const T& v) { f.write(reinterpret_cast<const char*>(&v), sizeof v); }
template <class T> bool get(std::ifstream& f, T& v) { return (bool)f.read(reinterpret_cast<char*>(&v), sizeof v); }
}

bool write_scene_file(const std::string& path, const SceneData& s, std::string& err)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    f.write(kMagic, 8);
    put<uint32_t>(f, s.ntri()); put<uint32_t>(f, (uint32_t)s.materials.size()); put<uint32_t>(f, (uint32_t)s.lights.size());
    put<uint32_t>(f, (uint32_t)s.textures.size()); put<uint32_t>(f, (uint32_t)s.cameras.size());
    f.write(reinterpret_cast<const char*>(s.tri_verts.data()), s.tri_verts.size() * sizeof(float));
    f.write(reinterpret_cast<const char*>(s.tri_geom.data()), s.tri_geom.size() * sizeof(uint32_t));
    for (auto& m : s.materials) { put(f, m.kind); f.write(reinterpret_cast<const char*>(m.rgb), 12); put(f, m.tex_id); }
    for (auto& l : s.lights) { f.write(reinterpret_cast<const char*>(l.pos), 12); f.write(reinterpret_cast<const char*>(l.color), 12); }
    for (auto& c : s.cameras) { f.write(reinterpret_cast<const char*>(c.orientation), 64); put(f, c.fov_deg); }
    for (auto& t : s.textures) {
        // texels that are exactly byte/256 (every PNG-loaded texture) are stored as bytes
        bool bytes = true;
        for (float v : t.rgb) { float b = v * 256.0f; if (!(b >= 0.0f && b <= 255.0f && b == (float)(int)b)) { bytes = false; break; } }
        put(f, t.width); put(f, t.height); put<uint32_t>(f, bytes ? 1u : 0u);
        if (bytes) {
            std::vector<unsigned char> q(t.rgb.size());
            for (size_t i = 0; i < q.size(); ++i) q[i] = (unsigned char)(t.rgb[i] * 256.0f);
            f.write(reinterpret_cast<const char*>(q.data()), q.size());
        } else {
            f.write(reinterpret_cast<const char*>(t.rgb.data()), t.rgb.size() * sizeof(float));
        }
    }
    return (bool)f;
}

bool read_scene_file(const std::string& path, SceneData& s, std::string& err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    char magic[8];
    if (!f.read(magic, 8) || std::memcmp(magic, kMagic, 8) != 0) { err = "not a scene file: " + path; return false; }
    uint32_t ntri, nmat, nlight, ntex, ncam;
    if (!get(f, ntri) || !get(f, nmat) || !get(f, nlight) || !get(f, ntex) || !get(f, ncam)) { err = "truncated scene file"; return false; }
    s = SceneData();
    s.tri_verts.resize((size_t)ntri * 9); s.tri_geom.resize(ntri);
    f.read(reinterpret_cast<char*>(s.tri_verts.data()), s.tri_verts.size() * sizeof(float));
    f.read(reinterpret_cast<char*>(s.tri_geom.data()), s.tri_geom.size() * sizeof(uint32_t));
    s.materials.resize(nmat); s.lights.resize(nlight); s.cameras.resize(ncam); s.textures.resize(ntex);
    for (auto& m : s.materials) { get(f, m.kind); f.read(reinterpret_cast<char*>(m.rgb), 12); get(f, m.tex_id); }
    for (auto& l : s.lights) { f.read(reinterpret_cast<char*>(l.pos), 12); f.read(reinterpret_cast<char*>(l.color), 12); }
    for (auto& c : s.cameras) { f.read(reinterpret_cast<char*>(c.orientation), 64); get(f, c.fov_deg); }
    for (auto& t : s.textures) {
        uint32_t bytes = 0;
        get(f, t.width); get(f, t.height); get(f, bytes);
        t.rgb.resize((size_t)t.width * t.height * 3);
        if (bytes) {
            std::vector<unsigned char> q(t.rgb.size());
            f.read(reinterpret_cast<char*>(q.data()), q.size());
            for (size_t i = 0; i < q.size(); ++i) t.rgb[i] = (float)q[i] / 256.0f;
        } else {
            f.read(reinterpret_cast<char*>(t.rgb.data()), t.rgb.size() * sizeof(float));
        }
    }
    if (!f) { err = "truncated scene file"; return false; }
    return true;
}

}  // namespace mi355rt
