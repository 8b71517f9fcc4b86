// capi.cpp — the extern "C" surface declared in include/mi355rt.h.
#include <cstring>
#include <memory>
#include <string>
#include "lightmap.hpp"
#include "../../include/mi355rt.h"
#include "group.hpp"
#include "renderer.hpp"

using namespace mi355rt;

namespace mi355rt { bool comm_unique_id(uint8_t* id128, std::string& err); }

// A handle is a device group (csrc/group.hpp): one renderer per HIP device of this process, usually exactly one.
// `r` is the primary renderer (device 0 of the group): entry points that do not depend on the decomposition use it.
struct mi355rt_handle {
    std::unique_ptr<DeviceGroup> g;
    Renderer* r;
    std::string error;      // error text of group-level failures
};

namespace {
thread_local std::string g_create_error;

int finish_create(const SceneData& scene, const mi355rt_config* cfg, mi355rt_handle** out)
{
    int code = MI355RT_E_INVALID;
    std::string err;
    std::unique_ptr<DeviceGroup> g = DeviceGroup::create(scene, *cfg, err, code);
    if (!g) { g_create_error = err; return code; }
    Renderer* r = g->primary();
    *out = new mi355rt_handle{ std::move(g), r, std::string() };
    return MI355RT_OK;
}
int bad(const char* msg) { g_create_error = msg; return MI355RT_E_INVALID; }
}  // namespace

extern "C" {

void mi355rt_default_config(mi355rt_config* cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->width = 1024; cfg->height = 768;                         // DEFAULT_WIDTH / DEFAULT_HEIGHT, main.rs:13-14
    cfg->triangles_per_leaf = MI355RT_DEFAULT_TRIANGLES_PER_LEAF;
    cfg->recursions = 2; cfg->spread = 1;                         // mod.rs:81-82
    cfg->seed = 1; cfg->device = 0;
    cfg->stripe_rows = MI355RT_DEFAULT_STRIPE_ROWS; cfg->stripe_rank = 0; cfg->stripe_world = 1;     // 4-row stripes: measured against 8 / 2 / 1 (profiles/r02_notes.md)
    cfg->device_count = 1;
}

int mi355rt_create(const mi355rt_scene_desc* s, const mi355rt_config* cfg, mi355rt_handle** out)
{
    if (!s || !cfg || !out) return bad("null argument");
    *out = nullptr;
    if ((s->ntri && (!s->tri_verts || !s->tri_geom)) || (s->nmaterials && !s->materials) || (s->nlights && !s->lights) || (s->ntextures && !s->textures))
        return bad("null array with a non-zero count");
    SceneData sd;
    sd.tri_verts.assign(s->tri_verts, s->tri_verts + (size_t)s->ntri * 9);
    sd.tri_geom.assign(s->tri_geom, s->tri_geom + s->ntri);
    for (uint32_t i = 0; i < s->nmaterials; ++i) {
        MaterialData m;
        m.kind = s->materials[i].kind; std::memcpy(m.rgb, s->materials[i].rgb, 12); m.tex_id = s->materials[i].tex_id;
        if (m.kind > 1) return bad("material kind must be 0 (colour) or 1 (texture)");
        sd.materials.push_back(m);
    }
    for (uint32_t i = 0; i < s->nlights; ++i) {
        LightData l;
        std::memcpy(l.pos, s->lights[i].pos, 12); std::memcpy(l.color, s->lights[i].color, 12);
        sd.lights.push_back(l);
    }
    for (uint32_t i = 0; i < s->ntextures; ++i) {
        TextureData t;
        t.width = s->textures[i].width; t.height = s->textures[i].height;
        if (!s->textures[i].rgb || !t.width || !t.height) return bad("empty texture");
        t.rgb.assign(s->textures[i].rgb, s->textures[i].rgb + (size_t)t.width * t.height * 3);
        sd.textures.push_back(std::move(t));
    }
    CameraData c;
    std::memcpy(c.orientation, s->camera_orientation, 64); c.fov_deg = s->camera_fov_deg;
    sd.cameras.push_back(c);
    return finish_create(sd, cfg, out);
}

int mi355rt_create_from_collada_str(const char* doc, size_t len, const char* data_dir, const mi355rt_config* cfg, mi355rt_handle** out)
{
    if (!doc || !cfg || !out) return bad("null argument");
    *out = nullptr;
    SceneData sd; std::string err;
    if (!load_collada_str(std::string(doc, len), data_dir, sd, err)) { g_create_error = err; return MI355RT_E_LOAD; }
    return finish_create(sd, cfg, out);
}

int mi355rt_create_from_collada_file(const char* path, const mi355rt_config* cfg, mi355rt_handle** out)
{
    if (!path || !cfg || !out) return bad("null argument");
    *out = nullptr;
    SceneData sd; std::string err;
    if (!load_collada_file(path, sd, err)) { g_create_error = err; return MI355RT_E_LOAD; }
    return finish_create(sd, cfg, out);
}

int mi355rt_create_from_scene_file(const char* path, const mi355rt_config* cfg, mi355rt_handle** out)
{
    if (!path || !cfg || !out) return bad("null argument");
    *out = nullptr;
    SceneData sd; std::string err;
    if (!read_scene_file(path, sd, err)) { g_create_error = err; return MI355RT_E_LOAD; }
    return finish_create(sd, cfg, out);
}

void mi355rt_destroy(mi355rt_handle* h) { delete h; }

const char* mi355rt_last_error(const mi355rt_handle* h) { return h ? h->g->last_error().c_str() : g_create_error.c_str(); }

uint32_t mi355rt_trace_frame_additive(mi355rt_handle* h) { return h ? h->g->trace_frame_additive() : 0u; }

int mi355rt_render(mi355rt_handle* h, uint32_t spp, mi355rt_ray_counts* counts)
{
    if (!h) return MI355RT_E_INVALID;
    bool ok = h->g->render(spp);
    if (counts) { if (h->g->size() == 1) *counts = h->r->counts; else (void)h->g->last_counts(*counts); }
    return ok ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_render_async(mi355rt_handle* h, uint32_t spp)
{
    if (!h) return MI355RT_E_INVALID;
    return h->g->render(spp, false) ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_last_counts(mi355rt_handle* h, mi355rt_ray_counts* counts)
{
    if (!h || !counts) return MI355RT_E_INVALID;
    return h->g->last_counts(*counts) ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_get_tonemapped_pixels(mi355rt_handle* h, uint32_t* out, size_t n)
{
    if (!h) return MI355RT_E_INVALID;
    return h->g->get_tonemapped(out, n) ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_tonemap_owned_rows_device(mi355rt_handle* h, uint32_t* device_out, size_t n)
{
    if (!h) return MI355RT_E_INVALID;
    if (h->g->size() > 1) { h->r->last_error = "not available on a device group: use mi355rt_get_tonemapped_pixels"; return MI355RT_E_INVALID; }
    return h->r->tonemap_owned_rows_device(device_out, n) ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_tonemap_owned_rows_device_on_stream(mi355rt_handle* h, uint32_t* device_out, size_t n, void* hip_stream)
{
    if (!h) return MI355RT_E_INVALID;
    if (!hip_stream) { h->r->last_error = "null stream: use mi355rt_tonemap_owned_rows_device"; return MI355RT_E_INVALID; }
    if (h->g->size() > 1) { h->r->last_error = "not available on a device group: use mi355rt_get_tonemapped_pixels"; return MI355RT_E_INVALID; }
    return h->r->tonemap_owned_rows_device(device_out, n, (hipStream_t)hip_stream) ? MI355RT_OK : MI355RT_E_HIP;
}

// a device group owns every row of the frame (its devices share them out among themselves)
uint32_t mi355rt_owned_rows(const mi355rt_handle* h) { return !h ? 0u : h->g->size() > 1 ? h->r->cfg.height : (uint32_t)h->r->owned_rows.size(); }

int mi355rt_owned_row_list(const mi355rt_handle* h, uint32_t* rows, size_t n)
{
    if (!h || !rows || n < mi355rt_owned_rows(h)) return MI355RT_E_INVALID;
    if (h->g->size() > 1) { for (uint32_t r = 0; r < h->r->cfg.height; ++r) rows[r] = r; return MI355RT_OK; }
    std::memcpy(rows, h->r->owned_rows.data(), h->r->owned_rows.size() * 4);
    return MI355RT_OK;
}

int mi355rt_film_get(mi355rt_handle* h, float* sum_rgb, float* sumsq_rgb, uint32_t* n)
{
    if (!h) return MI355RT_E_INVALID;
    return h->g->film_get(sum_rgb, sumsq_rgb, n) ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_film_clear(mi355rt_handle* h)
{
    if (!h) return MI355RT_E_INVALID;
    return h->g->film_clear() ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_film_get_pixels(mi355rt_handle* h, float* rgb)
{
    if (!h || !rgb) return MI355RT_E_INVALID;
    return h->g->film_stat(false, rgb) ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_film_get_estimated_variances(mi355rt_handle* h, float* rgb)
{
    if (!h || !rgb) return MI355RT_E_INVALID;
    return h->g->film_stat(true, rgb) ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_camera_move_rel(mi355rt_handle* h, float x, float y, float z)
{
    if (!h) return MI355RT_E_INVALID;
    h->g->camera_move_rel(x, y, z);
    return MI355RT_OK;
}
int mi355rt_camera_add_x_angle(mi355rt_handle* h, float radians)
{
    if (!h) return MI355RT_E_INVALID;
    h->g->camera_add_x_angle(radians);
    return MI355RT_OK;
}
int mi355rt_camera_add_y_angle(mi355rt_handle* h, float radians)
{
    if (!h) return MI355RT_E_INVALID;
    h->g->camera_add_y_angle(radians);
    return MI355RT_OK;
}
int mi355rt_camera_get(const mi355rt_handle* h, float rot16[16], float orient16[16], float max_xy[2])
{
    if (!h) return MI355RT_E_INVALID;
    if (rot16) std::memcpy(rot16, h->r->camera.rotation().e, 64);
    if (orient16) std::memcpy(orient16, h->r->camera.orientation().e, 64);
    if (max_xy) { max_xy[0] = h->r->camera.max_x(); max_xy[1] = h->r->camera.max_y(); }
    return MI355RT_OK;
}
int mi355rt_camera_get_ray(const mi355rt_handle* h, uint32_t u, uint32_t v, float xi1, float xi2, float ray6[6])
{
    if (!h || !ray6) return MI355RT_E_INVALID;
    Ray r = h->r->camera.get_ray(u, v, xi1, xi2);
    ray6[0] = r.pos.x; ray6[1] = r.pos.y; ray6[2] = r.pos.z; ray6[3] = r.dir.x; ray6[4] = r.dir.y; ray6[5] = r.dir.z;
    return MI355RT_OK;
}

int mi355rt_set_seed(mi355rt_handle* h, uint64_t seed)
{
    if (!h) return MI355RT_E_INVALID;
    return h->g->set_seed(seed) ? MI355RT_OK : MI355RT_E_HIP;
}
int mi355rt_set_flags(mi355rt_handle* h, uint32_t flags)
{
    if (!h) return MI355RT_E_INVALID;
    return h->g->set_flags(flags) ? MI355RT_OK : MI355RT_E_INVALID;
}

int mi355rt_set_slices(mi355rt_handle* h, uint32_t slices)
{
    if (!h || slices < 1 || slices > 8) return MI355RT_E_INVALID;
    h->g->set_slices(slices);
    return MI355RT_OK;
}
uint32_t mi355rt_get_slices(const mi355rt_handle* h) { return h ? h->r->slices : 0; }

int mi355rt_intersect_rays(mi355rt_handle* h, const float* rays6, size_t n, float* tuv, uint32_t* prim)
{
    if (!h || (n && (!rays6 || !tuv || !prim))) return MI355RT_E_INVALID;
    return h->r->intersect(rays6, n, tuv, prim, nullptr) ? MI355RT_OK : MI355RT_E_HIP;
}
int mi355rt_occluded_rays(mi355rt_handle* h, const float* rays6, size_t n, uint8_t* blocked)
{
    if (!h || (n && (!rays6 || !blocked))) return MI355RT_E_INVALID;
    return h->r->intersect(rays6, n, nullptr, nullptr, blocked) ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_get_sample_table(const mi355rt_handle* h, float* out)
{
    if (!h || !out) return MI355RT_E_INVALID;
    std::memcpy(out, h->r->table.data(), h->r->table.size() * sizeof(float));
    return MI355RT_OK;
}

int mi355rt_debug_sample(mi355rt_handle* h, uint32_t pixel, uint32_t sampleno, float color3[3], float* node_L, size_t nodes)
{
    if (!h || !color3 || !node_L) return MI355RT_E_INVALID;
    return h->r->debug_sample(pixel, sampleno, color3, node_L, nodes) ? MI355RT_OK : MI355RT_E_HIP;
}
int mi355rt_debug_numerics(mi355rt_handle* h, const float* a, const float* b, size_t n, float* quot, float* root, float* pow32)
{
    if (!h || (n && (!a || !b || !quot || !root || !pow32))) return MI355RT_E_INVALID;
    return h->r->debug_numerics(a, b, n, quot, root, pow32) ? MI355RT_OK : MI355RT_E_HIP;
}
int mi355rt_debug_slab(mi355rt_handle* h, const float* inv_rays6, const float* cubes6, size_t n, uint8_t* hit, float* tmin)
{
    if (!h || (n && (!inv_rays6 || !cubes6 || !hit || !tmin))) return MI355RT_E_INVALID;
    return h->r->debug_slab(inv_rays6, cubes6, n, hit, tmin) ? MI355RT_OK : MI355RT_E_HIP;
}
uint32_t mi355rt_tree_nodes(const mi355rt_handle* h) { return h ? h->r->nodes_per_sample : 0u; }
int mi355rt_debug_speculation(const mi355rt_handle* h, uint64_t out[2])
{
    if (!h || !out) return MI355RT_E_INVALID;
    h->r->speculation_stats(out);
    return MI355RT_OK;
}
int mi355rt_debug_light_map(const float* tri_verts, uint32_t ntri, const float light[3], double pad, uint32_t res, float* out_dist2, double* nearest)
{
    if ((ntri && !tri_verts) || !light || !out_dist2 || res == 0 || res > 4096 || !(pad >= 0.0)) return MI355RT_E_INVALID;
    mi355rt::LightMap lm;
    mi355rt::build_light_map(tri_verts, ntri, light, pad, res, lm);
    std::memcpy(out_dist2, lm.dist2.data(), lm.dist2.size() * sizeof(float));
    if (nearest) *nearest = lm.nearest;
    return MI355RT_OK;
}

namespace {
// bounds of every vertex below a reference of the wide tree; counts what the walk reaches and which child boxes fail to hold their subtree
struct WideWalk {
    const mi355rt::Bvh& b; std::vector<uint32_t> seen; uint32_t children = 0, bad_boxes = 0;
    void below(int32_t ref, double mn[3], double mx[3])
    {
        for (int a = 0; a < 3; ++a) { mn[a] = 1e300; mx[a] = -1e300; }
        if (ref < 0) {
            const uint32_t code = ~(uint32_t)ref, first = code >> 3, cnt = (code & 7u) + 1u;
            for (uint32_t t = first; t < first + cnt && t < b.tris.size(); ++t) {
                const mi355rt::BvhTri& r = b.tris[t];
                if (r.prim < seen.size()) ++seen[r.prim];
                for (int v = 0; v < 3; ++v) for (int a = 0; a < 3; ++a) {
                    const double x = (double)r.v0[a] + (v == 1 ? (double)r.e1[a] : v == 2 ? (double)r.e2[a] : 0.0);
                    mn[a] = std::min(mn[a], x); mx[a] = std::max(mx[a], x);
                }
            }
            return;
        }
        const mi355rt::BvhNode4& o = b.nodes4[(size_t)ref];
        const uint32_t cbm = (o.bases & 0xFFFFFu) - 128u, nb = ~(((o.bases >> 20) | ((o.ew >> 24) << 12)) << 3);      // traverse.hpp, wide_children
        const uint32_t lo[3] = { o.lox, o.loy, o.loz }, hi[3] = { o.hix, o.hiy, o.hiz };
        for (int i = 0; i < 4; ++i) {
            if (((lo[0] >> (8 * i)) & 0xFFu) > ((hi[0] >> (8 * i)) & 0xFFu)) continue;                                  // unused slot (inverted box)
            ++children;
            const uint32_t t = (o.meta >> (8 * i)) & 0xFFu;
            const int32_t cref = (t & 0x80u) ? (int32_t)(cbm + t) : (int32_t)(nb - t);
            double cmn[3], cmx[3];
            below(cref, cmn, cmx);
            for (int a = 0; a < 3; ++a) {
                const double scale = std::ldexp(1.0, (int)((o.ew >> (8 * a)) & 0xFFu) - 127);
                const double blo = (double)o.org[a] + (double)((lo[a] >> (8 * i)) & 0xFFu) * scale, bhi = (double)o.org[a] + (double)((hi[a] >> (8 * i)) & 0xFFu) * scale;
                if (cmn[a] <= cmx[a] && (blo > cmn[a] || bhi < cmx[a])) { ++bad_boxes; break; }
                mn[a] = std::min(mn[a], cmn[a]); mx[a] = std::max(mx[a], cmx[a]);
            }
        }
    }
};
}  // namespace

int mi355rt_debug_wide_bvh(const float* tri_verts, uint32_t ntri, uint32_t out[8])
{
    if ((ntri && !tri_verts) || !out) return MI355RT_E_INVALID;
    std::vector<uint32_t> geom(std::max(ntri, 1u), 0u);
    mi355rt::Bvh b;
    mi355rt::build_bvh(tri_verts, geom.data(), ntri, b);
    mi355rt::build_wide(b);
    for (int k = 0; k < 8; ++k) out[k] = 0u;
    out[1] = (uint32_t)b.nodes.size(); out[3] = b.max_depth;
    if (b.nodes4.empty()) return MI355RT_OK;
    out[0] = (uint32_t)b.nodes4.size(); out[2] = b.stack_need4;
    WideWalk w{ b, std::vector<uint32_t>(ntri, 0u) };
    double mn[3], mx[3];
    w.below(0, mn, mx);
    out[4] = w.children; out[6] = w.bad_boxes;
    for (uint32_t c : w.seen) out[5] += c == 1u ? 1u : 0u;
    std::vector<uint32_t> seen2(ntri, 0u);
    std::vector<int32_t> st(1, b.root);
    while (!st.empty()) {
        const int32_t r = st.back(); st.pop_back();
        if (r < 0) { const uint32_t code = ~(uint32_t)r, first = code >> 3, cnt = (code & 7u) + 1u; for (uint32_t t = first; t < first + cnt && t < b.tris.size(); ++t) if (b.tris[t].prim < ntri) ++seen2[b.tris[t].prim]; }
        else { st.push_back(b.nodes[(size_t)r].child0); st.push_back(b.nodes[(size_t)r].child1); }
    }
    for (uint32_t c : seen2) out[7] += c == 1u ? 1u : 0u;
    return MI355RT_OK;
}

int mi355rt_accel_stats(const mi355rt_handle* h, uint32_t out[8])
{
    if (!h || !out) return MI355RT_E_INVALID;
    const Bvh& b = h->r->bvh;
    out[0] = (uint32_t)b.nodes.size(); out[1] = b.leaves; out[2] = b.max_depth; out[3] = b.max_leaf;
    out[4] = (uint32_t)(b.nodes.size() * sizeof(BvhNode)); out[5] = (uint32_t)(b.tris.size() * sizeof(BvhTri));
    out[6] = (uint32_t)(h->r->build_ms_[0] * 1000.0); out[7] = (uint32_t)(h->r->build_ms_[1] * 1000.0);
    return MI355RT_OK;
}
int mi355rt_bvh_build_info(const mi355rt_handle* h, uint32_t out[2])
{
    if (!h || !out) return MI355RT_E_INVALID;
    out[0] = h->r->bvh_on_device_ ? 1u : 0u; out[1] = (uint32_t)(h->r->lbvh_device_ms_ * 1000.0);
    return MI355RT_OK;
}
int mi355rt_octree_stats(const mi355rt_handle* h, uint32_t out[8])
{
    if (!h || !out) return MI355RT_E_INVALID;
    std::memcpy(out, h->r->oct_stats_, sizeof h->r->oct_stats_);
    return MI355RT_OK;
}
uint32_t mi355rt_device_count(const mi355rt_handle* h) { return h ? (uint32_t)h->g->size() : 0u; }

int mi355rt_debug_gather_rate(mi355rt_handle* h, uint32_t table_nodes, uint32_t steps, double out[3])
{
    if (!h || !out) return MI355RT_E_INVALID;
    return h->r->debug_gather_rate(table_nodes, steps, out) ? MI355RT_OK : MI355RT_E_HIP;
}

int64_t mi355rt_debug_check_guards(mi355rt_handle* h)
{
    if (!h) return -1;
    int64_t bad = 0;
    for (size_t i = 0; i < h->g->size(); ++i) { const long b = h->g->device(i)->check_guards(); if (b < 0) return -1; bad += b; }
    return bad;
}

uint64_t mi355rt_hbm_allocated_bytes(const mi355rt_handle* h)
{
    if (!h) return 0;
    uint64_t b = 0;
    for (size_t i = 0; i < h->g->size(); ++i) b += h->g->device(i)->hbm_allocated_bytes();
    return b;
}

int mi355rt_synchronize(mi355rt_handle* h)
{
    if (!h) return MI355RT_E_INVALID;
    return h->g->synchronize() ? MI355RT_OK : MI355RT_E_HIP;
}

int mi355rt_comm_unique_id(uint8_t* id128)
{
    if (!id128) return bad("null argument");
    std::string err;
    if (!comm_unique_id(id128, err)) { g_create_error = err; return MI355RT_E_HIP; }
    return MI355RT_OK;
}
int mi355rt_comm_available(mi355rt_handle* h)
{
    if (!h) return MI355RT_E_INVALID;
    if (h->g->size() > 1) { h->r->last_error = "a device group gathers inside the process; RCCL communicators are for one-device handles"; return MI355RT_E_INVALID; }
    return h->r->comm_available() ? MI355RT_OK : MI355RT_E_HIP;
}
int mi355rt_comm_init(mi355rt_handle* h, const uint8_t* id128)
{
    if (!h || !id128) return MI355RT_E_INVALID;
    if (h->g->size() > 1) { h->r->last_error = "a device group gathers inside the process; RCCL communicators are for one-device handles"; return MI355RT_E_INVALID; }
    return h->r->comm_init(id128) ? MI355RT_OK : MI355RT_E_HIP;
}
int mi355rt_comm_gather_frame(mi355rt_handle* h, uint32_t root, uint32_t* host_out, size_t n)
{
    if (!h) return MI355RT_E_INVALID;
    return h->r->comm_gather(root, host_out, n) ? MI355RT_OK : MI355RT_E_HIP;
}
int mi355rt_comm_destroy(mi355rt_handle* h)
{
    if (!h) return MI355RT_E_INVALID;
    h->r->comm_destroy();
    return MI355RT_OK;
}
uint32_t mi355rt_comm_ranks(mi355rt_handle* h) { return h ? h->r->comm_ranks() : 0u; }

uint32_t mi355rt_width(const mi355rt_handle* h) { return h ? h->r->cfg.width : 0u; }
uint32_t mi355rt_height(const mi355rt_handle* h) { return h ? h->r->cfg.height : 0u; }
uint32_t mi355rt_triangle_count(const mi355rt_handle* h) { return h ? h->r->ntri : 0u; }
uint32_t mi355rt_current_row(const mi355rt_handle* h) { return h ? h->r->current_row : 0u; }

}  // extern "C"
