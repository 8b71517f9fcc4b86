/*
 * mi355rt.h — C ABI of libmi355rt.so, the MI355X (gfx950) render path behind
 * raytracer-rs's `raytracer_lib` API.
 *
 * Every entry point names the reference interface it replaces (paths relative to
 * /root/reference/raytracer_lib/src).  Conventions:
 *   - plain pointers and sizes only; the caller owns every input pointer, the library copies
 *     during `create`; output buffers are caller-allocated;
 *   - functions return 0 on success or a negative MI355RT_E_* code, and
 *     mi355rt_last_error() then returns the message the reference would carry in its
 *     `Result<_, String>` (lib.rs:15-27);
 *   - a handle may be created on one thread and used on another (main.rs:183,194-196): every
 *     entry binds the handle's HIP device; calls on ONE handle must not overlap;
 *   - there is NO CPU fallback: without a usable HIP device `create` fails with
 *     MI355RT_E_NO_DEVICE.
 */
#ifndef MI355RT_H
#define MI355RT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355RT_OK            0
#define MI355RT_E_INVALID    -1   /* bad argument */
#define MI355RT_E_NO_DEVICE  -2   /* no HIP device / HIP runtime error */
#define MI355RT_E_LOAD       -3   /* scene load error (SceneLoadError, loaders/mod.rs:20-25) */
#define MI355RT_E_HIP        -4   /* HIP runtime error after creation */

/* DEFAULT_TRIANGLES_PER_LEAF, oct_tree_intersector.rs:12 / lib.rs:7 */
#define MI355RT_DEFAULT_TRIANGLES_PER_LEAF 70u
/* rows per stripe of the deal over ranks / devices / frame slices when config.stripe_rows is 0 (and what mi355rt_default_config sets) */
#define MI355RT_DEFAULT_STRIPE_ROWS 4u

/* Config.flags */
#define MI355RT_FLAG_FIX_ROW_INDEX  1u  /* v = idx / width instead of the reference's idx / height (mod.rs:93-96) */
#define MI355RT_FLAG_COUNT_STEPS    2u  /* instrumented traversal: count BVH nodes visited / triangles tested */
#define MI355RT_FLAG_TIME_KERNELS   4u  /* bracket every trace-kernel launch with HIP events */
/* Intersector semantics — CREATE-time flags.
 *   default (neither flag): the reference's DEFAULT intersector, OctTreeIntersector (lib.rs:29-44 wires it): sorted
 *     front-to-back children, first leaf whose closest triangle's hit point lies inside the leaf cube wins
 *     (oct_tree_intersector.rs:148-206).  Served by the BVH (true closest hit) plus the octree CONFIRM step, which derives
 *     the octree's answer from the true closest hit exactly (csrc/traverse.hpp, confirm_walk); the octree is built with
 *     config.triangles_per_leaf.
 *   MI355RT_FLAG_OCTREE_SEMANTICS: the same semantics by walking the reference's octree directly, bug for bug — slow;
 *     kept as the independent cross-check of the confirm step.
 *   MI355RT_FLAG_TRUE_CLOSEST_HIT: the true closest hit, i.e. the reference's NoAccelerationIntersector
 *     (no_acceleration_intersector.rs:13-41); no octree is built, triangles_per_leaf is ignored. */
#define MI355RT_FLAG_OCTREE_SEMANTICS 8u
#define MI355RT_FLAG_TRUE_CLOSEST_HIT 32u
/* CREATE-time flag, testing only: the members of a device group (config.device_count > 1) all use config.device
 * instead of consecutive devices, so that the group's decomposition and gather run on a single-GPU machine. */
#define MI355RT_FLAG_GROUP_SHARES_DEVICE 16u
/* Create-time: build the BVH on the device (Morton order, Karras' parallel hierarchy, bottom-up refit: csrc/lbvh.hip) instead
 * of the host's binned-SAH build — the replacement north_star names for oct_tree_intersector.rs:66-146.  Results are identical
 * (any conservative tree gives the same hits); a frame is slower (Morton-order trees cost more node visits per ray) and create
 * is faster on large scenes.  Falls back to the host build when the device tree would be deeper than the traversal stack or the
 * scene fits one leaf; mi355rt_accel_stats out[6] then reports the host build's time and mi355rt_last_error says why. */
#define MI355RT_FLAG_DEVICE_LBVH 64u

typedef struct mi355rt_handle mi355rt_handle;

/* Material.diffuse, scene/mod.rs:63-69 + color.rs:98-108.  kind 0 = Diffuse::Color(rgb),
 * kind 1 = Diffuse::TextureId(tex_id).  emissive/specular/ior are never read by shading. */
typedef struct mi355rt_material {
    uint32_t kind;
    float rgb[3];
    uint32_t tex_id;
} mi355rt_material;

/* Light, scene/mod.rs:12-16 */
typedef struct mi355rt_light {
    float pos[3];
    float color[3];
} mi355rt_light;

/* Texture, scene/texture.rs:6-10: width*height RGB f32 texels (byte/256.0), row-major */
typedef struct mi355rt_texture {
    uint32_t width, height;
    const float* rgb;
} mi355rt_texture;

/* Scene, scene/mod.rs:18-23, flattened: one triangle soup in geometry order (= visual-scene
 * node order), tri_geom[i] = index of the geometry (and of its material) triangle i belongs to;
 * camera = the arguments of Camera::from_orientation_matrix for scene.cameras[0]
 * (camera.rs:22-27, lib.rs:39). */
typedef struct mi355rt_scene_desc {
    const float* tri_verts;          /* ntri * 9 floats, world space */
    const uint32_t* tri_geom;        /* ntri */
    uint32_t ntri;
    const mi355rt_material* materials;
    uint32_t nmaterials;
    const mi355rt_light* lights;
    uint32_t nlights;
    const mi355rt_texture* textures;
    uint32_t ntextures;
    float camera_orientation[16];    /* vecmath Matrix (row-vector convention) */
    float camera_fov_deg;
} mi355rt_scene_desc;

typedef struct mi355rt_config {
    uint32_t width, height;          /* create_raytracer(.., width, height), lib.rs:15 */
    uint32_t triangles_per_leaf;     /* leaf size of the reference's octree (the BVH underneath has its own) */
    uint32_t recursions;             /* RECURSIONS = 2, mod.rs:81 (0 selects the default) */
    uint32_t spread;                 /* SUB_SPREAD = 1, mod.rs:82 (0 selects the default) */
    uint32_t flags;                  /* MI355RT_FLAG_* */
    uint64_t seed;                   /* counter-RNG seed (the reference draws OS entropy); low 32 bits used */
    int32_t device;                  /* HIP device ordinal */
    /* Row-stripe ownership for multi-GPU rendering: this handle renders the stripes of
     * `stripe_rows` rows whose index is congruent to stripe_rank modulo stripe_world.
     * stripe_world <= 1 renders every row.  mi355rt_default_config sets stripe_rows = 4: thin stripes balance the ranks (the
     * slowest rank sets the frame), and the pixel tiles of a pass are cut to the stripe height, so they cost nothing per ray. */
    uint32_t stripe_rows, stripe_rank, stripe_world;
    uint32_t samples_per_pass;       /* samples per pixel traced per wavefront pass (0 = auto) */
    /* Device group: the handle drives `device_count` HIP devices of THIS process (device, device + 1, ...).  Rows
     * are dealt to them in stripes of stripe_rows; every entry point works on the group as on one device (the
     * reference's callers see one RayTracer, main.rs:183-216); mi355rt_get_tonemapped_pixels gathers the packed
     * stripes on the first device with hipMemcpyPeerAsync over xGMI.  0 or 1: one device.  Not combinable with
     * stripe_world > 1 (that is the one-process-per-GPU decomposition, gathered with mi355rt_comm_*). */
    uint32_t device_count;
} mi355rt_config;

/* Ray counters of one mi355rt_render / mi355rt_trace_frame_additive call. */
typedef struct mi355rt_ray_counts {
    uint64_t primary;        /* primary samples (the reference's own rays/s metric, stats.rs:27) */
    uint64_t bounce;         /* reflection rays, mod.rs:156-158 */
    uint64_t shadow;         /* shadow rays, mod.rs:226 */
    uint64_t primary_hits;
    uint64_t primary_culled; /* primary samples never traced: their 256-sample chunk lies outside the screen bounds of the
                              * top BVH boxes (all of them miss, mod.rs:99-100); traced primary rays = primary - primary_culled */
    uint64_t nodes_visited;  /* only with MI355RT_FLAG_COUNT_STEPS */
    uint64_t tris_tested;    /* only with MI355RT_FLAG_COUNT_STEPS */
    uint64_t trace_launches; /* trace-kernel launches (a 50-row frame: fused launches) */
    uint64_t inner_execs;    /* only with MI355RT_FLAG_COUNT_STEPS: wave-level executions of the inner-node section */
    uint64_t leaf_execs;     /* only with MI355RT_FLAG_COUNT_STEPS: wave-level executions of the triangle section */
    double trace_ms;         /* summed HIP-event time of the trace kernels (MI355RT_FLAG_TIME_KERNELS) */
    double total_ms;         /* HIP-event time of the whole call on the handle's stream */
    double trace_secondary_ms;        /* the part of trace_ms spent in launches of rounds >= 1 (reflection + shadow rays) */
    uint64_t trace_secondary_launches;
    double shader_clock_mhz; /* only with MI355RT_FLAG_COUNT_STEPS: the clock the trace waves actually ran at, sum of delta s_memtime over
                              * sum of delta s_memrealtime (100 MHz) of all waves of the call's trace launches; 0 when not measured */
    uint64_t shadow_skipped; /* shadow rays (counted in `shadow`) never traced: the depth cube map around their light proves that nothing lies
                              * between the shaded point and the light, so no intersector could block them; traced shadow rays = shadow - shadow_skipped */
} mi355rt_ray_counts;

void mi355rt_default_config(mi355rt_config* cfg);

/* build_raytracer, lib.rs:29-44 (octree build replaced by a BVH build + upload). */
int mi355rt_create(const mi355rt_scene_desc* scene, const mi355rt_config* cfg, mi355rt_handle** out);
/* create_raytracer(collada_doc, triangles_per_leaf, width, height), lib.rs:15-20.
 * data_dir (may be NULL) is where texture files are looked up (colladaloader.rs:146-150). */
int mi355rt_create_from_collada_str(const char* doc, size_t len, const char* data_dir,
                                    const mi355rt_config* cfg, mi355rt_handle** out);
/* create_raytracer_from_file(collada_filename, ...), lib.rs:22-27 */
int mi355rt_create_from_collada_file(const char* path, const mi355rt_config* cfg, mi355rt_handle** out);
/* flat scene container written by tools/dae2scene (no reference counterpart) */
int mi355rt_create_from_scene_file(const char* path, const mi355rt_config* cfg, mi355rt_handle** out);
void mi355rt_destroy(mi355rt_handle* h);

/* Error text of the last failed call; h == NULL returns the last creation error of this thread. */
const char* mi355rt_last_error(const mi355rt_handle* h);

/* RayTracer::trace_frame_additive, mod.rs:80-117: 50 rows x width pixels x 1 sample, row cursor
 * wraps modulo height; returns 50*width (0 on error).  With stripes, only owned rows are traced. */
uint32_t mi355rt_trace_frame_additive(mi355rt_handle* h);
/* Whole frame (owned stripes) x spp samples per pixel — the benchmark entry; no reference
 * counterpart (the reference has no spp concept).  counts may be NULL. */
int mi355rt_render(mi355rt_handle* h, uint32_t spp, mi355rt_ray_counts* counts);
/* The same frame, QUEUED only: returns as soon as the launches are on the handle's stream(s), like mi355rt_trace_frame_additive does;
 * mi355rt_last_counts / mi355rt_synchronize / any read-out waits for it.  Consecutive frames then run back to back on the device
 * (every later call on the handle is ordered behind it).  A device group of several devices still waits. */
int mi355rt_render_async(mi355rt_handle* h, uint32_t spp);
/* counters of the last trace_frame_additive / render call (waits for an asynchronous call to finish) */
int mi355rt_last_counts(mi355rt_handle* h, mi355rt_ray_counts* counts);

/* RayTracer::get_tonemapped_pixels, mod.rs:120-128: width*height u32 0xAARRGGBB (A = 255). */
int mi355rt_get_tonemapped_pixels(mi355rt_handle* h, uint32_t* out, size_t n);
/* Same, written to DEVICE memory on the handle's device (e.g. a buffer owned by the caller's
 * collective library).  Rows owned by this handle only, packed in ascending row order:
 * mi355rt_owned_rows(h) * width values.  Synchronous with respect to the host. */
int mi355rt_tonemap_owned_rows_device(mi355rt_handle* h, uint32_t* device_out, size_t n);
/* Same, ASYNCHRONOUS: the kernel is launched on the caller's HIP stream (`hip_stream` is a hipStream_t),
 * after everything this handle has queued; the call returns at once and the write is ordered with the
 * caller's other work on that stream (buffer initialisation before, the collective after).  Later calls on
 * this handle wait for it. */
int mi355rt_tonemap_owned_rows_device_on_stream(mi355rt_handle* h, uint32_t* device_out, size_t n, void* hip_stream);
uint32_t mi355rt_owned_rows(const mi355rt_handle* h);
/* ascending list of the rows this handle owns */
int mi355rt_owned_row_list(const mi355rt_handle* h, uint32_t* rows, size_t n);

/* Film, film.rs:27-68.  pixel_datas: sum_rgb / sumsq_rgb hold width*height*3 floats, n holds
 * width*height counts; any pointer may be NULL. */
int mi355rt_film_get(mi355rt_handle* h, float* sum_rgb, float* sumsq_rgb, uint32_t* n);
int mi355rt_film_clear(mi355rt_handle* h);                                   /* Film::clear, film.rs:37-41 */
int mi355rt_film_get_pixels(mi355rt_handle* h, float* rgb);                  /* Film::get_pixels, film.rs:43-47 */
int mi355rt_film_get_estimated_variances(mi355rt_handle* h, float* rgb);     /* film.rs:51-67 */

/* Camera, camera.rs:63-78 (the `pub camera` field of RayTracer, mod.rs:38). */
int mi355rt_camera_move_rel(mi355rt_handle* h, float x, float y, float z);
int mi355rt_camera_add_x_angle(mi355rt_handle* h, float radians);
int mi355rt_camera_add_y_angle(mi355rt_handle* h, float radians);
/* rotation_matrix / orientation_matrix after update_matrices (camera.rs:92-98), max_x, max_y */
int mi355rt_camera_get(const mi355rt_handle* h, float rot16[16], float orient16[16], float max_xy[2]);
/* Camera::get_ray, camera.rs:80-90, with explicit jitter (xi1, xi2 in [0,1)): out = pos3, dir3 */
int mi355rt_camera_get_ray(const mi355rt_handle* h, uint32_t u, uint32_t v, float xi1, float xi2, float ray6[6]);

/* Re-seed: afterwards the handle renders what a handle created with this seed renders (the per-sample hash key AND
 * the 65 536-entry direction table are functions of the seed; only its low 32 bits are used).  The film is kept. */
int mi355rt_set_seed(mi355rt_handle* h, uint64_t seed);
/* Run-time flags (FIX_ROW_INDEX, COUNT_STEPS, TIME_KERNELS).  MI355RT_FLAG_OCTREE_SEMANTICS is fixed at creation:
 * pass the bit as it was created, a call that would change it fails with MI355RT_E_INVALID and changes nothing. */
int mi355rt_set_flags(mi355rt_handle* h, uint32_t flags);
/* Number of concurrent frame slices mi355rt_render splits its rows into (1..8, default 3, or the
 * environment variable MI355RT_SLICES).  Each slice runs its wavefront passes on its own HIP stream, so the
 * drain window of one slice's trace launch is filled by another slice's kernels; results do not depend on
 * it.  With MI355RT_FLAG_TIME_KERNELS the per-kernel times only mean something with 1 slice. */
int mi355rt_set_slices(mi355rt_handle* h, uint32_t slices);
uint32_t mi355rt_get_slices(const mi355rt_handle* h);

/* Intersector::intersect_ray, accel_intersect.rs:10-13, batched on the device: rays6 = n x
 * (pos3, dir3); out tuv = n x 3 (untouched on a miss), prim = n global triangle indices
 * (0xFFFFFFFF on a miss; geometry_index = tri_geom[prim], vertex_index = 3 * index within
 * the geometry, mod.rs:17-21).  Semantics of the handle's intersector (see the flags above): by default the
 * reference's OctTreeIntersector, with MI355RT_FLAG_TRUE_CLOSEST_HIT the true closest hit (lowest t, ties to
 * the lowest triangle index). */
int mi355rt_intersect_rays(mi355rt_handle* h, const float* rays6, size_t n, float* tuv, uint32_t* prim);
/* shadow-ray predicate of shade(), mod.rs:222-230: blocked[i] = 1 iff the closest hit of ray i
 * has 0.01 < t < 1.0 */
int mi355rt_occluded_rays(mi355rt_handle* h, const float* rays6, size_t n, uint8_t* blocked);

/* SampleGenerator table, sample_generator.rs:15-24: 65 536 x 3 floats */
int mi355rt_get_sample_table(const mi355rt_handle* h, float* out);
/* Per-node direct-light terms of one primary sample, computed on the device by the same
 * kernels as a frame (stage-level parity): node_L = nodes x 3 floats (breadth-first radiance
 * tree, black where not reached), color3 = the sample's radiance.  Does not touch the film. */
int mi355rt_debug_sample(mi355rt_handle* h, uint32_t pixel, uint32_t sampleno, float color3[3], float* node_L, size_t nodes);
/* Device arithmetic self-check: quot = a/b, root = sqrt(a), pow32 = a^32 computed exactly as the
 * kernels compute them (IEEE division and square root; powf(x, 32.0) of mod.rs:255). */
int mi355rt_debug_numerics(mi355rt_handle* h, const float* a, const float* b, size_t n, float* quot, float* root, float* pow32);
/* intersect_cube_inverse_ray, oct_tree_intersector.rs:348-372, exactly as the reference-exact intersector runs it
 * on the device: inv_rays6 = n x (origin3, 1/dir3), cubes6 = n x (min3, max3); hit[i] = 1 iff the slab test
 * passes, tmin[i] = the entry distance it returns (negative when the origin is inside).  Exists so that the
 * reference's own known-answer vectors (oct_tree_intersector.rs:475-512) run through the HIP path. */
int mi355rt_debug_slab(mi355rt_handle* h, const float* inv_rays6, const float* cubes6, size_t n, uint8_t* hit, float* tmin);
uint32_t mi355rt_tree_nodes(const mi355rt_handle* h);
/* Speculation of the drop-in loop: out[0] = 50-row frames launched ahead of the call that asks for them, out[1] = how many of those the next
 * mi355rt_trace_frame_additive call took over (the others were given back: the caller did something else first).  Test hook; device 0 of a group. */
int mi355rt_debug_speculation(const mi355rt_handle* h, uint64_t out[2]);
/* The depth cube map the library builds around a point light (csrc/lightmap.hpp; what lets the shade kernels leave out the shadow rays of
 * mod.rs:224-232 whose way to the light is provably free).  HOST code, no device needed: out_dist2[6 * res * res] receives, per direction texel
 * (face = 2 * major axis + (component negative), then i over the lower and j over the higher of the two other axes), a lower bound of the squared
 * distance from `light` to any of the ntri triangles (tri_verts: ntri x 9 floats) padded by `pad`, +inf where no triangle is seen; *nearest the
 * distance to the nearest triangle.  Exists so that the bound can be checked against brute force without a GPU. */
int mi355rt_debug_light_map(const float* tri_verts, uint32_t ntri, const float light[3], double pad, uint32_t res, float* out_dist2, double* nearest);

/* Test hook (host only, no device): the 4-wide tree of 48-byte nodes that experiment builds of the kernels walk (-DMI355RT_WIDE=1; bvh.hpp,
 * BvhNode4: 8-bit child boxes in a per-node frame), built from the ntri triangles like mi355rt_create builds the binary tree, then checked by a
 * walk that decodes the node words the way the kernels do.  out[0] wide nodes (0: the format does not serve this scene), [1] binary nodes,
 * [2] most deferred children of any walk (stack rows), [3] binary depth, [4] child slots in use, [5] triangles the walk of the wide tree reaches
 * exactly once, [6] child boxes that fail to contain a vertex of a triangle below them (must be 0), [7] triangles the re-pointed binary tree
 * reaches exactly once. */
int mi355rt_debug_wide_bvh(const float* tri_verts, uint32_t ntri, uint32_t out[8]);

/* acceleration-structure facts: out[0] nodes, [1] leaves, [2] max depth, [3] max leaf size,
 * [4] node bytes, [5] triangle bytes, [6] BVH build time inside create (wall, microseconds; host SAH build, or the device
 * build with its uploads and read-back), [7] host octree build time inside create (microseconds; 0 with
 * MI355RT_FLAG_TRUE_CLOSEST_HIT) */
int mi355rt_accel_stats(const mi355rt_handle* h, uint32_t out[8]);
/* which builder made the BVH: out[0] 1 = the device (MI355RT_FLAG_DEVICE_LBVH served the scene), 0 = the host;
 * out[1] device time of the build kernels + sort (HIP events, microseconds; 0 for a host build) */
int mi355rt_bvh_build_info(const mi355rt_handle* h, uint32_t out[2]);
/* the reference's octree (absent with MI355RT_FLAG_TRUE_CLOSEST_HIT): out[0] octree nodes, [1] inner, [2] leaves, [3] empty leaves, [4] depth,
 * [5] triangle references (the quantities of SURVEY.md 6.2) */
int mi355rt_octree_stats(const mi355rt_handle* h, uint32_t out[8]);
/* devices of the handle's group (1 for an ordinary handle) */
uint32_t mi355rt_device_count(const mi355rt_handle* h);
/* measurement hook behind bench.py's roofline (no reference counterpart): the rate at which the device's vector memory pipe serves
 * the trace kernels' kind of fetch and nothing else — every lane of 8 waves per SIMD walks `steps` random 32-byte nodes of an
 * L2-resident table of table_nodes nodes with two 16-byte loads per node (profiles/r03_notes.md).  out[0] cache-line accesses per
 * second (one per lane and load), out[1] kernel time in ms, out[2] node fetches per second. */
int mi355rt_debug_gather_rate(mi355rt_handle* h, uint32_t table_nodes, uint32_t steps, double out[3]);
/* test hook: with MI355RT_DEBUG_GUARD set in the environment every pass buffer is allocated with a 256-byte tail of 0xA5;
 * this returns how many of those bytes a launch has overwritten (0 = nothing wrote past a buffer; -1: error) */
int64_t mi355rt_debug_check_guards(mi355rt_handle* h);
/* device memory (HBM) the handle holds right now, summed over its devices: scene + acceleration structures + film + the
 * pass buffers of the largest pass rendered so far + gather slots */
uint64_t mi355rt_hbm_allocated_bytes(const mi355rt_handle* h);
/* wait until everything queued on the handle (50-row frames, gathers) has finished on its device(s) */
int mi355rt_synchronize(mi355rt_handle* h);

/* ---- one process per GPU: the framebuffer gather over RCCL / xGMI (no reference counterpart: the reference is
 * one process on one CPU).  Every process creates its handle with stripe_rank / stripe_world = its rank / the
 * number of processes.  Rank 0 obtains a 128-byte id (ncclUniqueId) with mi355rt_comm_unique_id and hands it to the
 * others by whatever means the host application has; then ALL ranks call mi355rt_comm_init (collective).
 * mi355rt_comm_gather_frame (collective) maps every rank's rows to packed u32 and moves them to `root` with grouped
 * ncclSend / ncclRecv on the handle's stream; the root places them into its frame.  host_out (root only, may be
 * NULL): width*height u32 copied out after a synchronisation; with NULL the call only queues the work
 * (mi355rt_synchronize waits for it).  librccl.so is loaded on first use. */
#define MI355RT_COMM_ID_BYTES 128
int mi355rt_comm_unique_id(uint8_t* id128);
/* local, communication-free pre-check of mi355rt_comm_init (librccl.so loadable with every symbol, device bindable, no live
 * communicator): the ranks should agree on this BEFORE any of them enters the collective mi355rt_comm_init, so that nobody
 * blocks inside RCCL's bootstrap waiting for a rank that cannot join */
int mi355rt_comm_available(mi355rt_handle* h);
int mi355rt_comm_init(mi355rt_handle* h, const uint8_t* id128);
int mi355rt_comm_gather_frame(mi355rt_handle* h, uint32_t root, uint32_t* host_out, size_t n);
int mi355rt_comm_destroy(mi355rt_handle* h);
/* ranks of the handle's live communicator as RCCL counts them (ncclCommCount; 0: no communicator, or it is not the one
 * this handle's stripe_rank was dealt into) */
uint32_t mi355rt_comm_ranks(mi355rt_handle* h);

uint32_t mi355rt_width(const mi355rt_handle* h);
uint32_t mi355rt_height(const mi355rt_handle* h);
uint32_t mi355rt_triangle_count(const mi355rt_handle* h);
uint32_t mi355rt_current_row(const mi355rt_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* MI355RT_H */
