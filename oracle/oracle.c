/*
 * oracle.c — CPU restatement of the raytracer-rs render hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under raytracer-rs_amd/ (the product) may
 * include, link, import or execute this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * What it restates (all citations relative to /root/reference/raytracer_lib/src):
 *   raytracer/mod.rs                          trace_frame_additive, compute_radiance,
 *                                             randomize_reflection_ray, calc_normal, shade
 *   raytracer/accel_intersect/oct_tree_intersector.rs   octree build (SAT) + traversal
 *   raytracer/accel_intersect/no_acceleration_intersector.rs   brute-force closest hit
 *   raytracer/intersect.rs:62-98              Moller-Trumbore "late out"
 *   raytracer/sample_generator.rs             65 536-entry unit-vector table + walk
 *   raytracer/film.rs, raytracer/tonemap.rs:4-10, scene/color.rs:89-95
 *   scene/camera.rs, vecmath.rs
 *
 * Parity status ("pinning"): the reference's own tests pin only the slab test
 * (oct_tree_intersector.rs:475-512), two matrix identities (vecmath.rs:343-359) and two
 * COLLADA matrix conversions (collada_types.rs:98-125); tests/test_oracle_kat.py checks
 * those against this file.  Triangle intersection, traversal, shading, sampling, film and
 * packing are NOT covered by any reference test or golden image, and the Rust reference
 * cannot be built in this environment (no rustc/cargo): for those functions parity is
 * UNPINNED — this file follows the reference statement by statement instead.
 *
 * Deliberate, documented deviations (the reference cannot be seeded at all):
 *   - every `StdRng::from_os_rng()` / `rand::rng()` draw is replaced by a counter-based
 *     hash RNG (pcg4d, Jarzynski & Olano 2020) keyed by (pixel, sample#, dimension, seed);
 *     the distributions are the reference's (rand 0.9.1): U[0,1) with 23 random bits for
 *     the pixel jitter, uniform integer in [0, 65534] for the table start index,
 *     rejection sampling in [-1,1)^3 for the table entries.
 *   - x.powf(32.0) (mod.rs:255) is evaluated as five squarings in double precision and
 *     rounded once to f32 (libm-independent; agrees with a correctly rounded powf).
 *   - flags select the fixed pixel->ray row index (v = idx / width) and the brute-force
 *     intersector (the reference's NoAccelerationIntersector) instead of the octree.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, like rustc), see oracle/Makefile.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_FLAG_FIX_ROW_INDEX   1u
#define ORACLE_FLAG_BRUTE_FORCE     2u

#define NUM_SAMPLES 65536            /* sample_generator.rs:5-7 */
#define SAMPLE_MAX  65535

typedef struct { float x, y, z; } v3;
typedef struct { v3 pos, dir; } ray_t;
typedef struct { float r, g, b; } rgb_t;

/* ---------------------------------------------------------------- vecmath.rs */
static inline v3 v3new(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return v3new(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return v3new(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float s) { return v3new(a.x * s, a.y * s, a.z * s); }   /* Vec3 * f32 */
static inline v3 sscale(float s, v3 a) { return v3new(s * a.x, s * a.y, s * a.z); }   /* f32 * Vec3 */
/* vecmath.rs:74-76 */
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vecmath.rs:79-85 */
static inline v3 cross(v3 a, v3 b)
{
    return v3new(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vecmath.rs:23-26 */
static inline v3 normalized(v3 a)
{
    float len = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return v3new(a.x / len, a.y / len, a.z / len);
}

/* 4x4 matrix, vecmath.rs:87-160.  e[16] row-major storage, row-vector convention. */
static void mat_ident(float* m)
{
    memset(m, 0, 16 * sizeof(float));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}
/* vecmath.rs:237-313 */
void oracle_mat_mul(const float* s, const float* r, float* out)
{
    float t[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            t[4 * i + j] = s[4 * i + 0] * r[0 + j] + s[4 * i + 1] * r[4 + j]
                         + s[4 * i + 2] * r[8 + j] + s[4 * i + 3] * r[12 + j];
    memcpy(out, t, sizeof t);
}
/* vecmath.rs:200-211 : Matrix * Vec4 */
void oracle_mat_vec(const float* e, const float* v, float* out)
{
    float x = v[0] * e[0] + v[1] * e[4] + v[2] * e[8] + v[3] * e[12];
    float y = v[0] * e[1] + v[1] * e[5] + v[2] * e[9] + v[3] * e[13];
    float z = v[0] * e[2] + v[1] * e[6] + v[2] * e[10] + v[3] * e[14];
    float w = v[0] * e[3] + v[1] * e[7] + v[2] * e[11] + v[3] * e[15];
    out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}
/* vecmath.rs:141-159 */
static void mat_transpose(const float* s, float* m)
{
    memcpy(m, s, 16 * sizeof(float));
    m[1] = s[4];  m[2] = s[8];  m[3] = s[12];
    m[4] = s[1];  m[6] = s[9];  m[7] = s[13];
    m[8] = s[2];  m[9] = s[6];  m[11] = s[14];
    m[12] = s[3]; m[13] = s[7]; m[14] = s[11];
}
static void mat_rot_x(float rad, float* m)      /* vecmath.rs:116-123 */
{
    mat_ident(m);
    m[5] = cosf(rad); m[6] = -sinf(rad); m[9] = sinf(rad); m[10] = cosf(rad);
}
static void mat_rot_y(float rad, float* m)      /* vecmath.rs:124-131 */
{
    mat_ident(m);
    m[0] = cosf(rad); m[2] = sinf(rad); m[8] = -sinf(rad); m[10] = cosf(rad);
}
static void mat_translate(v3 v, float* m)       /* vecmath.rs:133-139 */
{
    mat_ident(m);
    m[12] = v.x; m[13] = v.y; m[14] = v.z;
}
/* scene/loaders/colladaloader/collada_types.rs:76-90 : reflect_z * transpose(C) * swap_yz */
void oracle_collada_matrix_to_vecmath(const float* collada16, float* out16)
{
    static const float swap_yz[16] = { 1, 0, 0, 0,  0, 0, 1, 0,  0, 1, 0, 0,  0, 0, 0, 1 };
    static const float reflect_z[16] = { 1, 0, 0, 0,  0, 1, 0, 0,  0, 0, -1, 0,  0, 0, 0, 1 };
    float row_major[16], t[16];
    mat_transpose(collada16, row_major);
    oracle_mat_mul(reflect_z, row_major, t);
    oracle_mat_mul(t, swap_yz, out16);
}

/* ---------------------------------------------------------------- counter RNG
 * Replaces rand 0.9.1 StdRng::from_os_rng() (mod.rs:84, mod.rs:152) and rand::rng()
 * (sample_generator.rs:37).  pcg4d: Jarzynski & Olano, "Hash Functions for GPU
 * Rendering", JCGT 9(3) 2020.  Integer-only, so CPU and GPU agree bit for bit. */
static void pcg4d(uint32_t v[4])
{
    for (int i = 0; i < 4; ++i) v[i] = v[i] * 1664525u + 1013904223u;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    for (int i = 0; i < 4; ++i) v[i] ^= v[i] >> 16;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
}
void oracle_pcg4d(const uint32_t* in, uint32_t* out) { memcpy(out, in, 16); pcg4d(out); }
/* rand 0.9.1 UniformFloat<f32>::sample_single for 0.0..1.0: 23 mantissa bits -> [0,1) */
static inline float u01(uint32_t bits) { return (float)(bits >> 9) * (1.0f / 8388608.0f); }
/* uniform integer in [0, 65534] (sample_generator.rs:32: random_range(0..NUM_SAMPLES-1)),
 * widening-multiply range reduction */
static inline uint32_t table_index(uint32_t bits) { return (uint32_t)(((uint64_t)bits * 65535u) >> 32); }

#define DIM_JITTER 0u                /* hash word 2: 0 = pixel jitter, 1 + child node id = bounce */
#define DIM_TABLE  0xFFFFFFFFu       /* hash word 2 for the sample-table stream */

/* ---------------------------------------------------------------- scene + state */
typedef struct { uint32_t kind; rgb_t rgb; uint32_t tex; } material_t;  /* kind 0 colour, 1 texture */
typedef struct { v3 pos; rgb_t color; } light_t;
typedef struct { uint32_t w, h; const rgb_t* data; } texture_t;

typedef struct { v3 min, max; } cube_t;
typedef struct {
    int is_leaf;
    uint32_t child[8];
    uint32_t* tris;        /* global triangle ids, list order = reference list order */
    uint32_t ntris;
} octnode_t;

typedef struct {
    float x_angle, y_angle;
    v3 pos;
    uint32_t width, height;
    float base_orient[16], base_rot[16], orient[16], rot[16];
    float max_x, max_y;
} camera_t;

typedef struct {
    uint64_t rays_primary, rays_bounce, rays_shadow;
    uint64_t nodes_visited, cubes_tested, tris_tested;
    uint64_t primary_hits;
} counters_t;

typedef struct oracle {
    uint32_t ntri, ngeom, nlights, ntex;
    v3* verts;             /* ntri*3, world space (== geometry.vertices == transformed_vertices) */
    uint32_t* tri_geom;    /* geometry index per triangle */
    uint32_t* geom_first;  /* first global triangle of each geometry */
    material_t* mats;
    light_t* lights;
    texture_t* texs;
    rgb_t* texdata;

    uint32_t width, height, tris_per_leaf, recursions, spread, flags;
    uint64_t seed;
    camera_t cam;

    cube_t* cubes; octnode_t* nodes; uint32_t nnodes, capnodes;

    v3* table;             /* NUM_SAMPLES unit vectors */

    rgb_t* sum; rgb_t* sumsq; uint32_t* nsamp;   /* film.rs:3-8 */
    uint32_t current_row;

    counters_t cnt;        /* only updated by single-threaded entry points */
} oracle_t;

typedef struct { float t, u, v; uint32_t prim; int hit; } hit_t;   /* mod.rs:17-21 + intersect.rs:3-8 */

/* ---------------------------------------------------------------- intersect.rs:62-98 */
static int mt_late_out(const ray_t* ray, v3 v0, v3 v1, v3 v2, float* t_out, float* u_out, float* v_out)
{
    v3 v0v1 = vsub(v1, v0);
    v3 v0v2 = vsub(v2, v0);
    v3 pvec = cross(ray->dir, v0v2);
    float det = dot(v0v1, pvec);
    if (fabsf(det) < 1.1920929e-7f) return 0;          /* f32::EPSILON */
    float inv_det = 1.0f / det;
    v3 tvec = vsub(ray->pos, v0);
    float u = dot(tvec, pvec) * inv_det;
    v3 qvec = cross(tvec, v0v1);
    float v = dot(ray->dir, qvec) * inv_det;
    float t = dot(v0v2, qvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return 0;
    if (v < 0.0f || u + v > 1.0f) return 0;
    if (t < 0.0f) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}
int oracle_mt(const float* ray6, const float* tri9, float* tuv)
{
    ray_t r = { { ray6[0], ray6[1], ray6[2] }, { ray6[3], ray6[4], ray6[5] } };
    return mt_late_out(&r, v3new(tri9[0], tri9[1], tri9[2]), v3new(tri9[3], tri9[4], tri9[5]),
                       v3new(tri9[6], tri9[7], tri9[8]), &tuv[0], &tuv[1], &tuv[2]);
}

/* ---------------------------------------------------------------- oct_tree_intersector.rs:348-372 */
static int slab(const ray_t* inv_ray, const cube_t* c, float* t_out)
{
    float tx1 = (c->min.x - inv_ray->pos.x) * inv_ray->dir.x;
    float tx2 = (c->max.x - inv_ray->pos.x) * inv_ray->dir.x;
    float tmin = fminf(tx1, tx2);
    float tmax = fmaxf(tx1, tx2);
    float ty1 = (c->min.y - inv_ray->pos.y) * inv_ray->dir.y;
    float ty2 = (c->max.y - inv_ray->pos.y) * inv_ray->dir.y;
    tmin = fmaxf(tmin, fminf(ty1, ty2));
    tmax = fminf(tmax, fmaxf(ty1, ty2));
    float tz1 = (c->min.z - inv_ray->pos.z) * inv_ray->dir.z;
    float tz2 = (c->max.z - inv_ray->pos.z) * inv_ray->dir.z;
    tmin = fmaxf(tmin, fminf(tz1, tz2));
    tmax = fminf(tmax, fmaxf(tz1, tz2));
    if (tmax >= tmin && tmax > 0.0f) { *t_out = tmin; return 1; }
    return 0;
}
/* inv_ray6 = origin + ALREADY INVERTED direction, as in the reference tests */
int oracle_slab(const float* inv_ray6, const float* cube6, float* t_out)
{
    ray_t r = { { inv_ray6[0], inv_ray6[1], inv_ray6[2] }, { inv_ray6[3], inv_ray6[4], inv_ray6[5] } };
    cube_t c = { { cube6[0], cube6[1], cube6[2] }, { cube6[3], cube6[4], cube6[5] } };
    return slab(&r, &c, t_out);
}

/* ---------------------------------------------------------------- octree build: OCT:66-146, 274-330, 374-469 */
static void project(const v3* pts, int n, v3 axis, float* mn, float* mx)   /* OCT:460-469 */
{
    float lo = 3.40282347e+38f, hi = -3.40282347e+38f;
    for (int i = 0; i < n; ++i) {
        float val = dot(axis, pts[i]);
        lo = fminf(lo, val);
        hi = fmaxf(hi, val);
    }
    *mn = lo; *mx = hi;
}
static int tri_cube_sat(const cube_t* cube, const v3* tv)                   /* OCT:393-458 */
{
    const v3 xa = { 1, 0, 0 }, ya = { 0, 1, 0 }, za = { 0, 0, 1 };
    float tmin, tmax, cmin, cmax;
    project(tv, 3, xa, &tmin, &tmax);
    if (tmax < cube->min.x || tmin > cube->max.x) return 0;
    project(tv, 3, ya, &tmin, &tmax);
    if (tmax < cube->min.y || tmin > cube->max.y) return 0;
    project(tv, 3, za, &tmin, &tmax);
    if (tmax < cube->min.z || tmin > cube->max.z) return 0;

    v3 cv[8] = {
        cube->min,
        { cube->max.x, cube->min.y, cube->min.z },
        { cube->min.x, cube->max.y, cube->min.z },
        { cube->min.x, cube->min.y, cube->max.z },
        { cube->min.x, cube->max.y, cube->max.z },
        { cube->max.x, cube->min.y, cube->max.z },
        { cube->max.x, cube->max.y, cube->min.z },
        cube->max,
    };
    v3 e1 = vsub(tv[0], tv[1]);
    v3 e2 = vsub(tv[1], tv[2]);
    v3 n = cross(e1, e2);
    float off = dot(n, tv[0]);
    project(cv, 8, n, &cmin, &cmax);
    if (cmax < off || cmin > off) return 0;

    v3 e3 = vsub(tv[2], tv[0]);
    v3 axes[9] = {
        cross(e1, xa), cross(e1, ya), cross(e1, za),
        cross(e2, xa), cross(e2, ya), cross(e2, za),
        cross(e3, xa), cross(e3, ya), cross(e3, za),
    };
    for (int i = 0; i < 9; ++i) {
        project(cv, 8, axes[i], &cmin, &cmax);
        project(tv, 3, axes[i], &tmin, &tmax);
        if (cmax < tmin || cmin > tmax) return 0;
    }
    return 1;
}
static void child_cubes(const cube_t* c, cube_t out[8])                      /* OCT:274-313 */
{
    v3 mid = sscale(0.5f, vadd(c->max, c->min));
    v3 mn = c->min, mx = c->max;
    out[0].min = v3new(mn.x, mn.y, mn.z);   out[0].max = v3new(mid.x, mid.y, mid.z);
    out[1].min = v3new(mid.x, mn.y, mn.z);  out[1].max = v3new(mx.x, mid.y, mid.z);
    out[2].min = v3new(mn.x, mid.y, mn.z);  out[2].max = v3new(mid.x, mx.y, mid.z);
    out[3].min = v3new(mid.x, mid.y, mn.z); out[3].max = v3new(mx.x, mx.y, mid.z);
    out[4].min = v3new(mn.x, mn.y, mid.z);  out[4].max = v3new(mid.x, mid.y, mx.z);
    out[5].min = v3new(mid.x, mn.y, mid.z); out[5].max = v3new(mx.x, mid.y, mx.z);
    out[6].min = v3new(mn.x, mid.y, mid.z); out[6].max = v3new(mid.x, mx.y, mx.z);
    out[7].min = v3new(mid.x, mid.y, mid.z); out[7].max = v3new(mx.x, mx.y, mx.z);
}
static uint32_t push_node(oracle_t* o, const cube_t* cube, uint32_t* tris, uint32_t ntris)
{
    if (o->nnodes == o->capnodes) {
        o->capnodes = o->capnodes ? o->capnodes * 2 : 64;
        o->nodes = (octnode_t*)realloc(o->nodes, o->capnodes * sizeof(octnode_t));
        o->cubes = (cube_t*)realloc(o->cubes, o->capnodes * sizeof(cube_t));
    }
    uint32_t idx = o->nnodes++;
    o->cubes[idx] = *cube;
    memset(&o->nodes[idx], 0, sizeof(octnode_t));
    o->nodes[idx].is_leaf = 1;
    o->nodes[idx].tris = tris;
    o->nodes[idx].ntris = ntris;
    return idx;
}
static void split_node(oracle_t* o, uint32_t node_idx, uint32_t level)       /* OCT:94-146 */
{
    if (!o->nodes[node_idx].is_leaf) return;
    if (o->nodes[node_idx].ntris <= o->tris_per_leaf || level > 8) return;

    cube_t kids[8];
    child_cubes(&o->cubes[node_idx], kids);
    uint32_t* parent_tris = o->nodes[node_idx].tris;
    uint32_t parent_n = o->nodes[node_idx].ntris;
    uint32_t child_idx[8];
    for (int i = 0; i < 8; ++i) {
        uint32_t* list = (uint32_t*)malloc((parent_n ? parent_n : 1) * sizeof(uint32_t));
        uint32_t n = 0;
        for (uint32_t k = 0; k < parent_n; ++k)                              /* OCT:374-391 */
            if (tri_cube_sat(&kids[i], &o->verts[3 * parent_tris[k]])) list[n++] = parent_tris[k];
        child_idx[i] = push_node(o, &kids[i], list, n);                      /* cube index == node index */
    }
    free(parent_tris);
    o->nodes[node_idx].is_leaf = 0;
    o->nodes[node_idx].tris = NULL;
    o->nodes[node_idx].ntris = 0;
    memcpy(o->nodes[node_idx].child, child_idx, sizeof child_idx);
    for (int i = 0; i < 8; ++i) split_node(o, child_idx[i], level + 1);
}
static void build_octree(oracle_t* o)                                        /* OCT:66-81, 315-342 */
{
    cube_t trunk = { { 3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f },
                     { -3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f } };
    for (uint32_t i = 0; i < o->ntri * 3; ++i) {
        v3 p = o->verts[i];
        trunk.min.x = fminf(trunk.min.x, p.x); trunk.min.y = fminf(trunk.min.y, p.y); trunk.min.z = fminf(trunk.min.z, p.z);
        trunk.max.x = fmaxf(trunk.max.x, p.x); trunk.max.y = fmaxf(trunk.max.y, p.y); trunk.max.z = fmaxf(trunk.max.z, p.z);
    }
    uint32_t* all = (uint32_t*)malloc((o->ntri ? o->ntri : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < o->ntri; ++i) all[i] = i;
    push_node(o, &trunk, all, o->ntri);
    split_node(o, 0, 0);
}

/* ---------------------------------------------------------------- traversal */
static void leaf_closest(const oracle_t* o, const ray_t* ray, const uint32_t* tris, uint32_t n,
                         hit_t* best, counters_t* cnt)                       /* OCT:249-272 */
{
    best->hit = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const v3* tv = &o->verts[3 * tris[k]];
        float t, u, v;
        if (cnt) cnt->tris_tested++;
        if (!mt_late_out(ray, tv[0], tv[1], tv[2], &t, &u, &v)) continue;
        if (!best->hit || t < best->t) { best->hit = 1; best->t = t; best->u = u; best->v = v; best->prim = tris[k]; }
    }
}
static int cube_contains(const cube_t* c, v3 p)                              /* OCT:34-45 */
{
    if (p.x < c->min.x || p.x > c->max.x || p.y < c->min.y || p.y > c->max.y || p.z < c->min.z || p.z > c->max.z)
        return 0;
    return 1;
}
static int intersect_node(const oracle_t* o, const ray_t* ray, const ray_t* inv_ray, uint32_t node_idx,
                          hit_t* out, counters_t* cnt)                       /* OCT:148-206 */
{
    const octnode_t* nd = &o->nodes[node_idx];
    if (nd->is_leaf) {
        hit_t h;
        leaf_closest(o, ray, nd->tris, nd->ntris, &h, cnt);
        if (!h.hit) return 0;
        v3 hp = vadd(ray->pos, vscale(ray->dir, h.t));                       /* OCT:164 */
        if (!cube_contains(&o->cubes[node_idx], hp)) return 0;
        *out = h;
        return 1;
    }
    if (cnt) cnt->nodes_visited++;
    uint32_t idx[8]; float dist[8]; int n = 0;
    for (int i = 0; i < 8; ++i) {
        float t;
        if (cnt) cnt->cubes_tested++;
        if (slab(inv_ray, &o->cubes[nd->child[i]], &t)) { idx[n] = nd->child[i]; dist[n] = t; ++n; }
    }
    /* stable sort by tmin ascending (Rust sort_by is stable; OCT:183).  A NaN distance would
     * panic in the reference (partial_cmp().unwrap()); here it compares as "not less". */
    for (int i = 1; i < n; ++i) {
        uint32_t ci = idx[i]; float di = dist[i]; int j = i - 1;
        while (j >= 0 && di < dist[j]) { idx[j + 1] = idx[j]; dist[j + 1] = dist[j]; --j; }
        idx[j + 1] = ci; dist[j + 1] = di;
    }
    for (int i = 0; i < n; ++i)
        if (intersect_node(o, ray, inv_ray, idx[i], out, cnt)) return 1;
    return 0;
}
/* Intersector::intersect_ray — OCT:240-246 (octree) or no_acceleration_intersector.rs:13-41 */
static int intersect_ray(const oracle_t* o, const ray_t* ray, int brute, hit_t* out, counters_t* cnt)
{
    if (brute) {
        /* geometries in order, triangles in order, strict `<` keeps the first of equal t */
        hit_t best; best.hit = 0;
        for (uint32_t k = 0; k < o->ntri; ++k) {
            const v3* tv = &o->verts[3 * k];
            float t, u, v;
            if (cnt) cnt->tris_tested++;
            if (!mt_late_out(ray, tv[0], tv[1], tv[2], &t, &u, &v)) continue;
            if (!best.hit || t < best.t) { best.hit = 1; best.t = t; best.u = u; best.v = v; best.prim = k; }
        }
        *out = best;
        return best.hit;
    }
    ray_t inv = { ray->pos, { 1.0f / ray->dir.x, 1.0f / ray->dir.y, 1.0f / ray->dir.z } };
    out->hit = 0;
    return intersect_node(o, ray, &inv, 0, out, cnt);
}

/* ---------------------------------------------------------------- camera.rs */
static void cam_update(camera_t* c)                                          /* camera.rs:92-98 */
{
    float rx[16], ry[16], t[16], tr[16];
    mat_rot_x(c->x_angle, rx);
    mat_rot_y(c->y_angle, ry);
    oracle_mat_mul(rx, ry, t);
    oracle_mat_mul(t, c->base_rot, c->rot);
    mat_translate(c->pos, tr);
    oracle_mat_mul(c->rot, tr, t);
    oracle_mat_mul(t, c->base_orient, c->orient);
}
static void cam_init(camera_t* c, uint32_t w, uint32_t h, const float* m, float fov_deg)  /* camera.rs:22-61 */
{
    memset(c, 0, sizeof *c);
    memcpy(c->base_orient, m, sizeof c->base_orient);
    memcpy(c->base_rot, m, sizeof c->base_rot);
    c->base_rot[3] = 0.0f; c->base_rot[7] = 0.0f; c->base_rot[11] = 0.0f;
    c->base_rot[12] = 0.0f; c->base_rot[13] = 0.0f; c->base_rot[14] = 0.0f; c->base_rot[15] = 1.0f;
    float fov = fov_deg * 3.14159274101257324f / 180.0f;                     /* std::f32::consts::PI */
    float half_fov = 0.5f * fov;
    c->max_x = 1.0f * tanf(half_fov);
    c->max_y = 1.0f * tanf(half_fov);
    c->width = w; c->height = h;
    cam_update(c);
}
static ray_t cam_get_ray(const camera_t* c, uint32_t u, uint32_t v, float xi1, float xi2)   /* camera.rs:80-90 */
{
    float dir_x = -c->max_x + 2.0f * c->max_x * (((float)u + xi1) / (float)c->width);
    float dir_y = -c->max_y + 2.0f * c->max_y * (((float)v + xi2) / (float)c->height);
    float d4[4] = { dir_x, -dir_y, 1.0f, 1.0f }, o4[4] = { 0.0f, 0.0f, 0.0f, 1.0f }, rd[4], ro[4];
    oracle_mat_vec(c->rot, d4, rd);
    oracle_mat_vec(c->orient, o4, ro);
    ray_t r = { { ro[0], ro[1], ro[2] }, { rd[0], rd[1], rd[2] } };
    return r;
}

/* ---------------------------------------------------------------- sample_generator.rs */
static void build_table(oracle_t* o)                                         /* :15-24, :36-52 */
{
    o->table = (v3*)malloc(NUM_SAMPLES * sizeof(v3));
    for (uint32_t i = 0; i < NUM_SAMPLES; ++i) {
        for (uint32_t attempt = 0;; ++attempt) {
            uint32_t h[4] = { i, attempt, DIM_TABLE, (uint32_t)o->seed };
            pcg4d(h);
            /* random_range(-1.0..1.0): value0_1 * (high - low) + low */
            v3 d = v3new(u01(h[0]) * 2.0f + -1.0f, u01(h[1]) * 2.0f + -1.0f, u01(h[2]) * 2.0f + -1.0f);
            if (dot(d, d) < 1.0f) { o->table[i] = normalized(d); break; }
        }
    }
}

/* ---------------------------------------------------------------- shading */
static inline rgb_t rgb_new(float r, float g, float b) { rgb_t c = { r, g, b }; return c; }
static inline rgb_t rgb_add(rgb_t a, rgb_t b) { return rgb_new(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline rgb_t rgb_scale(rgb_t a, float s) { return rgb_new(a.r * s, a.g * s, a.b * s); }
static inline rgb_t rgb_mul(rgb_t a, rgb_t b) { return rgb_new(a.r * b.r, a.g * b.g, a.b * b.b); }

/* x.powf(32.0), mod.rs:255: five squarings in double, one rounding to f32. */
float oracle_pow32(float x)
{
    double d = (double)x;
    d = d * d; d = d * d; d = d * d; d = d * d; d = d * d;
    return (float)d;
}

static v3 calc_normal(const oracle_t* o, const hit_t* h)                     /* mod.rs:198-205 */
{
    const v3* tv = &o->verts[3 * h->prim];
    return normalized(cross(vsub(tv[1], tv[0]), vsub(tv[2], tv[0])));
}
static rgb_t get_texel(const texture_t* t, float u, float v)                 /* texture.rs:21-27 */
{
    /* Rust `as usize` saturates and maps NaN to 0; an index past the end panics in the
     * reference — here it is clamped to the last texel (documented divergence on UB input). */
    float fx = u * (float)t->w, fy = v * (float)t->h;
    uint64_t x = fx > 0.0f ? (fx >= 1.8446744e19f ? UINT64_MAX : (uint64_t)fx) : 0;
    uint64_t y = fy > 0.0f ? (fy >= 1.8446744e19f ? UINT64_MAX : (uint64_t)fy) : 0;
    uint64_t n = (uint64_t)t->w * t->h;
    uint64_t idx = (y > n ? n : y) * t->w + (x > n ? n : x);
    if (idx >= n) idx = n - 1;
    return t->data[idx];
}
static rgb_t shade(const oracle_t* o, const ray_t* ray, const hit_t* hit, v3 normal, int brute, counters_t* cnt)  /* mod.rs:207-261 */
{
    rgb_t accum = rgb_new(0, 0, 0);
    v3 hit_point = vadd(ray->pos, sscale(hit->t, ray->dir));
    for (uint32_t li = 0; li < o->nlights; ++li) {
        const light_t* light = &o->lights[li];
        ray_t to_light = { hit_point, vsub(light->pos, hit_point) };
        float ndl = dot(normal, normalized(to_light.dir));
        if (ndl < 0.0f) continue;
        int blocked = 0;
        ray_t off = { vadd(to_light.pos, vscale(to_light.dir, 0.01f)), to_light.dir };
        hit_t sh;
        if (cnt) cnt->rays_shadow++;
        if (intersect_ray(o, &off, brute, &sh, cnt))
            if (sh.t > 0.01f && sh.t < 1.0f) blocked = 1;
        if (!blocked) {
            const material_t* m = &o->mats[o->tri_geom[hit->prim]];
            rgb_t diffuse = m->kind == 0 ? m->rgb : get_texel(&o->texs[m->tex], hit->u, hit->v);
            v3 view_ray = normalized(ray->dir);
            v3 refl = vsub(sscale(2.0f * ndl, normal), normalized(to_light.dir));
            float spec = oracle_pow32(dot(view_ray, refl));
            rgb_t c = rgb_add(rgb_scale(diffuse, ndl), rgb_scale(rgb_new(1.0f, 1.0f, 1.0f), spec));
            accum = rgb_add(accum, rgb_mul(c, light->color));
        }
    }
    return accum;
}
static ray_t reflection_ray(const oracle_t* o, const hit_t* hit, const ray_t* ray, v3 normal, uint32_t start)  /* mod.rs:178-196 */
{
    uint32_t idx = start;                                                    /* sample_generator.rs:31-34 */
    v3 d = o->table[idx];
    while (dot(d, normal) <= 0.0f) {                                         /* sample_generator.rs:26-29 */
        idx = (idx + 1) % SAMPLE_MAX;
        d = o->table[idx];
    }
    v3 hp = vadd(ray->pos, sscale(hit->t, ray->dir));
    hp = vadd(hp, sscale(0.00001f, d));
    ray_t r = { hp, d };
    return r;
}

/* Radiance tree bookkeeping.  Nodes are numbered breadth-first: the primary hit is node 0,
 * a node at level l has k_l = spread*(recursions-l) children.  The node id is the RNG
 * dimension of the bounce ray that reaches it, and the slot of its direct-light term. */
typedef struct {
    uint32_t pixel, sampleno;
    rgb_t* node_L;        /* optional: direct-light term per node (black when not reached) */
    uint8_t* node_hit;    /* optional */
    uint32_t level_first[16];
} tree_ctx_t;

static rgb_t compute_radiance(const oracle_t* o, const ray_t* ray, const hit_t* hit, uint32_t recursions,
                              uint32_t level, uint32_t index_in_level, tree_ctx_t* tc, int brute, counters_t* cnt)  /* mod.rs:132-176 */
{
    v3 normal = calc_normal(o, hit);
    rgb_t radiance = shade(o, ray, hit, normal, brute, cnt);
    uint32_t node = tc->level_first[level] + index_in_level;
    if (tc->node_L) { tc->node_L[node] = radiance; tc->node_hit[node] = 1; }
    if (recursions < 1) return radiance;
    uint32_t num_sub = o->spread * recursions;
    rgb_t sum = rgb_new(0, 0, 0);
    for (uint32_t k = 0; k < num_sub; ++k) {
        uint32_t child_index = index_in_level * num_sub + k;
        uint32_t child_node = tc->level_first[level + 1] + child_index;
        uint32_t h[4] = { tc->pixel, tc->sampleno, 1u + child_node, (uint32_t)o->seed };
        pcg4d(h);
        ray_t sub = reflection_ray(o, hit, ray, normal, table_index(h[0]));
        hit_t sh;
        if (cnt) cnt->rays_bounce++;
        rgb_t x = rgb_new(0, 0, 0);
        if (intersect_ray(o, &sub, brute, &sh, cnt))
            x = compute_radiance(o, &sub, &sh, recursions - 1, level + 1, child_index, tc, brute, cnt);
        sum = rgb_add(sum, x);
    }
    rgb_t sub_radiance = rgb_scale(sum, 1.0f / (float)num_sub);
    return rgb_add(radiance, sub_radiance);
}
static uint32_t tree_levels(const oracle_t* o, uint32_t* level_first)
{
    uint32_t count = 1, first = 0;
    for (uint32_t l = 0; l <= o->recursions; ++l) {
        level_first[l] = first;
        first += count;
        count *= o->spread * (o->recursions - l);
    }
    return first;       /* total nodes */
}
uint32_t oracle_tree_nodes(const oracle_t* o) { uint32_t lf[16]; return tree_levels(o, lf); }

static ray_t primary_ray(const oracle_t* o, uint32_t pixel, uint32_t sampleno)   /* mod.rs:93-96 */
{
    uint32_t h[4] = { pixel, sampleno, DIM_JITTER, (uint32_t)o->seed };
    pcg4d(h);
    uint32_t u = pixel % o->width;
    uint32_t v = (o->flags & ORACLE_FLAG_FIX_ROW_INDEX) ? pixel / o->width : pixel / o->height;
    return cam_get_ray(&o->cam, u, v, u01(h[0]), u01(h[1]));
}
static rgb_t sample_pixel(const oracle_t* o, uint32_t pixel, uint32_t sampleno, rgb_t* node_L, uint8_t* node_hit, counters_t* cnt)
{
    int brute = (o->flags & ORACLE_FLAG_BRUTE_FORCE) != 0;
    ray_t ray = primary_ray(o, pixel, sampleno);
    hit_t hit;
    tree_ctx_t tc;
    tc.pixel = pixel; tc.sampleno = sampleno; tc.node_L = node_L; tc.node_hit = node_hit;
    tree_levels(o, tc.level_first);
    if (cnt) cnt->rays_primary++;
    if (!intersect_ray(o, &ray, brute, &hit, cnt)) return rgb_new(0, 0, 0);
    if (cnt) cnt->primary_hits++;
    return compute_radiance(o, &ray, &hit, o->recursions, 0, 0, &tc, brute, cnt);
}
static void add_sample(oracle_t* o, uint32_t pixel, rgb_t c)                 /* film.rs:20-24 */
{
    o->sum[pixel] = rgb_add(o->sum[pixel], c);
    o->sumsq[pixel] = rgb_add(o->sumsq[pixel], rgb_new(c.r * c.r, c.g * c.g, c.b * c.b));
    o->nsamp[pixel] += 1;
}

/* ---------------------------------------------------------------- public API */
oracle_t* oracle_create(const float* tri_verts, const uint32_t* tri_geom, uint32_t ntri,
                        const uint32_t* mat_kind, const float* mat_rgb, const uint32_t* mat_tex, uint32_t nmats,
                        const float* light_pos_color, uint32_t nlights,
                        const uint32_t* tex_dims, const float* tex_data, uint32_t ntex,
                        const float* cam_matrix16, float fov_deg,
                        uint32_t width, uint32_t height, uint32_t tris_per_leaf,
                        uint32_t recursions, uint32_t spread, uint64_t seed, uint32_t flags)
{
    oracle_t* o = (oracle_t*)calloc(1, sizeof(oracle_t));
    o->ntri = ntri; o->ngeom = nmats; o->nlights = nlights; o->ntex = ntex;
    o->verts = (v3*)malloc((size_t)(ntri ? ntri : 1) * 3 * sizeof(v3));
    memcpy(o->verts, tri_verts, (size_t)ntri * 9 * sizeof(float));
    o->tri_geom = (uint32_t*)malloc((size_t)(ntri ? ntri : 1) * sizeof(uint32_t));
    memcpy(o->tri_geom, tri_geom, (size_t)ntri * sizeof(uint32_t));
    o->mats = (material_t*)calloc(nmats ? nmats : 1, sizeof(material_t));
    for (uint32_t i = 0; i < nmats; ++i) {
        o->mats[i].kind = mat_kind[i];
        o->mats[i].rgb = rgb_new(mat_rgb[3 * i], mat_rgb[3 * i + 1], mat_rgb[3 * i + 2]);
        o->mats[i].tex = mat_tex[i];
    }
    o->lights = (light_t*)calloc(nlights ? nlights : 1, sizeof(light_t));
    for (uint32_t i = 0; i < nlights; ++i) {
        o->lights[i].pos = v3new(light_pos_color[6 * i], light_pos_color[6 * i + 1], light_pos_color[6 * i + 2]);
        o->lights[i].color = rgb_new(light_pos_color[6 * i + 3], light_pos_color[6 * i + 4], light_pos_color[6 * i + 5]);
    }
    size_t texels = 0;
    for (uint32_t i = 0; i < ntex; ++i) texels += (size_t)tex_dims[2 * i] * tex_dims[2 * i + 1];
    o->texdata = (rgb_t*)malloc((texels ? texels : 1) * sizeof(rgb_t));
    if (texels) memcpy(o->texdata, tex_data, texels * sizeof(rgb_t));
    o->texs = (texture_t*)calloc(ntex ? ntex : 1, sizeof(texture_t));
    size_t off = 0;
    for (uint32_t i = 0; i < ntex; ++i) {
        o->texs[i].w = tex_dims[2 * i]; o->texs[i].h = tex_dims[2 * i + 1];
        o->texs[i].data = o->texdata + off;
        off += (size_t)o->texs[i].w * o->texs[i].h;
    }
    o->width = width; o->height = height; o->tris_per_leaf = tris_per_leaf;
    o->recursions = recursions; o->spread = spread; o->seed = seed; o->flags = flags;
    cam_init(&o->cam, width, height, cam_matrix16, fov_deg);
    build_octree(o);
    build_table(o);
    size_t npix = (size_t)width * height;
    o->sum = (rgb_t*)calloc(npix ? npix : 1, sizeof(rgb_t));
    o->sumsq = (rgb_t*)calloc(npix ? npix : 1, sizeof(rgb_t));
    o->nsamp = (uint32_t*)calloc(npix ? npix : 1, sizeof(uint32_t));
    return o;
}
void oracle_destroy(oracle_t* o)
{
    if (!o) return;
    for (uint32_t i = 0; i < o->nnodes; ++i) free(o->nodes[i].tris);
    free(o->nodes); free(o->cubes); free(o->verts); free(o->tri_geom); free(o->mats); free(o->lights);
    free(o->texs); free(o->texdata); free(o->table); free(o->sum); free(o->sumsq); free(o->nsamp);
    free(o);
}
/* a re-seeded oracle equals one created with that seed: the direction table is a function of the seed too */
void oracle_set_seed(oracle_t* o, uint64_t seed) { o->seed = seed; free(o->table); build_table(o); }
void oracle_set_flags(oracle_t* o, uint32_t flags) { o->flags = flags; }

/* mod.rs:80-117 : 50 rows x width pixels x 1 sample; sample# of a pixel = samples it already holds */
uint32_t oracle_trace_frame_additive(oracle_t* o)
{
    uint32_t num_primary_rays = 0;
    for (int k = 0; k < 50; ++k) {
        for (uint32_t i = 0; i < o->width; ++i) {
            uint32_t idx = o->current_row * o->width + i;
            rgb_t c = sample_pixel(o, idx, o->nsamp[idx], NULL, NULL, &o->cnt);
            add_sample(o, idx, c);
        }
        num_primary_rays += o->width;
        o->current_row = (o->current_row + 1) % o->height;
    }
    return num_primary_rays;
}

typedef struct { oracle_t* o; uint32_t row_end, spp; volatile uint32_t* next_row; counters_t cnt; } job_t;
static void* render_worker(void* p)
{
    job_t* j = (job_t*)p;
    oracle_t* o = j->o;
    for (;;) {
        uint32_t row = __sync_fetch_and_add(j->next_row, 1);
        if (row >= j->row_end) break;
        for (uint32_t i = 0; i < o->width; ++i) {
            uint32_t idx = row * o->width + i;
            for (uint32_t s = 0; s < j->spp; ++s) {
                rgb_t c = sample_pixel(o, idx, o->nsamp[idx], NULL, NULL, &j->cnt);
                add_sample(o, idx, c);
            }
        }
    }
    return NULL;
}
/* rows [row_begin,row_end) x spp samples per pixel, rows handed out dynamically to nthreads.
 * counts8 (optional) = {primary, bounce, shadow, nodes, cubes, tris, primary_hits, 0}. */
void oracle_render_rows(oracle_t* o, uint32_t row_begin, uint32_t row_end, uint32_t spp, uint32_t nthreads, uint64_t* counts8)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    volatile uint32_t next = row_begin;
    pthread_t th[256]; job_t jobs[256];
    for (uint32_t t = 0; t < nthreads; ++t) {
        memset(&jobs[t], 0, sizeof(job_t));
        jobs[t].o = o; jobs[t].row_end = row_end; jobs[t].spp = spp; jobs[t].next_row = &next;
    }
    if (nthreads == 1) render_worker(&jobs[0]);
    else {
        for (uint32_t t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, render_worker, &jobs[t]);
        for (uint32_t t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    counters_t c; memset(&c, 0, sizeof c);
    for (uint32_t t = 0; t < nthreads; ++t) {
        c.rays_primary += jobs[t].cnt.rays_primary; c.rays_bounce += jobs[t].cnt.rays_bounce; c.rays_shadow += jobs[t].cnt.rays_shadow;
        c.nodes_visited += jobs[t].cnt.nodes_visited; c.cubes_tested += jobs[t].cnt.cubes_tested; c.tris_tested += jobs[t].cnt.tris_tested;
        c.primary_hits += jobs[t].cnt.primary_hits;
    }
    if (counts8) {
        counts8[0] = c.rays_primary; counts8[1] = c.rays_bounce; counts8[2] = c.rays_shadow;
        counts8[3] = c.nodes_visited; counts8[4] = c.cubes_tested; counts8[5] = c.tris_tested;
        counts8[6] = c.primary_hits; counts8[7] = 0;
    }
}
void oracle_render(oracle_t* o, uint32_t spp, uint32_t nthreads, uint64_t* counts8)
{
    oracle_render_rows(o, 0, o->height, spp, nthreads, counts8);
}
/* one primary sample, returning the per-node direct-light terms too (stage-level parity) */
void oracle_sample_debug(oracle_t* o, uint32_t pixel, uint32_t sampleno, float* color3, float* node_L, uint8_t* node_hit)
{
    uint32_t n = oracle_tree_nodes(o);
    memset(node_L, 0, n * 3 * sizeof(float));
    memset(node_hit, 0, n);
    rgb_t c = sample_pixel(o, pixel, sampleno, (rgb_t*)node_L, node_hit, NULL);
    color3[0] = c.r; color3[1] = c.g; color3[2] = c.b;
}
void oracle_primary_ray(const oracle_t* o, uint32_t pixel, uint32_t sampleno, float* ray6)
{
    ray_t r = primary_ray(o, pixel, sampleno);
    ray6[0] = r.pos.x; ray6[1] = r.pos.y; ray6[2] = r.pos.z; ray6[3] = r.dir.x; ray6[4] = r.dir.y; ray6[5] = r.dir.z;
}
void oracle_get_ray(const oracle_t* o, uint32_t u, uint32_t v, float xi1, float xi2, float* ray6)
{
    ray_t r = cam_get_ray(&o->cam, u, v, xi1, xi2);
    ray6[0] = r.pos.x; ray6[1] = r.pos.y; ray6[2] = r.pos.z; ray6[3] = r.dir.x; ray6[4] = r.dir.y; ray6[5] = r.dir.z;
}
/* batched Intersector::intersect_ray.  mode 0 = octree (reference default), 1 = brute force.
 * out: tuv[3n] (untouched on miss), prim[n] (0xFFFFFFFF on miss). */
void oracle_intersect(const oracle_t* o, const float* rays6, uint32_t n, int mode, float* tuv, uint32_t* prim)
{
    for (uint32_t i = 0; i < n; ++i) {
        ray_t r = { { rays6[6 * i], rays6[6 * i + 1], rays6[6 * i + 2] }, { rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5] } };
        hit_t h;
        if (intersect_ray(o, &r, mode, &h, NULL)) { tuv[3 * i] = h.t; tuv[3 * i + 1] = h.u; tuv[3 * i + 2] = h.v; prim[i] = h.prim; }
        else prim[i] = 0xFFFFFFFFu;
    }
}
typedef struct { const oracle_t* o; const float* rays6; uint32_t n; int mode; float* tuv; uint32_t* prim; volatile uint32_t* next; } ijob_t;
static void* intersect_worker(void* p)
{
    ijob_t* j = (ijob_t*)p;
    for (;;) {
        uint32_t b = __sync_fetch_and_add(j->next, 256);
        if (b >= j->n) break;
        uint32_t e = b + 256 < j->n ? b + 256 : j->n;
        oracle_intersect(j->o, j->rays6 + 6 * (size_t)b, e - b, j->mode, j->tuv + 3 * (size_t)b, j->prim + b);
    }
    return NULL;
}
void oracle_intersect_mt(const oracle_t* o, const float* rays6, uint32_t n, int mode, float* tuv, uint32_t* prim, uint32_t nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    volatile uint32_t next = 0;
    ijob_t job = { o, rays6, n, mode, tuv, prim, &next };
    pthread_t th[256];
    for (uint32_t t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, intersect_worker, &job);
    for (uint32_t t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

void oracle_film_clear(oracle_t* o)                                          /* film.rs:37-41 */
{
    size_t npix = (size_t)o->width * o->height;
    memset(o->sum, 0, npix * sizeof(rgb_t)); memset(o->sumsq, 0, npix * sizeof(rgb_t)); memset(o->nsamp, 0, npix * sizeof(uint32_t));
}
void oracle_film_get(const oracle_t* o, float* sum3, float* sumsq3, uint32_t* n)
{
    size_t npix = (size_t)o->width * o->height;
    if (sum3) memcpy(sum3, o->sum, npix * sizeof(rgb_t));
    if (sumsq3) memcpy(sumsq3, o->sumsq, npix * sizeof(rgb_t));
    if (n) memcpy(n, o->nsamp, npix * sizeof(uint32_t));
}
void oracle_get_pixels(const oracle_t* o, float* rgb3)                       /* film.rs:43-47 */
{
    size_t npix = (size_t)o->width * o->height;
    for (size_t i = 0; i < npix; ++i) {
        float inv = 1.0f / (float)o->nsamp[i];
        rgb3[3 * i] = o->sum[i].r * inv; rgb3[3 * i + 1] = o->sum[i].g * inv; rgb3[3 * i + 2] = o->sum[i].b * inv;
    }
}
void oracle_get_estimated_variances(const oracle_t* o, float* rgb3)          /* film.rs:51-67 */
{
    size_t npix = (size_t)o->width * o->height;
    for (size_t i = 0; i < npix; ++i) {
        uint32_t n = o->nsamp[i];
        float nn1 = (float)(uint32_t)(n * (n - 1u));        /* u32 arithmetic; wraps where the reference would overflow */
        float n2n1 = (float)n * nn1;
        rgb3[3 * i]     = (o->sumsq[i].r / nn1 - o->sum[i].r * o->sum[i].r / n2n1) * 50.0f;
        rgb3[3 * i + 1] = (o->sumsq[i].g / nn1 - o->sum[i].g * o->sum[i].g / n2n1) * 50.0f;
        rgb3[3 * i + 2] = (o->sumsq[i].b / nn1 - o->sum[i].b * o->sum[i].b / n2n1) * 50.0f;
    }
}
/* tonemap.rs:4-10 + color.rs:85-95.  Rust f32::min/max return the non-NaN operand (fminf/fmaxf);
 * `as u8` truncates toward zero. */
uint32_t oracle_tonemap_pack(float r, float g, float b)
{
    float c[4] = { r / (1.0f + r), g / (1.0f + g), b / (1.0f + b), 1.0f };
    uint32_t q[4];
    for (int i = 0; i < 4; ++i) q[i] = (uint32_t)(uint8_t)(fmaxf(fminf(c[i], 1.0f), 0.0f) * 255.0f);
    return q[2] | q[1] << 8 | q[0] << 16 | q[3] << 24;
}
void oracle_get_tonemapped(const oracle_t* o, uint32_t* out)                 /* mod.rs:120-128 */
{
    size_t npix = (size_t)o->width * o->height;
    for (size_t i = 0; i < npix; ++i) {
        float inv = 1.0f / (float)o->nsamp[i];
        out[i] = oracle_tonemap_pack(o->sum[i].r * inv, o->sum[i].g * inv, o->sum[i].b * inv);
    }
}
void oracle_camera_move_rel(oracle_t* o, float x, float y, float z)          /* camera.rs:73-78 */
{
    o->cam.pos.x += x; o->cam.pos.y += y; o->cam.pos.z += z; cam_update(&o->cam);
}
void oracle_camera_add_x_angle(oracle_t* o, float r) { o->cam.x_angle += r; cam_update(&o->cam); }   /* camera.rs:63-66 */
void oracle_camera_add_y_angle(oracle_t* o, float r) { o->cam.y_angle += r; cam_update(&o->cam); }   /* camera.rs:68-71 */
void oracle_camera_get(const oracle_t* o, float* rot16, float* orient16, float* max_xy)
{
    memcpy(rot16, o->cam.rot, 64); memcpy(orient16, o->cam.orient, 64);
    max_xy[0] = o->cam.max_x; max_xy[1] = o->cam.max_y;
}
void oracle_sample_table(const oracle_t* o, float* out) { memcpy(out, o->table, NUM_SAMPLES * sizeof(v3)); }
uint32_t oracle_current_row(const oracle_t* o) { return o->current_row; }
/* out[8] = nodes, inner, leaves, empty leaves, max depth, triangle refs, max leaf size, 0 */
static void stats_rec(const oracle_t* o, uint32_t idx, uint32_t depth, uint32_t* s)
{
    const octnode_t* nd = &o->nodes[idx];
    if (nd->is_leaf) {
        s[2]++; if (nd->ntris == 0) s[3]++;
        if (depth > s[4]) s[4] = depth;
        s[5] += nd->ntris; if (nd->ntris > s[6]) s[6] = nd->ntris;
    } else {
        s[1]++;
        for (int i = 0; i < 8; ++i) stats_rec(o, nd->child[i], depth + 1, s);
    }
}
void oracle_octree_stats(const oracle_t* o, uint32_t* out8)
{
    memset(out8, 0, 8 * sizeof(uint32_t));
    out8[0] = o->nnodes;
    stats_rec(o, 0, 0, out8);
}
void oracle_counters(const oracle_t* o, uint64_t* c8)
{
    c8[0] = o->cnt.rays_primary; c8[1] = o->cnt.rays_bounce; c8[2] = o->cnt.rays_shadow; c8[3] = o->cnt.nodes_visited;
    c8[4] = o->cnt.cubes_tested; c8[5] = o->cnt.tris_tested; c8[6] = o->cnt.primary_hits; c8[7] = 0;
}
