// kernels.hip — gfx950 (CDNA4, wave64) kernels of the render path.
//
// Wavefront structure of one pass over N primary samples.  The samples are cut into
// chunks of ps.chunk; a chunk keeps its identity through every round: its rays of round r live in
// region `chunk` of that round's queue (ps.region records, the worst case) — radiance rays packed
// from the front, shadow rays packed from the back — with (n_radiance, n_shadow) in counts[chunk].
//
//   trace  round r : closest hit of every ray of the round (round 0: ray-gen from the sample index,
//                    mod.rs:93-98).  Persistent waves; a lane that finishes its ray is refilled from
//                    the wave's current chunk (wave64 ballot + prefix popcount, no atomics; one atomic
//                    per CHUNK to pull the next one), so lanes stay busy although ray lengths differ.
//                    Finished rays wait in their lanes; at the next refill a radiance ray stores its
//                    hit record and (true-closest mode) an unblocked shadow ray its light term
//                    (mod.rs:232-256).  A shadow ray is rebuilt from its record's hit point and the light.
//   confirm round r: reference-default semantics only: the octree's answer derived from the true
//                    closest hit (traverse.hpp, confirm_walk); settles the shadow rays of the round.
//   shade  round r : one wave per chunk, chunks pulled like the trace kernel pulls them.  Hits are
//                    compacted into an LDS index list (ballot + prefix popcount), then shade() set-up —
//                    normal, Phong terms, shadow record (mod.rs:198-257) — and the level r+1 reflection
//                    rays (mod.rs:178-196) are appended to the chunk's region of the next queue with
//                    wave-local counters (32-byte records, non-temporal stores: "ray records" below).
//   resolve        : per pixel, combine the per-node light terms in the reference's own summation
//                    order (mod.rs:154-175) and add the samples to the film in sample order
//                    (film.rs:20-24).
// A frame is rendered as several such passes at once: renderer.cpp deals the rows to concurrent slices,
// each with its own stream and pass buffers, so the kernels of one slice fill the drain window at the end
// of another slice's (persistent) trace launch.  The pixels of a pass are walked in 8x8 tiles (pass_pixel).
//
// Numerics: compiled with -ffp-contract=off; everything that decides a result (Moller-Trumbore,
// shading, camera, film, tonemap) is IEEE f32 in the reference's operation order.  Only the BVH box
// tests are free-form: boxes are padded (bvh.cpp) and the test is conservative.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>
#include "device_math.hpp"
#include "device_types.hpp"
#include "kernels.hpp"
#include "traverse.hpp"

namespace mi355rt {

typedef const uint32_t __attribute__((address_space(4))) * const_u1_ptr;    // constant address space: a wave-uniform address loads through the scalar cache
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr uint32_t kMiss = 0xFFFFFFFFu;
#ifndef MI355RT_FUSED_FENCE_AGENT
#define MI355RT_FUSED_FENCE_AGENT 0
#endif
#ifndef MI355RT_PRIMARY_BLOCKS
#define MI355RT_PRIMARY_BLOCKS 7                // blocks per CU the primary trace kernel is compiled for: 69 VGPRs; 8 blocks = 64 VGPRs + 20 B of scratch, measured in profiles/r03_notes.md
#endif
#ifndef MI355RT_WIDE_BLOCKS
#define MI355RT_WIDE_BLOCKS 5                   // blocks per CU the trace kernels are compiled for when they walk the 4-wide tree (deeper stacks: ~32 LDS rows per block)
#endif
#ifndef MI355RT_CONFIRM_BLOCKS
#define MI355RT_CONFIRM_BLOCKS 5               // blocks per CU the confirm kernel is compiled for (A/B knob, profiles/r02_notes.md)
#endif
#ifndef MI355RT_SHADE_P_BLOCKS
#define MI355RT_SHADE_P_BLOCKS 5               // blocks per CU the shade kernels are compiled for and launched with (A/B knobs)
#endif
#ifndef MI355RT_SHADE_PW_BLOCKS
#define MI355RT_SHADE_PW_BLOCKS 4              // primary shade kernel with the confirm walk inside: 128 VGPRs, no scratch (5 blocks: 80 B of scratch, 5 % slower)
#endif
#ifndef MI355RT_SHADE_SW_BLOCKS
#define MI355RT_SHADE_SW_BLOCKS 5              // secondary shade kernel with the confirm walk of the radiance hits inside
#endif
#ifndef MI355RT_SHADE_S_BLOCKS
#define MI355RT_SHADE_S_BLOCKS 7
#endif
#ifndef MI355RT_SHADE_PULL
#define MI355RT_SHADE_PULL 0                   // 0: like the trace kernel (ps.pull_mode); 2: static striding
#endif
constexpr uint32_t kShadePullMode = MI355RT_SHADE_PULL;
#ifndef MI355RT_CONFIRM_PULL
#define MI355RT_CONFIRM_PULL 0                 // 0: like the trace kernel (ps.pull_mode); 2: static striding (A/B knob)
#endif
constexpr uint32_t kConfirmPullMode = MI355RT_CONFIRM_PULL;
constexpr bool kFusedFenceAgent = MI355RT_FUSED_FENCE_AGENT != 0;      // A/B knob of the build (see phase_fence)
#ifndef MI355RT_INNER_STEPS
#define MI355RT_INNER_STEPS 2
#endif
constexpr int kInnerStepsPerIteration = MI355RT_INNER_STEPS;     // measured: 1 -> 2 takes 9 % off the trace kernel, 3 and 4 add nothing

// One device word sustains only ~88 atomics/us on this chip: the statistics counters that every wave
// flushes into exist in kShards copies (the host adds them up), and the work cursor is touched
// sparingly (pull_chunk).
__device__ __forceinline__ uint32_t global_wave_id() { return blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); }

// Pull the next chunk for this wave (wave-uniform).
//   mode 4 (default): dynamic, ps.ncursors (64) cursors that lie 64 KiB apart.  Cursor k hands out the
//     chunks k, k + ncursors, ... (every cursor sees a uniform sample of the image, so they run dry
//     together); a wave starts on cursor (wave % ncursors) and moves on to the next one when its own is
//     dry, at most 8 of them.  Returning atomics to ONE word — or to several words in one cache line, or
//     256 B apart — serialise at ~88/us for the whole chip: with one cursor the 131 072 chunk pulls of a
//     launch put a 1.5 ms floor under every trace launch.
//   mode 2: static striding, no atomics (wave w takes chunks w, w + nwaves, ...).  Measured on thai2:
//     mean wave busy 2.56 ms but last wave done at 3.89 ms — a third of the machine idles on imbalance.
//   mode 0: one cursor (kept for the A/B in profiles/r01_notes.md).
// Tried and dropped: a relaxed agent-scope load in front of the atomic (2.5x slower), 3/4 static + a
// dynamic rest, several chunks per pull, draining more than 8 cursors, heaviest-chunks-first order.
constexpr uint32_t kCursorStride = kCtrlWordsPerRound / kMaxCursors;   // u32 words between cursors (64 KiB): atomics to nearby lines serialise on one memory channel
static_assert(kCursorStride == 16384 && kShadeCursorOffset == 2 * kConfirmCursorOffset && kLiveCountOffset == 3 * kConfirmCursorOffset, "cursor layout (device_types.hpp) and the reset loop of resolve_kernel");
struct PullState { bool first = true; uint32_t shard = 0u, tries = 0u, left = 0u, part = 0xFFFFFFFFu; };
// the live-chunk lists of a pass (DPass::live): cursor k hands out the entries of list k instead of the chunk numbers k, k + ncursors, ...
struct LiveLists { const uint32_t* list = nullptr; const uint32_t* count = nullptr; uint32_t cap = 0u; };
// (the build knobs that make the shade / confirm launches stride over ALL chunks switch the lists off for every launch: a launch that visits chunks the
// others skipped would read ray counts nobody wrote)
constexpr bool kLiveListsOk = kShadePullMode == 0u && kConfirmPullMode == 0u;
__device__ __forceinline__ LiveLists live_lists(const DPass& ps) { LiveLists l; if (kLiveListsOk) { l.list = ps.live; l.count = ps.live_count; l.cap = ps.live_cap; } return l; }
constexpr uint32_t kWholeChunk = 0xFFFFFFFFu;

// tail / split (trace kernels only; 0 / 1 elsewhere): the LAST `tail` chunks of every cursor's sequence are handed out in `split`
// parts (st.part = 0 .. split-1; kWholeChunk otherwise), several waves sharing a chunk.  A persistent launch ends when its slowest
// wave does, and with whole chunks the waves run out of work up to ~1.5 chunk durations apart — a fifth of a launch of 8 chunks per
// wave (one rank's share of a strong-scaled frame), a tenth of a full-size one (profiles/r03_notes.md).  Splitting only the tail keeps
// the per-pull cost (count load, culling test, a colder start) off the bulk of the chunks.
__device__ __forceinline__ bool pull_chunk(uint32_t* cursor, uint32_t nchunks, uint32_t mode, uint32_t ncursors, uint32_t group, PullState& st, uint32_t& chunk,
                                           uint32_t tail = 0u, uint32_t split_shift = 0u, const LiveLists live = LiveLists())
{
    if (mode == 4u && live.list != nullptr) {
        // live lists: cursor k's sequence is list k (filled by the primary shade launch); otherwise as below
        if (st.first) { st.first = false; st.shard = global_wave_id() % ncursors; }
        const uint32_t max_tries = ncursors < 8u ? ncursors : 8u;
        while (st.tries < max_tries) {
            // (lists and counts were written by an EARLIER launch: wave-uniform addresses, read through the scalar cache, which every launch starts with empty)
            const uint32_t len = ((const_u1_ptr)(uintptr_t)live.count)[(size_t)st.shard * kCursorStride + kLiveCountOffset];
            uint32_t v = 0xFFFFFFFFu;
            if (len != 0u) { if (lane_id() == 0) v = atomicAdd(&cursor[(size_t)st.shard * kCursorStride], 1u); v = bcast_first(v); }
            uint32_t j = v; st.part = kWholeChunk;
            if (tail != 0u && v != 0xFFFFFFFFu) {
                const uint32_t head = len - (tail < len ? tail : len);
                if (v >= head) { const uint32_t jj = v - head; j = head + (jj >> split_shift); st.part = jj & ((1u << split_shift) - 1u); }
            }
            if (j < len) { chunk = ((const_u1_ptr)(uintptr_t)live.list)[(size_t)st.shard * live.cap + j]; return true; }
            st.shard = (st.shard + 1u) % ncursors;      // dry for good
            ++st.tries;
        }
        return false;
    }
    if (mode == 4u) {
        if (st.first) { st.first = false; st.shard = global_wave_id() % ncursors; }
        else if (st.left > 0u && chunk + 1u < nchunks) { --st.left; ++chunk; return true; }     // rest of the group pulled last time
        const uint32_t max_tries = ncursors < 8u ? ncursors : 8u;
        while (st.tries < max_tries) {
            uint32_t v = 0u;
            if (lane_id() == 0) v = atomicAdd(&cursor[(size_t)st.shard * kCursorStride], 1u);
            v = bcast_first(v);
            if (tail != 0u && group == 1u) {
                const uint32_t len = st.shard < nchunks ? (nchunks - st.shard + ncursors - 1u) / ncursors : 0u;     // chunks of this cursor
                const uint32_t head = len - (tail < len ? tail : len);
                uint32_t j = v; st.part = kWholeChunk;
                if (v >= head) { const uint32_t jj = v - head; j = head + (jj >> split_shift); st.part = jj & ((1u << split_shift) - 1u); }
                if (j < len) { chunk = st.shard + j * ncursors; return true; }
            } else {
                const uint32_t c = (st.shard + v * ncursors) * group;
                if (c < nchunks) { chunk = c; st.left = group - 1u; return true; }
            }
            st.shard = (st.shard + 1u) % ncursors;      // dry for good
            ++st.tries;
        }
        return false;
    }
    if (mode == 2u) {
        if (st.first) { st.first = false; chunk = global_wave_id(); }
        else chunk += gridDim.x * kWavesPerBlock;
        return chunk < nchunks;
    }
    uint32_t v = 0u;
    if (lane_id() == 0) v = atomicAdd(cursor, 1u);
    chunk = bcast_first(v);
    return chunk < nchunks;
}

// Streamed data (ray records, hit records, slot bookkeeping: written once, read once, gigabytes per pass) can be marked
// non-temporal so that it does not evict the BVH from the L2.  MI355RT_NT is a build knob (bit 0: ray-record loads of the
// trace kernel, 1: its hit stores, 2: the shade kernel's streams, 3: the confirm kernel's loads); see profiles/r02_notes.md.
#ifndef MI355RT_NT
#define MI355RT_NT 4
#endif
constexpr uint32_t kNT = MI355RT_NT;
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <uint32_t BIT> __device__ __forceinline__ float4 ld4(const float4* p)
{
    if constexpr ((kNT >> BIT) & 1u) { const v4f_t v = __builtin_nontemporal_load((const v4f_t*)p); return make_float4(v.x, v.y, v.z, v.w); }
    else return *p;
}
template <uint32_t BIT> __device__ __forceinline__ void st4(float4* p, float4 a)
{
    if constexpr ((kNT >> BIT) & 1u) { v4f_t v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w; __builtin_nontemporal_store(v, (v4f_t*)p); }
    else *p = a;
}
template <uint32_t BIT> __device__ __forceinline__ uint32_t ld1(const uint32_t* p)
{
    if constexpr ((kNT >> BIT) & 1u) return __builtin_nontemporal_load(p); else return *p;
}
template <uint32_t BIT> __device__ __forceinline__ void st1(uint32_t* p, uint32_t a)
{
    if constexpr ((kNT >> BIT) & 1u) __builtin_nontemporal_store(a, p); else *p = a;
}

// wave64 compaction: every lane of the wave calls this; lanes with want == true get consecutive
// indices after the wave's running count `n` (ballot + prefix popcount, no atomic: the wave owns the
// region it appends to).
__device__ __forceinline__ uint32_t wave_append(bool want, uint32_t& n, uint32_t& n_new)
{
    const unsigned long long mask = __ballot(want);
    n_new = (uint32_t)__popcll(mask);
    const uint32_t idx = n + (uint32_t)__popcll(mask & lanemask_lt());
    n += n_new;
    return idx;
}

__device__ __forceinline__ f3 fetch_texel(const DScene& sc, uint32_t tex, float u, float v)     // texture.rs:21-27
{
    const DTexture t = sc.textures[tex];
    // `as usize`: truncation, saturating, NaN -> 0.  An index past the end panics in the reference;
    // here it is clamped to the last texel.
    const float fx = u * (float)t.width, fy = v * (float)t.height;
    unsigned long long x = fx > 0.0f ? (fx >= 1.8446744e19f ? ~0ull : (unsigned long long)fx) : 0ull;
    unsigned long long y = fy > 0.0f ? (fy >= 1.8446744e19f ? ~0ull : (unsigned long long)fy) : 0ull;
    const unsigned long long n = (unsigned long long)t.width * t.height;
    unsigned long long i = (y > n ? n : y) * t.width + (x > n ? n : x);
    if (i >= n) i = n - 1;
    const float* p = sc.texels + 3ull * (t.offset + i);
    return mk3(p[0], p[1], p[2]);
}

// Pass order of the pixels.  The rows of a pass are taken in groups of ps.row_group (8) and walked column by column
// inside a group, so that 64 consecutive samples are an 8x8 pixel tile and a chunk of 256 a 32x8 one: the
// primary rays of a wave are as coherent as they can be, and the chunk culling below tests compact tiles.
// (The film does not depend on the order: every pixel accumulates its own samples in sample order.)
// (ps.row_group: 8 rows, or the stripe height when the rows are dealt in stripes of fewer rows — a group never spans two stripes)
// entry i of the pass's row list; the list is a cyclic window of ps.row_wrap entries when the pass is a 50-row
// frame of trace_frame_additive (its rows are a run of the handle's device-resident owned-row list that may wrap)
__device__ __forceinline__ uint32_t pass_row(const DPass& ps, uint32_t i) { return ps.rows[i >= ps.row_wrap ? i - ps.row_wrap : i]; }
__device__ __forceinline__ void pass_column(const DPass& ps, uint32_t width, uint32_t p, uint32_t& first_row, uint32_t& nrows_in_group, uint32_t& x, uint32_t& y)
{
    const uint32_t rg = ps.row_group;                               // a power of two
    const uint32_t gs = width * rg, g = p / gs, q = p - g * gs;
    nrows_in_group = min(rg, ps.npix / width - g * rg);
    x = nrows_in_group == rg ? q >> ps.row_group_shift : q / nrows_in_group;
    y = q - x * nrows_in_group;
    first_row = ps.row0 + g * rg;
}
__device__ __forceinline__ uint32_t pass_pixel(const DPass& ps, uint32_t width, uint32_t p)
{
    uint32_t first_row, nr, x, y;
    pass_column(ps, width, p, first_row, nr, x, y);
    return pass_row(ps, first_row + y) * width + x;
}

// Order of the primary samples of a pass: sample gi is sample number s of the pass's p-th pixel, gi = ((s / G) * npix + p) * G + s % G
// with G = ps.sample_group consecutive samples of a pixel kept together (G divides the pass's samples per pixel):
//   G = 1 (rounds 1-2): one sample of every pixel, then the next: a wave is an 8x8 (128x2 ...) pixel tile at one sample number.
//   G = spp: ALL samples of a pixel are consecutive: the 64 lanes of a wave trace 64 samples of one pixel.
// Lanes that hold samples of the same pixel walk the tree together — their primary rays differ by a sub-pixel jitter, their shadow rays run
// from one pixel's footprint to the same light — and a fetch whose lanes share cache lines is cheaper on the pipe that bounds the trace
// kernels; but the work of a chunk gets lumpier (a chunk is all hits or all misses), which costs the tails of the launches.
// Measured in profiles/r03_notes.md.  The film does not depend on the order: every pixel accumulates its own samples in sample order.
__device__ __forceinline__ void sample_of(const DPass& ps, uint32_t gi, uint32_t& s, uint32_t& p)
{
    const uint32_t G = ps.sample_group, per = ps.npix * G;
    const uint32_t g = gi / per, r = gi - g * per;
    p = r / G;
    s = g * G + (r - p * G);
}
__device__ __forceinline__ size_t sample_index(const DPass& ps, uint32_t s, uint32_t p)
{
    const uint32_t G = ps.sample_group, g = s / G;
    return ((size_t)g * ps.npix + p) * G + (s - g * G);
}

// pixel -> primary ray, mod.rs:93-96 + camera.rs:80-90.  gi = index of the primary sample in the pass.
__device__ __forceinline__ void primary_sample(const DCamera& cam, const DPass& ps, const uint32_t* __restrict__ film_n, uint32_t gi,
                                               uint32_t& pixel, uint32_t& sampleno, f3& o, f3& d)
{
    if (ps.use_explicit) { pixel = ps.explicit_pixel; sampleno = ps.explicit_sampleno; }
    else {
        uint32_t s, p;
        sample_of(ps, gi, s, p);
        pixel = pass_pixel(ps, cam.width, p);
        sampleno = film_n[pixel] + s;
    }
    uint32_t h0 = pixel, h1 = sampleno, h2 = 0u, h3 = ps.seed;
    pcg4d(h0, h1, h2, h3);
    const uint32_t cu = pixel % cam.width;
    const uint32_t cv = (ps.flags & 1u) ? pixel / cam.width : pixel / cam.height;   // reference: idx / height
    const float dir_x = -cam.max_x + 2.0f * cam.max_x * div_rn((float)cu + u01(h0), (float)cam.width);
    const float dir_y = -cam.max_y + 2.0f * cam.max_y * div_rn((float)cv + u01(h1), (float)cam.height);
    const float vx = dir_x, vy = -dir_y, vz = 1.0f, vw = 1.0f;
    d.x = vx * cam.rot[0] + vy * cam.rot[4] + vz * cam.rot[8] + vw * cam.rot[12];
    d.y = vx * cam.rot[1] + vy * cam.rot[5] + vz * cam.rot[9] + vw * cam.rot[13];
    d.z = vx * cam.rot[2] + vy * cam.rot[6] + vz * cam.rot[10] + vw * cam.rot[14];
    o = mk3(cam.origin[0], cam.origin[1], cam.origin[2]);
}

// Frustum culling of a whole chunk of primary samples.  The samples of a chunk are a run of columns of one
// row group at one sample index; their (dir_x, dir_y) of camera.rs:81-84 lie in a rectangle whatever the
// jitter.  If that rectangle misses the screen-space bounds of every one of the (up to 16) top BVH subtree
// boxes, no ray of the chunk can hit anything: all of them miss (mod.rs:99-100).  Purely conservative: it
// only ever skips work whose outcome is "miss".
__device__ __forceinline__ bool chunk_is_culled(const DCamera& cam, const DPass& ps, uint32_t chunk, uint32_t n)
{
    if (!cam.cull_valid || ps.use_explicit || n == 0u) return false;
    const uint32_t g0 = chunk * ps.chunk, g1 = g0 + n - 1u;
    uint32_t s0, p0, s1, p1;
    sample_of(ps, g0, s0, p0); sample_of(ps, g1, s1, p1);
    // the chunk must lie inside ONE sample group (npix * G consecutive samples: the pass's pixels once, G samples each).  A chunk that runs into the next
    // group holds pixels from the end AND the start of the pixel order — with an image of fewer than chunk / G pixels it can even wrap past its own
    // first pixel, so p1 >= p0 alone does not tell
    const uint32_t per = ps.npix * ps.sample_group;
    if (p1 < p0 || g0 / per != g1 / per) return false;
    uint32_t fr0, nr0, xa, ya, fr1, nr1, xb, yb;
    pass_column(ps, cam.width, p0, fr0, nr0, xa, ya);
    pass_column(ps, cam.width, p1, fr1, nr1, xb, yb);
    if (fr0 != fr1) return false;                                    // straddles two row groups
    uint32_t row_lo = 0xFFFFFFFFu, row_hi = 0u;                      // all rows of the group (a superset of the chunk's)
    for (uint32_t y = 0; y < nr0; ++y) { const uint32_t r = pass_row(ps, fr0 + y); row_lo = min(row_lo, r); row_hi = max(row_hi, r); }
    const uint32_t ia = row_lo * cam.width + xa, ib = row_hi * cam.width + xb;
    const uint32_t va = (ps.flags & 1u) ? ia / cam.width : ia / cam.height, vb = (ps.flags & 1u) ? ib / cam.width : ib / cam.height;
    // same expressions as primary_sample with jitter 0 and 1 (monotonic in u, v), widened a little
    const float x_lo = -cam.max_x + 2.0f * cam.max_x * ((float)xa / (float)cam.width);
    const float x_hi = -cam.max_x + 2.0f * cam.max_x * (((float)xb + 1.0f) / (float)cam.width);
    const float y_lo = -cam.max_y + 2.0f * cam.max_y * ((float)va / (float)cam.height);
    const float y_hi = -cam.max_y + 2.0f * cam.max_y * (((float)vb + 1.0f) / (float)cam.height);
    const float ex = 1e-5f * cam.max_x, ey = 1e-5f * cam.max_y;
    bool all_outside = true;
    for (uint32_t k = 0; k < cam.cull_valid && all_outside; ++k) {
        const bool outside = x_hi + ex < cam.cull_rect[k][0] || x_lo - ex > cam.cull_rect[k][1] || y_hi + ey < cam.cull_rect[k][2] || y_lo - ey > cam.cull_rect[k][3];
        if (!outside) all_outside = false;
    }
    if (all_outside) return true;
    // second stage: the coverage mask (renderer.cpp, refresh_cull_mask).  The footprint's cells, one (row, word) pair per lane; the wave must be
    // whole for the ballot to see every pair (every caller is in wave-uniform control flow; if not: no culling).
    if (cam.cull_mask == nullptr || __builtin_amdgcn_read_exec() != ~0ull) return false;
    constexpr int G = (int)kCullGrid, wpr = G / 32;
    const float fx0 = (x_lo - ex - cam.mask_x0) * cam.mask_inv_cx, fx1 = (x_hi + ex - cam.mask_x0) * cam.mask_inv_cx;
    const float fy0 = (y_lo - ey - cam.mask_y0) * cam.mask_inv_cy, fy1 = (y_hi + ey - cam.mask_y0) * cam.mask_inv_cy;
    if (!(fx0 == fx0) || !(fx1 == fx1) || !(fy0 == fy0) || !(fy1 == fy1)) return false;
    // (the host widened every rectangle by 1/100 of a cell: the f32 rounding of these indices, < 1e-4 cells, cannot lose a touched cell)
    const int i0 = max((int)floorf(fminf(fmaxf(fx0, -4.0f), (float)G + 4.0f)), 0), i1 = min((int)floorf(fminf(fmaxf(fx1, -4.0f), (float)G + 4.0f)), G - 1);
    const int j0 = max((int)floorf(fminf(fmaxf(fy0, -4.0f), (float)G + 4.0f)), 0), j1 = min((int)floorf(fminf(fmaxf(fy1, -4.0f), (float)G + 4.0f)), G - 1);
    if (i1 < i0 || j1 < j0) return true;                                 // wholly outside the mask's domain: outside every rectangle of the first stage too
    const int w0 = i0 >> 5, nw = (i1 >> 5) - w0 + 1, nrows = j1 - j0 + 1;
    if (nw * nrows > 64) return false;                                   // an unusually large footprint: not culled
    const int lane = lane_id();
    uint32_t word = 0u;
    if (lane < nw * nrows) {
        const int row = lane / nw, wi = w0 + (lane - row * nw);
        word = cam.cull_mask[(j0 + row) * wpr + wi];
        const int lo = max(i0 - wi * 32, 0), hi = min(i1 - wi * 32, 31);  // bits lo .. hi of this word belong to the footprint
        word &= (0xFFFFFFFFu >> (31 - hi)) & (0xFFFFFFFFu << lo);
    }
    return __ballot(word != 0u) == 0ull;
}

// the verdict, cached per pixel block of the pass when the host had it computed (DPass::block_culled), else computed here
__device__ __forceinline__ bool chunk_culled(const DCamera& cam, const DPass& ps, uint32_t chunk, uint32_t n)
{
    if (ps.block_culled != nullptr) return ((const_u1_ptr)(uintptr_t)ps.block_culled)[chunk % ps.cull_blocks] != 0u;
    return chunk_is_culled(cam, ps, chunk, n);
}

// record index of ray i of a chunk: radiance rays from the front, shadow rays from the back
__device__ __forceinline__ size_t record_index(const DPass& ps, uint32_t chunk, uint32_t i, uint32_t n_rad)
{
    const size_t base = (size_t)chunk * ps.region;
    return i < n_rad ? base + i : base + (ps.region - 1u - (i - n_rad));
}

// ---- ray records: two float4 planes q0[], q1[] per queue (32 B per ray) --------------------------------------------
//   radiance ray: q0 = { o.xyz, d.x },    q1 = { d.y, d.z, light-term slot of its sample, level << 4 | tree node << 8 }
//   shadow ray:   q0 = { hp.xyz, term },  q1 unused   hp: the shaded hit point, term: float index of its light term in slot_L.
//                 The kernel that shades writes the finished term into slot_L at once — OPTIMISTICALLY: most shadow rays reach
//                 the light — and whoever finds the ray BLOCKED (mod.rs:226-232) puts the zero back (store_blocked): the
//                 unblocked majority costs no load and no store after the trace.  The ray itself is rebuilt from hp and the
//                 light by the kernel that traces it — the expressions of mod.rs:215, 224-225 in the order shade()
//                 evaluates them, so the floats are the ones the reference traces.
// (pixel, sample number) of a sample — the reflection sampler's hash inputs — live in ps.slot_ps, per light-term slot.
__device__ __forceinline__ void shadow_ray_of(const DScene& sc, const DPass& ps, const float4 q0, f3& o, f3& d)
{
    uint32_t li = 0u;
    if (sc.nlights > 1u) li = (__float_as_uint(q0.w) / 3u / ps.nslots) % sc.nlights;      // term = 3 * ((node * nlights + li) * nslots + slot)
    const DLight lt = sc.lights[li];
    const f3 hp = mk3(q0.x, q0.y, q0.z);
    const f3 l = sub3(mk3(lt.px, lt.py, lt.pz), hp);                           // mod.rs:215
    o = add3(hp, vscale(l, 0.01f)); d = l;                                      // mod.rs:224-225
}

// a shadow ray that IS blocked (its intersector hit lies in (0.01, 1), mod.rs:226-232): the light term the shade kernel wrote
// optimistically does not count — back to black (mod.rs:232 `continue`)
__device__ __forceinline__ void store_blocked(float* __restrict__ slot_L, uint32_t term)
{
    // three dword stores of ONE zero register (the compiler's own choice is a dwordx3 store of three zero registers: in the
    // trace kernels, which sit at the 64-VGPR limit of 8 waves per SIMD, those two extra registers are a scratch spill)
    float* dst = slot_L + term;
    const float zero = 0.0f;
    asm volatile("global_store_dword %0, %1, off\n\tglobal_store_dword %0, %1, off offset:4\n\tglobal_store_dword %0, %1, off offset:8" : : "v"(dst), "v"(zero) : "memory");
}

// ---- trace: closest hit of every ray of a round --------------------------------------------------
// The trace loop of one wave.  SINGLE == false: persistent wave of trace_kernel, pulls chunks from `cursor`.
// SINGLE == true: the wave traces exactly the rays of chunk `single_chunk` (fused_pass_kernel).
// CONFIRM == true (reference-default semantics): shadow rays are traced like radiance rays — closest hit on [0, 1) —
// and every ray only records its hit; the octree confirm step (confirm_chunk) turns true closest hits into the reference
// intersector's answers and settles the shadow predicate.  CONFIRM == false (MI355RT_FLAG_TRUE_CLOSEST_HIT): the shadow
// predicate is decided here with the interval trick of traverse.hpp and unblocked light terms are stored directly.
template <bool PRIMARY, bool COUNT, bool SINGLE, bool CONFIRM>
__device__ __forceinline__ void trace_wave(const DScene& sc, const DCamera& cam, const DPass& ps,
                                           const float4* __restrict__ in_q, const uint2* __restrict__ in_counts,
                                           float4* __restrict__ hits, uint32_t* cursor,
                                           float* __restrict__ slot_L, const uint32_t* __restrict__ film_n,
                                           DCounters* counters, int* stack, uint32_t single_chunk, uint32_t single_nrad, uint32_t single_nshadow)
{
    uint32_t acc_nodes = 0, acc_tris = 0, acc_ie = 0, acc_le = 0;
    uint32_t acc_below[6] = { 0, 0, 0, 0, 0, 0 };        // COUNT only: inner-node visits by node index
    uint32_t acc_li = 0, acc_ll = 0, acc_ld = 0, acc_it = 0, acc_rf = 0, acc_rp = 0, acc_rr = 0;       // COUNT only: lane census per iteration, refill statistics
    unsigned long long t_begin = 0, c_begin = 0;
    if (COUNT) { t_begin = __builtin_amdgcn_s_memrealtime(); c_begin = __builtin_amdgcn_s_memtime(); }

    // wave-uniform work state: the chunk being handed out
    uint32_t w_chunk = 0u, w_next = 0u, w_nrad = 0u, w_ntot = 0u;
    bool exhausted = false;
    PullState w_pull;
    // lane state
    RayState rs;
    uint32_t rec = 0;                                // radiance: hit-record index; shadow: float index into slot_L (both < 2^32)
    ray_init(rs, mk3(0, 0, 0), mk3(0, 0, 1), false, sc.root);
    rs.node = kNodeIdle;
    stack[0] = kNodeFin;                             // sentinel row of this lane's stack column (traverse.hpp, pop_or_finish)

    for (;;) {
        // ---- refill idle lanes from the current chunk (pull a new chunk when it runs dry)
        // A refill makes the wave wait for ray records that come from HBM, so it is done only when
        // at least ps.refill_threshold lanes are idle (or nothing is left to do): the other waves of
        // the SIMD then have enough work to cover the wait.
        unsigned long long idle = __ballot(rs.node <= kNodeFin);        // no ray, or a finished one (the two most negative codes)
        if ((uint32_t)__popcll(idle) < ps.refill_threshold && idle != ~0ull) idle = 0ull;
        if (COUNT && idle != 0ull) ++acc_rf;
        if (idle != 0ull) {
            if (rs.node == kNodeFin) {     // ---- results of the rays that ended since the last refill (they waited in their lanes: one store section per refill, well filled)
                rs.node = kNodeIdle;
                if (CONFIRM && sc.oct_single_leaf) {
                    // The reference's octree is ONE leaf (at most triangles_per_leaf triangles in the scene, e.g. 4boxes): the leaf
                    // lists every triangle, so its closest hit IS the true closest hit and the octree's whole answer is the
                    // contains test of OCT:160-169 on the root cube.  Settled right here: no confirm launch for such scenes.
                    bool hit = rs.prim != kMiss;
                    if (hit) hit = cube_contains(mk3(sc.oct_root[0], sc.oct_root[1], sc.oct_root[2]), mk3(sc.oct_root[3], sc.oct_root[4], sc.oct_root[5]),
                                                 add3(rs.o, vscale(rs.d, rs.t)));
                    if (rs.occ < 0) {                                           // radiance ray
                        st1<1>((uint32_t*)((char*)ps.hit_prim + (rec << 2)), hit ? rs.prim : kMiss);
                        if (hit) st4<1>((float4*)((char*)hits + ((size_t)rec << 4)), make_float4(rs.t, rs.u, rs.v, __uint_as_float(rs.prim)));
                    } else if (hit && rs.t > 0.01f && rs.t < 1.0f) {              // shadow ray (rec: its term), blocked (mod.rs:226-232)
                        store_blocked(slot_L, rec);
                    }
                } else
                if (CONFIRM || rs.occ < 0) {                                    // radiance ray (CONFIRM: every ray)
                    st1<1>((uint32_t*)((char*)ps.hit_prim + (rec << 2)), rs.prim);     // 4 B for every ray, the 16 B record only for hits
                    if (rs.prim != kMiss) st4<1>((float4*)((char*)hits + ((size_t)rec << 4)), make_float4(rs.t, rs.u, rs.v, __uint_as_float(rs.prim)));
                } else if (rs.occ == 1) {                              // blocked, mod.rs:232
                    store_blocked(slot_L, rec);
                }
            }
        }
        while (idle != 0ull && !exhausted) {
            if (w_next >= w_ntot) {
                uint32_t c = 0u;
                c = w_chunk;
                if (SINGLE) {
                    if (!w_pull.first) { exhausted = true; break; }
                    w_pull.first = false; c = single_chunk;
                }
                else if (!pull_chunk(cursor, ps.nchunks, ps.pull_mode, ps.ncursors, ps.pull_group, w_pull, c, ps.tail_chunks, ps.tail_split_shift, PRIMARY ? LiveLists() : live_lists(ps))) { exhausted = true; break; }
                // bcast_first: these are wave-uniform by construction; saying so keeps them in SGPRs
                w_chunk = bcast_first(c); w_next = 0u;
                if (PRIMARY) {
                    w_nrad = min(ps.chunk, ps.nsamples - w_chunk * ps.chunk);
                    if (chunk_culled(cam, ps, w_chunk, w_nrad)) w_nrad = 0u;    // the shade kernel makes the same decision
                    w_nrad = bcast_first(w_nrad);
                    w_ntot = w_nrad;
                }
                else if (SINGLE) { w_nrad = single_nrad; w_ntot = single_nrad + single_nshadow; }      // from the shade phase, in registers
                else { const uint2 n = in_counts[w_chunk]; w_nrad = bcast_first(n.x); w_ntot = w_nrad + bcast_first(n.y); }
                if (!SINGLE && w_pull.part != kWholeChunk) {       // a part of a tail chunk: rays [part * per, (part + 1) * per) of it, per a multiple of 64
                    const uint32_t per = ((w_ntot + (64u << ps.tail_split_shift) - 1u) >> (6u + ps.tail_split_shift)) << 6;
                    w_next = bcast_first(min(w_pull.part * per, w_ntot));
                    w_ntot = bcast_first(min(w_next + per, w_ntot));
                }
                continue;
            }
            const uint32_t avail = w_ntot - w_next;
            if (COUNT) { ++acc_rp; acc_rr += min((uint32_t)__popcll(idle), avail); }
            const uint32_t rank = (uint32_t)__popcll(idle & lanemask_lt());
            if (rs.node == kNodeIdle && rank < avail) {
                const uint32_t i = w_next + rank;
                f3 o, d;
                bool shadow = false;
                if (PRIMARY) {
                    uint32_t pixel, sampleno;
                    primary_sample(cam, ps, film_n, w_chunk * ps.chunk + i, pixel, sampleno, o, d);
                    rec = w_chunk * ps.region + i;
                } else {
                    // record index in 32 bits, byte offsets in 64 (a whole 1080p x 64 spp frame in ONE pass is 531 M records of 16 B per plane);
                    // the planes of the queue are wave-uniform base pointers
                    shadow = i >= w_nrad;
                    const uint32_t r = w_chunk * ps.region + (shadow ? ps.region - 1u - (i - w_nrad) : i);
                    const char* __restrict__ p0 = (const char*)in_q;
                    const char* __restrict__ p1 = (const char*)(in_q + ps.qstride);
                    const float4 r0 = ld4<0>((const float4*)(p0 + ((size_t)r << 4)));
                    float4 r1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (!shadow) r1 = ld4<0>((const float4*)(p1 + ((size_t)r << 4)));                      // a shadow record has no second plane
                    rec = r;
                    if (shadow) {
                        shadow_ray_of(sc, ps, r0, o, d);
                        // a shadow ray whose verdict is reached right here (true-closest semantics; one-leaf octrees) needs its term, not its record
                        if (!CONFIRM || sc.oct_single_leaf) rec = __float_as_uint(r0.w);           // float index of its light term in slot_L (shade kernel)
                    } else { o = mk3(r0.x, r0.y, r0.z); d = mk3(r0.w, r1.x, r1.y); }
                }
                ray_init(rs, o, d, shadow, sc.root);
#ifdef MI355RT_EXP_NOSHADOW      // timing experiment (wrong results): shadow rays are not traced at all — what do they cost?
                if (shadow) rs.node = kNodeFin;
#endif
#ifdef MI355RT_EXP_PATHLOADS     // timing experiment (same results): what would reading a 13-slot path record per secondary ray ADD?  Rays that start within 0.01 of each other read the same record
                if (!PRIMARY) {
                    const uint32_t hx = (uint32_t)(int)floorf(o.x * 100.0f) * 73856093u ^ (uint32_t)(int)floorf(o.y * 100.0f) * 19349663u ^ (uint32_t)(int)floorf(o.z * 100.0f) * 83492791u;
                    const uint4* __restrict__ t4 = (const uint4*)sc.tris;
                    const uint32_t b0 = hx % (sc.ntri * 3u - 16u);
                    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                    for (uint32_t k = 0; k < 13u; ++k) { const uint4 q = t4[b0 + k]; acc.x ^= q.x; acc.y ^= q.y; acc.z ^= q.z; acc.w ^= q.w; }
                    asm volatile("" : : "v"(acc.x), "v"(acc.y), "v"(acc.z), "v"(acc.w));
                }
#endif
            }
            w_next += min((uint32_t)__popcll(idle), avail);
            idle = __ballot(rs.node == kNodeIdle);
        }
        if (__ballot(rs.node != kNodeIdle) == 0ull) break;
        if (COUNT) { acc_li += (uint32_t)__popcll(__ballot(lane_at_inner(rs))); acc_ll += (uint32_t)__popcll(__ballot(lane_at_leaf(rs))); acc_ld += (uint32_t)__popcll(__ballot(rs.node <= kNodeFin)); ++acc_it; }

        // ---- while-while scheduling: inner-node steps run for the lanes at inner nodes; lanes that
        // reached a leaf wait until enough of them are there (or nobody is left at an inner node),
        // so that the expensive triangle code always runs with a well-filled wave.  The sections are
        // entered on wave-uniform conditions; inside, lanes are predicated, not branched, and every
        // section pops (or finishes) its own lanes, so a lane that leaves a node is busy again at once.
#pragma unroll
        for (int u = 0; u < kInnerStepsPerIteration; ++u) {
            if (__ballot(lane_at_inner(rs)) == 0ull) break;
            inner_pred<COUNT>(sc, rs, stack, kBlock, acc_nodes, COUNT ? acc_below : nullptr);
            if (COUNT) {
                ++acc_ie;
                // instrumented builds check what the shipped loop takes from the host (renderer.cpp: BVH depth <= kBvhMaxDepth, stack rows = depth + 2):
                // a push writes row sp + 1 of stack_depth + 1 rows.  The call then fails with "traversal stack overflow" instead of walking on garbage.
                if (__ballot(rs.sp >= (int)ps.stack_depth) != 0ull && lane_id() == 0) atomicOr(&counters->overflow, 2u);
            }
        }
#ifdef MI355RT_EXP_REDESCEND     // timing experiment (same results): a secondary ray that reaches its FIRST leaf starts over at the root — what does the first descent cost, in place?
        if (!PRIMARY) { const bool at_leaf = lane_at_leaf(rs); const bool redo = at_leaf & (rs.tri == 0u); rs.tri = at_leaf ? 1u : rs.tri; if (redo) { rs.node = sc.root; rs.sp = 0; } }
#endif
        const unsigned long long m_leaf = __ballot(lane_at_leaf(rs));
        if (m_leaf != 0ull && ((uint32_t)__popcll(m_leaf) >= ps.leaf_threshold || __ballot(lane_at_inner(rs)) == 0ull))
            { leaf_pred<COUNT, CONFIRM || PRIMARY>(sc, rs, stack, kBlock, acc_tris); if (COUNT) ++acc_le; }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) { acc_nodes += __shfl_down((int)acc_nodes, off, 64); acc_tris += __shfl_down((int)acc_tris, off, 64); for (int k = 0; k < 6; ++k) acc_below[k] += __shfl_down((int)acc_below[k], off, 64); }
        DCounters* cs = &counters[global_wave_id() % kShards];
        if (lane_id() == 0) {
            atomicAdd(&cs->nodes_visited, (unsigned long long)acc_nodes); atomicAdd(&cs->tris_tested, (unsigned long long)acc_tris);
            for (int k = 0; k < 6; ++k) atomicAdd(&cs->visits_below[k], (unsigned long long)acc_below[k]);
            atomicAdd(&cs->inner_execs, (unsigned long long)acc_ie); atomicAdd(&cs->leaf_execs, (unsigned long long)acc_le);
            atomicAdd(&cs->lanes_inner, (unsigned long long)acc_li); atomicAdd(&cs->lanes_leaf, (unsigned long long)acc_ll); atomicAdd(&cs->lanes_done, (unsigned long long)acc_ld);
            atomicAdd(&cs->lane_samples, (unsigned long long)acc_it); atomicAdd(&cs->refills, (unsigned long long)acc_rf); atomicAdd(&cs->refill_passes, (unsigned long long)acc_rp);
            atomicAdd(&cs->refill_rays, (unsigned long long)acc_rr);
            // load-balance diagnostics: when did this wave run out of work, relative to the first wave's start
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime(), c_end = __builtin_amdgcn_s_memtime();
            DCounters* c0 = &counters[0];
            atomicAdd(&cs->t_sum_cycles, c_end - c_begin); atomicAdd(&cs->t_sum_real, t_end - t_begin);
            atomicMin(&c0->t_start, t_begin); atomicMin(&c0->t_first_end, t_end); atomicMax(&c0->t_last_end, t_end);
            atomicAdd(&c0->t_sum_end, t_end - t_begin); atomicAdd(&c0->n_waves, 1ull);
        }
    }
}

template <bool PRIMARY, bool COUNT, bool CONFIRM>
__global__ __launch_bounds__(kBlock, MI355RT_WIDE ? MI355RT_WIDE_BLOCKS : PRIMARY ? MI355RT_PRIMARY_BLOCKS : 8) void trace_kernel(DScene sc, DCamera cam, DPass ps,
                                                      const float4* __restrict__ in_q, const uint2* __restrict__ in_counts,
                                                      float4* __restrict__ hits, uint32_t* cursor,
                                                      float* __restrict__ slot_L, const uint32_t* __restrict__ film_n,
                                                      DCounters* counters)
{
    extern __shared__ int s_stack[];                 // ps.stack_depth rows of kBlock ints
    trace_wave<PRIMARY, COUNT, false, CONFIRM>(sc, cam, ps, in_q, in_counts, hits, cursor, slot_L, film_n, counters, &s_stack[threadIdx.x], 0u, 0u, 0u);
}

// ---- primary rays through the screen-space triangle bins (DCamera::tile_ofs; renderer.cpp, refresh_tile_bins) ----------------------------
// The closest hit of the primary rays, found without the tree: the 64 samples a wave takes together are one tile of the image, the tile's list
// holds every triangle a ray of the tile can hit (nearest first), and every lane tests the SAME triangle at the same time — the triangle comes
// through the scalar cache into SGPRs, the vector memory pipe that bounds the tree walk is not used at all, and what is left is the
// reference's own Moller-Trumbore arithmetic (intersect.rs:62-98: same operations, same order, same tie rule as leaf_pred) at one triangle per ~70
// vector instructions for 64 rays.  A list ends early once every lane holds a hit nearer than anything the rest of the list can offer.
// Output: what trace_kernel<PRIMARY> writes — hit flags and hit records of the chunk's region; culled chunks are skipped alike.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef const u32x4_t __attribute__((address_space(4))) * const_u4_ptr;    // constant address space: a wave-uniform address loads through the scalar cache
typedef const u32x2_t __attribute__((address_space(4))) * const_u2_ptr;
// closest hit of the primary rays of ONE tile: the 64 samples from sample i0 of `chunk` on (lane l: sample i0 + l; valid: it exists).  Every lane
// regenerates its ray; (bt, bu, bv, bprim) is its closest hit over the tile's list (kMiss: none).  Returns the number of list entries tested.
__device__ __forceinline__ uint32_t raster_tile(const DScene& sc, const DCamera& cam, const DPass& ps, const uint32_t* __restrict__ film_n, uint32_t chunk, uint32_t i0, bool valid,
                                                f3& o, f3& d, float& bt, float& bu, float& bv, uint32_t& bprim)
{
    const const_u2_ptr tile_ofs = (const_u2_ptr)(uintptr_t)cam.tile_ofs, entries = (const_u2_ptr)(uintptr_t)cam.tile_entries;
    const const_u4_ptr tris = (const_u4_ptr)(uintptr_t)sc.tris;
    const uint32_t gi = chunk * ps.chunk + i0 + (valid ? (uint32_t)lane_id() : 0u);
    uint32_t pixel, sampleno;
    primary_sample(cam, ps, film_n, gi, pixel, sampleno, o, d);
    // the tile of these 64 samples (wave-uniform: the host built the bins for exactly this layout)
    uint32_t s0, p0, first_row, nr, x, y;
    sample_of(ps, chunk * ps.chunk + i0, s0, p0);
    pass_column(ps, cam.width, p0, first_row, nr, x, y);
    const uint32_t tile = bcast_first((pass_row(ps, first_row) / cam.tile_rg) * cam.tile_nblocks + x / cam.tile_cols);
    const u32x2_t oc = tile_ofs[tile];
    const float dlen = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z) * 1.00001f;      // t * |d| = distance from the camera; a little long: the early out stays on the safe side
    bt = __builtin_inff(); bu = 0.0f; bv = 0.0f; bprim = kMiss;
    uint32_t tested = 0u;
    for (uint32_t e = 0; e < oc.y; ++e) {
        const u32x2_t en = entries[oc.x + e];
        // nothing from here on is nearer than en.y (the list is sorted): done when every lane's hit is nearer still
        if (__ballot(valid && !(bt * dlen < __uint_as_float(en.y))) == 0ull) break;
        ++tested;
        const uint32_t ti = en.x * 3u;
        const u32x4_t q0 = tris[ti], q1 = tris[ti + 1u], q2 = tris[ti + 2u];
        const f3 v0 = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        const f3 v0v1 = mk3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)), v0v2 = mk3(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z));
        // Moller-Trumbore "late out", intersect.rs:62-98, same operation order (leaf_pred)
        const f3 pvec = cross3(d, v0v2);
        const float det = dot3(v0v1, pvec);
        const float inv_det = div_rn(1.0f, det);
        const f3 tvec = sub3(o, v0);
        const float u = dot3(tvec, pvec) * inv_det;
        const f3 qvec = cross3(tvec, v0v1);
        const float v = dot3(d, qvec) * inv_det;
        const float t = dot3(v0v2, qvec) * inv_det;
        const bool ok = !(fabsf(det) < 1.1920929e-7f) & !((u < 0.0f) | (u > 1.0f)) & !((v < 0.0f) | (u + v > 1.0f)) & !(t < 0.0f);
        const uint32_t prim = q0.w;
        const bool better = ok & ((bprim == kMiss) | (t < bt) | ((t == bt) & (prim < bprim)));
        bt = better ? t : bt; bu = better ? u : bu; bv = better ? v : bv; bprim = better ? prim : bprim;
    }
    // a one-leaf octree: its whole answer is the contains test on the root cube (trace_wave; the flag is only ever set with the reference-default semantics)
    if (sc.oct_single_leaf && bprim != kMiss &&
        !cube_contains(mk3(sc.oct_root[0], sc.oct_root[1], sc.oct_root[2]), mk3(sc.oct_root[3], sc.oct_root[4], sc.oct_root[5]), add3(o, vscale(d, bt)))) bprim = kMiss;
    return tested;
}

template <bool COUNT, bool CONFIRM>
__global__ __launch_bounds__(kBlock) void raster_kernel(DScene sc, DCamera cam, DPass ps, float4* __restrict__ hits, uint32_t* cursor,
                                                        const uint32_t* __restrict__ film_n, DCounters* counters)
{
    const int lane = lane_id();
    unsigned long long acc_tris = 0;
    PullState pull; uint32_t chunk = 0u;
    while (pull_chunk(cursor, ps.nchunks, ps.pull_mode, ps.ncursors, ps.pull_group, pull, chunk)) {
        const uint32_t n = min(ps.chunk, ps.nsamples - chunk * ps.chunk);
        if (chunk_culled(cam, ps, chunk, n)) continue;               // the shade kernel makes the same decision
        for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool valid = i < n;
            f3 o, d; float bt, bu, bv; uint32_t bprim;
            const uint32_t tested = raster_tile(sc, cam, ps, film_n, chunk, i0, valid, o, d, bt, bu, bv, bprim);
            if (COUNT) acc_tris += (unsigned long long)tested * (unsigned long long)__popcll(__ballot(valid));
            if (valid) {
                const uint32_t rec = chunk * ps.region + i;
                st1<1>((uint32_t*)((char*)ps.hit_prim + ((size_t)rec << 2)), bprim);
                if (bprim != kMiss) st4<1>((float4*)((char*)hits + ((size_t)rec << 4)), make_float4(bt, bu, bv, __uint_as_float(bprim)));
            }
        }
    }
    if (COUNT && lane == 0 && acc_tris) atomicAdd(&counters[global_wave_id() % kShards].tris_tested, acc_tris);
}

// ---- trace with the reference-exact octree intersector (parity path, MI355RT_FLAG_OCTREE_SEMANTICS) ----
template <bool PRIMARY>
__global__ __launch_bounds__(kBlock) void trace_octree_kernel(DScene sc, DCamera cam, DPass ps,
                                                             const float4* __restrict__ in_q, const uint2* __restrict__ in_counts,
                                                             float4* __restrict__ hits, float* __restrict__ slot_L,
                                                             const uint32_t* __restrict__ film_n)
{
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    for (uint32_t chunk = global_wave_id(); chunk < ps.nchunks; chunk += nwaves) {
        uint32_t n_rad, n_tot;
        if (PRIMARY) { n_rad = min(ps.chunk, ps.nsamples - chunk * ps.chunk); n_tot = n_rad; }
        else { const uint2 n = in_counts[chunk]; n_rad = n.x; n_tot = n.x + n.y; }
        for (uint32_t i = (uint32_t)lane_id(); i < n_tot; i += 64u) {
            f3 o, d;
            size_t r = 0;
            if (PRIMARY) {
                uint32_t pixel, sampleno;
                primary_sample(cam, ps, film_n, chunk * ps.chunk + i, pixel, sampleno, o, d);
                r = (size_t)chunk * ps.region + i;
            } else {
                r = record_index(ps, chunk, i, n_rad);
                const float4 r0 = in_q[r];
                if (i < n_rad) { const float4 r1 = in_q[ps.qstride + r]; o = mk3(r0.x, r0.y, r0.z); d = mk3(r0.w, r1.x, r1.y); }
                else shadow_ray_of(sc, ps, r0, o, d);
            }
            float t, u, v; uint32_t prim;
            octree_intersect(sc, o, d, t, u, v, prim);
            if (i < n_rad) {
                ps.hit_prim[r] = prim;
                if (prim != kMiss) hits[r] = make_float4(t, u, v, __uint_as_float(prim));
            } else if (prim != kMiss && t > 0.01f && t < 1.0f) {              // blocked, mod.rs:226-232
                store_blocked(slot_L, ((const uint32_t*)in_q)[4ull * r + 3u]);
            }
        }
    }
}

// The wave's LDS list of hit indices: contiguous (shade_kernel) or spread over the wave's 64 columns of the
// block's traversal-stack rows (fused_pass_kernel, where the same LDS serves both phases).
struct LinearList { uint32_t* p; __device__ __forceinline__ uint32_t& operator[](uint32_t i) const { return p[i]; } };
struct ColumnList { int* col0; __device__ __forceinline__ uint32_t& operator[](uint32_t i) const { return *(uint32_t*)&col0[(i >> 6) * kBlock + (i & 63u)]; } };

// ---- octree confirm: true closest hits -> the reference intersector's answers (traverse.hpp, confirm_walk) --------
// One wave per chunk, after the trace of a round.  Records that hit something are compacted into the wave's LDS list
// (ballot + prefix popcount), then every lane walks the octree for one of them.  Radiance records: the hit record is
// rewritten where the reference's octree returns something else (another triangle, or nothing).  Shadow records: the
// predicate of mod.rs:226-232 on the confirmed hit; the light term the record carries is stored when the ray is NOT blocked
// — also for the shadow rays that hit nothing at all.
// one record that hit something: radiance -> rewrite the hit record where the octree answers differently; shadow -> the predicate
template <bool PRIMARY>
__device__ __forceinline__ void confirm_record(const DScene& sc, const DCamera& cam, const DPass& ps, uint32_t r, uint32_t sample_index, bool shadow,
                                               const float4* __restrict__ in_q, float4* __restrict__ hits, float* __restrict__ slot_L, const uint32_t* __restrict__ film_n)
{
    f3 o, d;
    float term_bits = 0.0f;
    if (PRIMARY) {
        uint32_t pixel, sampleno;
        primary_sample(cam, ps, film_n, sample_index, pixel, sampleno, o, d);
    } else {
        const float4 r0 = ld4<3>(&in_q[r]);
        term_bits = r0.w;
        if (shadow) shadow_ray_of(sc, ps, r0, o, d);
        else { const float4 r1 = ld4<3>(&in_q[ps.qstride + r]); o = mk3(r0.x, r0.y, r0.z); d = mk3(r0.w, r1.x, r1.y); }
    }
    const float4 h = ld4<3>(&hits[r]);
    float t = h.x, u = h.y, v = h.z; uint32_t prim = __float_as_uint(h.w);
    const uint32_t prim_in = prim;
#ifndef MI355RT_EXP_NOWALK          // timing experiment (wrong results): the confirm kernel's gathers and stores without the walk
    confirm_walk(sc, o, d, t, u, v, prim);
#endif
    if (!shadow) {
        if (prim != prim_in) {
            ps.hit_prim[r] = prim;
            if (prim != kMiss) hits[r] = make_float4(t, u, v, __uint_as_float(prim));
        }
    } else if (prim != kMiss && t > 0.01f && t < 1.0f) {                   // blocked, mod.rs:226-232
        store_blocked(slot_L, __float_as_uint(term_bits));
    }
}

// confirm step of ONE chunk by one wave (fused_pass_kernel): compact the records that hit something, then walk
template <bool PRIMARY, class List>
__device__ __forceinline__ void confirm_chunk(const DScene& sc, const DCamera& cam, const DPass& ps, uint32_t chunk, const List list,
                                              const float4* __restrict__ in_q, uint32_t n_rad, uint32_t n_sh,
                                              float4* __restrict__ hits, float* __restrict__ slot_L, const uint32_t* __restrict__ film_n)
{
    const int lane = lane_id();
    if (PRIMARY) {
        n_rad = min(ps.chunk, ps.nsamples - chunk * ps.chunk); n_sh = 0u;
        if (chunk_culled(cam, ps, chunk, n_rad)) return;                 // nothing was traced, nothing was hit
    }
    const uint32_t n_tot = n_rad + n_sh;
    uint32_t cnt = 0u;
    for (uint32_t it = 0; it < n_tot; it += 64u) {
        const uint32_t i = it + (uint32_t)lane;
        const bool valid = i < n_tot;
        const uint32_t r = valid ? (uint32_t)record_index(ps, chunk, i, n_rad) : 0u;
        const bool hit = valid && ld1<3>(&ps.hit_prim[r]) != kMiss;
        uint32_t n_new;
        const uint32_t pos = wave_append(hit, cnt, n_new);
        if (hit) list[pos] = i;
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = 0; j < cnt; j += 64u) {
        if (j + (uint32_t)lane >= cnt) continue;
        const uint32_t i = list[j + (uint32_t)lane];
        confirm_record<PRIMARY>(sc, cam, ps, (uint32_t)record_index(ps, chunk, i, n_rad), chunk * ps.chunk + i, i >= n_rad, in_q, hits, slot_L, film_n);
    }
    __builtin_amdgcn_wave_barrier();
}

// The confirm launch of a round.  A chunk holds ~60-90 records that hit something: walked chunk by chunk, the second
// batch of 64 lanes of every chunk would be mostly empty.  So a wave collects the hit records of its chunks in ONE LDS
// list and walks 64 of them whenever it has 64 — across chunk borders (the writes are per record, in place).
template <bool PRIMARY>
__global__ __launch_bounds__(kBlock, MI355RT_CONFIRM_BLOCKS) void confirm_kernel(DScene sc, DCamera cam, DPass ps, const float4* __restrict__ in_q, const uint2* __restrict__ in_counts,
                                                         float4* __restrict__ hits, uint32_t* cursor, float* __restrict__ slot_L, const uint32_t* __restrict__ film_n, uint32_t shadow_only)
{
    __shared__ uint32_t s_list[kWavesPerBlock][2][128];   // per wave: up to 127 pending entries of { record index | shadow << 31, sample index }
    const int lane = lane_id();
    uint32_t* list_r = s_list[threadIdx.x >> 6][0];
    uint32_t* list_s = s_list[threadIdx.x >> 6][1];
    uint32_t cnt = 0u;                                // < 64 between the steps below
    // chunks are pulled like the trace kernel pulls them (the hits sit in a part of the image: static striding leaves waves idle)
    PullState pull; uint32_t chunk = 0u;
    while (pull_chunk(cursor, ps.nchunks, kConfirmPullMode ? kConfirmPullMode : ps.pull_mode, ps.ncursors, ps.pull_group, pull, chunk, 0u, 0u, PRIMARY ? LiveLists() : live_lists(ps))) {
        uint32_t n_rad = 0u, n_sh = 0u;
        if (PRIMARY) {
            n_rad = min(ps.chunk, ps.nsamples - chunk * ps.chunk);
            if (chunk_culled(cam, ps, chunk, n_rad)) continue;
        } else { const uint2 n = in_counts[chunk]; n_rad = n.x; n_sh = n.y; }
        const uint32_t n_tot = n_rad + n_sh;
        for (uint32_t it = shadow_only ? n_rad : 0u; it < n_tot; it += 64u) {     // shadow_only: the shade kernel of the round confirms the radiance hits itself
            const uint32_t i = it + (uint32_t)lane;
            const bool valid = i < n_tot;
            const uint32_t r = valid ? (uint32_t)record_index(ps, chunk, i, n_rad) : 0u;
            const bool hit = valid && ld1<3>(&ps.hit_prim[r]) != kMiss;
                uint32_t n_new;
            const uint32_t pos = wave_append(hit, cnt, n_new);
            if (hit) { list_r[pos] = r | (i >= n_rad ? 0x80000000u : 0u); if (PRIMARY) list_s[pos] = chunk * ps.chunk + i; }
            __builtin_amdgcn_wave_barrier();
            if (cnt >= 64u) {                         // a full batch, taken from the end of the list
                cnt -= 64u;
                const uint32_t code = list_r[cnt + (uint32_t)lane];
                const uint32_t si = PRIMARY ? list_s[cnt + (uint32_t)lane] : 0u;
                __builtin_amdgcn_wave_barrier();
                confirm_record<PRIMARY>(sc, cam, ps, code & 0x7FFFFFFFu, si, (code >> 31) != 0u, in_q, hits, slot_L, film_n);
            }
        }
    }
    if ((uint32_t)lane < cnt) {                       // the last, partial batch
        const uint32_t code = list_r[lane];
        confirm_record<PRIMARY>(sc, cam, ps, code & 0x7FFFFFFFu, PRIMARY ? list_s[lane] : 0u, (code >> 31) != 0u, in_q, hits, slot_L, film_n);
    }
}

// The walk of mod.rs:187-189 / sample_generator.rs:27 — take the first entry from jx on (wrapping at 65535) that lies in the
// hemisphere of n.  tv = table[jx] is already loaded; the rest is fetched FOUR entries at a time: as a loop of single dependent
// loads it runs as long as the unluckiest lane of the wave (~7 cache latencies); four consecutive entries are one or two lines.
__device__ __forceinline__ void hemisphere_walk(const float4* __restrict__ table, const f3 n, uint32_t& jx, float4& tv)
{
    for (uint32_t guard = 0u; tv.x * n.x + tv.y * n.y + tv.z * n.z <= 0.0f && guard < kNumSamples; guard += 4u) {
        const uint32_t j1 = jx + 1u == kSampleMax ? 0u : jx + 1u;
        const uint32_t j2 = j1 + 1u == kSampleMax ? 0u : j1 + 1u;
        const uint32_t j3 = j2 + 1u == kSampleMax ? 0u : j2 + 1u;
        const uint32_t j4 = j3 + 1u == kSampleMax ? 0u : j3 + 1u;
        const float4 t1 = table[j1], t2 = table[j2], t3 = table[j3], t4 = table[j4];
        const bool r1 = t1.x * n.x + t1.y * n.y + t1.z * n.z <= 0.0f, r2 = t2.x * n.x + t2.y * n.y + t2.z * n.z <= 0.0f;
        const bool r3 = t3.x * n.x + t3.y * n.y + t3.z * n.z <= 0.0f;
        // the reference stops at the first accepted entry, or after kNumSamples steps with whatever it holds
        const uint32_t left = kNumSamples - guard;           // steps still allowed (>= 1)
        if (!r1 || left == 1u) { tv = t1; jx = j1; break; }
        if (!r2 || left == 2u) { tv = t2; jx = j2; break; }
        if (!r3 || left == 3u) { tv = t3; jx = j3; break; }
        tv = t4; jx = j4;
    }
}

#ifndef MI355RT_SORT_BINS
#define MI355RT_SORT_BINS 8                    // reflection rays of a shading batch grouped by direction: 0 off, 8 octants, 24 octant x dominant axis (A/B knob, profiles/r03_notes.md)
#endif
constexpr uint32_t kSortBins = MI355RT_SORT_BINS;
__device__ __forceinline__ uint32_t dir_bin(const f3 d)
{
    const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    if (kSortBins <= 8u) return oct;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const uint32_t major = ax >= ay ? (ax >= az ? 0u : 2u) : (ay >= az ? 1u : 2u);
    return oct * 3u + major;
}

// Is the way from the shaded point to light li provably free?  l = light - hp (mod.rs:215).  The shadow ray of mod.rs:224-225 starts at hp + 0.01 l and
// is searched on t in [0, 1): from 0.99 |l| in front of the light, through it, to 0.01 |l| behind it — every point of it in the direction -l (or +l)
// from the light.  The light's depth cube map (lightmap.hpp) bounds from below the squared distance to anything seen in the texel of -l, the
// light's tail2 the distance to anything at all: with both bounds beyond the ray's reach no triangle can be hit, so the ray is not blocked whatever
// intersector answers (mod.rs:226-232) and it need not be traced.  Conservative like the BVH's boxes: the maps are padded (renderer.cpp).
__device__ __forceinline__ bool light_proves_unoccluded(const DScene& sc, uint32_t li, const DLight& lt, const f3 l)
{
    const uint32_t R = sc.light_map_res;
    if (R == 0u) return false;
    const float l2 = l.x * l.x + l.y * l.y + l.z * l.z;
    const float vx = -l.x, vy = -l.y, vz = -l.z;                      // from the light to the shaded point
    const float ax = fabsf(vx), ay = fabsf(vy), az = fabsf(vz);
    uint32_t m; float vm, va, vb;                                       // major axis, the two others in ascending order (lightmap.cpp)
    if (ax >= ay && ax >= az) { m = 0u; vm = vx; va = vy; vb = vz; }
    else if (ay >= az) { m = 1u; vm = vy; va = vx; vb = vz; }
    else { m = 2u; vm = vz; va = vx; vb = vy; }
    const float am = fabsf(vm);
    if (!(am > 0.0f) || !(l2 < 3.0e38f)) return false;
    const float fR = (float)R, u = div_rn(va, am), v = div_rn(vb, am);
    const int i = min(max((int)((u * 0.5f + 0.5f) * fR), 0), (int)R - 1), j = min(max((int)((v * 0.5f + 0.5f) * fR), 0), (int)R - 1);
    const uint32_t face = 2u * m + (vm < 0.0f ? 1u : 0u);
    const float bound = sc.light_maps[((size_t)(li * 6u + face) * R + (uint32_t)i) * R + (uint32_t)j];
    // 0.98011 > 0.99^2 and 1.0002e-4 > 0.01^2: the rounding of l2 and of the ray's origin (1e-7) stays inside
    return (l2 * 0.98011f < bound) & (l2 * 1.0002e-4f < lt.tail2);
}

// ---- shade: hit records -> light terms, shadow rays and reflection rays -----------------------------
// Shade the hits of one chunk (one wave): see the header of this file.
// in_nrad: radiance rays of the chunk in in_q (ignored for PRIMARY); out_nrad / out_nshadow: what was appended to out_q.
// WALK (primary round of the wavefront kernels, reference-default semantics): the octree confirm step of the hits is done
// here, on the ray this kernel regenerates anyway, instead of in a confirm launch of its own; a hit the octree drops keeps
// its (zeroed) light-term slot and is shaded no further: it resolves to black like a miss (mod.rs:99-100).
// RASTER (primary round of the wavefront kernels, tile bins built): the closest hits of the chunk's primary rays are found right here (raster_tile) and
// handed to the shading loop through LDS (lds_hits: one float4 per sample of a chunk) — no primary trace launch, no hit flags and no hit records in HBM.
template <bool PRIMARY, class List, bool WALK = false, bool RASTER = false>
__device__ __forceinline__ void shade_chunk(const DScene& sc, const DCamera& cam, const DPass& ps, uint32_t level, uint32_t chunk, const List list,
                                            const float4* __restrict__ in_q, uint32_t in_nrad, uint32_t& out_nrad, uint32_t& out_nshadow,
                                            const float4* __restrict__ hits,
                                            float4* __restrict__ out_q, uint2* __restrict__ out_counts,
                                            float* __restrict__ slot_L, uint32_t* __restrict__ sample_slot,
                                            const uint32_t* __restrict__ film_n, DCounters* counters,
                                            unsigned long long& acc_bounce, unsigned long long& acc_shadow, unsigned long long& acc_hits,
                                            float4* lds_hits = nullptr, unsigned long long* acc_tris = nullptr)
{
    const int lane = lane_id();
    {
        uint32_t n_rad = PRIMARY ? min(ps.chunk, ps.nsamples - chunk * ps.chunk) : in_nrad;
        const size_t base = (size_t)chunk * ps.region;
        if (PRIMARY && chunk_culled(cam, ps, chunk, n_rad)) {
            // no hit records were written for this chunk: every sample is a miss (with cached verdicts the resolve launch knows that too and reads no slot)
            if (ps.block_culled == nullptr) for (uint32_t i = (uint32_t)lane; i < n_rad; i += 64u) sample_slot[chunk * ps.chunk + i] = kMiss;
            acc_hits += (unsigned long long)n_rad << 32;         // high half: primary samples skipped by the frustum culling
            n_rad = 0u;
        }
        // ---- compact the rays that hit something (wave64 ballot + prefix popcount into LDS)
        uint32_t cnt = 0u;
        for (uint32_t it = 0; it < n_rad; it += 64u) {
            const uint32_t i = it + (uint32_t)lane;
            bool valid;
            float4 found = make_float4(0, 0, 0, 0);
            if (RASTER) {
                f3 ro, rd; float bt, bu, bv; uint32_t bprim;
                const uint32_t tested = raster_tile(sc, cam, ps, film_n, chunk, it, i < n_rad, ro, rd, bt, bu, bv, bprim);
                if (acc_tris) *acc_tris += (unsigned long long)tested * (unsigned long long)__popcll(__ballot(i < n_rad));
                valid = i < n_rad && bprim != kMiss;
                found = make_float4(bt, bu, bv, __uint_as_float(bprim));
            } else valid = i < n_rad && ld1<2>(&ps.hit_prim[base + i]) != kMiss;
            uint32_t n_new;
            const uint32_t pos = wave_append(valid, cnt, n_new);
            if (valid) list[pos] = i;
            if (RASTER && valid) lds_hits[pos] = found;
            // primary round: light-term slots are handed out per chunk to the samples that hit something
            // (74 % of the primary samples miss and need neither a slot nor zero-filling)
            if (PRIMARY && i < n_rad) st1<2>(&sample_slot[chunk * ps.chunk + i], valid ? chunk * ps.chunk + pos : kMiss);
        }
        if (PRIMARY) {
            // zero the slots this chunk uses: a node whose shadow ray is blocked, or that is never
            // reached, must read back as black (mod.rs:99-100, 170)
            // (slot_L is node-major: plane q = node * nlights + light holds the term of every slot, 12 B each, so that the
            // stores of a round — one node level — fill whole cache lines instead of 12 B of every 60)
            // The planes of node 0 (one per light) are not zeroed here: this wave writes every one of their entries itself below — the term, or the zero.
            const uint32_t planes = ps.nodes_per_sample * sc.nlights, total = cnt * 3u;
#ifdef MI355RT_EXP_SHADE_NOZERO
            for (uint32_t q = planes; q < planes; ++q) {
#else
            for (uint32_t q = sc.nlights; q < planes; ++q) {
#endif
                float* z = slot_L + 3ull * ((size_t)q * ps.nslots + (size_t)chunk * ps.chunk);     // 16-byte aligned: nslots and ps.chunk are multiples of 4
                float4* z4 = (float4*)z;
                for (uint32_t k = (uint32_t)lane; k < total / 4u; k += 64u) st4<2>(&z4[k], make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                for (uint32_t k = (total & ~3u) + (uint32_t)lane; k < total; k += 64u) z[k] = 0.0f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t out_front = 0u, out_back = 0u, dropped = 0u, skipped = 0u;
        for (uint32_t j = 0; j < cnt; j += 64u) {
            bool active = j + (uint32_t)lane < cnt;
            const bool in_batch = active;                      // PRIMARY: owns the light-term slot chunk * ps.chunk + j + lane, shaded or not
            f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), hp = mk3(0, 0, 0), n = mk3(0, 0, 1);
            uint32_t slot = 0u, node = 0u, pixel = 0u, sampleno = 0u, geom = 0u;
            float4 h = make_float4(0, 0, 0, 0);
            if (active) {
                const uint32_t i = list[j + (uint32_t)lane];
                h = RASTER ? lds_hits[j + (uint32_t)lane] : ld4<2>(&hits[base + i]);
                if (PRIMARY) {
                    slot = chunk * ps.chunk + j + (uint32_t)lane;       // == sample_slot[chunk * ps.chunk + i]
                    primary_sample(cam, ps, film_n, chunk * ps.chunk + i, pixel, sampleno, o, d);
                } else {
                    const float4 r0 = ld4<2>(&in_q[base + i]), r1 = ld4<2>(&in_q[ps.qstride + base + i]);
                    o = mk3(r0.x, r0.y, r0.z); d = mk3(r0.w, r1.x, r1.y);
                    slot = __float_as_uint(r1.z); node = (__float_as_uint(r1.w) >> 8) & 0xFFFFu;
                    const uint2 px = ps.slot_ps[slot];
                    pixel = px.x; sampleno = px.y;
                }
                if (PRIMARY && level < ps.recursions) ps.slot_ps[slot] = make_uint2(pixel, sampleno);   // what the deeper levels hash with
                if (WALK) {
                    float wt = h.x, wu = h.y, wv = h.z; uint32_t wprim = __float_as_uint(h.w);
#ifndef MI355RT_EXP_SHADE_NOWALK     // timing experiments (wrong picture): what do the parts of the shade kernels cost?
                    confirm_walk(sc, o, d, wt, wu, wv, wprim);
#endif
                    h = make_float4(wt, wu, wv, __uint_as_float(wprim));
                    active = wprim != kMiss;
                }
            }
            if (WALK) dropped += (uint32_t)__popcll(__ballot(j + (uint32_t)lane < cnt && !active));
            if (active) {
                hp = add3(o, sscale(h.x, d));                                          // mod.rs:212
                const float4 nn = ((const float4*)sc.normals)[__float_as_uint(h.w)];   // calc_normal, mod.rs:198-205 (precomputed)
                n = mk3(nn.x, nn.y, nn.z);
                geom = __float_as_uint(nn.w);
            }
            // ---- shade() set-up per light, mod.rs:214-257; the shadow ray carries the finished term
            for (uint32_t li = 0; li < sc.nlights; ++li) {
                bool want = false, is_free = false;
                f3 c = mk3(0, 0, 0);
                if (active) {
                    const DLight lt = sc.lights[li];
                    const f3 l = sub3(mk3(lt.px, lt.py, lt.pz), hp);                   // mod.rs:215
                    const f3 ln = normalized3(l);
                    const float ndl = dot3(n, ln);                                     // mod.rs:216
                    if (!(ndl < 0.0f)) {                                               // mod.rs:218
                        want = true;                                                   // the shadow ray of mod.rs:224-225: rebuilt from hp by its tracer (shadow_ray_of)
                        is_free = light_proves_unoccluded(sc, li, lt, l);
                        const DMaterial m = sc.materials[geom];
                        f3 diffuse = mk3(m.r, m.g, m.b);
                        if (m.kind_tex & 0x80000000u) diffuse = fetch_texel(sc, m.kind_tex & 0x7FFFFFFFu, h.y, h.z);
                        const f3 view = normalized3(d);                                // mod.rs:251
                        const f3 refl = sub3(sscale(2.0f * ndl, n), ln);               // mod.rs:252-253
                        const float spec = pow32(dot3(view, refl));                    // mod.rs:255, SPECULAR = white
                        c = mk3((diffuse.x * ndl + spec) * lt.cr, (diffuse.y * ndl + spec) * lt.cg, (diffuse.z * ndl + spec) * lt.cb);
                    }
                }
                uint32_t n_new;
#ifdef MI355RT_EXP_SHADE_NOLIGHT
                want = false;
#endif
                // a shadow ray the light's depth map proves free is never made: its term stands as written
                const bool ray = want && !is_free;
                skipped += (uint32_t)__popcll(__ballot(want && is_free));
                const uint32_t oi = wave_append(ray, out_back, n_new);
                if (want && (!ray || out_front + oi < ps.region)) {
                    const uint32_t term = 3u * ((node * sc.nlights + li) * ps.nslots + slot);             // float index in slot_L (< 2^32: renderer.cpp)
                    if (ray) st4<2>(&out_q[base + (ps.region - 1u - oi)], make_float4(hp.x, hp.y, hp.z, __uint_as_float(term)));
                    float* dst = slot_L + term;                    // optimistic: whoever finds the shadow ray blocked zeroes it again (store_blocked)
                    dst[0] = c.x; dst[1] = c.y; dst[2] = c.z;
                } else if (PRIMARY && in_batch) {
                    // no shadow ray for this light (mod.rs:218), or the octree dropped the hit: the term of node 0 reads back as black (mod.rs:99-100, 211)
                    float* dst = slot_L + 3ull * ((size_t)li * ps.nslots + (size_t)chunk * ps.chunk + j + (uint32_t)lane);
                    dst[0] = 0.0f; dst[1] = 0.0f; dst[2] = 0.0f;
                }
            }
            // ---- reflection rays, mod.rs:146-158 + 178-196
#ifdef MI355RT_EXP_SHADE_NOREFL
            if (false) {
#else
            if (level < ps.recursions) {
#endif
                const uint32_t k = ps.spread * (ps.recursions - level);                // num_sub_rays, mod.rs:150
                const uint32_t index_in_level = node - ps.level_first[level];
                // children in pairs: both first table entries are in flight before either walk starts (the walks are chains of
                // dependent cache accesses; one after the other they were most of this kernel's latency)
                for (uint32_t ci = 0; ci < k; ci += 2u) {
                    const bool two = ci + 1u < k;
                    const uint32_t child0 = ps.level_first[level + 1] + index_in_level * k + ci, child1 = child0 + 1u;
                    f3 bo0 = mk3(0, 0, 0), bd0 = mk3(0, 0, 1), bo1 = bo0, bd1 = bd0;
                    if (active) {
                        const float4* __restrict__ table = (const float4*)sc.table;
                        uint32_t a0 = pixel, a1 = sampleno, a2 = 1u + child0, a3 = ps.seed;
                        pcg4d(a0, a1, a2, a3);
                        uint32_t jx0 = __umulhi(a0, 65535u), jx1 = jx0;                  // uniform in [0, 65534], sample_generator.rs:32
                        if (two) {
                            uint32_t b0 = pixel, b1 = sampleno, b2 = 1u + child1, b3 = ps.seed;
                            pcg4d(b0, b1, b2, b3);
                            jx1 = __umulhi(b0, 65535u);
                        }
                        float4 tv0 = table[jx0], tv1 = table[jx1];
#if defined(MI355RT_EXP_DIRCOH) && MI355RT_EXP_DIRCOH == 1      // timing experiment (wrong picture): every reflection ray of a chunk starts its table walk at the same entry
                        jx0 = (chunk * 2654435761u) >> 17; jx1 = jx0 + 7u; tv0 = table[jx0]; tv1 = table[jx1];
#endif
                        hemisphere_walk(table, n, jx0, tv0);
                        if (two) hemisphere_walk(table, n, jx1, tv1);
#if defined(MI355RT_EXP_DIRCOH) && MI355RT_EXP_DIRCOH == 2      // timing experiment (wrong picture): every reflection ray of a chunk points into ONE octant (that of the first lane's normal): what would sorting a chunk's rays by octant buy at best?
                        {
                            const uint32_t oct = (uint32_t)__builtin_amdgcn_readfirstlane((int)((n.x < 0.0f ? 1u : 0u) | (n.y < 0.0f ? 2u : 0u) | (n.z < 0.0f ? 4u : 0u)));
                            for (int w = 0; w < 2; ++w) {
                                uint32_t& jx = w ? jx1 : jx0; float4& tv = w ? tv1 : tv0;
                                for (uint32_t g = 0; g < 400u; ++g) {
                                    const uint32_t o2 = (tv.x < 0.0f ? 1u : 0u) | (tv.y < 0.0f ? 2u : 0u) | (tv.z < 0.0f ? 4u : 0u);
                                    if (o2 == oct && tv.x * n.x + tv.y * n.y + tv.z * n.z > 0.0f) break;
                                    jx = jx + 1u == kSampleMax ? 0u : jx + 1u; tv = table[jx];
                                }
                                if (!(tv.x * n.x + tv.y * n.y + tv.z * n.z > 0.0f)) hemisphere_walk(table, n, jx, tv);
                            }
                        }
#endif
                        bd0 = mk3(tv0.x, tv0.y, tv0.z); bo0 = add3(hp, sscale(0.00001f, bd0));   // mod.rs:192-193
                        bd1 = mk3(tv1.x, tv1.y, tv1.z); bo1 = add3(hp, sscale(0.00001f, bd1));
                    }
                    // Where the batch's reflection rays go in the chunk's region: grouped by the OCTANT of their direction.  The rays of a batch start at
                    // neighbouring hit points; those that also agree in the signs of their direction make the same near / far choice at every node, so they
                    // walk the tree together for longer, and lanes of a quad that fetch the same node cost the memory pipe one access instead of four
                    // (profiles/r03_notes.md: all rays of a chunk in one octant would take 14 % off the secondary trace launches).  Eight ballots per child;
                    // the order of the records inside a region means nothing to anyone else.
                    uint32_t pos[2] = { 0u, 0u };
                    if (kSortBins != 0u) {
                        const uint32_t key0 = active ? dir_bin(bd0) : kSortBins, key1 = (active && two) ? dir_bin(bd1) : kSortBins;
                        uint32_t run = out_front;
                        for (uint32_t b = 0; b < kSortBins; ++b) {
                            const unsigned long long m0 = __ballot(key0 == b), m1 = __ballot(key1 == b);
                            if (key0 == b) pos[0] = run + (uint32_t)__popcll(m0 & lanemask_lt());
                            run += (uint32_t)__popcll(m0);
                            if (key1 == b) pos[1] = run + (uint32_t)__popcll(m1 & lanemask_lt());
                            run += (uint32_t)__popcll(m1);
                        }
                        out_front = run;
                    }
                    for (uint32_t w = 0; w < (two ? 2u : 1u); ++w) {
                        const f3 bo = w ? bo1 : bo0, bd = w ? bd1 : bd0;
                        const uint32_t child_node = w ? child1 : child0;
                        uint32_t n_new;
                        const uint32_t oi = kSortBins != 0u ? pos[w] : wave_append(active, out_front, n_new);
                        if (active && oi + out_back < ps.region + 0u) {
                            const size_t r = base + oi;
                            st4<2>(&out_q[r], make_float4(bo.x, bo.y, bo.z, bd.x));
                            st4<2>(&out_q[ps.qstride + r], make_float4(bd.y, bd.z, __uint_as_float(slot), __uint_as_float(((level + 1u) << 4) | (child_node << 8))));
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (out_front + out_back > ps.region) { if (lane == 0) counters->overflow = 1u; out_front = 0u; out_back = 0u; }
        if (out_counts != nullptr && lane == 0) out_counts[chunk] = make_uint2(out_front, out_back);     // fused launch: null, the counts travel in registers
        out_nrad = out_front; out_nshadow = out_back;
        acc_bounce += out_front; acc_shadow += out_back + ((unsigned long long)skipped << 32); acc_hits += cnt - dropped;     // high half of acc_shadow: shadow rays never made
    }
}

__device__ __forceinline__ void flush_shade_counters(DCounters* counters, uint32_t wave, unsigned long long acc_bounce, unsigned long long acc_shadow, unsigned long long acc_hits)
{
    if (lane_id() == 0) {
        DCounters* cs = &counters[wave % kShards];
        if (acc_bounce) atomicAdd(&cs->bounce, acc_bounce);
        if (acc_shadow & 0xFFFFFFFFull) atomicAdd(&cs->shadow, acc_shadow & 0xFFFFFFFFull);
        if (acc_shadow >> 32) atomicAdd(&cs->shadow_skipped, acc_shadow >> 32);
        if (acc_hits & 0xFFFFFFFFull) atomicAdd(&cs->primary_hits, acc_hits & 0xFFFFFFFFull);
        if (acc_hits >> 32) atomicAdd(&cs->primary_culled, acc_hits >> 32);
    }
}

template <bool PRIMARY, bool WALK, bool RASTER = false>
__global__ __launch_bounds__(kBlock, PRIMARY ? (WALK ? MI355RT_SHADE_PW_BLOCKS : MI355RT_SHADE_P_BLOCKS) : (WALK ? MI355RT_SHADE_SW_BLOCKS : MI355RT_SHADE_S_BLOCKS)) void shade_kernel(DScene sc, DCamera cam, DPass ps, uint32_t level,
                                                      const float4* __restrict__ in_q, const uint2* __restrict__ in_counts,
                                                      const float4* __restrict__ hits,
                                                      float4* __restrict__ out_q, uint2* __restrict__ out_counts, uint32_t* cursor,
                                                      float* __restrict__ slot_L, uint32_t* __restrict__ sample_slot,
                                                      const uint32_t* __restrict__ film_n, DCounters* counters)
{
    extern __shared__ uint32_t s_list[];             // kWavesPerBlock lists of ps.list_cap hit indices (+ RASTER: kWavesPerBlock x ps.chunk hit records of 16 B)
    const LinearList list{ &s_list[(threadIdx.x >> 6) * ps.list_cap] };
    float4* lds_hits = RASTER ? (float4*)&s_list[kWavesPerBlock * ps.list_cap] + (threadIdx.x >> 6) * ps.chunk : nullptr;
    const uint32_t wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    unsigned long long acc_bounce = 0, acc_shadow = 0, acc_hits = 0, acc_tris = 0;
    // chunks are pulled from the round's cursors: the hits sit in a part of the image, culled / empty chunks cost nothing
    PullState pull; uint32_t chunk = 0u;
    while (pull_chunk(cursor, ps.nchunks, kShadePullMode ? kShadePullMode : ps.pull_mode, ps.ncursors, ps.pull_group, pull, chunk, 0u, 0u, PRIMARY ? LiveLists() : live_lists(ps))) {
        uint32_t o_rad, o_sh;
        shade_chunk<PRIMARY, LinearList, WALK, RASTER>(sc, cam, ps, level, chunk, list, in_q, PRIMARY ? 0u : in_counts[chunk].x, o_rad, o_sh, hits, out_q, out_counts, slot_L, sample_slot, film_n, counters, acc_bounce, acc_shadow, acc_hits,
                                                       lds_hits, (RASTER && (ps.flags & 2u)) ? &acc_tris : nullptr);      // flags & 2: MI355RT_FLAG_COUNT_STEPS
        // a chunk that leaves the primary round with rays goes on the live list of its cursor: the later launches of the pass visit only those (DPass::live)
        if (PRIMARY && kLiveListsOk && ps.live != nullptr && (o_rad | o_sh) != 0u && lane_id() == 0) {
            const uint32_t k = chunk % ps.ncursors;
            const uint32_t at = atomicAdd(&ps.live_count[(size_t)k * kCursorStride + kLiveCountOffset], 1u);
            ps.live[(size_t)k * ps.live_cap + at] = chunk;
        }
    }
    flush_shade_counters(counters, wave, acc_bounce, acc_shadow, PRIMARY ? acc_hits : 0ull);
    if (RASTER && acc_tris && lane_id() == 0) atomicAdd(&counters[wave % kShards].tris_tested, acc_tris);
}

// ---- resolve: radiance tree -> sample colour -> film ------------------------------------------
template <int REC>
__device__ f3 node_radiance(const float* __restrict__ L, const DPass& ps, uint32_t nlights, uint32_t level, uint32_t index)
{
    const uint32_t node = ps.level_first[level] + index;
    f3 rad = mk3(0.0f, 0.0f, 0.0f);                                    // accum_color, mod.rs:211
    for (uint32_t li = 0; li < nlights; ++li) {
        const float* p = L + 3ull * ((size_t)node * nlights + li) * ps.nslots;
        rad = add3(rad, mk3(p[0], p[1], p[2]));                         // mod.rs:254
    }
    if constexpr (REC == 0) {
        return rad;                                                     // mod.rs:146-148
    } else {
        const uint32_t k = ps.spread * (uint32_t)REC;                   // mod.rs:150
        f3 sum = mk3(0.0f, 0.0f, 0.0f);
        for (uint32_t c = 0; c < k; ++c) sum = add3(sum, node_radiance<REC - 1>(L, ps, nlights, level + 1, index * k + c));   // fold, mod.rs:173
        const f3 sub = vscale(sum, div_rn(1.0f, (float)k));          // mod.rs:174
        return add3(rad, sub);                                          // mod.rs:175
    }
}

// kResolveLanes consecutive lanes work on one pixel: lane j evaluates the radiance trees of samples j,
// j + kResolveLanes, ... (the loads), then the values of one round are exchanged inside the wave and every
// lane of the group ADDS them in sample order (bit-exactness) — redundantly, so there is no divergence.
// With one thread per pixel, a pass with few pixels and many samples per pixel (one rank's stripes of an
// 8-GPU frame: 512 spp) is a latency-bound loop of dependent loads on a nearly empty chip; with many lanes per
// pixel and few samples most lanes of a group have nothing to load.  The launcher picks the group size from
// the samples per pixel of the pass (21-spp passes of a 1080p frame: 1 / 2 / 4 / 8 / 16 lanes -> 0.22 / 0.21 /
// 0.25 / 0.31 / 0.55 ms).
template <uint32_t kResolveLanes>
__global__ __launch_bounds__(256) void resolve_kernel(DPass ps, uint32_t width, uint32_t nlights, const float* __restrict__ slot_L,
                                                     const uint32_t* __restrict__ sample_slot,
                                                     float* film_sum, float* film_sumsq, uint32_t* film_n, float* debug_color, uint32_t* ctrl)
{
    // last kernel of a pass: leave the pass's work cursors zeroed for the next one (960 words; a 20 MiB memset otherwise)
    if (ctrl != nullptr && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < kMaxRounds * kMaxCursors * 4u; i += blockDim.x)     // trace, confirm, shade cursors and the live count of every (round, cursor)
            ctrl[(size_t)(i / (kMaxCursors * 4u)) * kCtrlWordsPerRound + (size_t)((i / 4u) % kMaxCursors) * kCursorStride + (i % 4u) * kConfirmCursorOffset] = 0u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = gid % kResolveLanes;
    const bool live = gid / kResolveLanes < ps.npix;
    const uint32_t p = live ? gid / kResolveLanes : ps.npix - 1u;        // surplus groups of the last block shadow the last pixel
    const uint32_t spp = ps.nsamples / ps.npix;
    const uint32_t pixel = ps.use_explicit ? ps.explicit_pixel : pass_pixel(ps, width, p);
    f3 sum = mk3(0, 0, 0), sumsq = mk3(0, 0, 0);
    uint32_t n = 0;
    if (!ps.use_explicit) {
        sum = mk3(film_sum[3ull * pixel], film_sum[3ull * pixel + 1], film_sum[3ull * pixel + 2]);
        sumsq = mk3(film_sumsq[3ull * pixel], film_sumsq[3ull * pixel + 1], film_sumsq[3ull * pixel + 2]);
        n = film_n[pixel];
    }
    const int group_lane0 = lane_id() & ~(int)(kResolveLanes - 1u);
    // every sample of a pixel in a culled block is a miss: no slot to look up (the primary shade launch wrote none)
    const bool dead = ps.block_culled != nullptr && ps.block_culled[(uint32_t)(sample_index(ps, 0u, p) / ps.chunk) % ps.cull_blocks] != 0u;
    for (uint32_t s0 = 0; s0 < spp; s0 += kResolveLanes) {
        const uint32_t sl = (s0 + j < spp && !dead) ? sample_slot[sample_index(ps, s0 + j, p)] : 0xFFFFFFFFu;
        const float* L = slot_L + 3ull * (size_t)sl;       // the slot's entry of plane 0; plane q is 3 * q * nslots floats further
        f3 c = mk3(0.0f, 0.0f, 0.0f);                                   // primary miss: RGB::black(), mod.rs:100
        if (sl != 0xFFFFFFFFu) switch (ps.recursions) {
            case 0: c = node_radiance<0>(L, ps, nlights, 0, 0); break;
            case 1: c = node_radiance<1>(L, ps, nlights, 0, 0); break;
            case 2: c = node_radiance<2>(L, ps, nlights, 0, 0); break;
            default: c = node_radiance<3>(L, ps, nlights, 0, 0); break;
        }
#pragma unroll
        for (uint32_t k = 0; k < kResolveLanes; ++k) {
            const f3 ck = mk3(__shfl(c.x, group_lane0 + (int)k, 64), __shfl(c.y, group_lane0 + (int)k, 64), __shfl(c.z, group_lane0 + (int)k, 64));
            if (s0 + k < spp) {
                // PixelData::add_sample, film.rs:20-24
                sum = add3(sum, ck);
                sumsq = add3(sumsq, mk3(ck.x * ck.x, ck.y * ck.y, ck.z * ck.z));
                n += 1u;
                if (ps.use_explicit && j == 0u) { debug_color[0] = ck.x; debug_color[1] = ck.y; debug_color[2] = ck.z; }
            }
        }
    }
    if (!ps.use_explicit && live && j == 0u) {
        film_sum[3ull * pixel] = sum.x; film_sum[3ull * pixel + 1] = sum.y; film_sum[3ull * pixel + 2] = sum.z;
        film_sumsq[3ull * pixel] = sumsq.x; film_sumsq[3ull * pixel + 1] = sumsq.y; film_sumsq[3ull * pixel + 2] = sumsq.z;
        film_n[pixel] = n;
    }
}

// ---- one 50-row frame in ONE launch (trace_frame_additive, mod.rs:80-117) ---------------------------------
// A call of the reference's entry traces 50 rows x 1 sample: ~51 k primary samples at the reference binary's
// 1024x768, a few microseconds of work for this chip.  As wavefront rounds that is 8 dependent launches of a
// persistent grid: launch latency and drain, nothing else.  Here every wave takes a chunk of 64 samples (an 8x8
// pixel tile) through ALL rounds by itself — trace, shade, trace, ... , resolve — with the same device functions,
// the same queue regions (its chunk's, in global memory) and therefore the same arithmetic as the wavefront
// kernels; between phases a fence makes the wave's own stores visible to its loads.  No cursor, no memset, no
// inter-wave dependency; samples per pixel in such a pass is 1 (film.rs:20-24: one add per pixel).
// The wave reads back only what IT stored (its chunk's queue region, hit records, light terms): the stores must have
// left the wave (vmcnt) and the loads must not be served from a stale line of this CU's vector cache.  WORKGROUP scope
// is exactly that on gfx950 (the waves of a workgroup share the CU's cache, which stores write through): a wait, no
// L2 write-back / invalidate.  An agent-scope fence here flushes and invalidates the XCD's whole L2 eight times per
// wave — it made this kernel 3.5x slower (0.35 ms -> see profiles/r02_notes.md), evicting the BVH each time.
__device__ __forceinline__ void phase_fence()
{
    if (kFusedFenceAgent) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
}

__device__ __forceinline__ void resolve_chunk_1spp(const DPass& ps, uint32_t width, uint32_t nlights, uint32_t chunk, const float* slot_L, const uint32_t* sample_slot,
                                                   float* film_sum, float* film_sumsq, uint32_t* film_n)
{
    const uint32_t n = min(ps.chunk, ps.nsamples - chunk * ps.chunk);
    for (uint32_t i = (uint32_t)lane_id(); i < n; i += 64u) {
        const uint32_t p = chunk * ps.chunk + i;                         // 1 sample per pixel: sample index == pixel index of the pass
        const uint32_t pixel = pass_pixel(ps, width, p);
        const uint32_t sl = sample_slot[p];
        const float* L = slot_L + 3ull * (size_t)sl;       // the slot's entry of plane 0; plane q is 3 * q * nslots floats further
        f3 c = mk3(0.0f, 0.0f, 0.0f);                                    // primary miss: RGB::black(), mod.rs:100
        if (sl != 0xFFFFFFFFu) switch (ps.recursions) {
            case 0: c = node_radiance<0>(L, ps, nlights, 0, 0); break;
            case 1: c = node_radiance<1>(L, ps, nlights, 0, 0); break;
            case 2: c = node_radiance<2>(L, ps, nlights, 0, 0); break;
            default: c = node_radiance<3>(L, ps, nlights, 0, 0); break;
        }
        // PixelData::add_sample, film.rs:20-24
        float* ps_ = film_sum + 3ull * pixel; float* pq = film_sumsq + 3ull * pixel;
        ps_[0] = ps_[0] + c.x; ps_[1] = ps_[1] + c.y; ps_[2] = ps_[2] + c.z;
        pq[0] = pq[0] + c.x * c.x; pq[1] = pq[1] + c.y * c.y; pq[2] = pq[2] + c.z * c.z;
        film_n[pixel] = film_n[pixel] + 1u;
    }
}

#ifndef MI355RT_FUSED_BLOCKS
#define MI355RT_FUSED_BLOCKS 1                 // blocks per CU the fused 50-row kernel is compiled for (A/B knob: 4 = 128 VGPRs)
#endif
template <bool CONFIRM>
__global__ __launch_bounds__(kBlock, MI355RT_FUSED_BLOCKS) void fused_pass_kernel(DScene sc, DCamera cam, DPass ps, float4* q0, float4* q1,
                                                            float4* hits, float* slot_L, uint32_t* sample_slot,
                                                            float* film_sum, float* film_sumsq, uint32_t* film_n, DCounters* counters)
{
    extern __shared__ int s_stack[];                 // max(traversal stack rows, hit-list rows) x kBlock ints
    int* stack = &s_stack[threadIdx.x];
    const ColumnList list{ &s_stack[threadIdx.x & ~63u] };
    const uint32_t wave = global_wave_id(), nwaves = gridDim.x * kWavesPerBlock;
    unsigned long long acc_bounce = 0, acc_shadow = 0, acc_hits = 0;
    for (uint32_t chunk = wave; chunk < ps.nchunks; chunk += nwaves) {
        // ray counts of the chunk travel from phase to phase in registers: a wave-uniform load of the count the wave
        // itself stored a moment ago would go through the scalar cache, which may hold the line from a neighbour's read
        uint32_t n_rad = 0u, n_sh = 0u;
        trace_wave<true, false, true, true>(sc, cam, ps, nullptr, nullptr, hits, nullptr, slot_L, film_n, counters, stack, chunk, 0u, 0u);
        phase_fence();
        if (CONFIRM && !sc.oct_single_leaf) { confirm_chunk<true>(sc, cam, ps, chunk, list, nullptr, 0u, 0u, hits, slot_L, film_n); phase_fence(); }
        shade_chunk<true>(sc, cam, ps, 0u, chunk, list, nullptr, 0u, n_rad, n_sh, hits, q0, nullptr, slot_L, sample_slot, film_n, counters, acc_bounce, acc_shadow, acc_hits);
        phase_fence();
        for (uint32_t r = 1; r < ps.recursions + 2u; ++r) {
            float4* in_q = (r - 1u) & 1u ? q1 : q0;
            trace_wave<false, false, true, CONFIRM>(sc, cam, ps, in_q, nullptr, hits, nullptr, slot_L, film_n, counters, stack, chunk, n_rad, n_sh);
            phase_fence();
            if (CONFIRM && !sc.oct_single_leaf) { confirm_chunk<false>(sc, cam, ps, chunk, list, in_q, n_rad, n_sh, hits, slot_L, film_n); phase_fence(); }
            if (r <= ps.recursions) {
                unsigned long long unused = 0;
                uint32_t o_rad = 0u, o_sh = 0u;
                shade_chunk<false>(sc, cam, ps, r, chunk, list, in_q, n_rad, o_rad, o_sh, hits, r & 1u ? q1 : q0, nullptr, slot_L, sample_slot, film_n, counters, acc_bounce, acc_shadow, unused);
                n_rad = o_rad; n_sh = o_sh;
                phase_fence();
            }
        }
        resolve_chunk_1spp(ps, cam.width, sc.nlights, chunk, slot_L, sample_slot, film_sum, film_sumsq, film_n);
    }
    flush_shade_counters(counters, wave, acc_bounce, acc_shadow, acc_hits);
}

// ---- get_tonemapped_pixels, mod.rs:120-128 = film.rs:43-47 + tonemap.rs:4-10 + color.rs:85-95 --
__device__ __forceinline__ uint32_t to_u8(float c)
{
    // Rust f32::min/max return the non-NaN operand (so NaN -> 1.0 -> 255); `as u8` truncates
    const float m = fmaxf(fminf(c, 1.0f), 0.0f) * 255.0f;
    return (uint32_t)m & 0xFFu;
}
__global__ __launch_bounds__(256) void tonemap_kernel(const uint32_t* __restrict__ rows, uint32_t row_base, uint32_t nrows, uint32_t width, int packed,
                                                     const float* __restrict__ film_sum, const uint32_t* __restrict__ film_n, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nrows * width) return;
    const uint32_t r = (uint32_t)(i / width), x = (uint32_t)(i % width);
    const size_t pixel = (size_t)(rows ? rows[r] : row_base + r) * width + x;      // rows == null: the rows row_base .. row_base + nrows - 1
    const float inv = div_rn(1.0f, (float)film_n[pixel]);                     // film.rs:46
    const float cr = film_sum[3 * pixel] * inv, cg = film_sum[3 * pixel + 1] * inv, cb = film_sum[3 * pixel + 2] * inv;
    const uint32_t R = to_u8(div_rn(cr, 1.0f + cr)), G = to_u8(div_rn(cg, 1.0f + cg)), B = to_u8(div_rn(cb, 1.0f + cb));
    out[packed ? i : pixel] = B | (G << 8) | (R << 16) | (255u << 24);           // color.rs:94
}

// ---- batched Intersector seam (accel_intersect.rs:10-13) --------------------------------------
// mode 0: BVH true closest hit + octree confirm (reference-default semantics); 1: the reference's octree walked
// directly (cross-check path); 2: BVH true closest hit only (NoAccelerationIntersector semantics)
__global__ __launch_bounds__(kBlock) void intersect_kernel(DScene sc, const float* __restrict__ rays6, uint32_t n, int shadow_mode, int mode,
                                                          float* tuv, uint32_t* prim, uint8_t* blocked)
{
    extern __shared__ int s_stack[];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 o = mk3(rays6[6ull * i], rays6[6ull * i + 1], rays6[6ull * i + 2]);
    const f3 d = mk3(rays6[6ull * i + 3], rays6[6ull * i + 4], rays6[6ull * i + 5]);
    if (mode == 1) {             // reference-exact intersector, walked directly
        float t, u, v; uint32_t p;
        octree_intersect(sc, o, d, t, u, v, p);
        if (shadow_mode) blocked[i] = (p != kMiss && t > 0.01f && t < 1.0f) ? 1 : 0;
        else {
            prim[i] = p;
            if (p != kMiss) { tuv[3ull * i] = t; tuv[3ull * i + 1] = u; tuv[3ull * i + 2] = v; }
        }
        return;
    }
    RayState rs;
    uint32_t a = 0, b = 0;
    if (mode == 0) {
        // closest hit (shadow rays: on [0, 1)), then the reference's octree decides what it would have returned
        ray_init(rs, o, d, false, sc.root);
        if (shadow_mode) rs.tlimit = 0x1.fffffep-1f;
        ray_run<false>(sc, rs, &s_stack[threadIdx.x], kBlock, a, b);
        float t = rs.t, u = rs.u, v = rs.v; uint32_t p = rs.prim;
        if (p != kMiss) confirm_walk(sc, o, d, t, u, v, p);
        if (shadow_mode) blocked[i] = (p != kMiss && t > 0.01f && t < 1.0f) ? 1 : 0;
        else {
            prim[i] = p;
            if (p != kMiss) { tuv[3ull * i] = t; tuv[3ull * i + 1] = u; tuv[3ull * i + 2] = v; }
        }
        return;
    }
    ray_init(rs, o, d, shadow_mode != 0, sc.root);
    ray_run<false>(sc, rs, &s_stack[threadIdx.x], kBlock, a, b);
    if (shadow_mode) blocked[i] = rs.occ == 1 ? 1 : 0;
    else {
        prim[i] = rs.prim;
        if (rs.prim != kMiss) { tuv[3ull * i] = rs.t; tuv[3ull * i + 1] = rs.u; tuv[3ull * i + 2] = rs.v; }
    }
}

// ---- device arithmetic self-check: a/b, sqrt(a), a^32 as the kernels compute them ----------------
__global__ void numerics_kernel(const float* __restrict__ a, const float* __restrict__ b, uint32_t n, float* q, float* r, float* p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q[i] = div_rn(a[i], b[i]);
    r[i] = sqrt_rn(a[i]);
    p[i] = pow32(a[i]);
}
// ---- the reference's slab test as the device runs it (known-answer vectors of oct_tree_intersector.rs:475-512) ----
__global__ void slab_kernel(const float* __restrict__ inv_rays6, const float* __restrict__ cubes6, uint32_t n, uint8_t* hit, float* tmin)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = inv_rays6 + 6ull * i; const float* c = cubes6 + 6ull * i;
    float t = 0.0f;
    const bool h = cube_slab(mk3(c[0], c[1], c[2]), mk3(c[3], c[4], c[5]), mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), t);
    hit[i] = h ? 1 : 0; tmin[i] = t;
}
hipError_t launch_slab(hipStream_t stream, const float* inv_rays6, const float* cubes6, uint32_t n, uint8_t* hit, float* tmin)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(slab_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, inv_rays6, cubes6, n, hit, tmin);
    return hipGetLastError();
}

// ---- the roofline of the trace kernels, measured: what the CU's vector memory pipe delivers when it does nothing but the
// trace kernel's kind of fetch (profiles/r03_notes.md, tools/micro/gather_bench.hip shape A).  Every lane walks a dependent chain
// of random 32-byte "nodes" of an L2-resident table with the inner step's two 16-byte loads; 8 waves per SIMD, grid = the chip.
// Each lane touches its own cache line per step: 64 lines per load instruction (rocprofv3: TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD = 64.0).
__global__ __launch_bounds__(kBlock, 8) void gather_rate_kernel(const uint4* __restrict__ table, uint32_t nnodes, uint32_t steps, uint32_t* sink)
{
    const uint32_t ray = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t node = (ray * 2654435761u) % nnodes, acc = 0u;
    for (uint32_t s = 0; s < steps; ++s) {
        const uint4 q0 = table[2u * node], q1 = table[2u * node + 1u];
        acc += q0.x ^ q1.y;
        node = (q0.w + q1.w + s * 40503u + ray) % nnodes;              // dependent: the next node comes from the data
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
hipError_t launch_gather_rate(hipStream_t stream, int num_cus, const void* table, uint32_t nnodes, uint32_t steps, uint32_t* sink)
{
    hipLaunchKernelGGL(gather_rate_kernel, dim3((unsigned)num_cus * 8u), dim3(kBlock), 0, stream, (const uint4*)table, nnodes, steps, sink);
    return hipGetLastError();
}

// ---- Film::get_pixels (film.rs:43-47) and Film::get_estimated_variances (film.rs:51-67) on the device ----
__global__ __launch_bounds__(256) void film_stat_kernel(int variances, size_t npix, const float* __restrict__ film_sum, const float* __restrict__ film_sumsq,
                                                       const uint32_t* __restrict__ film_n, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint32_t n = film_n[i];
    if (!variances) {
        const float inv = div_rn(1.0f, (float)n);                                   // film.rs:46
        out[3 * i] = film_sum[3 * i] * inv; out[3 * i + 1] = film_sum[3 * i + 1] * inv; out[3 * i + 2] = film_sum[3 * i + 2] * inv;
        return;
    }
    const float nn1 = (float)(uint32_t)(n * (n - 1u));                              // film.rs:55-56 (u32 product)
    const float n2n1 = (float)n * nn1;                                              // film.rs:57
    for (int c = 0; c < 3; ++c) {
        const float s = film_sum[3 * i + c], q = film_sumsq[3 * i + c];
        out[3 * i + c] = (div_rn(q, nn1) - div_rn(s * s, n2n1)) * 50.0f;            // film.rs:58-64
    }
}
hipError_t launch_film_stat(hipStream_t stream, bool variances, size_t npix, const float* film_sum, const float* film_sumsq, const uint32_t* film_n, float* out)
{
    if (npix == 0) return hipSuccess;
    hipLaunchKernelGGL(film_stat_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, variances ? 1 : 0, npix, film_sum, film_sumsq, film_n, out);
    return hipGetLastError();
}

hipError_t launch_numerics(hipStream_t stream, const float* a, const float* b, uint32_t n, float* q, float* r, float* p)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(numerics_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, n, q, r, p);
    return hipGetLastError();
}

// ---- launchers --------------------------------------------------------------------------------
bool kernels_walk_wide_nodes() { return MI355RT_WIDE != 0; }

static size_t stack_bytes(uint32_t depth) { return (size_t)((depth ? depth : 1u) + 1u) * kBlock * sizeof(int); }   // sentinel row + one row per level (the deepest level's row doubles as the spare row above the top)

template <bool P, bool C, bool F>
static int trace_blocks_per_cu(size_t lds)
{
    // the occupancy query costs ~0.3 ms of host time: ask once per (kernel, device, LDS size).  Device groups launch from one
    // host thread per device, so the cache is guarded.
    static std::mutex mu;
    static std::map<std::pair<int, size_t>, int> cache;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({ dev, lds });
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel<P, C, F>, kBlock, lds) != hipSuccess || nb < 1) nb = 1;
    nb = nb > 8 ? 8 : nb;
    cache[{ dev, lds }] = nb;
    return nb;
}

template <bool P, bool C, bool F>
static hipError_t launch_trace_variant(hipStream_t stream, int num_cus, int blocks_per_cu_cap, const DScene& sc, const DCamera& cam, const DPass& ps,
                                       const void* in_q, const void* in_counts, void* hits, uint32_t* cursor,
                                       float* slot_L, const uint32_t* film_n, DCounters* counters)
{
    // persistent grid: as many blocks as the chip holds; waves pull chunks from `cursor`
    const size_t lds = stack_bytes(ps.stack_depth);
    const int per_cu = trace_blocks_per_cu<P, C, F>(lds);
    int use_per_cu = per_cu;
    if (blocks_per_cu_cap > 0 && blocks_per_cu_cap < use_per_cu) use_per_cu = blocks_per_cu_cap;      // leave room for another stream's kernels
    if (const char* e = getenv("MI355RT_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= per_cu) use_per_cu = v; }   // occupancy experiment
    // tail of every cursor's chunk sequence that is handed out in parts (pull_chunk): ps.tail_chunks comes in as "chunks per WAVE"
    DPass pt = ps;
    const uint32_t waves_per_cursor = ((uint32_t)(num_cus * use_per_cu) * kWavesPerBlock + ps.ncursors - 1u) / ps.ncursors;
    pt.tail_chunks = (ps.pull_mode == 4u && ps.pull_group == 1u && ps.tail_split_shift != 0u) ? ps.tail_chunks * waves_per_cursor : 0u;
    hipLaunchKernelGGL((trace_kernel<P, C, F>), dim3((unsigned)(num_cus * use_per_cu)), dim3(kBlock), lds, stream, sc, cam, pt,
                       (const float4*)in_q, (const uint2*)in_counts, (float4*)hits, cursor, slot_L, film_n, counters);
    return hipGetLastError();
}

// confirm: the octree confirm step follows (reference-default semantics); primary rays are radiance rays either way
hipError_t launch_trace(hipStream_t stream, int num_cus, int blocks_per_cu_cap, bool primary, bool count, bool confirm, const DScene& sc, const DCamera& cam, const DPass& ps,
                        const void* in_q, const void* in_counts, void* hits, uint32_t* cursor,
                        float* slot_L, const uint32_t* film_n, DCounters* counters)
{
#define MI355RT_TRACE_ARGS stream, num_cus, blocks_per_cu_cap, sc, cam, ps, in_q, in_counts, hits, cursor, slot_L, film_n, counters
    if (primary) return count ? launch_trace_variant<true, true, true>(MI355RT_TRACE_ARGS) : launch_trace_variant<true, false, true>(MI355RT_TRACE_ARGS);
    if (confirm) return count ? launch_trace_variant<false, true, true>(MI355RT_TRACE_ARGS) : launch_trace_variant<false, false, true>(MI355RT_TRACE_ARGS);
    return count ? launch_trace_variant<false, true, false>(MI355RT_TRACE_ARGS) : launch_trace_variant<false, false, false>(MI355RT_TRACE_ARGS);
#undef MI355RT_TRACE_ARGS
}

// one wave per pixel block of the pass: the culling verdict of its chunk in the first sample group (DPass::block_culled)
__global__ __launch_bounds__(kBlock) void cull_blocks_kernel(DCamera cam, DPass ps, uint32_t nblocks, uint32_t* out)
{
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    for (uint32_t b = global_wave_id(); b < nblocks; b += nwaves) {
        const bool culled = chunk_is_culled(cam, ps, b, ps.chunk);
        if (lane_id() == 0) out[b] = culled ? 1u : 0u;
    }
}
hipError_t launch_cull_blocks(hipStream_t stream, const DCamera& cam, const DPass& ps, uint32_t nblocks, uint32_t* out)
{
    if (nblocks == 0) return hipSuccess;
    DPass plain = ps; plain.block_culled = nullptr; plain.cull_blocks = 0;
    const unsigned blocks = std::min<unsigned>((nblocks + kWavesPerBlock - 1) / kWavesPerBlock, 4096u);
    hipLaunchKernelGGL(cull_blocks_kernel, dim3(blocks), dim3(kBlock), 0, stream, cam, plain, nblocks, out);
    return hipGetLastError();
}

hipError_t launch_raster(hipStream_t stream, int num_cus, bool count, bool confirm, const DScene& sc, const DCamera& cam, const DPass& ps,
                         void* hits, uint32_t* cursor, const uint32_t* film_n, DCounters* counters)
{
    // persistent grid like the trace launch; no LDS, few registers: 8 blocks per CU
    const dim3 grid((unsigned)(num_cus * 8)), block(kBlock);
    if (confirm) { if (count) hipLaunchKernelGGL((raster_kernel<true, true>), grid, block, 0, stream, sc, cam, ps, (float4*)hits, cursor, film_n, counters);
                   else hipLaunchKernelGGL((raster_kernel<false, true>), grid, block, 0, stream, sc, cam, ps, (float4*)hits, cursor, film_n, counters); }
    else { if (count) hipLaunchKernelGGL((raster_kernel<true, false>), grid, block, 0, stream, sc, cam, ps, (float4*)hits, cursor, film_n, counters);
           else hipLaunchKernelGGL((raster_kernel<false, false>), grid, block, 0, stream, sc, cam, ps, (float4*)hits, cursor, film_n, counters); }
    return hipGetLastError();
}

hipError_t launch_trace_octree(hipStream_t stream, int num_cus, bool primary, const DScene& sc, const DCamera& cam, const DPass& ps,
                               const void* in_q, const void* in_counts, void* hits, float* slot_L, const uint32_t* film_n)
{
    unsigned blocks = (ps.nchunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned cap = (unsigned)num_cus * 4u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 grid(blocks), block(kBlock);
    if (primary) hipLaunchKernelGGL((trace_octree_kernel<true>), grid, block, 0, stream, sc, cam, ps, (const float4*)in_q, (const uint2*)in_counts, (float4*)hits, slot_L, film_n);
    else hipLaunchKernelGGL((trace_octree_kernel<false>), grid, block, 0, stream, sc, cam, ps, (const float4*)in_q, (const uint2*)in_counts, (float4*)hits, slot_L, film_n);
    return hipGetLastError();
}

hipError_t launch_shade(hipStream_t stream, int num_cus, bool primary, bool walk, const DScene& sc, const DCamera& cam, const DPass& ps, uint32_t level,
                        const void* in_q, const void* in_counts, const void* hits, void* out_q, void* out_counts, uint32_t* cursor,
                        float* slot_L, uint32_t* sample_slot, const uint32_t* film_n, DCounters* counters, bool raster)
{
    // raster (primary round only): the kernel finds the primary rays' closest hits itself, through the tile bins (cam.tile_ofs), and keeps them in LDS
    const size_t lds = (size_t)ps.list_cap * kWavesPerBlock * sizeof(uint32_t) + (raster ? (size_t)ps.chunk * kWavesPerBlock * sizeof(float4) : 0);
    unsigned blocks = (ps.nchunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned cap = (unsigned)num_cus * (primary ? (walk ? MI355RT_SHADE_PW_BLOCKS : MI355RT_SHADE_P_BLOCKS) : (walk ? MI355RT_SHADE_SW_BLOCKS : MI355RT_SHADE_S_BLOCKS));    // what the chip holds at once
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 grid(blocks), block(kBlock);
#define MI355RT_SHADE_ARGS grid, block, lds, stream, sc, cam, ps, level, (const float4*)in_q, (const uint2*)in_counts, (const float4*)hits, (float4*)out_q, (uint2*)out_counts, cursor, slot_L, sample_slot, film_n, counters
    if (primary && raster && walk) hipLaunchKernelGGL((shade_kernel<true, true, true>), MI355RT_SHADE_ARGS);
    else if (primary && raster) hipLaunchKernelGGL((shade_kernel<true, false, true>), MI355RT_SHADE_ARGS);
    else if (primary && walk) hipLaunchKernelGGL((shade_kernel<true, true>), MI355RT_SHADE_ARGS);
    else if (primary) hipLaunchKernelGGL((shade_kernel<true, false>), MI355RT_SHADE_ARGS);
    else if (walk) hipLaunchKernelGGL((shade_kernel<false, true>), MI355RT_SHADE_ARGS);
    else hipLaunchKernelGGL((shade_kernel<false, false>), MI355RT_SHADE_ARGS);
#undef MI355RT_SHADE_ARGS
    return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t stream, const DPass& ps, uint32_t width, uint32_t nlights, const float* slot_L, const uint32_t* sample_slot,
                          float* film_sum, float* film_sumsq, uint32_t* film_n, float* debug_color, uint32_t* ctrl)
{
    const uint32_t spp = ps.npix ? ps.nsamples / ps.npix : 1u;
    uint32_t lanes = spp <= 32u ? 2u : (spp <= 96u ? 4u : 8u);
    if (const char* e = getenv("MI355RT_RESOLVE_LANES")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) lanes = (uint32_t)v; }
    dim3 block(256), grid((unsigned)(((size_t)ps.npix * lanes + 255) / 256));
#define MI355RT_RESOLVE_ARGS grid, block, 0, stream, ps, width, nlights, slot_L, sample_slot, film_sum, film_sumsq, film_n, debug_color, ctrl
    switch (lanes) {
        case 1: hipLaunchKernelGGL(resolve_kernel<1>, MI355RT_RESOLVE_ARGS); break;
        case 2: hipLaunchKernelGGL(resolve_kernel<2>, MI355RT_RESOLVE_ARGS); break;
        case 4: hipLaunchKernelGGL(resolve_kernel<4>, MI355RT_RESOLVE_ARGS); break;
        case 16: hipLaunchKernelGGL(resolve_kernel<16>, MI355RT_RESOLVE_ARGS); break;
        default: hipLaunchKernelGGL(resolve_kernel<8>, MI355RT_RESOLVE_ARGS); break;
    }
#undef MI355RT_RESOLVE_ARGS
    return hipGetLastError();
}

hipError_t launch_tonemap(hipStream_t stream, const uint32_t* rows, uint32_t row_base, uint32_t nrows, uint32_t width, bool packed,
                          const float* film_sum, const uint32_t* film_n, uint32_t* out)
{
    const size_t n = (size_t)nrows * width;
    if (n == 0) return hipSuccess;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(tonemap_kernel, grid, block, 0, stream, rows, row_base, nrows, width, packed ? 1 : 0, film_sum, film_n, out);
    return hipGetLastError();
}

// ---- Film::clear (film.rs:37-41) of the rows a striped handle owns: one launch instead of three whole-film memsets ----
__global__ __launch_bounds__(256) void film_clear_rows_kernel(const uint32_t* __restrict__ rows, uint32_t nrows, uint32_t width, float* film_sum, float* film_sumsq, uint32_t* film_n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nrows * width) return;
    const size_t pixel = (size_t)rows[i / width] * width + i % width;
    film_sum[3 * pixel] = 0.0f; film_sum[3 * pixel + 1] = 0.0f; film_sum[3 * pixel + 2] = 0.0f;
    film_sumsq[3 * pixel] = 0.0f; film_sumsq[3 * pixel + 1] = 0.0f; film_sumsq[3 * pixel + 2] = 0.0f;
    film_n[pixel] = 0u;
}
// copy the film entries of `total` rows of the owned-row list (entries first, first + 1, ... cyclically) to a packed backup, or back
// (Renderer::trace_frame_additive: the rows a speculatively launched 50-row frame is about to change)
__global__ __launch_bounds__(256) void film_rows_copy_kernel(const uint32_t* __restrict__ rows, uint32_t first, uint32_t total, uint32_t nown, uint32_t width,
                                                            float* film_sum, float* film_sumsq, uint32_t* film_n, float* bk_sum, float* bk_sumsq, uint32_t* bk_n, int restore)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)total * width) return;
    const uint32_t k = (uint32_t)(i / width), x = (uint32_t)(i % width);
    const size_t pixel = (size_t)rows[(first + k) % nown] * width + x;
    if (restore) {
        film_sum[3 * pixel] = bk_sum[3 * i]; film_sum[3 * pixel + 1] = bk_sum[3 * i + 1]; film_sum[3 * pixel + 2] = bk_sum[3 * i + 2];
        film_sumsq[3 * pixel] = bk_sumsq[3 * i]; film_sumsq[3 * pixel + 1] = bk_sumsq[3 * i + 1]; film_sumsq[3 * pixel + 2] = bk_sumsq[3 * i + 2];
        film_n[pixel] = bk_n[i];
    } else {
        bk_sum[3 * i] = film_sum[3 * pixel]; bk_sum[3 * i + 1] = film_sum[3 * pixel + 1]; bk_sum[3 * i + 2] = film_sum[3 * pixel + 2];
        bk_sumsq[3 * i] = film_sumsq[3 * pixel]; bk_sumsq[3 * i + 1] = film_sumsq[3 * pixel + 1]; bk_sumsq[3 * i + 2] = film_sumsq[3 * pixel + 2];
        bk_n[i] = film_n[pixel];
    }
}
hipError_t launch_film_rows_copy(hipStream_t stream, const uint32_t* rows, uint32_t first, uint32_t total, uint32_t nown, uint32_t width,
                                 float* film_sum, float* film_sumsq, uint32_t* film_n, float* bk_sum, float* bk_sumsq, uint32_t* bk_n, bool restore)
{
    const size_t n = (size_t)total * width;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(film_rows_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows, first, total, nown, width, film_sum, film_sumsq, film_n, bk_sum, bk_sumsq, bk_n, restore ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_film_clear_rows(hipStream_t stream, const uint32_t* rows, uint32_t nrows, uint32_t width, float* film_sum, float* film_sumsq, uint32_t* film_n)
{
    const size_t n = (size_t)nrows * width;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(film_clear_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows, nrows, width, film_sum, film_sumsq, film_n);
    return hipGetLastError();
}

// ---- multi-GPU gather, last step on the root: packed stripes of every rank -> the full frame -----------------
// gathered = world slots of slot_rows * width u32; slot r holds rank r's owned rows in ascending order (rows are dealt
// in stripes of stripe_rows, stripe s to rank s % world: Renderer::init).  One thread per pixel of the frame.
__global__ __launch_bounds__(256) void place_stripes_kernel(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frame,
                                                           uint32_t width, uint32_t height, uint32_t stripe_rows, uint32_t world, uint32_t slot_rows)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)y * width);
    const uint32_t stripe = y / stripe_rows, rank = stripe % world;
    const uint32_t local_row = (stripe / world) * stripe_rows + (y - stripe * stripe_rows);
    frame[i] = gathered[((size_t)rank * slot_rows + local_row) * width + x];
}
hipError_t launch_place_stripes(hipStream_t stream, const uint32_t* gathered, uint32_t* frame, uint32_t width, uint32_t height,
                                uint32_t stripe_rows, uint32_t world, uint32_t slot_rows)
{
    const size_t n = (size_t)width * height;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(place_stripes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, gathered, frame, width, height, stripe_rows, world, slot_rows);
    return hipGetLastError();
}

// rows of LDS the fused kernel needs: the traversal stack (+ trash row) or the wave's hit list, whichever is larger
uint32_t fused_pass_lds_rows(uint32_t stack_depth, uint32_t max_level_nodes, uint32_t records_per_sample)
{
    const uint32_t a = (stack_depth ? stack_depth : 1u) + 1u;      // traversal stack + trash row
    const uint32_t b = max_level_nodes > records_per_sample ? max_level_nodes : records_per_sample;   // shade hit list / confirm record list (64-sample chunks: one row per record of a sample)
    return a > b ? a : b;
}

hipError_t launch_fused_pass(hipStream_t stream, int num_cus, bool confirm, const DScene& sc, const DCamera& cam, const DPass& ps, uint32_t max_level_nodes, uint32_t records_per_sample,
                             void* q0, void* q1, void* hits, float* slot_L, uint32_t* sample_slot,
                             float* film_sum, float* film_sumsq, uint32_t* film_n, DCounters* counters)
{
    if (ps.nchunks == 0) return hipSuccess;
    if (ps.chunk > 64u) return hipErrorInvalidValue;          // the LDS lists hold one row of 64 entries per record of a sample
    const size_t lds = (size_t)fused_pass_lds_rows(ps.stack_depth, max_level_nodes, records_per_sample) * kBlock * sizeof(int);
    unsigned blocks = (ps.nchunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned cap = (unsigned)num_cus * 8u;
    if (blocks > cap) blocks = cap;
    if (confirm) hipLaunchKernelGGL(fused_pass_kernel<true>, dim3(blocks), dim3(kBlock), lds, stream, sc, cam, ps, (float4*)q0, (float4*)q1,
                                    (float4*)hits, slot_L, sample_slot, film_sum, film_sumsq, film_n, counters);
    else hipLaunchKernelGGL(fused_pass_kernel<false>, dim3(blocks), dim3(kBlock), lds, stream, sc, cam, ps, (float4*)q0, (float4*)q1,
                            (float4*)hits, slot_L, sample_slot, film_sum, film_sumsq, film_n, counters);
    return hipGetLastError();
}

// the octree confirm step of one round (reference-default semantics), between its trace and its shade launch
hipError_t launch_confirm(hipStream_t stream, int num_cus, bool primary, bool shadow_only, const DScene& sc, const DCamera& cam, const DPass& ps,
                          const void* in_q, const void* in_counts, void* hits, uint32_t* cursor, float* slot_L, const uint32_t* film_n)
{
    const size_t lds = 0;                                  // the pending lists are static LDS (128 entries per wave)
    unsigned blocks = (ps.nchunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned cap = (unsigned)num_cus * MI355RT_CONFIRM_BLOCKS;   // several chunks per wave, so that batches fill up across chunks
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (primary) hipLaunchKernelGGL(confirm_kernel<true>, dim3(blocks), dim3(kBlock), lds, stream, sc, cam, ps, (const float4*)in_q, (const uint2*)in_counts, (float4*)hits, cursor, slot_L, film_n, 0u);
    else hipLaunchKernelGGL(confirm_kernel<false>, dim3(blocks), dim3(kBlock), lds, stream, sc, cam, ps, (const float4*)in_q, (const uint2*)in_counts, (float4*)hits, cursor, slot_L, film_n, shadow_only ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_intersect(hipStream_t stream, const DScene& sc, uint32_t stack_depth, int mode, const float* rays6, uint32_t n, bool shadow_mode,
                            float* tuv, uint32_t* prim, uint8_t* blocked)
{
    if (n == 0) return hipSuccess;
    dim3 block(kBlock), grid((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(intersect_kernel, grid, block, stack_bytes(stack_depth), stream, sc, rays6, n, shadow_mode ? 1 : 0, mode, tuv, prim, blocked);
    return hipGetLastError();
}

}  // namespace mi355rt
