// device_types.hpp — plain structs shared by the host renderer and the HIP kernels.
#pragma once
#include <cstdint>
#include <hip/hip_vector_types.h>

namespace mi355rt {

constexpr uint32_t kNumSamples = 65536;     // sample_generator.rs:5-7
constexpr uint32_t kSampleMax = 65535;
constexpr uint32_t kMaxRecursions = 3;
constexpr uint32_t kMaxLevels = kMaxRecursions + 1;

// Ray-queue record, 32 bytes = 2 x float4, stored as two planes (q0 of every record, then q1) so that a wave's 64
// consecutive records are 1 KiB contiguous per plane.  Layout of the two kinds: kernels.hip ("ray records").
// (Round 1-2a: 48 bytes; the shadow ray's origin and direction are now rebuilt from the hit point by the kernel that
// traces it, and (pixel, sample number) moved to a per-slot side array: profiles/r02_notes.md.)
constexpr uint32_t kRayRecordBytes = 32;
constexpr uint32_t kShards = 64;            // copies of every contended device word (work cursors, counters)
constexpr uint32_t kMaxRounds = kMaxRecursions + 2;
constexpr uint32_t kCullRects = 16;
constexpr uint32_t kMaxCursors = 64;
constexpr uint32_t kCtrlWordsPerRound = kMaxCursors * 16384;   // work cursors of one round, 64 KiB apart
constexpr uint32_t kShadeCursorOffset = 128;                   // the shade launch of a round: 512 B after
constexpr uint32_t kConfirmCursorOffset = 64;                  // the confirm launch of a round: its cursors sit 256 B after the trace launch's
constexpr uint32_t kLiveCountOffset = 192;                     // round 0's cursor words + 768 B: how many chunks cursor k's live list holds (DPass::live)

struct DMaterial { float r, g, b; uint32_t kind_tex; };       // kind_tex: bit 31 = texture, low bits = texture id
struct DLight { float px, py, pz, cr, cg, cb;
                float tail2;   // (distance to the nearest triangle, padded)^2: a shadow ray runs 1 % of its length PAST the light (t in (0.99, 1)); nothing is there while 1e-4 |l|^2 < tail2
                float pad; };
struct DTexture { uint32_t width, height; uint64_t offset; }; // offset in texels into the texel pool

struct DScene {
    const void* nodes;        // BvhNode[]
    const void* tris;         // BvhTri[] (leaf order)
    const void* normals;      // float4[ntri]: (n.xyz, geometry index bits), original triangle order
    const DMaterial* materials;
    const DLight* lights;
    const DTexture* textures;
    const float* texels;      // RGB f32 pool
    const void* table;        // float4[65536] unit vectors
    const void* oct_nodes;    // OctNodeFlat[] (reference-exact intersector only, else null)
    const uint32_t* oct_leaf_tris;
    const void* prim_tris;    // BvhTri[] in ORIGINAL triangle order (reference-exact intersector only)
    const int2* oct_info;     // per octree node, what the confirm walk needs in ONE 8-byte load: x = first child (>= 0) or ~tri_first (leaf), y = leaf triangle count
    const uint32_t* tri_home; // per triangle: the one octree leaf that lists it, or 0xFFFFFFFF when several do
    const float* light_maps;  // per light a depth cube map [6][res][res]: lower bound of the squared distance from the light to anything seen in the texel's directions (lightmap.hpp); null: none
    uint32_t light_map_res;   // texels per face edge (0: no maps)
    int32_t root;
    uint32_t nlights;
    uint32_t ntri;
    uint32_t oct_single_leaf; // the reference's octree is one leaf (scene of <= triangles_per_leaf triangles): the trace kernels settle its semantics themselves
    float oct_root[6];        // root cube of the octree (scene extents): min.xyz, max.xyz
};

struct DCamera {
    float rot[16];            // rotation_matrix, camera.rs:92-95
    float origin[3];          // orientation_matrix * (0,0,0,1)
    float max_x, max_y;
    uint32_t width, height;
    // screen-space bounds of the (padded) scene box in the camera's (dir_x, dir_y) plane: primary-ray
    // chunks whose pixel footprint lies outside cannot hit anything and are skipped (cull_valid == 0: off)
    uint32_t cull_valid;      // number of rectangles below (0: culling off)
    float cull_rect[kCullRects][4];   // x0, x1, y0, y1 of the top BVH subtrees
    // second stage (round 3): a kCullGrid x kCullGrid bit mask over the bounding rectangle of those rectangles — a cell's bit is set iff the (padded)
    // screen rectangle of some triangle touches it.  A chunk whose footprint touches no set cell cannot hit anything.  null: first stage only.
    const uint32_t* cull_mask;
    float mask_x0, mask_y0, mask_inv_cx, mask_inv_cy;      // cell (i, j) = floor((dir_x - mask_x0) * mask_inv_cx), floor((dir_y - mask_y0) * mask_inv_cy)
    // Screen-space triangle bins for the primary rays (round 3; kernels.hip, raster_kernel): the 64 samples a wave takes together are a tile of
    // tile_cols columns x tile_rg rows; tile (row / tile_rg) * tile_nblocks + column / tile_cols lists every triangle a primary ray of the tile can
    // hit, nearest first.  null: no bins (the primary rays walk the BVH).
    const uint2* tile_ofs;        // per tile: x = first entry, y = number of entries
    const uint2* tile_entries;    // x = triangle (index into DScene::tris), y = float bits of a lower bound of its distance from the camera
    uint32_t tile_cols, tile_rg, tile_nblocks;
};
constexpr uint32_t kCullGrid = 1024;                       // cells per axis (32 words per row, 128 KiB)

struct DPass {
    // primary-sample enumeration: thread i -> s = i / npix, p = i % npix,
    // row = rows[row0 + p / width], x = p % width, pixel = row * width + x
    const uint32_t* rows;
    size_t qstride;           // records per queue: a queue is two float4 planes q0[], q1[] (structure of arrays; layout: kernels.hip)
    uint2* slot_ps;           // per light-term slot: (pixel, sample number) of its sample
    uint32_t* hit_prim;       // per radiance record: triangle hit or 0xFFFFFFFF (the 16-byte hit record is written for hits only)
    uint32_t row0;
    uint32_t row_wrap;        // entries of the cyclic row list (50-row frames: rows[(row0 + i) % row_wrap]); 0xFFFFFFFF: no wrap
    uint32_t npix;            // pixels in this pass (rows_in_pass * width)
    uint32_t row_group, row_group_shift;   // rows per pixel-tile group of the pass order (a power of two <= 8) and its log2
    uint32_t nsamples;        // npix * samples per pixel in this pass
    uint32_t spp;             // samples per pixel in this pass
    uint32_t sample_group;    // consecutive samples of a pixel kept together in the pass order (kernels.hip, sample_of); divides spp
    uint32_t seed;
    uint32_t flags;
    uint32_t recursions, spread;
    uint32_t nodes_per_sample;
    uint32_t nslots;          // light-term slots of the pass (nchunks * chunk): the stride between the planes of slot_L
    uint32_t level_first[kMaxLevels + 1];
    uint32_t use_explicit, explicit_pixel, explicit_sampleno;
    uint32_t chunk;           // primary samples per chunk
    uint32_t nchunks;         // ceil(nsamples / chunk)
    uint32_t region;          // queue records reserved per chunk = chunk * worst-case records per sample
    uint32_t stack_depth;     // traversal stack rows in LDS (BVH max depth + 1)
    uint32_t leaf_threshold;  // trace kernel: run the triangle code once this many lanes wait at a leaf
    uint32_t refill_threshold; // trace kernel: refill idle lanes once this many are idle
    uint32_t ncursors;        // trace kernel: number of work cursors (pull mode 4)
    uint32_t pull_group;      // pull mode 4: consecutive chunks handed out per atomic
    uint32_t pull_mode;       // work distribution: 4 = cursors (default), 2 = static striding, 0 = one cursor
    uint32_t list_cap;        // shade kernel: LDS hit-list entries per wave (max radiance rays per chunk)
    uint32_t tail_chunks;     // trace kernel: the last tail_chunks chunks of every cursor's sequence are handed out in parts (host: chunks per wave; the launcher scales it by the waves per cursor)
    uint32_t tail_split_shift; // log2 of the parts a tail chunk is handed out in (0: whole chunks everywhere)
    // Live chunks (round 3).  Most chunks of a pass hold no ray after the primary round (thai2: 70 % — culled, or nothing hit), and every later launch
    // paid a pull, a count load and a loop set-up for each of them (1.35 ms per frame with every chunk empty).  The primary shade launch appends
    // every chunk it leaves rays in to the list of cursor (chunk % ncursors): live[k * live_cap ...], length in live_count[k * 16384 + kLiveCountOffset];
    // the launches of the later rounds hand out list entries instead of chunk numbers (pull_chunk).  null: every launch walks all chunks.
    // Cached culling verdicts (round 3): whether a chunk is culled depends on its pixels, not on its sample numbers, and a pass whose sample groups are
    // whole numbers of chunks repeats the same pixel blocks in every group.  block_culled[chunk % cull_blocks] != 0: the chunk's primary samples all miss
    // (chunk_is_culled, evaluated once per camera and layout by cull_blocks_kernel instead of by every launch for every chunk); the primary shade launch
    // then writes nothing for such a chunk and the resolve launch reads nothing.  null: every launch evaluates chunk_is_culled itself.
    const uint32_t* block_culled;
    uint32_t cull_blocks;
    uint32_t* live;
    uint32_t* live_count;
    uint32_t live_cap;
};

struct DCounters {            // one set per render call, zeroed at its start
    unsigned long long bounce, shadow, primary_hits, nodes_visited, tris_tested;
    unsigned long long inner_execs, leaf_execs;   // COUNT mode: wave-level executions of the inner / leaf section
    unsigned int overflow;
    unsigned int pad;
    unsigned long long primary_culled;            // primary samples in chunks the frustum culling skipped (never traced)
    unsigned long long shadow_skipped;            // shadow rays the light's depth map proved unoccluded (never traced; counted in `shadow` by the host)
    unsigned long long t_first_end, t_last_end, t_sum_end, t_start, n_waves;   // COUNT mode: wave end times (s_memrealtime ticks, 100 MHz)
    unsigned long long lanes_inner, lanes_leaf, lanes_done, lane_samples;   // COUNT mode: where the lanes are at every loop iteration (summed lane counts; samples = iterations)
    unsigned long long t_sum_cycles, t_sum_real;                                            // COUNT mode: summed s_memtime ticks (shader cycles) of the waves, against t_sum_end in s_memrealtime ticks
    unsigned long long visits_below[6];                                          // COUNT mode: inner-node visits with node index < 64, 128, 256, 512, 1024, 2048 (what an LDS copy of the first N nodes would serve)
    unsigned long long refills, refill_passes, refill_rays;                    // COUNT mode: refill sections entered, passes through the assignment code, rays handed out
};

}  // namespace mi355rt
