// kernels.hpp — host-callable launchers of kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include "device_types.hpp"

namespace mi355rt {

hipError_t launch_trace(hipStream_t stream, int num_cus, int blocks_per_cu_cap, bool primary, bool count, bool confirm, const DScene& sc, const DCamera& cam, const DPass& ps,
                        const void* in_q, const void* in_counts, void* hits, uint32_t* cursor,
                        float* slot_L, const uint32_t* film_n, DCounters* counters);
// the primary rays' closest hits through the screen-space triangle bins (cam.tile_ofs != null); same outputs as the primary trace launch
hipError_t launch_raster(hipStream_t stream, int num_cus, bool count, bool confirm, const DScene& sc, const DCamera& cam, const DPass& ps,
                         void* hits, uint32_t* cursor, const uint32_t* film_n, DCounters* counters);
// culling verdicts of the pass's pixel blocks (DPass::block_culled): out[b] = chunk b (first sample group) is culled
hipError_t launch_cull_blocks(hipStream_t stream, const DCamera& cam, const DPass& ps, uint32_t nblocks, uint32_t* out);
hipError_t launch_trace_octree(hipStream_t stream, int num_cus, bool primary, const DScene& sc, const DCamera& cam, const DPass& ps,
                               const void* in_q, const void* in_counts, void* hits, float* slot_L, const uint32_t* film_n);
hipError_t launch_shade(hipStream_t stream, int num_cus, bool primary, bool walk, const DScene& sc, const DCamera& cam, const DPass& ps, uint32_t level,
                        const void* in_q, const void* in_counts, const void* hits, void* out_q, void* out_counts, uint32_t* cursor,
                        float* slot_L, uint32_t* sample_slot, const uint32_t* film_n, DCounters* counters, bool raster = false);   // raster: primary round, hits through the tile bins inside the launch
hipError_t launch_resolve(hipStream_t stream, const DPass& ps, uint32_t width, uint32_t nlights, const float* slot_L, const uint32_t* sample_slot,
                          float* film_sum, float* film_sumsq, uint32_t* film_n, float* debug_color, uint32_t* ctrl);   // ctrl != null: zero the pass's work cursors on the way out
// one 50-row frame (1 sample per pixel) in a single launch: every wave takes a 64-sample chunk through all rounds
bool kernels_walk_wide_nodes();      // this build's trace loops read BvhNode4 slots (MI355RT_WIDE), not BvhNode
uint32_t fused_pass_lds_rows(uint32_t stack_depth, uint32_t max_level_nodes, uint32_t records_per_sample);
// reference-default semantics: true closest hits of a round -> the octree intersector's answers (+ the shadow predicate)
hipError_t launch_confirm(hipStream_t stream, int num_cus, bool primary, bool shadow_only, const DScene& sc, const DCamera& cam, const DPass& ps,
                          const void* in_q, const void* in_counts, void* hits, uint32_t* cursor, float* slot_L, const uint32_t* film_n);
hipError_t launch_fused_pass(hipStream_t stream, int num_cus, bool confirm, const DScene& sc, const DCamera& cam, const DPass& ps, uint32_t max_level_nodes, uint32_t records_per_sample,
                             void* q0, void* q1, void* hits, float* slot_L, uint32_t* sample_slot,
                             float* film_sum, float* film_sumsq, uint32_t* film_n, DCounters* counters);
// rows == null: the contiguous rows row_base .. row_base + nrows - 1
hipError_t launch_tonemap(hipStream_t stream, const uint32_t* rows, uint32_t row_base, uint32_t nrows, uint32_t width, bool packed,
                          const float* film_sum, const uint32_t* film_n, uint32_t* out);
hipError_t launch_intersect(hipStream_t stream, const DScene& sc, uint32_t stack_depth, int mode, const float* rays6, uint32_t n, bool shadow_mode,
                            float* tuv, uint32_t* prim, uint8_t* blocked);
// multi-GPU gather on the root: world slots of slot_rows packed rows each -> full frame
hipError_t launch_place_stripes(hipStream_t stream, const uint32_t* gathered, uint32_t* frame, uint32_t width, uint32_t height,
                                uint32_t stripe_rows, uint32_t world, uint32_t slot_rows);
// Film::clear restricted to the listed rows (a striped handle's own rows)
hipError_t launch_film_clear_rows(hipStream_t stream, const uint32_t* rows, uint32_t nrows, uint32_t width, float* film_sum, float* film_sumsq, uint32_t* film_n);
// film entries of `total` rows of the owned-row list (cyclic from entry `first`) -> packed backup, or back
hipError_t launch_film_rows_copy(hipStream_t stream, const uint32_t* rows, uint32_t first, uint32_t total, uint32_t nown, uint32_t width,
                                 float* film_sum, float* film_sumsq, uint32_t* film_n, float* bk_sum, float* bk_sumsq, uint32_t* bk_n, bool restore);
hipError_t launch_slab(hipStream_t stream, const float* inv_rays6, const float* cubes6, uint32_t n, uint8_t* hit, float* tmin);
hipError_t launch_film_stat(hipStream_t stream, bool variances, size_t npix, const float* film_sum, const float* film_sumsq, const uint32_t* film_n, float* out);
// the gather microbenchmark behind bench.py's roofline: num_cus * 8 blocks walk `steps` random nodes of `table` each
hipError_t launch_gather_rate(hipStream_t stream, int num_cus, const void* table, uint32_t nnodes, uint32_t steps, uint32_t* sink);
hipError_t launch_numerics(hipStream_t stream, const float* a, const float* b, uint32_t n, float* q, float* r, float* p);

}  // namespace mi355rt
