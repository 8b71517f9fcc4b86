// renderer.hpp — host side of the device render path: owns the HIP buffers, the camera and the
// film of one RayTracer (raytracer/mod.rs:32-47) and schedules the wavefront passes.
#pragma once
#include <hip/hip_runtime_api.h>
#include <array>
#include <memory>
#include <string>
#include <vector>
#include "../../include/mi355rt.h"
#include "bvh.hpp"
#include "camera.hpp"
#include "device_types.hpp"
#include "scene.hpp"

namespace mi355rt {

class Renderer {
public:
    static std::unique_ptr<Renderer> create(const SceneData& scene, const mi355rt_config& cfg, std::string& err, int& code);
    ~Renderer();

    uint32_t trace_frame_additive();                                   // mod.rs:80-117
    bool render(uint32_t spp, bool wait = true);                        // wait == false: queued only (mi355rt_render_async)
    bool get_tonemapped(uint32_t* out, size_t n);                       // mod.rs:120-128
    bool tonemap_owned_rows_device(uint32_t* device_out, size_t n, hipStream_t caller_stream = nullptr);
    bool last_counts(mi355rt_ray_counts& out);
    bool film_get(float* sum, float* sumsq, uint32_t* n);
    bool set_seed(uint64_t seed);
    bool set_flags(uint32_t flags);
    bool film_clear();                                                  // film.rs:37-41
    bool intersect(const float* rays6, size_t n, float* tuv, uint32_t* prim, uint8_t* blocked);
    bool synchronize();                                                 // wait for everything queued on this handle
    // ---- multi-GPU gather of the packed u32 stripes into the root's frame.  Two transports end in the
    // same place: the in-process device group (hipMemcpyPeerAsync, csrc/group.cpp) and RCCL between processes (comm_*).
    uint32_t slot_rows() const;                                         // rows of one rank's slot: ceil(stripes / world) * stripe_rows
    uint32_t rows_of_rank(uint32_t rank) const;
    bool gather_prepare(bool root);                                     // root: world slots, others: their own slot
    uint32_t* gather_slot(uint32_t rank) { return d_gather_ ? d_gather_ + (gather_is_root_ ? (size_t)rank : 0) * slot_rows() * cfg.width : nullptr; }
    bool tonemap_to_gather_slot();                                      // own rows, packed, into the own slot (on the handle's stream)
    bool finish_gather(uint32_t* host_out, size_t n);                   // root: slots -> frame (+ copy to the host when host_out != null)
    hipStream_t stream() const { return stream_; }
    hipEvent_t gather_event() const { return ev_gather_; }
    bool comm_available();                                              // local pre-check of comm_init (no communication)
    bool comm_init(const uint8_t* id128);                               // RCCL communicator over the stripe ranks (collective)
    bool comm_gather(uint32_t root, uint32_t* host_out, size_t n);      // collective: grouped ncclSend / ncclRecv to the root, then finish_gather there
    void comm_destroy();
    uint32_t comm_ranks();                                              // ncclCommCount of the live communicator, 0 without one
    long check_guards();                                                // MI355RT_DEBUG_GUARD: corrupted guard bytes behind the pass buffers
    size_t hbm_allocated_bytes() const;                                 // device memory this handle holds (scene, film, pass buffers, gather slots)
    bool debug_gather_rate(uint32_t table_nodes, uint32_t steps, double out[3]);
    bool debug_slab(const float* inv_rays6, const float* cubes6, size_t n, uint8_t* hit, float* tmin);
    bool film_stat(bool variances, float* rgb);
    void speculation_stats(uint64_t out[2]) const { out[0] = spec_launched_; out[1] = spec_adopted_; }
    bool debug_numerics(const float* a, const float* b, size_t n, float* q, float* r, float* p);
    bool debug_sample(uint32_t pixel, uint32_t sampleno, float* color3, float* node_L, size_t nodes);

    Camera camera;
    mi355rt_config cfg;
    mi355rt_ray_counts counts{};
    std::string last_error;
    std::vector<float> table;            // 65536 x 3
    std::vector<uint32_t> owned_rows;
    Bvh bvh;
    uint32_t ntri = 0;
    uint32_t current_row = 0;
    uint32_t slices = 1;                  // concurrent frame slices of render() (1..8), MI355RT_SLICES / mi355rt_set_slices.  Round 2 ran 3: they filled the
                                          // tails of each other's trace launches; with the tail chunks handed out in parts one slice is as fast and holds a third of the memory
    bool slices_explicit = false;         // set through the API: used as given, also for small frames
    uint32_t nodes_per_sample = 1;
    uint32_t level_first[kMaxLevels + 1] = { 0 };
    std::vector<std::array<float, 6>> cull_boxes_;   // top BVH subtree boxes for primary-chunk culling
    uint32_t* d_cull_mask_ = nullptr;    // kCullGrid x kCullGrid coverage bits (device_types.hpp), rebuilt when the camera changes
    std::vector<float> mask_key_;        // the camera the mask on the device was built for (rot, origin, max_x, max_y); empty: none
    float mask_dom_[4] = { 0, 0, 0, 0 }; // its domain: x0, y0, 1 / cell width, 1 / cell height
    bool mask_valid_ = false;
    uint2* d_tile_ofs_ = nullptr; uint2* d_tile_entries_ = nullptr;     // screen-space triangle bins of the primary rays (device_types.hpp), rebuilt with the mask
    size_t tile_entries_cap_ = 0, tile_ofs_cap_ = 0;
    std::vector<float> bins_key_;        // camera + layout the bins on the device were built for
    uint32_t bins_layout_[3] = { 0, 0, 0 };   // tile_cols, tile_rg, tile_nblocks
    bool bins_valid_ = false;
    double bins_ms_ = 0.0; size_t bins_entries_ = 0;
    uint32_t oct_stats_[8] = { 0 };       // the reference's octree: nodes, inner, leaves, empty, depth, triangle refs
    double build_ms_[2] = { 0.0, 0.0 };   // build times inside create (wall): BVH (host binned SAH, or the device build), octree (SAT, host)
    double light_map_ms_ = 0.0;           // build time of the lights' depth cube maps inside create
    bool bvh_on_device_ = false;          // MI355RT_FLAG_DEVICE_LBVH and the device build served the scene
    bool wide_ = false;                   // the device holds the 4-wide tree (bvh.nodes4): this build's kernels walk it
    uint32_t traversal_rows() const { return (wide_ ? bvh.stack_need4 : bvh.max_depth) + 1u; }       // stack rows the trace loops need (kernels.hip, stack_bytes)
    double lbvh_device_ms_ = 0.0;         // its device time (kernels + sort)
    enum Mode { kModeConfirm = 0, kModeOctreeWalk = 1, kModeTrueClosest = 2 };
    Mode mode_ = kModeConfirm;            // intersector semantics, fixed at creation

private:
    Renderer() = default;
    bool init(const SceneData& scene, std::string& err, int& code);
    bool bind();
    bool fail(hipError_t e, const char* what);
    // A frame slice: an interleaved share of the owned rows with its own stream and pass buffers.  The
    // slices of one render() call run concurrently, so that the drain window at the end of one slice's
    // (persistent) trace launch is filled by the other slices' kernels.  Pixels of different slices are
    // disjoint, so the film needs no ordering between them.
    struct Slice {
        hipStream_t stream = nullptr;            // slice 0 runs on the renderer's main stream
        hipEvent_t done = nullptr;
        hipEvent_t ev_ready = nullptr, ev_traced = nullptr;   // slice stream -> trace stream -> slice stream, per round (render())
        std::vector<uint32_t> rows;              // this slice's rows (host copy of d_rows)
        uint32_t* d_rows = nullptr;
        uint32_t* d_ctrl = nullptr;              // per round: chunk cursors
        bool ctrl_clean = true;                  // the cursors are zero (creation, or the last pass's resolve kernel left them so)
        void* d_queue[2] = { nullptr, nullptr };
        uint32_t* d_chunk_counts[2] = { nullptr, nullptr };   // rays per chunk in each queue
        void* d_hits = nullptr;
        uint32_t* d_hit_prim = nullptr;          // hit records of the round being processed (16 B per queue record)
        float* d_slot_L = nullptr;
        void* d_slot_ps = nullptr;               // light-term slot -> (pixel, sample number) of its sample (uint2)
        uint32_t* d_sample_slot = nullptr;       // primary sample -> slot of its light terms (0xFFFFFFFF: the primary ray missed)
        uint32_t* d_live = nullptr;              // live-chunk lists of the pass, one per work cursor (DPass::live)
        uint32_t* d_block_culled = nullptr;      // cached culling verdicts of the pass's pixel blocks (DPass::block_culled) for the camera / layout of cull_key
        size_t block_culled_cap = 0;
        std::vector<float> cull_key;
        size_t capacity = 0;                     // samples
        size_t queue_records = 0;
        size_t count_entries = 0;                // entries of each d_chunk_counts array
        size_t bytes = 0;                        // device memory of this slice's pass buffers
        std::vector<const uint8_t*> guards;      // MI355RT_DEBUG_GUARD: the 0xA5-filled tails of the pass buffers
    };
    bool ensure_pass_capacity(Slice& sl, size_t nsamples);
    void free_pass_buffers();
    bool assign_slice_rows(uint32_t nslices);
    struct PassRun { Slice* sl = nullptr; DPass ps; DCamera cam; uint32_t rounds = 0; bool live = false; };
    bool pass_begin(PassRun& run, Slice& sl, const uint32_t* d_rows, uint32_t row0, uint32_t nrows, uint32_t spp, bool explicit_sample, uint32_t epixel, uint32_t esample, uint32_t row_wrap);
    bool pass_round(PassRun& run, uint32_t r, hipStream_t trace_stream, int trace_blocks_per_cu);
    bool pass_end(PassRun& run);
    bool run_pass(Slice& sl, const uint32_t* d_rows, uint32_t row0, uint32_t nrows, uint32_t spp, bool explicit_sample, uint32_t epixel, uint32_t esample, uint32_t row_wrap = 0xFFFFFFFFu);
    void describe_pass(DPass& ps, const Slice& sl, const uint32_t* d_rows, uint32_t row0, uint32_t row_wrap, uint32_t npix, size_t nsamples, uint32_t chunk,
                       bool explicit_sample, uint32_t epixel, uint32_t esample) const;
    bool begin_call();
    bool end_call(uint64_t primary, bool wait = true);
    bool fetch_counts(uint64_t primary, bool timed_call);
    bool queue_counts_copy();
    void mark_dirty_window(uint32_t first, uint32_t total);
    // layout + rows: the pass whose primary rays the tile bins are for and the (host copy of the) row list it walks (null: no bins needed)
    DCamera device_camera(const DPass* layout = nullptr, const std::vector<uint32_t>* rows = nullptr);
    bool refresh_tile_bins(DCamera& c, const double inv[3][3], double pad, double zmin, const DPass& ps, const std::vector<uint32_t>& rows);
    bool refresh_cull_mask(DCamera& c, const double inv[3][3], double pad, double zmin, bool build);
    bool project_triangles(const DCamera& c, const double inv[3][3], double pad, double zmin);
    struct TriRect { double x0, x1, y0, y1; };
    std::vector<TriRect> tri_rects_;     // padded screen rectangles of the triangles (BVH order) for the camera of rects_key_
    std::vector<float> rects_key_;
    bool rects_ok_ = false;
    void collect_cull_boxes();
    void build_sample_table(std::vector<float>& table4);
    template <class T> bool upload(T*& dptr, const void* src, size_t bytes);

    int num_cus_ = 0;
    hipStream_t trace_stream_ = nullptr;         // the trace launches of a multi-slice frame, round by round
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_begin_ = nullptr, ev_end_ = nullptr;
    std::vector<hipEvent_t> ev_pool_;
    size_t ev_used_ = 0;
    std::vector<uint8_t> ev_secondary_;  // per event pair: the launch traced secondary rays (rounds >= 1)
    std::vector<void*> allocs_;
    size_t alloc_bytes_ = 0;             // bytes behind allocs_ (scene, acceleration structures, film, row lists, cursors)

    DScene dscene_{};
    float* d_film_sum_ = nullptr; float* d_film_sumsq_ = nullptr; uint32_t* d_film_n_ = nullptr;
    uint32_t* d_owned_rows_ = nullptr; uint32_t* d_all_rows_ = nullptr; uint32_t* d_tmp_rows_ = nullptr;
    uint32_t* d_ldr_ = nullptr;
    uint32_t* h_ldr_ = nullptr;          // pinned host mirror of d_ldr_ (get_tonemapped_pixels)
    std::vector<uint8_t> ldr_dirty_;     // per row: film changed since the row was last tone-mapped into d_ldr_ / h_ldr_
    hipEvent_t ev_tonemap_ = nullptr, ev_gather_ = nullptr;
    uint32_t* d_gather_ = nullptr;       // gather slots (see gather_prepare)
    bool gather_is_root_ = false;
    void* comm_ = nullptr;               // ncclComm_t
    bool counts_pending_ = false;        // the last call was an asynchronous 50-row frame: counters not fetched yet
    uint64_t pending_primary_ = 0;
    bool pending_timed_ = false;         // the pending call was a whole frame: its HIP-event time is read with the counters
    DCounters* d_counters_ = nullptr;
    // Speculation of the drop-in loop (trace_frame_additive): the NEXT 50-row frame is launched right behind the one just asked for, so that the device
    // traces it while the host reads out the current one (main.rs:197-207 alternates the two calls; one frame keeps the chip busy for 0.25 ms of latency,
    // not of work).  The rows it changes are backed up first; a next call that is not the predicted one (camera moved, film cleared, anything else
    // touched) puts them back.  Read-outs of the finished frame run on read_stream_, beside the speculative launch.
    struct Speculation { bool valid = false; uint32_t row = 0, first = 0, total = 0, next_row = 0; std::vector<float> cam_key; uint64_t seed = 0; uint32_t flags = 0; } spec_;
    DCounters* d_counters_spec_ = nullptr; DCounters* h_counters_spec_ = nullptr;
    float* d_bk_sum_ = nullptr; float* d_bk_sumsq_ = nullptr; uint32_t* d_bk_n_ = nullptr;
    hipStream_t read_stream_ = nullptr;
    hipEvent_t ev_call_done_ = nullptr, ev_spec_done_ = nullptr;      // behind the kernel (and the counters' copy) of the frame last asked for / of the speculative one
    bool call_done_valid_ = false;        // ev_call_done_ marks the last 50-row frame: read-outs wait for it, not for the stream
    uint64_t spec_launched_ = 0, spec_adopted_ = 0;
    bool settle_speculation();            // put the speculative frame's rows back (if one is out)
    bool launch_fused_window(uint32_t first, uint32_t total, const DCamera& cam, DCounters* dcounters);
    DCounters* h_counters_ = nullptr;    // pinned host mirror (queue_counts_copy)
    float* d_debug_color_ = nullptr;
    static constexpr uint32_t kMaxSlices = 8;
    Slice slices_[kMaxSlices];
    uint32_t rows_assigned_for_ = 0;     // number of slices the row lists were last split into
    uint32_t active_slices_ = 1;         // slices used by the call in flight (begin_call .. end_call)
    uint32_t chunk_ = 256;               // primary samples per work chunk
    static constexpr size_t kGuardBytes = 256;
    static constexpr uint32_t kMinChunk = 16;   // smallest chunk any launcher cuts a pass into (the fused 50-row launch: 16 / 32 / 64)
    uint32_t max_level_nodes_ = 1;
    uint32_t leaf_threshold_ = 16;
    mutable uint32_t refill_primary_ = 48;   // refill threshold of the primary trace launch (describe_pass)
    bool alloc_failed_ = false;          // the last ensure_pass_capacity failure was an out-of-memory
    uint32_t records_per_sample_ = 1;
    uint32_t nlights_ = 0;
    uint64_t launches_ = 0;
};

}  // namespace mi355rt
