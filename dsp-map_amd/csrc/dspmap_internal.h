// dspmap_internal.h -- the handle behind dspmap_t and helpers shared by dspmap_api.hip / dspmap_mgpu.hip
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/dspmap.h"
#include "dspmap_kernels.h"
#include "velocity_estimator.h"

#define DSPMAP_PTS_RING 4  // pinned cloud staging buffers in rotation
#define DSPMAP_RING 1024   // slots of the pinned frame-parameter ring (power of two)
#define DSPMAP_XQ_LIST 65536   // workgroups of the first birth kernel the estimator queue's deferral list holds (DevState::xq; 16 birth sources each)
#define DSPMAP_CLOUD_RING 64   // slots of the pinned, device-mapped CLOUD ring of the host-pointer update() (divides DSPMAP_RING)
struct dspmap {
    dspmap_config cfg;
    MapDims d;
    FilterParams fp;
    DevState s;
    KernelScratch k;
    bool device_ready = false;
    bool own_stream = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    unsigned frame_no = 0;           // replayed frames so far (every 32nd one is timed)
    int device = -1;
    std::string err;
    // parameters
    float p_stddev = 0.2f, v_stddev = 0.1f;  // :154-155
    float voxel_filter_res = 0.15f;          // :132
    int use_vel_est = 0;             // DSPMAP_P_VELOCITY_ESTIMATOR: 0 off, 1 host stage (velocity_estimator.cpp), 2 device (dspmap_velest.hip)
    VelEst ve = {};
    int ve_last_at = 0;              // where clusters_feature_vector_dynamic_last (:1401) lives: 0 nowhere yet, 1 host estimator, 2 device
    bool regen_tables = false;
    bool div_forced_off = false;     // DSPMAP_P_FAST_DIVISION = 0
    bool nb_frozen = false;                  // function statics of the birth stage (:808-811)
    // tables (host copies kept until upload)
    std::vector<float> h_ptab, h_vtab;
    std::vector<int> h_rtab;
    bool tables_injected = false, rtab_injected = false;
    int pend_cursor[3] = {0, 0, 0};
    // function statics of update() (:187-190)
    bool have_last = false;
    float last_p[3] = {0, 0, 0};
    double last_stamp = 0.0;
    float cur_pos[3] = {0, 0, 0};
    float quat[4] = {1, 0, 0, 0};
    float dt_last = 0.f;
    float update_time = 0.f;         // :634 (fp32 accumulation like the reference)
    int update_counter = 0;          // :635
    // capacities
    int pt_cap = 0, birth_cap = 0;
    float* pts_dev = nullptr;        // staging for host-fed clouds
    // pinned staging of host-fed clouds: a RING of buffers, each guarded by an event recorded behind its copy -- update()
    // returns while the frame (and the H2D copy in front of it) is still queued, and the caller may refill its cloud and
    // call again at once (the parameter ring lets the host run many frames ahead); pts_pin = the slot in use
    float* pts_pin = nullptr; int pts_pin_cap = 0;
    float* pts_ring[DSPMAP_PTS_RING] = {}; int pts_ring_cap[DSPMAP_PTS_RING] = {};
    hipEvent_t pts_ring_ev[DSPMAP_PTS_RING] = {}; bool pts_ring_busy[DSPMAP_PTS_RING] = {};
    unsigned pts_ring_pos = 0;
    // host-pointer update() as ONE graph launch: the caller's cloud is copied into a slot of a pinned, device-MAPPED ring and the
    // frame's first kernel (k_obs_points) reads the points over the bus -- no H2D copy node, no event record in front of the graph.
    // Slot (ring_head % DSPMAP_CLOUD_RING); the frame that read it last has published its ring position in hint_host[2]
    // (k_predict, i.e. after every workgroup of k_obs_points is done with the slot) before the host refills it.
    float* cring_host = nullptr; const float* cring_dev = nullptr; int cring_cap = 0;   // cring_cap: points per slot
    const float* host_cloud = nullptr; int host_cloud_stride = 0;                       // set by dspmap_update for the frame being queued
    BirthSrc* birth_pin = nullptr; int birth_pin_cap = 0;
    hipEvent_t birth_ev = nullptr; bool birth_ev_set = false;   // behind the last copy out of birth_pin
    // birth cloud supplied by the caller (estimator off) / produced by the estimator
    std::vector<dspmap_vpoint> h_birth;
    bool h_birth_valid = false;
    int last_n_birth = 0;            // grid bound of the last frame's birth launches (>= the number of birth sources)
    int static_hi = 0;               // largest synthesised birth cloud so far: a frame with an empty view re-uses the last
                                     // non-empty one's cloud, which may be longer than the frame's own point count
    bool last_birth_static = false;
    int vz_frames = 0;
    bool nb_dirty = false;           // newborn bits may be set outside a frame (pre-fill, import of flag-15 records, a birth stage
                                     // that no resampling followed): the next birth stage snapshots them (KernelScratch::nbsnap)
    u64* nbsnap_buf = nullptr;
    float4* res_true = nullptr;      // cube storage: voxels_objects_number[v][0..3] in the reference's voxel order, filled on demand (dspmap_get_results / dspmap_results_device)
    int last_n_points = 0;
    VelocityEstimator vel;
    // per-frame parameter block (host copy; pushed to s.fpar with one H2D copy per frame)
    FrameParams hp = {};
    // HIP graph of the device-resident frame (dspmap_update_device)
    bool use_graph = true;
    bool direct_ring = false;        // DSPMAP_P_USE_GRAPH = 2: plain launches, the frame's parameter block through the pinned ring like a replayed frame's
    bool est_queue = true;           // DSPMAP_P_ESTIMATOR_QUEUE: the device estimator's kernels on a queue of their own, tied to the captured frame through xq_dev
    unsigned long long api_seq = 0;  // entry points called on this handle (READY; the harmless ones take themselves off again: BENIGN)
    unsigned long long xq_chain_api = ~0ull;   // api_seq of the last frame whose estimator ran on its own queue: when the next such frame is the very
                                     // next call, its estimator waits for that frame's "birth stage has ended" word (xq_last_seq) instead of an event
    int xq_last_seq = 0;
    bool xq_break = false;           // this call has queued work on the handle's stream that the frame's estimator depends on (a staged cloud, the
                                     // estimator's state handed over from the host): its kernels are ordered behind the stream with an event
    bool xq_test_break = false;      // DSPMAP_XQ_TEST_BREAK (test hook, read at dspmap_create): every frame's estimator is ordered behind the stream with an event
    int xq_force = 0;                // DSPMAP_XQ_FORCE (test hook, read at dspmap_create): 1 "shared" = every candidate stream counts as sharing the main stream's
                                     // hardware queue, 2 "apart" = fail instead of falling back when none is apart
    bool xq_shared = false;          // no stream apart from the main stream's hardware queue was found (ensure_estimator_stream): frames keep the forked branch
    bool xq_failed = false;          // a cross-queue wait gave up (dspmap_check_estimator_queue): frames keep the forked branch
    int est_path = 0;                // the last device-estimator frame: 1 on its own stream, 2 forked branch because the stream would share the queue,
                                     // 3 forked branch (switched off / the map splits its placement / after a give-up); 0 none yet
    int xq_test_delay_us = 0;        // DSPMAP_XQ_TEST_DELAY_US (test hook): every third frame's estimator is held back this long, so that the
                                     // frame's first birth kernel finds its word missing and takes the deferral path
    long long xq_frames = 0;         // frames whose estimator ran that way (dspmap_debug_estimator_queue)
    int* xq_dev = nullptr;           // DevState::xq of the frames that use it (m->s.xq stays null: stages and sharded frames never wait on it)
    bool host_direct = true;         // DSPMAP_P_HOST_CLOUD_DIRECT: dspmap_update feeds the captured frame through the mapped cloud ring
    bool fut_clear_pending = false;   // clearOccupancyMapPrediction is lazy: done by the next frame's k_predict, or by the next reader
    hipStream_t stream2 = nullptr;   // fork/join branch inside the captured frame
    hipStream_t stream4 = nullptr;   // the bulk branch of a two-branch frame (DSPMAP_P_FRAME_BRANCHES)
    hipEvent_t ev_br[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // its fork, "predict(P) ended", "predict(not P) ended", its join, "place(not Q) ended"
    int tiling_req = -1;             // DSPMAP_P_TILING: -1 by size (derive_dims), 0 runs of 64 voxel indices, 1 cubes of 4 x 4 x 4
    int frame_branches = 0;          // DSPMAP_P_FRAME_BRANCHES: 0 never (default: measured slower), -1 the maps that would split their placement (cube storage), 1 whenever possible
    long long branch_frames = 0;     // frames that ran as two branches (dspmap_debug_frame_branches)
    bool branch_pending = false;
    float ptab_max = 0.f;            // largest |value| of the position table (FrameParams::birth_reach)
    hipStream_t stream3 = nullptr;   // the estimator's own stream (DSPMAP_P_ESTIMATOR_QUEUE): created at first use, tested not to share the main stream's
                                     // hardware queue (ensure_estimator_stream); plain launches only, never captured
    hipStream_t stream3_for = nullptr;   // the main stream it was paired with (dspmap_set_stream may change that one)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr;
    int n_cu = 256;
    // pinned parameter ring of the captured frames: slot (ring_head % DSPMAP_RING) is written by the host, read by the
    // frame's first kernel; ring_ev[q] marks the end of the last frame that used quarter q

    bool frame_ring = false;         // the frame being enqueued reads its parameters from the ring
    FrameParams* ring_host = nullptr;
    const FrameParams* ring_dev = nullptr;
    unsigned ring_head = 0;
    hipEvent_t ring_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ring_ev_set[4] = {false, false, false, false};
    volatile int* hint_host = nullptr;   // written by the device at every frame's start: 1-in-64 sample count of the non-empty tiles
    bool ro_kernel = false;          // small maps: launch k_rollout (many tiles hold hundreds of moving particles) instead of the inline rollout
    int ro_force = -1;               // DSPMAP_P_ROLLOUT_INLINE: -1 from the hint, 0 / 1 forced
    bool sparse_mode = false;        // launch k_predict's SPARSE variant (dspmap_pick_sweep_mode: from the hint, with hysteresis)
    int sparse_force = -1;           // DSPMAP_P_SPARSE_SWEEP: -1 from the hint, 0 / 1 forced
    bool tile_bitmaps = true;        // DSPMAP_P_TILE_BITMAPS: the sweeps of a sparse unsharded map's whole frames find their empty tiles in bitmaps (DevState::vis_bits)
    int resample_split = 0;          // DSPMAP_P_RESAMPLE_SPLIT: the tiles no newborn can reach are resampled on the side stream, beside the weight update and the births
    bool rsplit_enq = false;         // the frame enqueue_frame queued / captured last does so
    bool graph_rsplit[2] = {false, false};   // ... per captured graph
    long long rsplit_frames = 0;     // frames that ran that way (dspmap_debug_resample_split_frames)
    int side_fork = 1, side_wg = 3;  // split placement: where the side launch leaves the main chain / its workgroups per CU (DSPMAP_P_SIDE_PLACEMENT)
    int place_split_tiles = 8192;    // maps with at least this many tiles place the arrivals of the tiles outside the field of view
                                     // on the side stream, beside the pair kernels (DSPMAP_P_PLACE_SPLIT_TILES)
    int resample_wg_tiles = 8192;    // one-word maps with fewer tiles (and sparse ones of any size) run the four-waves-per-tile resampler (DSPMAP_P_RESAMPLE_WG_TILES)
    int last_resample_variant = 0;   // resample_variant() of the last frame / stage (dspmap_debug_rollout_paths)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};   // the captured frame, one per sweep direction (LaunchCtx::sweep_rev is a kernel argument)
    unsigned long long graph_key[2] = {~0ull, ~0ull};
    int sweep_alt = 2;               // DSPMAP_P_SWEEP_ALTERNATE: 2 (default) k_predict up, k_place down, k_resample down; 0 k_resample up; 1 all three flip
                                     // from frame to frame; -1 flip on maps of >= 4096 tiles
    unsigned frame_parity = 0;       // toggled by every prediction
    unsigned graph_epoch = 0;   // bumped whenever a baked-in kernel argument (pointer / parameter) changes
    // multi-GPU split-phase state
    float cull_sigmas = 9.f;           // DSPMAP_P_PAIR_CULL_SIGMAS
    bool mgpu_bound = false;
    bool mgpu_self_bound = false;      // the C++ driver (dspmap_dist.hip) works on the library's own Ck / n_static buffers
    struct dspmap_dist* dist = nullptr;
    bool mgpu_all_static = false;      // every birth source of the frame carries a zero-velocity tag: no velocity / rand() draws (:877-903)
    bool mgpu_birth_early = false;     // rank + children already queued by dspmap_mgpu_export_both
    bool mgpu_interior_done = false;   // dspmap_mgpu_place_interior placed the tiles [mgpu_tile_lo, mgpu_tile_hi)
    int mgpu_tile_lo = 0, mgpu_tile_hi = 0;
    bool mgpu_place_pending = false;   // k_predict ran, k_place waits for the imports
    bool mgpu_exact_lists = false;     // this frame selects the pyramid lists' cut over all ranks (dspmap_dist.hip)
    bool mgpu_split = false, mgpu_placed = false;   // between dspmap_mgpu_place_phase and dspmap_mgpu_ck_phase
    unsigned state_epoch = 0;          // bumped whenever particles are written outside a frame (seed / import / clear / checkpoint)
    bool mgpu_est_side = false;        // this frame's estimator (and the newborn children behind it) run on the side stream, beside the prediction
    bool mgpu_side_pending = false;    // the placement of the tiles without a view runs on the side stream (joined before the birth split)
    int vz_frames_at_begin = 0;
    int mgpu_nstatic_cap = 0;
    int* mgpu_count = nullptr;
    BirthSrc* mgpu_birth = nullptr;
    int last_exp[2] = {0, 0};   // particles exported down / up in the last frame
    // cloud pre-processing scratch (dspmap_preprocess.hip)
    void* pp_box = nullptr;
    void* pp_acc = nullptr;
    int* pp_blk = nullptr;
    size_t pp_cells_cap = 0;
    // per-stage profiling
    bool prof = false;
    hipEvent_t pev[DSPMAP_N_STAGES + 1] = {};
    double stage_ms[DSPMAP_N_STAGES] = {};
    int prof_frames = 0;
    bool prof_pending = false;
    float event_overhead_ms = 0.f;   // calibrated by dspmap_set_profiling(1): what an event bracket adds to the one kernel inside it
};

int dspmap_fail(dspmap* m, int code, const char* fmt, ...);
void dspmap_prof_mark(dspmap* m, int i);
void dspmap_prof_collect(dspmap* m);
LaunchCtx dspmap_ctx_of(dspmap* m);
void dspmap_mgpu_birth_early(dspmap* m, const LaunchCtx& c);   // the newborn children of a split-phase frame, on the stream the estimator ran on
void dspmap_resample(dspmap* m, const LaunchCtx& c);   // launch_resample + bookkeeping of the variant it ran
int dspmap_gate_and_delta(dspmap* m, const float pos[3], double stamp, const float q[4], float dp[3], float* dt);
int dspmap_check_estimator_queue(dspmap* m);   // first thing in every frame entry point: fails once if an earlier frame's cross-queue wait gave up
void dspmap_freeze_birth_statics(dspmap* m);
int dspmap_ensure_point_cap(dspmap* m, int n);
int dspmap_push_frame_params(dspmap* m);
void dspmap_flush_future_clear(dspmap* m);   // m->hp -> device
int dspmap_mark_nb_dirty(dspmap* m);
void dspmap_dist_free(dspmap* m);
const FrameParams* dspmap_ring_push(dspmap* m);   // the frame's parameter block into the pinned ring (direct launches); nullptr: no ring
void dspmap_ring_pushed(dspmap* m);               // after the launches that read the slot were queued
int dspmap_pts_slot_acquire(dspmap* m, int n);   // next pinned staging slot (waits for the copy that last used it) -> m->pts_pin
int dspmap_pts_slot_release(dspmap* m);          // after queueing the copy that reads / writes m->pts_pin
int dspmap_stage_points(dspmap* m, int n, int stride, const float* pts);   // host cloud -> m->pts_dev (pinned staging, async copy)
int dspmap_begin_cloud(dspmap* m, int n_points, bool static_birth);
int dspmap_upload_birth(dspmap* m, const dspmap_vpoint* pts, int n);   // host cloud -> DevState::birth (pinned staging, async copy)
int dspmap_ve_state_to_host(dspmap* m);     // clusters_feature_vector_dynamic_last (:1401) to the implementation that runs next
int dspmap_ve_state_to_device(dspmap* m);
int dspmap_mgpu_place_phase(dspmap* m);   // dspmap_mgpu_ck_partial in two halves: placement of the voxel-changing particles ...
int dspmap_mgpu_ck_phase(dspmap* m);      // ... list preparation (with DevState::pyr_kstar, if set) + the Ck pass   // bumps the frame epoch; returns the birth grid bound

#define HIPCHK(m, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return dspmap_fail((m), DSPMAP_E_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// entry points of the sharded frame (Z-slabs): their kernels take the slab's voxels in index order (layers are contiguous runs of it).  A whole-map
// handle that would pick the cube storage switches back while its device state does not exist yet; afterwards the call fails
int dspmap_need_index_order(dspmap* m);
#define INDEX_ORDER(m) do { if ((m) && (m)->d.tiling) { const int rc_io_ = dspmap_need_index_order(m); if (rc_io_ != DSPMAP_OK) return rc_io_; } } while (0)
#define BENIGN(m) (--(m)->api_seq)   /* (after READY) this entry point queues nothing that the velocity estimator's kernels read or write */
#define READY(m)                                       \
    do {                                               \
        if (!(m)) return DSPMAP_E_ARG;                 \
        ++(m)->api_seq;                                \
        if (!(m)->device_ready) {                      \
            int rc_ = dspmap_init_device(m);           \
            if (rc_ != DSPMAP_OK) return rc_;          \
        } else if ((m)->device >= 0) {                 \
            (void)hipSetDevice((m)->device);           \
        }                                              \
    } while (0)
