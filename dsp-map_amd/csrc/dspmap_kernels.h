// dspmap_kernels.h -- launchers of the gfx950 kernels (host-callable).
#pragma once
#include <hip/hip_runtime.h>
#include "dspmap_types.h"

// scratch owned by the runtime
struct KernelScratch {
    float4* mv_rec;     // [ntiles][64*slots][2] per-source-tile staging of k_predict (movers up, in-FOV stayers down)
    unsigned long long* vz_q;       // [v_loc*mw] (with vz0) slots that draw velocity noise in their first prediction
    unsigned long long* omask;      // [v_loc*mw] occupancy (mask | nbmask) before the frame's prediction (k_predict -> k_place)
    float4* in_rec;     // [ntiles][64*slots][2] per-destination-tile inbox of movers (k_predict tail -> k_place)
    int* in_cnt;        // [ntiles] inbox fill; zeroed again by k_place
    unsigned* tile_bits;  // [3][(ntiles + 63) / 64 * 2] the tile bitmaps (DevState::vis_bits / pred_bits / arr_bits point into it while they are in use)
    u64* expmask;       // [v_loc*mw] particles that left the slab (multi-GPU), or nullptr
    int* part_predict;  // [ntiles*4]
    int* tile_fov;      // [ntiles] 1 = a particle inside this tile may lie in the field of view (k_predict, conservative box test):
                        // the placement of the other tiles registers nothing in a pyramid and may run beside the pair kernels
    int* view_list;     // [ntiles] the tiles with a view on the field of view this frame, in no particular order (FrameScalars::n_view_tiles of them):
                        // written by extra workgroups of k_predict when the frame splits its placement, walked by the placement that precedes
                        // the pair kernels INSTEAD of all the tiles
    int* tile_cls;      // [ntiles] the TWO-BRANCH frame's tile classes, written by k_tile_class right after the binning (LaunchCtx::branches):
                        // bit 0 (Q) = a newborn of this frame can land in the tile (its rows, grown by the position table's reach, touch the field
                        // of view) -- and every tile in which a particle can be registered in a pyramid; bit 1 (P) = a particle of the tile can
                        // reach a Q tile this frame (Q's rows dilated by the frame's largest displacement).  The in-view chain predicts P,
                        // places and resamples Q; the bulk branch predicts / places / resamples the complements beside it
    int* part_resample; // [ntiles rounded up to 4] live particles per tile after resampling
    int* vb_cnt;        // [v_loc] children per destination voxel this frame (birth ordering)
    int* vb_idx;        // [v_loc*128] their birth indices
    int* ck_items;      // [np * ceil(capp/128)] work items of k_ck_partial (pyramid<<12 | chunk)
    int* wu_items;      // [np * (ceil(capp/256)+1)] work items of k_weight
    int* n_items;       // [2]
    int* nb_tab;        // [NB_TAB_STRIDE][np] neighbourhood of every pyramid: bins (h-major, -1 padded) + exclusive offsets of
                        // their observation counts, written by k_pyr_items once per frame
    int* part_birth;    // [ceil(birth_cap*32/256)*2] per-block {born, dropped} of k_birth_insert
    float4* child;      // [birth_cap*32] child position + destination voxel of this frame's births
    u64* nbsnap;        // [v_loc*mw] newborn bits as they were BEFORE this frame's birth stage, or nullptr when they are known
                        // to be all zero (every frame after a resampling).  Non-null after a constructor pre-fill / an import of
                        // flag-15 records / a second birth stage without resampling: addAParticle (:1184-1185) skips those slots
    float4* ro_rec;     // [ntiles][64*slots][2] the tile's MOVING old particles {px, py, vx, vy}, {w, local voxel} (k_resample -> k_rollout)
    int* ro_cnt;        // [2 * ntiles] per tile: records in ro_rec; then the float bits of their summed weight
    int* ro_sub;        // [4 * ntiles] (cube storage) lengths of the four runs -- one per layer of the cube -- k_resample writes a tile's records
                        // in; [4 t] = -1: ONE mixed run of ro_cnt[t] records (k_resample_wg)
    int* ro_stat;       // [2 * ceil(ntiles / 8)] per group of k_rollout: contributions through its LDS windows / straight to the accumulators
    int* work_list;     // [v_loc] scratch: per-voxel prefix of the constructor-seeded particles' noise ranks (k_vz_count)
    int ntiles;         // tiles of 64 voxels; k_predict / k_place run one workgroup per tile
    int nblk_sweep;
};

struct LaunchCtx {
    MapDims d;
    FilterParams fp;
    DevState s;
    KernelScratch k;
    hipStream_t stream;
    int pt_cap, birth_cap;
    VelEst ve;
    int n_cu = 256;      // compute units of the device (sizes launches that are meant to occupy only a share of it)
    bool ro_inline = true; // small maps (k_resample_wg): the tiles roll their moving particles out themselves, no k_rollout launch
    int resample_wg_tiles = 8192;   // one-word maps with fewer tiles (and sparse ones of any size) run k_resample_wg (dspmap::resample_wg_tiles, DSPMAP_P_RESAMPLE_WG_TILES)
    bool sweep_rev = false;   // this frame's k_predict / k_resample walk the tiles from the last one down and k_place from the first one up
                              // (the next frame the other way round): every tile sweep starts where its predecessor ended (Infinity Cache)
    bool resample_rev = false;   // k_resample walks the tiles from the last one down (after a k_place that ended there)
    bool place_split = false; // this frame places the arrivals of the tiles with a view first (launch_claim sel = 1) and the others beside the pair
                              // kernels (sel = 0): k_predict leaves the list of the tiles with a view (KernelScratch::view_list)
    bool tile_bits = false;   // this frame's sweeps go by the tile bitmaps (sparse whole frames of an unsharded map; DSPMAP_P_TILE_BITMAPS)
    int side_wg = 3;          // workgroups per CU of the side-stream placement (sel = 0; DSPMAP_P_SIDE_PLACEMENT)
    bool branches = false;   // this frame runs as two branches (DSPMAP_P_FRAME_BRANCHES; see KernelScratch::tile_cls)
    bool sparse = false; // most tiles hold nothing (dspmap::sparse_mode): k_predict's variant that leaves such tiles first
};

// frame setup: rotate boundary planes (:226-232), reset per-frame counters/bins (:235-238)
void launch_frame_setup(const LaunchCtx& c, bool reset_obs);
// observation binning (:244-290)
void launch_obs_bin(const LaunchCtx& c, int n_pts_grid);
void launch_setup_and_bin(const LaunchCtx& c, int n_pts_grid, bool gather = true, const FrameParams* ring = nullptr, int ring_mask = 0);   // ring: the frame's parameter block is read from this pinned ring   // gather = false: launch_predict*(c, true) does it
// mapPrediction (:627-701) incl. re-binning of movers (moveParticle :1206-1274)
void launch_predict(const LaunchCtx& c, bool with_gather = false);
void launch_spin(const LaunchCtx& c, int us);   // experiment aid: a one-wave kernel that waits `us` microseconds
void launch_predict_only(const LaunchCtx& c, bool with_gather = false, bool with_rank = false, int cls = 0);   // with_rank: k_birth_rank rides along
   // cls (two-branch frame): 0 every tile; +m only the tiles whose class has a bit of m, -m only those that have none (KernelScratch::tile_cls)
void launch_tile_class(const LaunchCtx& c);   // after the binning (needs the rotated planes and the frame's parameter block)
#define TILE_Q 1
#define TILE_P 2
void launch_scan_blocks(const LaunchCtx& c, int nblk);   // exclusive scan of s.blk_cnt[0..nblk), total -> fs->occupied_count
void launch_claim(const LaunchCtx& c, int n_birth_grid = 0, int part = 0, int tile_lo = 0, int tile_hi = 0, int sel = -1, int cls = 0);   // sel: -1 every tile of the part, 1 / 0 only the tiles with / without a view on the sensor's field of view (tile_fov)   // part: 0 all tiles, 1 [lo, hi), 2 the rest;   // > 0: k_birth_children rides along (after a launch with_rank)
void launch_reduce_counters(const LaunchCtx& c);
void launch_calib(const LaunchCtx& c, int mode, size_t n);
void launch_sweep_probe(const LaunchCtx& c, int what, int rows, int rows_per_batch);
// every float a in [0, amax]: does reciprocal + two FMAs give the IEEE quotient a / res?  *bad counts the exceptions
void launch_verify_div(hipStream_t stream, float res, float rcp_res, float amax, int* bad);
// multi-GPU: compact particles that left the slab / insert particles received from a neighbour
void launch_export_slab(const LaunchCtx& c, int dir, float* rec_out, int cap, int* count_dev, float* rec_out_down = nullptr);   // dir 0: both (up -> rec_out / count[0], down -> rec_out_down / count[1])
void launch_import_movers(const LaunchCtx& c, int n, const float* rec);  // folds the per-block partial counters into FrameScalars
// velocityEstimationThread (:1377-1544) on the device: view -> birth cloud in DevState::birth, FrameScalars::est_n
void launch_velocity_estimator(const LaunchCtx& c, bool with_rank);   // with_rank: + the birth stage's rank in the same workgroup
// ... on a queue of its own (c.stream = that queue; DevState::xq): k_ve_view (waits for "the previous frame's birth stage has ended" unless want
// == 0, then rotates and bins the view from the frame's ring slot), the two kernels on that picture, and the word the frame's first birth kernel
// waits for (= xq_seq)
void launch_velocity_estimator_xq(const LaunchCtx& c, bool with_rank, const FrameParams* slot, int* xq, int* gave_up, int want, int xq_seq);
int velocity_estimator_capacity();   // points per frame the device estimator handles
int velocity_estimator_slices();
// mapUpdate (:704-793)
void launch_place_fix(const LaunchCtx& c);     // re-slots the arrivals of voxels in which a full pyramid list turned a particle away (after launch_pyr_prepare)
// sharded maps: one pass (8 bits, most significant first) of the distributed selection of every pyramid's CAPP-th smallest sweep key;
// the caller sums `hist` ([np][256]) over the ranks between the two launches
void launch_pyr_hist(const LaunchCtx& c, int pass, const int2* sel, int* hist);
void launch_pyr_pick(const LaunchCtx& c, int pass, const int* hist, int2* sel, int* kstar);
int pyr_select_passes();
void launch_pyr_kept(const LaunchCtx& c, const int* kstar, int* kept);   // after the last pass: this rank's kept entries per pyramid
void launch_pyr_prepare(const LaunchCtx& c);   // range sort + full-list selection of the pyramid lists, work items (idempotent)
void launch_ck_partial(const LaunchCtx& c, bool prepared = false, bool with_fix = true);   // with_fix: the first workgroups run k_place_fix's pass     // launch_pyr_prepare (unless already queued) + the Ck pass
void launch_ck_finalize(const LaunchCtx& c);
void launch_weight_update(const LaunchCtx& c);
// mapAddNewBornParticlesByObservation (:796-921)
void launch_birth(const LaunchCtx& c, int n_birth, bool in_frame, bool all_static);  // in_frame: between k_weight and k_resample of a whole frame
void launch_birth_split(const LaunchCtx& c, int n_birth);
void launch_birth_split_cksum(const LaunchCtx& c, int n_birth);             // split + the 1/Ck reduction in one launch
void launch_birth_early(const LaunchCtx& c, int n_birth, bool with_rank = true);                      // split-phase frame: rank + children right after the prediction
void launch_birth_finish(const LaunchCtx& c, int n_birth, bool all_static);    // ... cursors + insert at its end
void launch_birth_late(const LaunchCtx& c, int n_birth, bool all_static, bool with_children = false);   // with_children: no k_birth_children launch preceded (the split's waves generate them);   // whole frame after launch_predict_only(with_rank) + launch_claim(n): split, 1/Ck sum, cursors, insert
void launch_birth_plan_insert(const LaunchCtx& c, int n_birth, bool in_frame, bool all_static);
void launch_birth_materialize(const LaunchCtx& c, BirthSrc* out, int cap, int* n_out);   // the frame's synthesised birth cloud, for host readback
// mapOccupancyCalculationAndResample (:924-1057)
void launch_resample(const LaunchCtx& c, int cls = 0, bool with_rollout = true, int part = 0);   // + the future rollout of the moving particles (k_rollout)
int rollout_groups(const MapDims& d, int ntiles);   // workgroup groups of k_rollout (KernelScratch::ro_stat holds 2 ints per workgroup: x 4 with cube storage and windows)
void launch_rollout(const LaunchCtx& c);    // the rollout alone (a two-branch frame: once, behind both branches' resampling)
int resample_variant(const LaunchCtx& c);   // bit 0: k_resample_wg; bits 1-2: rollout 0 inline, 1 k_rollout light, 2 k_rollout windows, 3 none
void kernels_init_device();                 // function attributes of the current device (dynamic LDS of k_rollout)
// readout (:385-438)
void launch_occupied_compact(const LaunchCtx& c, float thr);
void launch_clear_future(const LaunchCtx& c);
void launch_future_combine(const LaunchCtx& c);  // fold the static-particle future mass into the [V][T] grid
void launch_results_true(const LaunchCtx& c, float4* out);   // res4 (storage order) -> the reference's voxel order
// state helpers
void launch_seed_uniform(const LaunchCtx& c, int per_voxel, float weight, unsigned seed, float vmax);
void launch_import(const LaunchCtx& c, int n, const int* voxel_dev, const int* slot_dev, const float* rec8_dev, int* n_failed_dev);
void launch_export(const LaunchCtx& c, int* voxel_out, int* slot_out, float* rec8_out, int* count_dev, int cap);
void launch_add_random(const LaunchCtx& c, int n, float weight, int* slot_of_tmp /* [n] device scratch */);
