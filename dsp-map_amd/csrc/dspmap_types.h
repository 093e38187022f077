// dspmap_types.h -- structures shared by the host runtime and the gfx950 kernels.
//
// Data layout in HBM:
//   * voxel structure: dense slots like the reference's
//     voxels_with_particle[V][SLOTS][9] (dsp_dynamic.h:116) but SoA: one fp32
//     array per field, voxels grouped in tiles of 64 and stored slot-major inside
//     a tile: index = ((lv>>6)*SLOTS + slot)*64 + (lv&63); plus one 64-bit
//     occupancy word per voxel per 64 slots (bit = slot is live) and a second
//     word marking particles born this frame (the reference's flag 15,
//     dsp_dynamic.h:1186).  vz and update_time are not stored: vz is
//     identically 0 after the first prediction under LIMIT_MOVEMENT_IN_XY_PLANE
//     (:661-663) and update_time is never read.
//   * pyramid structure: instead of back-pointer lists (pyramids_in_fov, :124)
//     a per-pyramid staging area holding a copy {x,y,z,w} + slot id of each
//     particle inside that angular bin, rebuilt by the prediction kernel.
//   * result grid: float4 {mass, mean vx, vy, vz} per voxel + [V][T] future
//     accumulators (voxels_objects_number, :118-120).
#pragma once
#include <stdint.h>

#define DSP_MAX_PRED 16
#define DSP_DIRTY_CAP 65536          // voxels per frame that can be re-slotted after a pyramid list overflow
#define DSP_MAX_NBINS 25           // neighbourhood bins supported by the pair kernels (radius <= 2)
#define PS_NBK 128                 // range buckets per pyramid list (k_pyr_sort)
#define NB_TAB_STRIDE (2 * DSP_MAX_NBINS + 2)   // ints per pyramid in KernelScratch::nb_tab: bins, then offsets
#define DSP_OBS_CAP 100            // observation_max_points_num_one_pyramid :69
#define DSP_MAX_PLANES_H 129       // np_h + 1 boundary planes
#define DSP_MAX_PLANES_V 97

typedef unsigned long long u64;

struct MapDims {
    int nx, ny, nz;        // global grid
    int z_lo, z_hi;        // owned slab [z_lo, z_hi)
    int v_loc;             // voxels of the STORAGE order the device arrays are indexed with (mask[lv], res4[lv], cells of tile lv >> 6): the
                           // slab's nx*ny*(z_hi-z_lo) voxels in index order (tiling 0), or its 4x4x4 cubes, 64 voxels each, padding included (tiling 1)
    int v_true;            // the slab's voxels: nx*ny*(z_hi-z_lo) (what the caller's arrays hold, :118-120)
    int tiling;            // which 64 voxels share a tile (one wave, one lane per voxel): 0 = 64 consecutive voxel indices -- a run
                           // along x --, 1 = a cube of 4 x 4 x 4 voxels: storage index = cube * 64 + (z & 3) * 16 + (y & 3) * 4 + (x & 3), cubes x-fastest.
                           // A needle that points away from the sensor is cut by the field of view almost wherever it lies (58 % of the tiles of the
                           // 132x132x60 map have a view, against 19 % of its cubes): cubes are what lets the two-branch frame leave most tiles alone.
                           // Sweep keys, the caller's indices and every result array keep the reference's voxel index (:1081): lv_of_* / g_of_lv in
                           // dspmap_device.h translate
    int ncx, ncy, ncz;     // cubes per axis (tiling 1): ceil(nx / 4), ceil(ny / 4), ceil((z_hi - z_lo) / 4)
    int v_base;            // global index of local voxel 0 = z_lo*nx*ny
    int v_glob;            // nx*ny*nz
    int slots;             // SAFE_PARTICLE_NUM_VOXEL = 2*M  :65
    int mw;                // mask words per voxel = ceil(slots/64)
    int M;                 // MAX_PARTICLE_NUM_VOXEL :43
    int np_h, np_v, np;    // pyramids :58-60
    int capp;              // SAFE_PARTICLE_NUM_PYRAMID :66
    int capa;              // entries a pyramid's UNSORTED list can hold (2 x capp + 64): k_pyr_prepare keeps the capp smallest keys
    int T;                 // PREDICTION_TIMES :46
    int nn;                // pyramid neighbourhood radius: 1 = 3x3 (:1135-1136), 2 = 5x5 (dsp_dynamic_multiple_neighbors.h)
    int nbins;             // (2*nn+1)^2
    int static_model;      // dsp_static.h's motion model: velocities forced to 0 in prediction
    int tile_skip;         // DSPMAP_P_STATIC_TILE_SKIP: tiles of static particles are swept without their velocity rows (DevState::tile_moving)
    float res;
    float rcp_res;         // RN(1 / res)
    int div_ok;            // 1: a / res may be computed as reciprocal + two FMAs (verified exhaustively on the device, dspmap_device.h: div_res)
    float half_x, half_y, half_z; // :528-530
    float rng_inv_bw;             // range buckets of k_pyr_sort: bucket = (int)(|p| * rng_inv_bw), PS_NBK buckets to the map corner
    float pred_t[DSP_MAX_PRED];
};

struct FilterParams {
    float sigma_ob;        // :156
    float inv_sigma_ob;    // 1/sigma_ob (host fp32 division)
    float kappa;           // :157
    float p_det;           // :158
    float nb_weight;       // new_born_particle_weight :162
    int nb_num;            // new_born_particle_number_each_point :163
    int min_static_nb;     // (int)(n*0.15f), frozen at first birth call :808
    int model_nb;          // (int)(n*0.8f) :811
    float pdf_c;           // 1/sqrtf(2*pi/2): centre of the reference's LUT :1284
    float pdf_c3;          // pdf_c^3
    float occl_margin;     // obstacle_thickness_for_occlusion :70
    float cull_r;          // 9 * sigma_ob: a particle and an observation whose ranges differ by more contribute < 1e-19 (pair kernels)
    int tab_n;             // Gaussian table length :72
    int rtab_n;            // rand table length
};

// Device-side per-frame scalars and counters (one instance in HBM).
struct FrameScalars {
    int n_valid;        // valid_points :286
    int n_obs;
    int n_live_in;
    int n_moved;
    int n_out_of_map;
    int n_voxel_full;
    int n_pyramid_full;
    int n_fov;
    int n_born;
    int n_born_dropped;
    int n_live_out;
    int n_exp_up, n_exp_down;
    int occupied_count; // readout
    int n_voxel_full_import; // multi-GPU: movers received from a neighbour that found their voxel full
    int n_dirty;            // entries of DevState::dirty
    int bits_on;            // the last frame's sweeps went by the tile bitmaps (DevState::vis_bits): k_reduce_counters gates k_predict's per-tile counts with pred_bits
    int n_overflow_inexact; // diagnostics: voxels / arrivals the re-slotting pass could not treat exactly (see k_place_fix)
    int mv_acc;                // tiles in which k_resample_wg noted more than RO_INLINE_MAX moving particles (moved to hint_out[1] like live_acc)
    int live_acc, live_hint;   // every 64th tile that k_resample leaves non-empty counts itself in live_acc; the next frame's first kernel
                               // moves the count to live_hint: k_predict's estimate of how sparse the map is (a hint, never a result)
    int v_cur_in, r_cur_in;    // the velocity-table / rand() cursors before the frame's births (the fused insertion reads these, one workgroup writes the new ones)
    int n_place_vf, n_place_pf;   // arrivals k_place turned away: voxel full (:1227-1229) / pyramid list full (:1256-1259)
    int n_pyr_removed;  // particles k_pyr_prepare turned away because their pyramid's list was full (-2, :1256-1259)
    float expected_newborn;  // expected_new_born_objects :292
    float newborn_w;         // updated_weight_new_born :805
    float cur_pos[3];        // current_position :131
    int p_cur, v_cur, r_cur; // table cursors :483-484 (+ rand stream)
    int has_expected_override;
    // synthesised birth cloud (FrameParams::static_birth): frame epoch in which the view held at least one point, and
    // the length of the last non-empty view's cloud kept in DevState::birth (an empty view re-uses it, :1379-1381)
    int view_epoch, stale_n;
    int n_birth_ovf;    // entries of DevState::birth_ovf (reset by the birth rank, which precedes every generation of children)
    int est_n;          // length of the birth cloud the device velocity estimator wrote (kept when a view is empty, :1379)
    int n_view_tiles;   // entries of KernelScratch::view_list: the tiles whose box can intersect the field of view this frame (k_predict's extra
                        // workgroups of a split placement; reset with the pyramid lists)
    int vmax_bits;      // float bits of the largest |vx|, |vy| any particle of the map was ever given (births, imports, seeds, first-prediction
                        // noise; note_speed in dspmap_device.h; reset with the state): bounds how far a prediction can carry a particle --
                        // k_tile_class sizes the halo of the two-branch frame with it.  Monotone, conservative, never a result
    int pred_epoch;     // bumped by whatever resets the pyramid lists for a prediction (k_reset / k_obs_points): k_place stamps the tiles it
                        // served with it, k_place_fix only trusts a tile's inbox / pmask when the stamp is this prediction's
};

// Per-frame inputs, written by ONE small H2D copy per frame and read by the kernels from HBM, so that
// the kernel arguments of a frame never change and the whole frame can be replayed as a HIP graph.
struct FrameParams {
    float quat[4];      // sensor attitude w,x,y,z
    float cur_pos[3];   // current_position :131
    float od[3];        // -delta position (particles move opposite to the sensor, :300)
    float dt;
    int n_pts;          // points in `pts`
    int n_birth;        // entries in `birth`
    int static_birth;   // birth cloud: 0 = the caller's / host estimator's (birth, n_birth); 1 = synthesised from the frame's view
                        // (every in-FOV point a zero-velocity source); 2 = written by the device velocity estimator
    float res_filter;   // voxel_filtered_resolution :132 (ground split and cluster tolerance of the velocity estimator)
    int epoch;          // frame counter (bumped whenever a cloud is binned); FrameScalars::view_epoch refers to it
    int clear_fut;      // 1: k_predict zeroes the future accumulators first (a clearOccupancyMapPrediction is pending)
    int from_ring;      // 1: this block came through the pinned parameter ring
    unsigned ring_pos;  // its position in the ring: k_predict moves the ring's read position from ring_pos to ring_pos + 1 (once,
                        // whatever else replays a stale block afterwards)
    float birth_reach;  // the largest |value| of the position table (:871-873 adds three of them to a source point): how far from its
                        // observation a newborn can land (k_tile_class)
    const float* pts;   // n_pts x 3, sensor frame
    struct BirthSrc* birth;
};

struct DevState {
    u64* mask;     // [v_loc*mw] live bits
    u64* nbmask;   // [v_loc*mw] born-this-frame bits
    // particle fields, cell index = pidx(lv, slot); grouped by what is always written together, because a
    // scattered store costs one memory transaction per lane whatever its width (<= 16 B):
    float* pos;   // [S][3] {px, py, pz}   rewritten by every prediction
    float* vel;   // [S][2] {vx, vy}       written at birth / move / copy only
    float* w;     // [S]                   rewritten by the weight update and the resampler
    float* vz0;    // optional, only right after an import with vz != 0 (consumed by the next prediction)
    float4* res4;  // [v_loc] {mass, mean vx, mean vy, mean vz}
    u64* fut;      // [T][v_loc]  future mass scattered by moving particles, FIXED-POINT (units of 2^-24, fut_quantum in dspmap_device.h:
                   //             integer atomics -- the sum does not depend on the order of the adds, on the rollout variant or on
                   //             the sharding), HORIZON-major: the rollout flushes whole rows of neighbouring voxels of one horizon
                   //             (coalesced atomics)
    float* fut_out; // [v_loc][T] the caller's layout (voxels_objects_number[v][4..], :118-120): fut + fut_stat, written by
                   //             k_future_combine on demand (readout is not part of update())
    float* fut_stat; // [v_loc]   future mass of static particles (identical for every horizon; folded in at readout)
    // observations
    float4* obs;       // [np*100] {x,y,z,len}
    long long* obs_ck; // [np*100] sum over particles of P_d*w*g (pass 1), without the frame constant; fixed point,
                       // units of 2^-34 (ck_snap / CK_FIX_SCALE, dspmap_device.h): integer atomics are associative, so the sum does not depend on the
                       // order the workgroups (or the ranks of a sharded map) arrive in -- frames are reproducible
    float* obs_ckf;    // [np*100] final Ck = obs_ck + lambda + kappa (:737)
    float* part_inv;   // [np] per-pyramid sum of 1/Ck
    int* obs_cnt;      // [np]
    float* obs_maxlen; // [np]
    float* planes_h;   // [(np_h+1)*3] rotated
    float* planes_v;   // [(np_v+1)*3]
    float* planes_h0;  // un-rotated :563-578
    float* planes_v0;
    // input points
    float4* pt_rot;    // [pt_cap] rotated xyz + len
    int* pt_pyr;       // [pt_cap] pyramid id or -1
    // birth
    struct BirthSrc* birth; // [birth_cap]
    struct BirthPlan* plan; // [birth_cap]
    int2* birth_cvr;        // [birth_cap] draws of the point's children from the velocity table / the rand() stream (x, y); then, from
                            // index birth_cap on, the sums over the 16 points of each workgroup of the split (the fused insertion's prefix)
    unsigned* plan_inside;  // [birth_cap] bit k = child k of the point lies inside the map (:875), k_birth_children
    int* plan_pbase;        // [birth_cap] position-table cursor of each source point (k_birth_rank; kept apart from `plan`
                            // so that the rank and the split can run in the same launch)
    int* nstatic;           // [birth_cap] (multi-GPU all-reduce(max) buffer)
    int* birth_ovf;         // [birth_cap*32] birth indices of the children that found their destination voxel's bucket full
                            // (FrameScalars::n_birth_ovf entries; their voxel is in KernelScratch::child)
    // FOV staging
    float4* fov_rec;   // [np*capa] {x,y,z,w}
    int* fov_slot;     // [np*capa] cell index of the particle (pidx)
    int* fov_spos;     // [np*capa] where the range sort put the entry (index into fov_rec_s / fov_slot_s of its pyramid), -1 = not kept
    int* fov_key;      // [np*capa] its sweep key (source voxel * slots + slot): the reference registers a pyramid's particles in this order
    float4* fov_rec_s; // the same lists ordered by range bucket (k_pyr_sort), read by the pair kernels
    int* fov_slot_s;
    int* pyr_cnt;      // [np]
    // readout scratch
    int* blk_cnt;      // [ceil(v_loc/256)+1]
    float* occ_xyz;    // [v_loc*3] (allocated on first use)
    // tables
    float* p_tab; float* v_tab; int* r_tab;
    FrameScalars* fs;
    FrameParams* fpar;
    int* hint_out;      // host-mapped words for the caller's thread: [0] FrameScalars::live_hint (which k_predict variant to launch),
                        // [1] last frame's tiles with many moving particles (inline rollout or k_rollout)
    int* ring_seq;      // read position of the pinned parameter ring (frames replayed as a captured graph)
#define XQ_NDEC 64       // DevState::xq: counters of the first birth kernel's workgroups that have decided, one per 256-byte block from XQ_DEC on
#define XQ_DEC 64
#define XQ_LIST (XQ_DEC + XQ_NDEC * 64)   // ... and the list of the workgroups that left their share to workgroup 0
    int* xq;            // nullptr, or (DSPMAP_P_ESTIMATOR_QUEUE: the velocity estimator's kernels run on a queue of their own, tied to the
                        // captured frame by nothing but these words in HBM, agent-scope atomics) [0] = ring position + 1 of the last frame
                        // whose BIRTH STAGE has ended (published by the resampling kernel that follows it), polled by k_ve_view in front of
                        // the NEXT frame's estimator kernels (the rand() cursor and the birth buffers are theirs from then on);
                        // [1] = ring position + 1 of the last frame whose birth cloud the estimator has finished (k_ve_clusters, after a
                        // release fence), polled by the frame's first birth kernel; hint_out[3] notes a poll that gave up.
                        // Every wait is for work that was SUBMITTED EARLIER, so no mapping of streams to hardware queues can deadlock.
    // re-slotting after a full pyramid list has turned particles away (k_place_fix, dspmap_kernels.hip)
    u64* pmask;         // [v_loc*mw] occupancy after the prediction, before any arrival was placed (tiles with arrivals; k_place)
    u64* ta;            // [v_loc*mw] cells whose particle its pyramid's full list turned away this frame (all zero between frames)
    int* dflag;         // [v_loc] 1 = the voxel is in the dirty list
    int* dirty;         // [DSP_DIRTY_CAP] local voxels that lost a particle to a full pyramid list this frame
    // sharded maps: the pyramid-list capacity is the reference's GLOBAL one (:64-66,1256-1259)
    const int* pyr_kstar; // [np] or nullptr: the CAPP-th smallest sweep key of every pyramid over ALL ranks (0x7fffffff: no cut), found by the
                        // distributed radix select of dspmap_dist.hip; k_pyr_prepare cuts with it instead of selecting among the rank's own entries
    const int* pyr_kept; // [np] or nullptr (set together with pyr_kstar): entries of this rank's list that the global cut keeps = the length of
                        // the range-sorted list the pair kernels read (without a global cut: min(pyr_cnt, capp))
    long long* pyr_gcnt; // [np] or nullptr: this rank's list lengths before the cut, summed over the ranks by the Ck all-reduce they ride on
    int* in_n;          // [2 * tiles] {arrivals the last placement served in the tile, FrameScalars::pred_epoch of that placement}
    // TILE BITMAPS of a sparse map's whole frames (round 6; nullptr: not in use).  A sparse map's sweeps launch a workgroup per tile to find
    // most of them empty (87 120 tiles at 264x264x80, ~12 k with particles): what an empty tile costs is the round trip of the scalar loads
    // that tell -- three words at three addresses per tile.  One BIT per tile instead, 32 tiles a word: the whole table (11 kB at that size)
    // sits in the scalar cache, and an empty tile's workgroup is gone as fast as the dispatcher can start the next one.
    //   vis_bits   tiles k_predict / k_resample visit: live or with dirty future accumulators when the frame began (k_obs_points' extra
    //              workgroups rebuild it from tile_live / fut_dirty every frame -- the flags stay the truth, whoever wrote them), plus the
    //              tiles that received their first particle during the frame (k_place, k_birth_insert set the bit with the flag)
    //   pred_bits  the same table as the frame began: the tiles k_predict visited (k_reduce_counters adds up THEIR counts only)
    //   arr_bits   tiles with arrivals: set by k_predict's tail for the first record of an inbox, zeroed with the rebuild
    unsigned* vis_bits;
    unsigned* pred_bits;
    unsigned* arr_bits;
    int* fut_dirty;     // [tiles] 1 = something was added to the tile's future accumulators (fut, fut_stat) since they were zeroed
    int* tile_moving;   // [tiles] 0 = every LIVE particle of the tile has velocity (0, 0): rewritten by k_predict for every tile it visits (it reads
                        // the velocities anyway), set by whoever puts a moving particle there afterwards (k_place: arrivals, k_birth_insert:
                        // newborns of matched clusters; 1 everywhere after particles were written outside a frame).  The sweeps of such a tile
                        // do not fetch its velocity rows (8 of the 24 / 12 bytes k_predict / k_resample read per cell): a map is mostly
                        // static particles -- that is what the reference's static / dynamic split is about.  Conservative: 1 promises nothing.
    int* tile_live;     // [tiles] 0 = the 64-voxel tile holds no particle (k_resample found it empty and nothing was placed, born or
                        // imported there since): the sweeps skip it without reading its occupancy words.  Conservative: nonzero
                        // does not promise a particle.
};

// device velocity estimator (dspmap_velest.hip): one cluster = the reference's ClusterFeature :98-109 + bookkeeping
struct VeCluster {
    float cx, cy, cz;
    int point_num;
    float vx, vy, vz, intensity;
    int root, start, is_dyn, dyn_idx;
};
struct VelEst {
    float4* w;        // [cap] world position of the view points, in view (= input) order
    int* root;        // [cap] view index of the first point of the point's connected component (-1: ground point)
    int* ng_view;     // [cap] view index of the non-ground points
    unsigned* edges;  // [slices][cap] spanning-forest edges (g << 16 | root) of every slice of k_ve_components
    int* ecnt;        // [slices]
    int* rank;        // [cap/5+8]
    int* by_rank;     // [cap/5+8]
    int* dyn_list;    // [cap/5+8]
    VeCluster* cl;    // [cap/5+8]
    float* last;      // [(cap/5+8) * 5] clusters_feature_vector_dynamic_last :1401: cx, cy, cz, point_num (int bits), intensity
    int* n;           // [4] view points, non-ground points, kept clusters of the last frame
    // DSPMAP_P_ESTIMATOR_QUEUE: the estimator's own picture of the frame (it may run before the frame's first kernel has): the view rotated
    // and binned by k_ve_view, and a copy of the frame's parameter block, taken from the pinned ring
    float4* v_rot;    // [cap]
    int* v_pyr;       // [cap]
    FrameParams* v_fpar;
    int cap;          // view points the estimator handles (one workgroup holds them in LDS)
};

struct BirthSrc {   // == dspmap_vpoint
    float x, y, z, nx, ny, nz, intensity;
};
struct BirthPlan {
    float cx, cy, cz;   // source point in the sensor-centred frame :818-820
    int gvox;           // global voxel of the source point, -1 = outside map / invalid
    int n_static;       // :862-866
    unsigned inside;    // bit k: child k landed inside the map :875 (set with atomicOr)
    int pbase, vbase, rbase; // table cursors of this point's first draw (pbase: see DevState::plan_pbase)
};
