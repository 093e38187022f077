/*
 * dspmap.h -- C ABI of libdspmap_hip.so: the MI355X-native particle-based
 * dynamic occupancy mapper (hand-written HIP kernels for gfx950).
 *
 * This is the drop-in boundary for the per-frame loop of g-ch/DSP-map's
 * include/dsp_dynamic.h (class DSPMap).  Every entry point names the reference
 * interface it replaces (file:line relative to the reference tree).  Plain
 * pointers and sizes only; no C++/torch types cross this boundary.  The C++
 * class surface (`class DSPMap`, include/dsp_dynamic.h in THIS repo) and the
 * Python/ctypes binding (dsp-map_amd/capi.py) are thin forwards to it; the
 * binding a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - All functions return DSPMAP_OK (1) on success unless stated otherwise;
 *    0 = "frame rejected, state unchanged" for update (as the reference's
 *    `return 0`, dsp_dynamic.h:193-208); negative = error, text via
 *    dspmap_last_error().  Nothing throws.  There is NO CPU fallback: if no
 *    HIP device is usable every compute entry point fails with DSPMAP_E_DEVICE.
 *  - `host` pointers are caller-owned host memory, never retained past return.
 *    `dev` pointers are device memory on the handle's device.
 *  - One frame in flight per handle; a handle is not re-entrant (the reference
 *    is not either: function statics + file-scope state, dsp_dynamic.h:116-140,187-190).
 *  - Voxel indexing, slot capacity, pyramid (angular bin) layout and all
 *    constants follow the reference (dsp_dynamic.h:38-70).
 */
#ifndef DSPMAP_H
#define DSPMAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSPMAP_OK 1
#define DSPMAP_REJECTED 0
#define DSPMAP_E_ARG (-1)
#define DSPMAP_E_DEVICE (-2)
#define DSPMAP_E_STATE (-3)

#define DSPMAP_MAX_PRED_TIMES 16

typedef struct dspmap dspmap_t;

/* Run-time replacement of the reference's compile-time macros
 * (dsp_dynamic.h:38-50) plus what a multi-GPU shard needs. */
typedef struct dspmap_config {
    int nx, ny, nz;             /* MAP_LENGTH/WIDTH/HEIGHT_VOXEL_NUM  :38-40 */
    float voxel_resolution;     /* VOXEL_RESOLUTION                   :41 */
    int angle_resolution;       /* ANGLE_RESOLUTION, degrees          :42 */
    int max_particle_num_voxel; /* MAX_PARTICLE_NUM_VOXEL             :43 */
    int half_fov_h, half_fov_v; /* :49-50, degrees */
    int prediction_times;       /* PREDICTION_TIMES                   :46 */
    float prediction_future_time[DSPMAP_MAX_PRED_TIMES]; /* :47 */
    /* Z-slab owned by this handle: voxel layers [z_lo, z_hi) of the global
     * grid (voxel index is z-major, :1081).  z_lo == z_hi == 0 means all. */
    int z_lo, z_hi;
    int device;                 /* HIP device ordinal, -1 = current device */
    int gaussian_table_size;    /* GAUSSIAN_RANDOMS_NUM :72 (10,000,000); 0 = default */
    unsigned seed;              /* table/rand seed; 0 = time(NULL) like :586,1151 */
    /* the reference's two other headers as run-time parameters of the same kernels; 0 = dsp_dynamic.h's value */
    int pyramid_neighbor_n;     /* PYRAMID_NEIGHBOR_N (dsp_dynamic_multiple_neighbors.h:43: 2 -> 5x5 neighbourhood); default 1 */
    int safe_particle_factor;   /* SAFE_PARTICLE_NUM_VOXEL / MAX_PARTICLE_NUM_VOXEL: default 2 (:65); dsp_static.h:63 uses 5 */
    int static_model;           /* 1 = dsp_static.h: velocities forced to zero in prediction, every birth source static */
} dspmap_config;

/* Birth-source point = one entry of the reference's input_cloud_with_velocity
 * (pcl::PointXYZINormal, dsp_dynamic.h:134,1510-1539): world-frame xyz,
 * normal = estimated velocity (-10000 when the cluster is unmatched, :104-106),
 * intensity = cluster tag (0 = static / ground). */
typedef struct dspmap_vpoint {
    float x, y, z;
    float nx, ny, nz;
    float intensity;
} dspmap_vpoint;

/* Per-frame device counters (the reference computes similar counters and
 * never prints them, dsp_dynamic.h:629-632,926-927). */
typedef struct dspmap_counters {
    int n_points_in;       /* points handed to update() */
    int n_valid;           /* valid_points :286 (in FOV, incl. overflowed) */
    int n_obs;             /* stored observations (<= 99 per pyramid, :279-284) */
    int n_live_in;         /* live particles entering prediction */
    int n_moved;           /* particles that changed voxel */
    int n_out_of_map;      /* removed: left the map :688 */
    int n_voxel_full;      /* removed: destination voxel full (-1, :1227) */
    int n_pyramid_full;    /* removed: pyramid list full (-2, :1256) */
    int n_fov;             /* particles registered in pyramids after prediction */
    int n_born;            /* newborn particles inserted :911 */
    int n_born_dropped;    /* newborn dropped: voxel full :1198 */
    int n_live_out;        /* live particles after resampling */
    int n_exported_up, n_exported_down; /* multi-GPU: left the slab through z_hi / z_lo */
    int n_reslotted;       /* voxels whose arrivals were re-slotted because a full pyramid list turned a particle away (:1256-1259) */
    int n_overflow_inexact;/* diagnostics: arrivals / voxels that pass could not treat exactly (destination voxel full AND list full) */
    float newborn_weight;  /* updated_weight_new_born :805 */
    float update_ms;       /* device time of the last TIMED update (HIP events): every frame of the host-staged path, every 32nd frame
                              of dspmap_update_device's replayed graph (an event record between two replays costs ~5 us) */
} dspmap_counters;

enum dspmap_param {
    DSPMAP_P_POSITION_STDDEV = 1,   /* setPredictionVariance arg 1   :355 */
    DSPMAP_P_VELOCITY_STDDEV = 2,   /* setPredictionVariance arg 2   :355 */
    DSPMAP_P_OBSERVATION_STDDEV = 3,/* setObservationStdDev          :362 */
    DSPMAP_P_NEWBORN_WEIGHT = 4,    /* setNewBornParticleWeight      :366 */
    DSPMAP_P_NEWBORN_NUMBER = 5,    /* setNewBornParticleNumberofEachPoint :370 */
    DSPMAP_P_VOXEL_FILTER_RES = 6,  /* setOriginalVoxelFilterResolution :380 */
    DSPMAP_P_KAPPA = 7,             /* kappa :157 */
    DSPMAP_P_DETECTION = 8,         /* P_detection :158 */
    DSPMAP_P_VELOCITY_ESTIMATOR = 9,/* velocityEstimationThread (:297,1377-1544) inside update(): 2 = on the device, a side branch of the
                                       captured frame (dspmap_velest.hip; the drop-in class's default); 1 = host stage overlapped
                                       with the kernels (velocity_estimator.cpp); 0 = births use the caller's cloud, or tag every
                                       point in view as a static source */
    DSPMAP_P_REGENERATE_TABLES = 10,/* 1 = setPredictionVariance regenerates the Gaussian tables (:359) */
    DSPMAP_P_USE_GRAPH = 11,        /* 1 (default) = dspmap_update / dspmap_update_device replay the frame as a captured HIP graph (one launch, ~18 us of host time per
                                       frame); 2 = the same kernels as plain launches, parameter block through the same pinned ring: no graph boundary between two
                                       frames (it costs 8.7 us on this runtime: 66x66x40 0.151 -> 0.144 ms, -4.4 .. -5.9 % in-process; larger maps +-1 %) but ~100 us of
                                       host time per frame for its twelve launches (the host-pointer update() becomes host-bound: 4 800 frames/s), which is why it
                                       is not the default; 0 = plain launches with a copied parameter block (rounds 1-2; what stage profiling uses).  Same result */
    DSPMAP_P_OCCLUSION_MARGIN = 12, /* obstacle_thickness_for_occlusion :70 (0.3 m); the reference's two other headers use VOXEL_RESOLUTION */
    DSPMAP_P_UPDATE_TIME = 14,      /* read-only: update_time, the sum of the accepted frames' delt_t (:634) */
    DSPMAP_P_UPDATE_COUNTER = 15,   /* read-only: update_counter, the number of predictions run (:635) */
    DSPMAP_P_PLACE_SPLIT_TILES = 16,/* maps with at least this many 64-voxel tiles (default 8192) give the voxel-changing particles of the tiles
                                       that cannot see the sensor's field of view their slots on a side stream, beside the weight update
                                       (same result; a scheduling knob: 1 = always, a huge value = never; not while the handle sweeps the map
                                       as a sparse one -- most tiles empty, DSPMAP_P_SPARSE_SWEEP -- unless the value is 1; the environment
                                       variable DSPMAP_PLACE_SPLIT_TILES presets it at dspmap_create) */
    DSPMAP_P_FAST_DIVISION = 17,    /* read: 1 if (p + half) / VOXEL_RESOLUTION (:1062-1088) is computed as reciprocal + two FMAs -- only after a
                                       kernel has compared that quotient with the IEEE division, bit for bit, for this resolution
                                       (device initialisation); write 0: force the IEEE division (same results by construction) */
    DSPMAP_P_SPARSE_SWEEP = 18,     /* which variant of the prediction sweep runs: -1 (default) the handle decides from a running estimate of how
                                       many 64-voxel tiles hold particles (most empty: empty tiles are left after one scalar load), 0 / 1 force
                                       one (same result either way; read: the variant of the last frame) */
    DSPMAP_P_ROLLOUT_INLINE = 19,   /* maps small enough for the four-waves-per-tile resampler: 1 = its tiles add their moving particles' future
                                       status themselves (one float atomic per particle and horizon; no k_rollout launch), 0 = k_rollout's LDS
                                       windows, -1 (default) the handle decides from the number of tiles with hundreds of moving particles; larger
                                       maps: 1 = k_rollout without windows, 0 = with them.  The future status is accumulated in fixed point, every
                                       particle adds the same integer on every path: the SAME result bit for bit (read: the last frame's choice) */
    DSPMAP_P_RESAMPLE_WG_TILES = 20,/* one-occupancy-word maps with FEWER 64-voxel tiles than this (default 8192) -- and larger ones while the handle takes
                                       them for sparse (most tiles empty, DSPMAP_P_SPARSE_SWEEP) -- run the four-waves-per-tile variant of the
                                       resampling stage, the others the one-wave-per-tile variant (same result slot for slot; a scheduling knob: 0 =
                                       never, a huge value = whenever the map qualifies; the environment variable DSPMAP_RESAMPLE_WG_TILES presets it) */
    DSPMAP_P_SWEEP_ALTERNATE = 21,  /* direction of the three sweeps over the map's 64-voxel tiles.  A large map's live rows are several times the 256 MB
                                       Infinity Cache, so a sweep that starts where its predecessor ENDED finds its first tiles there instead of in HBM:
                                       2 (default) = prediction upwards, placement of the voxel-changing particles downwards, resampling downwards (and
                                       the next frame's prediction starts where it ended); 0 = resampling upwards; 1 = all three flip from frame to frame
                                       (two captured graphs); -1 = 1 on maps of at least 4096 tiles.  132x132x60 saturated: placement -15 %, frame -3 %.
                                       Same result in every mode: no stage depends on the order in which the tiles are visited */
    DSPMAP_P_STATIC_TILE_SKIP = 22, /* 1 (default): a 64-voxel tile whose live particles all have velocity (0, 0) is swept without its velocity rows, keeps
                                       its velocity cells zeroed, and a static particle that arrives there is placed without a velocity store (the
                                       reference never gives a static particle a velocity, :653); 0 = every tile is treated as if something moved in it.
                                       Same result either way, bit for bit (the diagnostic the differential GPU test switches) */
    DSPMAP_P_HOST_CLOUD_DIRECT = 23,/* 1 (default): dspmap_update (the host-pointer update() of the reference, :181) copies the caller's cloud into a
                                       slot of a pinned, device-mapped ring and the captured frame's first kernel reads it over the bus: the frame is
                                       ONE graph launch (needs DSPMAP_P_USE_GRAPH and the device velocity estimator); 0 = pinned staging + one H2D
                                       copy + an event in front of the graph (rounds 1-4).  Same result either way */
    /* 24: early registration of the voxel-changing particles by the prediction sweep (round 5) -- measured slower on the maps it was
       built for and removed in round 6 (LOG.md); the value is not reused */
    DSPMAP_P_ESTIMATOR_QUEUE = 25,  /* captured frames with the device velocity estimator (DSPMAP_P_VELOCITY_ESTIMATOR = 2) on maps that do not split their
                                       placement: 1 (default) = the estimator's kernels are launched on a stream of their own, BEFORE the frame, and meet it
                                       through two words in device memory: the estimator makes its own picture of the view from the frame's slot of the
                                       parameter ring (k_ve_view) and waits only for "the PREVIOUS frame's birth stage has ended" (published by that frame's
                                       resampling kernel: the rand() cursor and the birth buffers are the estimator's from then on); the frame's first birth
                                       kernel waits for "the birth cloud is complete" (published by the estimator's last kernel behind a release fence).  The
                                       reference's helper thread (:297,311) without a fork / join inside the graph, which costs ~8 us of a 147-us frame on
                                       this runtime (tools/micro/fork_join.hip).  Every wait is for work submitted earlier and is bounded (200 ms): a wait
                                       that gives up calls the frame's birth stage off, the next call fails once and the handle goes on with 0.  The stream
                                       is tested not to share the main stream's hardware queue; if none is found the handle keeps 0's path
                                       (dspmap_debug_estimator_path).  0 = a forked branch of the captured graph (rounds 2-5).  Same result either way */
    DSPMAP_P_FRAME_BRANCHES = 26,   /* whole frames of dense large maps run as TWO BRANCHES (round 6): the part of the map the sensor can see this frame -- grown by
                                       the reach of a newborn (:871-873) and by the frame's largest displacement (:665-667) -- goes through prediction,
                                       placement, mapUpdate, births and resampling on the main stream, the rest of the map (most of it: prediction,
                                       placement, resampling only -- the bandwidth-bound sweeps) beside it on a forked branch.  -1 (default) = the maps that
                                       would split their placement (DSPMAP_P_PLACE_SPLIT_TILES), 0 = never (the serial frame of rounds 1-5), 1 = whenever
                                       the frame allows it (any size; what the differential tests force).  Same result slot for slot.  The environment
                                       variable DSPMAP_FRAME_BRANCHES presets it */
    DSPMAP_P_TILING = 27,           /* which 64 voxels share a tile of the particle store (one wave, one lane per voxel): 0 = 64 consecutive voxel indices (a run along
                                       x; rounds 1-5), 1 = a cube of 4 x 4 x 4 voxels, -1 (default) = cubes on unsharded maps large enough for the two-branch
                                       frame, runs otherwise.  A run that points away from the sensor is cut by the field of view almost wherever it lies (58 %
                                       of the 132x132x60 map's runs have a view, 19 % of its cubes).  Settable only before the handle's first use (read: the
                                       order in use); sharded handles (Z-slabs) keep index order.  Results, state records and sweep order are the reference's
                                       whatever the storage: the same result slot for slot.  The environment variable DSPMAP_TILING presets it */
    DSPMAP_P_SIDE_PLACEMENT = 28,   /* frames that split their placement (DSPMAP_P_PLACE_SPLIT_TILES): value = 16 * fork + footprint.  fork: where the launch that
                                       places the arrivals of the tiles without a view leaves the main chain -- 0 behind the list preparation (rounds 3-5), 1
                                       (default) behind the placement of the tiles with a view, 2 behind the prediction; footprint: its workgroups per compute
                                       unit (1 .. 15, default 3); a negative value restores the default (19).  Measured on identical maps in one process
                                       (tools/ab_maps.py): 132x132x60 saturated 19 against 3: -2.9 %, 35: +8 %; 264x264x80 saturated: -1.4 %.  A scheduling knob:
                                       same result slot for slot */
    DSPMAP_P_RESAMPLE_SPLIT = 29,   /* frames that split their placement on cube storage, with a birth cloud made on the device from the frame's own view
                                       (DSPMAP_P_VELOCITY_ESTIMATOR 2, or every point in view a static source): 1 = the resampling stage (:924-1057) runs as two
                                       launches -- the tiles no newborn of this frame can reach (outside the field of view grown by the position table's
                                       largest value) on the side stream, right behind the placement it carries, BESIDE the weight update and the birth
                                       stage of the main chain; the others behind the births; the rollout behind both.  A frame with an empty view (stale
                                       birth cloud, :1379-1381) resamples every tile behind the births.  0 = one launch behind the births.  Same result
                                       slot for slot.  The environment variable DSPMAP_RESAMPLE_SPLIT presets it */
    DSPMAP_P_TILE_BITMAPS = 30,     /* whole frames of an unsharded map the handle sweeps as a sparse one (most 64-voxel tiles empty, DSPMAP_P_SPARSE_SWEEP): 1
                                       (default) = the three sweeps (prediction, placement, resampling) learn that a tile has nothing for them from one BIT per
                                       tile -- tables of a few kB that stay in the scalar cache, rebuilt from the per-tile flags by the frame's first kernel
                                       and extended by whoever puts the first particle into an empty tile -- instead of from the tile's own words in
                                       memory: an empty tile's workgroup leaves as fast as the next one can be started (264x264x80 filled by the depth
                                       stream: 87 120 tiles, ~12 k with particles); 0 = every workgroup loads its tile's flags (rounds 3-5).  Same result */
    DSPMAP_P_PAIR_CULL_SIGMAS = 13  /* mapUpdate evaluates a (particle, observation) pair only if their ranges differ by at most this many
                                       sigma_ob (default 9: the dropped terms are < 1e-19 and zero on the fixed-point Ck grid);
                                       a huge value evaluates every pair of the neighbourhood like the reference's loops */
};

/* ---- lifecycle: DSPMap::DSPMap / ~DSPMap  dsp_dynamic.h:145-179 ---- */
void dspmap_default_config(dspmap_config* cfg);          /* the reference's shipped macro values */
dspmap_t* dspmap_create(const dspmap_config* cfg);        /* no HIP call: safe during static init (src/map_sim_example.cpp:39) */
void dspmap_destroy(dspmap_t* m);
int dspmap_init_device(dspmap_t* m);                      /* allocate device state now (otherwise lazily on first use) */
const char* dspmap_last_error(const dspmap_t* m);
int dspmap_sync(dspmap_t* m);                             /* wait for all queued work of this handle */
int dspmap_set_stream(dspmap_t* m, void* hip_stream);     /* run on a caller-owned hipStream_t (e.g. torch's) */

/* ---- setters: dsp_dynamic.h:355-382 ---- */
int dspmap_set_param(dspmap_t* m, int key, double value);
double dspmap_get_param(const dspmap_t* m, int key);

/* ---- randomness.  The reference pre-draws two N(0,sigma) tables
 * (dsp_dynamic.h:138-139,1150-1160) and uses libc rand() for uniform
 * velocities (:1551-1553).  Tables can be injected so that runs can be
 * compared with another implementation fed the same tables. ---- */
int dspmap_set_gaussian_tables(dspmap_t* m, const float* p_tab_host, const float* v_tab_host, int n);
int dspmap_set_rand_table(dspmap_t* m, const int* rand_ints_host, int n); /* values in [0, RAND_MAX] */
int dspmap_set_cursors(dspmap_t* m, int p_cursor, int v_cursor, int r_cursor);
int dspmap_get_cursors(dspmap_t* m, int* p_cursor, int* v_cursor, int* r_cursor);

/* ---- the frame: DSPMap::update  dsp_dynamic.h:181-353 ----
 * Same arguments and the same 1 / 0 contract (0 = invalid quaternion or
 * |dp| > 10 m or dt outside [0,10] s; state and the "last pose" untouched). */
int dspmap_update(dspmap_t* m, int point_cloud_num, int size_of_one_point, const float* point_cloud_host,
                  float sensor_px, float sensor_py, float sensor_pz, double time_stamp_second,
                  float qw, float qx, float qy, float qz);

/* Same frame with inputs already resident in HBM: `points_dev` = n x 3 floats
 * (sensor frame, packed xyz); `birth_dev`/n_birth = the birth-source cloud
 * (what the velocity estimator would output) or NULL/0 = every in-FOV point is
 * a static source (zero velocity tag) -- unless DSPMAP_P_VELOCITY_ESTIMATOR is set:
 * 2 = the device estimator tags the cloud inside the captured frame (still
 * asynchronous); 1 = the cloud makes one round trip to the host estimator (D2H of
 * <= 60 kB, clustering + matching while the prediction and weight kernels run, H2D
 * of the tagged cloud).  Asynchronous otherwise: returns after enqueue. */
int dspmap_update_device(dspmap_t* m, int n_points, const float* points_dev, int n_birth,
                         const dspmap_vpoint* birth_dev, const float sensor_pos[3],
                         double time_stamp_second, const float quat_wxyz[4]);

/* Supply the birth-source cloud used by the next dspmap_update /
 * dspmap_stage_birth when the velocity estimator is off (what the reference's
 * velocityEstimationThread leaves in input_cloud_with_velocity, :134). */
int dspmap_set_birth_cloud(dspmap_t* m, const dspmap_vpoint* pts_host, int n);
/* getKMClusterResult :441-445 : the birth-source cloud of the last frame */
int dspmap_get_birth_cloud(dspmap_t* m, dspmap_vpoint* out_host, int cap, int* n_out);

/* ---- readout: dsp_dynamic.h:385-438 ---- */
/* getOccupancyMap :385-402 : voxel centres with occupancy mass > thr, ascending voxel
 * index; ALSO zeroes the future accumulators like the reference (:397-400). */
int dspmap_get_occupancy(dspmap_t* m, float threshold, float* xyz_out_host, int cap, int* n_out);
/* getOccupancyMapWithFutureStatus :405-426 : as above + copies V x T future masses, then zeroes them */
int dspmap_get_occupancy_with_future(dspmap_t* m, float threshold, float* xyz_out_host, int cap, int* n_out,
                                     float* future_status_host /* [V_local][T] */);
/* north-star name getFutureStatus(): the V x T copy + zeroing only */
int dspmap_get_future(dspmap_t* m, float* future_status_host);
/* clearOccupancyMapPrediction :431-438 */
int dspmap_clear_future(dspmap_t* m);
/* voxels_objects_number[v][0..3] (:118-120): occupancy mass + mean velocity, V_local x 4 floats */
int dspmap_get_results(dspmap_t* m, float* out_host);
/* device-resident views for callers that keep the map on the GPU (no copy) */
const float* dspmap_results_device(dspmap_t* m); /* [V_local][4] */
const float* dspmap_future_device(dspmap_t* m);  /* [V_local][T] */

/* getVoxelPositionFromIndexPublic :1556-1572 / getPointVoxelsIndexPublic :1574-1584 (host math) */
void dspmap_voxel_center(const dspmap_t* m, int index, float* px, float* py, float* pz);
int dspmap_point_voxel_index(const dspmap_t* m, float px, float py, float pz, int* index);

/* ---- sizes ---- */
int dspmap_voxel_num(const dspmap_t* m);        /* global VOXEL_NUM :62 */
int dspmap_local_voxel_num(const dspmap_t* m);  /* voxels in this handle's slab */
int dspmap_local_voxel_base(const dspmap_t* m); /* global index of the slab's first voxel (0 for an unsharded map) */
int dspmap_slots_per_voxel(const dspmap_t* m);  /* SAFE_PARTICLE_NUM_VOXEL :65 */
int dspmap_pyramid_num(const dspmap_t* m);      /* observation_pyramid_num :60 */
int dspmap_pyramid_capacity(const dspmap_t* m); /* SAFE_PARTICLE_NUM_PYRAMID :66 */
int dspmap_get_counters(dspmap_t* m, dspmap_counters* out);

/* per-stage device timing (HIP events on the handle's stream around each kernel group).
 * Off by default; when on, every update records events and the elapsed times accumulate.
 * Stages: 0 setup+binning, 1 predict, 2 claim(movers), 3 Ck partial, 4 weight update,
 *         5 Ck sum (birth normaliser), 6 birth, 7 occupancy+resample. */
#define DSPMAP_N_STAGES 8
int dspmap_set_profiling(dspmap_t* m, int on);
int dspmap_get_stage_ms(dspmap_t* m, float ms_sum_out[DSPMAP_N_STAGES], int* n_frames_out); /* sums since enabling; syncs */
/* what one event bracket adds to the kernel inside it (record cost + launch gaps), calibrated when profiling is switched on with a
 * kernel of known duration: stage time - this = the kernel's own duration for the stages that are one launch */
int dspmap_get_event_overhead_ms(dspmap_t* m, float* ms_out);

/* profiling aid: streams the six particle field arrays once with the sweeps' access pattern
 * (4 B per lane, 256 B per wave) -- mode 0 reads them (known byte count = 6*4*capacity), mode 1
 * rewrites px in place -- to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for this pattern. */
int dspmap_debug_stream(dspmap_t* m, int mode, long long* bytes_out);
/* profiling aid: the bare memory skeleton of the prediction sweep -- one workgroup per 64-voxel tile, the first `rows` slot
 * rows of every tile, nothing computed -- timed with events over `reps` launches: what the memory system sustains for
 * this access pattern.  what: bit 0 read positions (12 B), bit 1 read velocities (8 B), bit 2 read weights (4 B),
 * bit 3 write positions back (12 B); rows_per_batch = rows a wave keeps in flight.  *bytes_out = bytes per launch. */
int dspmap_debug_sweep_probe(dspmap_t* m, int what, int rows, int rows_per_batch, int reps, float* ms_out, long long* bytes_out);
/* diagnostics: out[t] = 1 if a particle inside 64-voxel tile t could lie in the field of view of the last frame
 * (the conservative box test behind DSPMAP_P_PLACE_SPLIT_TILES); returns the number of tiles or an error */
int dspmap_debug_tile_view(dspmap_t* m, int* out, int cap);
/* diagnostics: the number of 64-voxel tiles of this handle's storage, and the tile each of n voxels (GLOBAL indices of the reference, :1081) lives
 * in (-1: not in this handle's slab).  A tile is a run of 64 voxel indices or a cube of 4 x 4 x 4 voxels (DSPMAP_P_TILING) */
int dspmap_debug_tile_count(dspmap_t* m);
int dspmap_debug_tile_of_voxels(dspmap_t* m, int n, const int* voxel_global_host, int* tile_out_host);
/* diagnostics: out[t] = the "somebody moves" flag of 64-voxel tile t: 0 = every live particle of the tile had velocity (0, 0)
 * when k_predict last swept it and nothing with a velocity has arrived or been born there since -- the sweeps of such a tile do
 * not fetch its velocity rows (k_predict).  Returns the number of tiles or an error */
int dspmap_debug_tile_moving(dspmap_t* m, int* out, int cap);
/* diagnostics: which kernels the last resampling stage ran and which path the future-status contributions took:
 * out[0] = bit 0: four-waves-per-tile resampler; bits 1-2: rollout 0 inside the resampler, 1 k_rollout without LDS windows,
 * 2 k_rollout with LDS windows, 3 none; out[1] / out[2] = contributions k_rollout sent through its windows / as single atomics */
int dspmap_debug_rollout_paths(dspmap_t* m, long long out[3]);
/* diagnostics of DSPMAP_P_ESTIMATOR_QUEUE: out[0] = frames of this handle whose velocity estimator ran on a queue of its own, out[1] / out[2] = the
 * two hand-over words (ring position + 1 of the last frame whose birth stage has ended / whose birth cloud the estimator has finished),
 * out[3] = nonzero if a cross-queue wait ever gave up, out[4] = frames whose first birth kernel found the birth cloud unfinished (its workgroup 0
 * waited, the other workgroups left their shares to it), out[5] = shares it did for them */
int dspmap_debug_estimator_queue(dspmap_t* m, long long out[6]);
/* where the last frame with the device velocity estimator ran it: 1 = on a stream of its own (DSPMAP_P_ESTIMATOR_QUEUE), 2 = as a forked branch of the
 * captured frame because no stream apart from the main stream's hardware queue was found (the fallback), 3 = as a forked branch (switched off, a map
 * that splits its placement, or after a cross-queue wait gave up), 0 = no such frame yet.  Test hooks, read at dspmap_create: DSPMAP_XQ_FORCE=shared
 * (every candidate stream counts as sharing the queue: the fallback runs), DSPMAP_XQ_FORCE=apart (fail instead of falling back) */
int dspmap_debug_estimator_path(dspmap_t* m);
/* diagnostics of DSPMAP_P_FRAME_BRANCHES: out[0] = frames of this handle that ran as two branches, out[1] / out[2] = tiles of class Q (a newborn of
 * the last such frame could land there) / P (a particle could reach a Q tile), out[3] = tiles of the map, out[4] = the largest speed any particle
 * of the map was ever given, mm/s (what sizes P) */
int dspmap_debug_frame_branches(dspmap_t* m, long long out[5]);
/* frames of this handle whose resampling stage ran as two launches (DSPMAP_P_RESAMPLE_SPLIT) */
long long dspmap_debug_resample_split_frames(dspmap_t* m);
/* test hooks of dspmap_mgpu_comm_init_from_env's rendezvous file (no device, no RCCL): what rank 0 publishes / what a rank != 0
 * waits for (this launch's nonce: DSPMAP_RDZV_NONCE or TORCHELASTIC_RUN_ID + the parent's pid).  1 = written / found, 0 = not */
int dspmap_debug_rdzv_publish(const char* path, const char id[128]);
int dspmap_debug_rdzv_wait(const char* path, int timeout_ms, char id_out[128]);

/* ---- state access (the reference's equivalent is direct access to its
 * file-scope arrays, dsp_dynamic.h:116).  A record is 8 floats
 * {flag, vx, vy, vz, px, py, pz, weight} (:114-115 minus the dead update_time).
 * `voxel` is the GLOBAL voxel index; slot < 0 = first free slot (:1184-1185). ---- */
int dspmap_clear_state(dspmap_t* m);
int dspmap_import_state(dspmap_t* m, int n, const int* voxel_host, const int* slot_host, const float* rec8_host);
int dspmap_export_state(dspmap_t* m, int cap, int* voxel_out_host, int* slot_out_host, float* rec8_out_host, int* n_out);
/* ---- binary checkpoint / restore (the reference has none).  Saves every live particle with its
 * slot, the function statics of update() and of the birth stage, the table cursors, the result grid and the future
 * accumulators; the random tables come from the configuration's seed (or are re-injected by the caller).  Loading
 * requires a handle created with the same configuration.  The host velocity estimator's previous clusters are not
 * saved: the first frame after a restore matches no cluster, like the first frame of a run. */
int dspmap_save_checkpoint(dspmap_t* m, const char* path);
int dspmap_load_checkpoint(dspmap_t* m, const char* path);
/* ---- caller-side pre-processing on the device (next to the hot path; reference src/map_sim_example.cpp:309-336)
 * points_dev: n points, xyz first, stride_floats floats apart, device memory, in the frame the sensor driver
 * delivers (swap_axes = 1: camera optical frame, mapped x = z, y = -x, z = -y like :321-323; 0: already x forward).
 * Voxel-grid centroid filter with leaf size `leaf` (pcl::VoxelGrid, :313-317), crop to the open map box (:325),
 * at most max_points points in the filter's output order (:332) -> out_dev (max_points x 3 floats, device memory,
 * ready for dspmap_update_device).  *n_out = points written, *n_leaves_out (optional) = occupied leaves that touch
 * the map box.  Non-finite points are ignored (PCL does the same for non-dense clouds).  Only leaves touching the
 * map box are accumulated (the others cannot survive the crop), so the cost does not depend on far returns;
 * refuses leaf sizes that put more than 2^27 leaves into the map box. */
int dspmap_preprocess_cloud(dspmap_t* m, int n, const float* points_dev, int stride_floats, float leaf, int swap_axes,
                            int max_points, float* out_dev, int* n_out, int* n_leaves_out);

/* addRandomParticles :594-624 (constructor pre-fill); uses the rand table */
int dspmap_add_random_particles(dspmap_t* m, int n, float weight);
/* benchmark fill: every voxel gets `per_voxel` zero-velocity particles,
 * uniform in-voxel positions, given weight; generated on device from `seed`. */
int dspmap_seed_uniform(dspmap_t* m, int per_voxel, float weight, unsigned seed);
/* same fill with velocities uniform in [-vmax, vmax] (benchmark of the future-status rollout) */
int dspmap_seed_uniform_moving(dspmap_t* m, int per_voxel, float weight, unsigned seed, float vmax);

/* ---- single stages on the current state (test hooks; the reference's
 * stages are private members made reachable the same way by its own author
 * for mapAddNewBornParticlesByObservation, :795-796) ---- */
int dspmap_stage_bin_points(dspmap_t* m, int n, int stride, const float* pts_host,
                            float qw, float qx, float qy, float qz);                  /* :220-293 */
int dspmap_set_current_position(dspmap_t* m, float x, float y, float z);             /* :213-215 */
int dspmap_stage_predict(dspmap_t* m, float odom_dx, float odom_dy, float odom_dz, float dt); /* mapPrediction :627 */
int dspmap_stage_update(dspmap_t* m);                                                /* mapUpdate :704 */
int dspmap_stage_birth(dspmap_t* m);                                                 /* mapAddNewBornParticlesByObservation :796 */
int dspmap_stage_resample(dspmap_t* m);                                              /* mapOccupancyCalculationAndResample :924 */
/* observation bins after binning / update: xyz+Ck+len per stored obs (:497-501,514-515) */
int dspmap_get_observations(dspmap_t* m, float* obs_out_host /* [NP][100][5] */, int* count_out_host /* [NP] */,
                            float* max_len_out_host /* [NP] */, float* expected_newborn_out);
int dspmap_set_expected_newborn(dspmap_t* m, float v);
/* particles registered per pyramid by the last prediction (size of each pyramids_in_fov list, :124) */
int dspmap_get_pyramid_counts(dspmap_t* m, int* count_out_host /* [NP] */);

/* ---- multi-GPU split-phase frame (Z-slab sharding; the single-process reference has no
 * counterpart).  One process per GPU owns the voxel layers [z_lo, z_hi) (dspmap_config).  Every
 * rank is fed the same cloud and pose; per frame the caller runs
 *     begin -> export(+1), export(-1) [or export_both] -> (place_interior) -> [send to rank+1 / rank-1] -> import
 *           -> ck_partial (places the movers, imported ones included, then the Ck pass) -> [all-reduce SUM over the bound Ck buffer]
 *           -> weights_and_split -> [all-reduce MAX over the bound n_static buffer]
 *           -> finish
 * and issues the collectives itself (RCCL through torch.distributed).  What crosses slabs:
 *  (1) particles whose new voxel lies in another slab after prediction (vz == 0, so only the
 *      ego-motion's z component moves particles across layers, dsp_dynamic.h:661-667);
 *  (2) the per-observation sums Ck, because pyramids cut across slabs (:709-735);
 *  (3) n_static of each birth source, known only to the rank owning the source's voxel (:827-866).
 * Records are 8 floats {global voxel index (int bits), vx, vy, px, py, pz, w, source key (int bits)}: the key
 * (source voxel * slots + slot) orders the arrivals of a voxel as the reference's sequential sweep would serve them;
 * imported records are placed together with the slab's own movers (the placement pass runs after the import), so
 * a sharded map fills exactly the slots of the unsharded one.
 * The Ck buffer holds 64-bit fixed-point sums (units of 2^-34): all-reduce it as int64.  Integer addition is
 * associative, so a sharded frame computes exactly the Ck of the unsharded one and frames are reproducible. */
int dspmap_mgpu_bind(dspmap_t* m, long long* ck_dev /* [NP*100] int64 */, int* nstatic_dev, int nstatic_cap);
/* optional, between the exports and the imports: places the movers of the slab's interior (the tiles no record of a
 * neighbour can reach this frame) so that the GPU works while the caller sizes the exchange on the host */
int dspmap_mgpu_place_interior(dspmap_t* m);
int dspmap_mgpu_begin(dspmap_t* m, int n_points, const float* points_dev, int n_birth,
                      const dspmap_vpoint* birth_dev, const float sensor_pos[3],
                      double time_stamp_second, const float quat_wxyz[4]);   /* 1 / 0 like dspmap_update */
int dspmap_mgpu_export(dspmap_t* m, int dir /* +1 through z_hi, -1 through z_lo */, float* rec_dev_out,
                       int cap, int* n_out);                                 /* synchronises */
/* stream-ordered variant of the two exports: no host synchronisation; counts_dev[0] = records written to
 * up_dev_out, counts_dev[1] = to down_dev_out (device memory).  A count above `cap` means the buffer was too small
 * (the surplus particles are lost): the caller must check after reading the counts, then report them with
 * dspmap_mgpu_set_export_counts (statistics only). */
int dspmap_mgpu_export_both(dspmap_t* m, float* up_dev_out, float* down_dev_out, int cap, int* counts_dev);
int dspmap_mgpu_set_export_counts(dspmap_t* m, int n_up, int n_down);
int dspmap_mgpu_import(dspmap_t* m, int n, const float* rec_dev);
int dspmap_mgpu_ck_partial(dspmap_t* m);
int dspmap_mgpu_weights_and_split(dspmap_t* m);
int dspmap_mgpu_finish(dspmap_t* m);

/* ---- the same frame driven from C++ (dspmap_dist.hip): one call per frame and rank, the collectives are issued by the
 * library on its own stream through RCCL (dlopen of librccl.so at communicator set-up), no host synchronisation inside a
 * frame.  Exchange of the boundary particles = fixed-size ncclSend / ncclRecv pairs with rank +- 1 whose first record
 * is a header carrying the record count; the size is the largest export of an earlier frame (all ranks) + 50 % + 1024,
 * agreed through one extra slot of the n_static all-reduce (MAX); particles that cross more than one slab are forwarded
 * in further rounds.  Ck: ncclAllReduce(SUM, int64); n_static: ncclAllReduce(MAX, int32).
 *   rank 0:      dspmap_mgpu_get_unique_id(id)   -> hand `id` to the other ranks (MPI, torch.distributed, a file, ...)
 *   every rank:  dspmap_mgpu_comm_init(m, world, rank, id)       [or dspmap_mgpu_comm_init_from_env: RANK / WORLD_SIZE +
 *                                                                 a rendezvous file, DSPMAP_RDZV_FILE]
 *   per frame:   dspmap_mgpu_update(m, ...)                       1 / 0 like dspmap_update; same cloud and pose on every rank
 * The handle's configuration carries the rank's slab [z_lo, z_hi).  dspmap_mgpu_update returns DSPMAP_E_STATE once, a
 * frame late, if a frame's export did not fit the message (vertical step much larger than the frames before). */
#define DSPMAP_UNIQUE_ID_BYTES 128
int dspmap_mgpu_get_unique_id(char id_out[DSPMAP_UNIQUE_ID_BYTES]);
int dspmap_mgpu_comm_init(dspmap_t* m, int world, int rank, const char id[DSPMAP_UNIQUE_ID_BYTES]);
int dspmap_mgpu_comm_init_from_env(dspmap_t* m);
int dspmap_mgpu_comm_destroy(dspmap_t* m);
int dspmap_mgpu_update(dspmap_t* m, int n_points, const float* points_dev, int n_birth, const dspmap_vpoint* birth_dev,
                       const float sensor_pos[3], double time_stamp_second, const float quat_wxyz[4]);
int dspmap_mgpu_update_host(dspmap_t* m, int point_cloud_num, int size_of_one_point, const float* point_cloud_ptr,
                            float sensor_px, float sensor_py, float sensor_pz, double time_stamp_second,
                            float qw, float qx, float qy, float qz);
int dspmap_mgpu_message_records(const dspmap_t* m);   /* records the next frame's exchange messages carry */
/* the same driver over several slabs inside ONE process (tests, single-GPU debugging): device-to-device copies and small
 * reduction kernels stand in for the collectives; the handles share the first one's stream */
int dspmap_mgpu_group_create(dspmap_t** handles, int n);
int dspmap_mgpu_group_update(dspmap_t** handles, int n, int n_points, const float* points_dev, int n_birth,
                             const dspmap_vpoint* birth_dev, const float sensor_pos[3], double time_stamp_second,
                             const float quat_wxyz[4]);

/* per-slab, per-phase device time of the group's frames (HIP events around every phase of every slab; slab index n = the group's
 * stand-ins for the collectives).  Inside the group every slab has the GPU to itself for the length of its phase -- what its own
 * GPU would spend on it in a one-process-per-GPU run -- so  sum over phases of max over slabs  is the frame's critical path on n GPUs
 * without the transport: the figure bench.py reports as projected_8gpu (a projection; N > 1 has not run on hardware).
 * phases: 0 begin (binning, prediction, estimator, export), 1 exchange + import, 2 placement, 3 list selection (only frames that
 * run it; group level), 4 list preparation + Ck, 5 weights + split, 6 births + resampling + rollout.
 * out: [n + 1][DSPMAP_GROUP_PHASES] summed ms since enabling */
#define DSPMAP_GROUP_PHASES 7
int dspmap_mgpu_group_set_profiling(dspmap_t** handles, int n, int on);
int dspmap_mgpu_group_get_phase_ms(dspmap_t** handles, int n, float* ms_sum_out, int* n_frames_out);

#ifdef __cplusplus
}
#endif
#endif /* DSPMAP_H */
