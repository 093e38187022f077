/*
 * dsp_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the per-frame loop of g-ch/DSP-map's
 * include/dsp_dynamic.h (reference tag 2024_10_08).  Every function in
 * dsp_oracle.c cites the reference file:line it follows.
 *
 * PARITY UNPINNED: the reference header cannot be compiled in this image
 * (it needs Eigen, PCL and munkres-cpp, all absent; writing stand-ins for them
 * is not allowed) and the reference ships no tests / golden vectors.  The
 * oracle is pinned only against the handful of known answers SURVEY.md records
 * from the real reference ([probe] values: PDF LUT centre value, neighbour
 * table rows, pyramid-index formula, CAPP sizes) -- see tests/test_oracle_kat.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.  The product path (libdspmap_hip.so) never does.
 *
 * Unlike the reference (compile-time macros, file-scope static arrays, one map
 * per process) the oracle is sized at run time so that every BASELINE.json
 * configuration is one binary; storage layout, sweep order and arithmetic
 * follow the reference (dense AoS slots, first-free-slot allocation, LUT pdf).
 */
#ifndef DSP_ORACLE_H
#define DSP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define DSPO_MAX_PRED_TIMES 16
#define DSPO_OBS_MAX_PER_PYRAMID 100 /* dsp_dynamic.h:69 */
#define DSPO_PDF_LUT_SIZE 20000      /* dsp_dynamic.h:140 */

typedef struct dspo_config {
    int nx, ny, nz;              /* MAP_LENGTH/WIDTH/HEIGHT_VOXEL_NUM  dsp_dynamic.h:38-40 */
    float voxel_resolution;      /* VOXEL_RESOLUTION                   :41 */
    int angle_resolution;        /* ANGLE_RESOLUTION (deg)             :42 */
    int max_particle_num_voxel;  /* MAX_PARTICLE_NUM_VOXEL             :43 */
    int half_fov_h, half_fov_v;  /* :49-50 (deg) */
    int prediction_times;        /* PREDICTION_TIMES                   :46 */
    float prediction_future_time[DSPO_MAX_PRED_TIMES]; /* :47 */
    /* variants of the reference as run-time parameters; 0 = the dsp_dynamic.h value */
    int pyramid_neighbor_n;      /* PYRAMID_NEIGHBOR_N of dsp_dynamic_multiple_neighbors.h:43 (there 2); default 1 = 3x3 */
    int safe_particle_factor;    /* SAFE_PARTICLE_NUM_VOXEL / MAX_PARTICLE_NUM_VOXEL: 2 (:65), 5 in dsp_static.h:63 */
    int static_model;            /* 1 = dsp_static.h's motion model: velocities forced to 0 in prediction and birth */
} dspo_config;

/* source point handed to the birth stage = reference's input_cloud_with_velocity
 * entry (pcl::PointXYZINormal; dsp_dynamic.h:134,1510-1539): world xyz,
 * normal = velocity estimate (-10000 sentinel when unmatched, :104-106),
 * intensity = cluster tag (0 for static/ground). */
typedef struct dspo_vpoint {
    float x, y, z;
    float nx, ny, nz;
    float intensity;
} dspo_vpoint;

typedef struct dsp_oracle dsp_oracle;

void dspo_default_config(dspo_config* c); /* the reference's shipped macros (66x66x40, 0.15, 3deg, 9 ppv, T=6) */

dsp_oracle* dspo_create(const dspo_config* cfg);
void dspo_destroy(dsp_oracle* o);

/* ---- derived sizes (dsp_dynamic.h:58-66) ---- */
int dspo_voxel_num(const dsp_oracle* o);
int dspo_slots_per_voxel(const dsp_oracle* o);     /* SAFE_PARTICLE_NUM_VOXEL   */
int dspo_pyramid_num(const dsp_oracle* o);         /* observation_pyramid_num   */
int dspo_pyramid_capacity(const dsp_oracle* o);    /* SAFE_PARTICLE_NUM_PYRAMID */
int dspo_result_dim(const dsp_oracle* o);          /* 4 + PREDICTION_TIMES      */

/* ---- setters (dsp_dynamic.h:355-382). setPredictionVariance does NOT
 * regenerate tables here: tables are injected (see below). ---- */
void dspo_set_prediction_variance(dsp_oracle* o, float p_stddev, float v_stddev);
void dspo_set_observation_stddev(dsp_oracle* o, float s);
void dspo_set_newborn_weight(dsp_oracle* o, float w);
void dspo_set_newborn_number(dsp_oracle* o, int n);
void dspo_set_voxel_filter_resolution(dsp_oracle* o, float r);

/* ---- randomness.  The reference draws N(0,sigma) from two 10M-entry tables
 * filled at construction from default_random_engine(time(NULL))
 * (dsp_dynamic.h:1150-1160) and uniforms from libc rand() (:1551-1553).  The
 * oracle takes the tables from the caller (borrowed pointers, must outlive the
 * oracle) so runs are reproducible; `rand_ints` replaces the rand() stream
 * (values in [0, RAND_MAX]); if never set, libc rand() is used. ---- */
void dspo_set_gaussian_tables(dsp_oracle* o, const float* p_tab, const float* v_tab, int n);
void dspo_set_rand_table(dsp_oracle* o, const int* rand_ints, int n);
void dspo_set_cursors(dsp_oracle* o, int p_cursor, int v_cursor, int r_cursor);
void dspo_get_cursors(const dsp_oracle* o, int* p_cursor, int* v_cursor, int* r_cursor);
/* fills tables the way the reference does (libstdc++ minstd_rand0 +
 * normal_distribution<double>) is C++ only; see oracle/gauss_tables.cpp */

/* ---- whole frame: DSPMap::update  dsp_dynamic.h:181-353 ----
 * If `use_velocity_estimator` is 0 the caller must have supplied the birth
 * source cloud with dspo_set_birth_cloud() (what the reference's velocity
 * thread would have produced); otherwise the restated estimator runs. */
int dspo_update(dsp_oracle* o, int n_pts, int stride, const float* pts,
                float sx, float sy, float sz, double stamp,
                float qw, float qx, float qy, float qz);
void dspo_use_velocity_estimator(dsp_oracle* o, int mode); /* 0 caller's cloud, 1 restated estimator, 2 all-static tags in view order */
void dspo_static_birth_cloud(dsp_oracle* o);
void dspo_set_birth_cloud(dsp_oracle* o, const dspo_vpoint* pts, int n);
int dspo_get_birth_cloud(const dsp_oracle* o, dspo_vpoint* out, int cap);

/* ---- stages, callable one by one on injected state ---- */
/* obs binning part of update(): dsp_dynamic.h:220-293.  Sets the rotated
 * boundary planes, bins, counters, expected_new_born_objects. Returns valid_points. */
int dspo_bin_points(dsp_oracle* o, int n_pts, int stride, const float* pts,
                    float qw, float qx, float qy, float qz);
void dspo_set_current_position(dsp_oracle* o, float x, float y, float z); /* current_position[] :213-215 */
void dspo_map_prediction(dsp_oracle* o, float dx, float dy, float dz, float dt); /* :627-701 */
void dspo_map_update(dsp_oracle* o);                                            /* :704-793 */
void dspo_add_newborn(dsp_oracle* o);                                           /* :796-921 */
/* halves of mapUpdate + n_static hooks: test infrastructure for the Z-slab sharding test */
void dspo_map_update_ck(dsp_oracle* o);      /* pass 1 (:709-735) without the constant of :737 */
void dspo_map_update_weights(dsp_oracle* o); /* += lambda+kappa (:737), then pass 2 (:743-790) */
void dspo_compute_nstatic(dsp_oracle* o, int* out);
void dspo_set_nstatic_override(dsp_oracle* o, const int* arr);
void dspo_occupancy_resample(dsp_oracle* o);                                    /* :924-1057 */
void dspo_velocity_estimation(dsp_oracle* o);                                   /* :1377-1544 */

/* ---- readout: dsp_dynamic.h:385-438 ---- */
int dspo_get_occupancy_map(dsp_oracle* o, float thr, float* xyz_out, int cap);
int dspo_get_occupancy_map_with_future(dsp_oracle* o, float thr, float* xyz_out, int cap, float* future_VxT);
void dspo_clear_future(dsp_oracle* o);

/* ---- primitives exposed for known-answer tests ---- */
float dspo_query_normal_pdf(const dsp_oracle* o, float x, float mu, float sigma); /* :1294-1301 */
const float* dspo_pdf_lut(const dsp_oracle* o);                                   /* :1288-1292 */
void dspo_rotate_vector(const float v[3], const float q_wxyz[4], float out[3]);   /* :1303-1322 */
int dspo_in_pyramids_area(const dsp_oracle* o, float x, float y, float z);        /* :1329-1339 */
int dspo_pyramid_h(const dsp_oracle* o, float x, float y, float z);               /* :1341-1353 */
int dspo_pyramid_v(const dsp_oracle* o, float x, float y, float z);               /* :1355-1367 */
int dspo_voxel_index(const dsp_oracle* o, float x, float y, float z, int* idx);   /* :1076-1088 */
void dspo_voxel_center(const dsp_oracle* o, int idx, float* x, float* y, float* z); /* :1090-1107 */
const int* dspo_neighbor_table(const dsp_oracle* o);  /* [NP][10], :126-127,1128-1147 */
float dspo_generate_random_float(dsp_oracle* o, float lo, float hi);              /* :1551-1553 */
void dspo_add_random_particles(dsp_oracle* o, int n, float w);                    /* :594-624 */
/* test helper: first-free-slot injection of n particles with a flag each (flags) or one for all (flag); returns the number placed */
int dspo_inject(dsp_oracle* o, int n, const float* px, const float* py, const float* pz, const float* vx, const float* vy,
                const float* vz, const float* w, const float* flags, float flag);

/* ---- raw state (the reference's file-scope arrays) ---- */
float* dspo_particles(dsp_oracle* o);       /* [V][SLOTS][9]  voxels_with_particle :116 */
float* dspo_results(dsp_oracle* o);         /* [V][4+T]       voxels_objects_number :120 */
int* dspo_pyramid_lists(dsp_oracle* o);     /* [NP][CAPP][3]  pyramids_in_fov :124 */
float* dspo_obs(dsp_oracle* o);             /* [NP][100][5]   point_cloud :498 */
int* dspo_obs_count(dsp_oracle* o);         /* [NP]           observation_num_each_pyramid :501 */
float* dspo_obs_max_length(dsp_oracle* o);  /* [NP]           point_cloud_max_length :515 */
float dspo_expected_newborn(const dsp_oracle* o); /* expected_new_born_objects :292 */
void dspo_set_expected_newborn(dsp_oracle* o, float v);
void dspo_set_occlusion_margin(dsp_oracle* o, float v); /* obstacle_thickness_for_occlusion :70 (0.3); the variants use voxel_resolution */
float dspo_update_time(const dsp_oracle* o);
int dspo_count_live(const dsp_oracle* o);

/* ---- caller-side pre-processing (src/map_sim_example.cpp:309-336; pcl::VoxelGrid restated, parity unpinned) ----
 * returns the number of points written to out (<= max_points); *n_leaves = occupied leaves before the crop */
int dspo_preprocess_cloud(int n, const float* pts, int stride, float leaf, int swap_axes, float hx, float hy, float hz,
                          int max_points, float* out, int* n_leaves);

/* ---- test hooks for the restated third-party algorithms of the velocity estimator (checked against independent
 * implementations by tests/test_oracle_kat.py) ----
 * munkres-cpp: minimum-cost assignment of an nr x nc matrix, assign[r] = column or -1 (:1474-1481 reads only which cells are assigned) */
void dspo_hungarian(const float* cost, int nr, int nc, int* assign);
/* pcl::EuclideanClusterExtraction (:1406-1417): label[i] = rank of point i's cluster by size (0 = largest) or -1; returns the cluster count */
int dspo_euclidean_clusters(const float* pts, int n, float tol, int min_sz, int max_sz, int* label);

#ifdef __cplusplus
}
#endif
#endif
