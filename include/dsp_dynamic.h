/*
 * dsp_dynamic.h -- drop-in replacement for g-ch/DSP-map's include/dsp_dynamic.h:
 * the same `class DSPMap` public surface (reference file:line cited per member),
 * forwarding to the MI355X-native library libdspmap_hip.so through the C ABI in
 * dspmap.h.  A caller written against the reference header -- e.g. its
 * src/map_sim_example.cpp -- compiles against this one unchanged and links with
 * -ldspmap_hip.
 *
 * What is kept for source compatibility (all used by src/map_sim_example.cpp):
 *   - the configuration macros MAP_LENGTH_VOXEL_NUM ... PREDICTION_TIMES and the
 *     constants VOXEL_NUM, prediction_future_time[] (:38-47,62); unlike the
 *     reference they can be overridden with -D (the library is sized at run time)
 *   - `using namespace std;` (:35) and the transitive Eigen / PCL includes (:27-31)
 *     when those libraries are installed; without PCL a minimal pcl::PointCloud
 *     is provided so that the header is usable on machines without ROS
 *   - construction during static initialisation is safe (:39 of the example):
 *     no HIP call happens before the first update()
 * New: getFutureStatus() (the name BASELINE.json's north_star uses) and
 * `dspmap_handle()` for callers that want the device-resident views of dspmap.h.
 */
#ifndef DSP_DYNAMIC_H_MI355X
#define DSP_DYNAMIC_H_MI355X

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#if defined(__has_include)
#if __has_include("Eigen/Eigen")
#include "Eigen/Eigen"
#endif
#if __has_include(<pcl/point_types.h>)
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#define DSPMAP_HAVE_PCL 1
#endif
#endif

#include "dspmap.h"

using namespace std;  // the reference header does this (:35) and its example relies on it

#ifndef DSPMAP_HAVE_PCL
namespace pcl {  // minimal containers with the members the DSPMap interface touches
struct PointXYZ { float x = 0, y = 0, z = 0; };
struct PointXYZINormal { float x = 0, y = 0, z = 0, intensity = 0, normal_x = 0, normal_y = 0, normal_z = 0; };
template <typename T>
struct PointCloud {
    std::vector<T> points;
    unsigned width = 0, height = 1;
    void push_back(const T& p) { points.push_back(p); width = (unsigned)points.size(); }
    void clear() { points.clear(); width = 0; }
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    T& operator[](size_t i) { return points[i]; }
    const T& operator[](size_t i) const { return points[i]; }
    typename std::vector<T>::iterator begin() { return points.begin(); }
    typename std::vector<T>::iterator end() { return points.end(); }
};
}  // namespace pcl
#endif

/** Parameters for the map (reference :38-47) **/
#ifndef MAP_LENGTH_VOXEL_NUM
#define MAP_LENGTH_VOXEL_NUM 66
#endif
#ifndef MAP_WIDTH_VOXEL_NUM
#define MAP_WIDTH_VOXEL_NUM 66
#endif
#ifndef MAP_HEIGHT_VOXEL_NUM
#define MAP_HEIGHT_VOXEL_NUM 40
#endif
#ifndef VOXEL_RESOLUTION
#define VOXEL_RESOLUTION 0.15
#endif
#ifndef ANGLE_RESOLUTION
#define ANGLE_RESOLUTION 3
#endif
#ifndef MAX_PARTICLE_NUM_VOXEL
#define MAX_PARTICLE_NUM_VOXEL 9
#endif
#ifndef LIMIT_MOVEMENT_IN_XY_PLANE
#define LIMIT_MOVEMENT_IN_XY_PLANE 1
#endif
#ifndef PREDICTION_TIMES
#define PREDICTION_TIMES 6
#define DSPMAP_PREDICTION_TIME_LIST 0.05f, 0.2f, 0.5f, 1.f, 1.5f, 2.f
#elif PREDICTION_TIMES == 1 && !defined(DSPMAP_PREDICTION_TIME_LIST)
#define DSPMAP_PREDICTION_TIME_LIST 0.05f   /* dsp_static.h:46-47 */
#endif
#ifdef DSPMAP_PREDICTION_TIME_LIST
static const float prediction_future_time[PREDICTION_TIMES] = {DSPMAP_PREDICTION_TIME_LIST};
#endif  /* otherwise the including file defines prediction_future_time[PREDICTION_TIMES] before this header */
#if !LIMIT_MOVEMENT_IN_XY_PLANE
#error "libdspmap_hip implements the reference's default LIMIT_MOVEMENT_IN_XY_PLANE 1 (vz == 0)"
#endif

static const int VOXEL_NUM = MAP_LENGTH_VOXEL_NUM * MAP_WIDTH_VOXEL_NUM * MAP_HEIGHT_VOXEL_NUM;  // :62
#ifndef DSPMAP_HALF_FOV_H
#define DSPMAP_HALF_FOV_H 42
#endif
#ifndef DSPMAP_HALF_FOV_V
#define DSPMAP_HALF_FOV_V 24   /* 27 in the reference's two other headers */
#endif
static const int half_fov_h = DSPMAP_HALF_FOV_H;  // :49
static const int half_fov_v = DSPMAP_HALF_FOV_V;  // :50

#ifndef DSPMAP_NO_SAVE_FOLDER_GLOBAL
static string particle_save_folder = ".";  // :55 (one per translation unit here: the reference's non-static global could not be included twice)
#endif

#ifndef DSPMAP_ESTIMATOR_MODE
#define DSPMAP_ESTIMATOR_MODE 2   /* velocityEstimationThread on the device (dspmap_velest.hip); 1 = the host stage.  A sharded map
                                     (-DDSPMAP_WORLD) runs the device estimator on every rank: same cloud, same tagged birth cloud */
#endif

class DSPMap {
public:
    /* :145-175.  init_particle_num > 0 pre-fills the map with random particles (addRandomParticles
     * :594-624) -- deferred to the first device use so that a global `DSPMap my_map;` is safe. */
    DSPMap(int init_particle_num = 0, float init_weight = 0.01f)
        : init_particles_(init_particle_num), init_weight_(init_weight) {
        dspmap_config cfg;
        dspmap_default_config(&cfg);
        cfg.nx = MAP_LENGTH_VOXEL_NUM; cfg.ny = MAP_WIDTH_VOXEL_NUM; cfg.nz = MAP_HEIGHT_VOXEL_NUM;
        cfg.voxel_resolution = (float)VOXEL_RESOLUTION;
        cfg.angle_resolution = ANGLE_RESOLUTION;
        cfg.max_particle_num_voxel = MAX_PARTICLE_NUM_VOXEL;
        cfg.half_fov_h = half_fov_h; cfg.half_fov_v = half_fov_v;
        cfg.prediction_times = PREDICTION_TIMES;
        for (int i = 0; i < PREDICTION_TIMES; i++) cfg.prediction_future_time[i] = prediction_future_time[i];
        // the reference's two other headers are parameter sets of the same kernels:
        //   dsp_dynamic_multiple_neighbors.h: -DANGLE_RESOLUTION=1 -DPYRAMID_NEIGHBOR_N=2 (+ its map macros, :38-51)
        //   dsp_static.h: -DDSPMAP_STATIC_MODEL=1 -DDSPMAP_SAFE_PARTICLE_FACTOR=5 -DPREDICTION_TIMES=1 (:46-47,63,640-646)
        // both test occlusion with VOXEL_RESOLUTION instead of 0.3 m (:761 there): -DDSPMAP_OCCLUSION_MARGIN=VOXEL_RESOLUTION
#ifdef PYRAMID_NEIGHBOR_N
        cfg.pyramid_neighbor_n = PYRAMID_NEIGHBOR_N;
#endif
#ifdef DSPMAP_SAFE_PARTICLE_FACTOR
        cfg.safe_particle_factor = DSPMAP_SAFE_PARTICLE_FACTOR;
#endif
#ifdef DSPMAP_STATIC_MODEL
        cfg.static_model = DSPMAP_STATIC_MODEL;
#endif
#ifdef DSPMAP_WORLD
        // Z-slab sharding across the GPUs of one node (-DDSPMAP_WORLD): one process per GPU, started by any launcher that
        // exports RANK / WORLD_SIZE / LOCAL_RANK (torchrun, mpirun wrappers).  This rank owns the layers [z_lo, z_hi); every
        // rank is fed the same cloud and pose; update() runs the C++ RCCL frame driver (dspmap_mgpu_update_host).  The
        // getters return this rank's slab (voxel indices stay global).
        {
            const char* wr = getenv("WORLD_SIZE"); const char* rk = getenv("RANK"); const char* lr = getenv("LOCAL_RANK");
            world_ = wr ? atoi(wr) : 1; rank_ = rk ? atoi(rk) : 0;
            if (world_ > 1) {
                const int base = cfg.nz / world_, rem = cfg.nz % world_;
                cfg.z_lo = rank_ * base + (rank_ < rem ? rank_ : rem);
                cfg.z_hi = cfg.z_lo + base + (rank_ < rem ? 1 : 0);
            }
            cfg.device = lr ? atoi(lr) : -1;
        }
#endif
        h_ = dspmap_create(&cfg);
        if (h_) dspmap_set_param(h_, DSPMAP_P_VELOCITY_ESTIMATOR, DSPMAP_ESTIMATOR_MODE);  // update() runs the velocity estimator like :297
#ifdef DSPMAP_OCCLUSION_MARGIN
        if (h_) dspmap_set_param(h_, DSPMAP_P_OCCLUSION_MARGIN, (double)(DSPMAP_OCCLUSION_MARGIN));
#endif
        cout << "Map is ready to update!" << endl;  // :174
    }
    ~DSPMap() {  // :177-179
        dspmap_destroy(h_);
        cout << "\n See you ;)" << endl;
    }
    DSPMap(const DSPMap&) = delete;
    DSPMap& operator=(const DSPMap&) = delete;

    /* :181-353.  Returns 1, or 0 when the frame is rejected (state untouched, :193-208). */
    int update(int point_cloud_num, int size_of_one_point, float* point_cloud_ptr, float sensor_px, float sensor_py,
               float sensor_pz, double time_stamp_second, float sensor_quaternion_w, float sensor_quaternion_x,
               float sensor_quaternion_y, float sensor_quaternion_z) {
        lazy_prefill();
#ifdef DSPMAP_WORLD
        if (!comm_ready_) {   // the communicator is created at the first frame: no HIP / RCCL call during static initialisation
            if (dspmap_mgpu_comm_init_from_env(h_) != DSPMAP_OK) { cerr << "DSPMap: " << dspmap_last_error(h_) << endl; return 0; }
            comm_ready_ = true;
        }
        const int rc = dspmap_mgpu_update_host(h_, point_cloud_num, size_of_one_point, point_cloud_ptr, sensor_px, sensor_py,
                                               sensor_pz, time_stamp_second, sensor_quaternion_w, sensor_quaternion_x,
                                               sensor_quaternion_y, sensor_quaternion_z);
#else
        const int rc = dspmap_update(h_, point_cloud_num, size_of_one_point, point_cloud_ptr, sensor_px, sensor_py,
                                     sensor_pz, time_stamp_second, sensor_quaternion_w, sensor_quaternion_x,
                                     sensor_quaternion_y, sensor_quaternion_z);
#endif
        if (rc < 0) { cerr << "DSPMap::update failed: " << dspmap_last_error(h_) << endl; return 0; }
        if (rc == 1 && record_flag_) {  // :326-350: every frame when the flag is negative, else once after record_time
            const float update_time = (float)dspmap_get_param(h_, DSPMAP_P_UPDATE_TIME);
            if (record_flag_ < 0 || (update_time > record_time_ && !recorded_once_)) {
                recorded_once_ = 1;
                const int update_counter = (int)dspmap_get_param(h_, DSPMAP_P_UPDATE_COUNTER);
                writeParticleCsv(particle_save_folder + "/particles_update_t_" + to_string(update_counter) + "_" +
                                 to_string((int)(update_time * 1000)) + ".csv");
            }
        }
        return rc;
    }

    void setPredictionVariance(float p_stddev, float v_stddev) {  // :355-360 (regenerates the Gaussian tables)
        dspmap_set_param(h_, DSPMAP_P_POSITION_STDDEV, p_stddev);
        dspmap_set_param(h_, DSPMAP_P_VELOCITY_STDDEV, v_stddev);
        dspmap_set_param(h_, DSPMAP_P_REGENERATE_TABLES, 1);
    }
    void setObservationStdDev(float ob_stddev) { dspmap_set_param(h_, DSPMAP_P_OBSERVATION_STDDEV, ob_stddev); }  // :362
    void setNewBornParticleWeight(float weight) { dspmap_set_param(h_, DSPMAP_P_NEWBORN_WEIGHT, weight); }        // :366
    void setNewBornParticleNumberofEachPoint(int num) { dspmap_set_param(h_, DSPMAP_P_NEWBORN_NUMBER, num); }      // :370
    /* :375-378.  Arms the particle CSV dump at the end of update() (:326-350); writeParticleCsv() writes one on request. */
    void setParticleRecordFlag(int record_particle_flag, float record_csv_time = 1.f) {
        record_flag_ = record_particle_flag; record_time_ = record_csv_time;
    }
    static void setOriginalVoxelFilterResolution(float res) { voxel_filter_res() = res; }  // :380 (static in the reference)

    void getOccupancyMap(int& obstacles_num, pcl::PointCloud<pcl::PointXYZ>& cloud, const float threshold = 0.7) {  // :385-402
        fetch(obstacles_num, cloud, nullptr, threshold);
    }
    void getOccupancyMapWithFutureStatus(int& obstacles_num, pcl::PointCloud<pcl::PointXYZ>& cloud, float* future_status,
                                         const float threshold = 0.7) {  // :405-426
        fetch(obstacles_num, cloud, future_status, threshold);
    }
    /* north-star name: the V x T future occupancy masses (then cleared, like the getters above) */
    void getFutureStatus(float* future_status) { sync_params(); dspmap_get_future(h_, future_status + (size_t)dspmap_local_voxel_base(h_) * PREDICTION_TIMES); }
    void clearOccupancyMapPrediction() { dspmap_clear_future(h_); }  // :431-438

    void getKMClusterResult(pcl::PointCloud<pcl::PointXYZINormal>& cluster_cloud) {  // :441-445
        int n = 0;
        dspmap_get_birth_cloud(h_, nullptr, 0, &n);
        std::vector<dspmap_vpoint> v((size_t)n);
        if (n) dspmap_get_birth_cloud(h_, v.data(), n, &n);
        for (const auto& q : v) {
            pcl::PointXYZINormal p;
            p.x = q.x; p.y = q.y; p.z = q.z;
            p.normal_x = q.nx; p.normal_y = q.ny; p.normal_z = q.nz; p.intensity = q.intensity;
            cluster_cloud.push_back(p);
        }
    }

    void mapAddNewBornParticlesByObservation() { dspmap_stage_birth(h_); }  // public in the reference (:796)

    static float generateRandomFloat(float min, float max) {  // :1551-1553
        return min + static_cast<float>(rand()) / (static_cast<float>(RAND_MAX / (max - min)));
    }
    void getVoxelPositionFromIndexPublic(const int& index, float& px, float& py, float& pz) const {  // :1556-1572
        dspmap_voxel_center(h_, index, &px, &py, &pz);
    }
    int getPointVoxelsIndexPublic(const float& px, const float& py, const float& pz, int& index) {  // :1574-1584
        return dspmap_point_voxel_index(h_, px, py, pz, &index);
    }

    /* particle CSV in the reference's column order (:336-347): flag,vx,vy,vz,px,py,pz,weight,voxel_index */
    int writeParticleCsv(const std::string& file_name) {
        int n = 0;
        dspmap_export_state(h_, 0, nullptr, nullptr, nullptr, &n);
        std::vector<int> vox((size_t)n), slot((size_t)n);
        std::vector<float> rec((size_t)n * 8);
        if (n) dspmap_export_state(h_, n, vox.data(), slot.data(), rec.data(), &n);
        // the reference walks voxels, then slots, in ascending order (:337-338); the device compaction is unordered
        std::vector<int> order((size_t)n);
        for (int i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return vox[a] != vox[b] ? vox[a] < vox[b] : slot[a] < slot[b]; });
        std::ofstream w(file_name, std::ios::out | std::ios::trunc);
        for (int q = 0; q < n; q++) {
            const int i = order[q];
            for (int k = 0; k < 8; k++) w << rec[(size_t)i * 8 + k] << ",";
            w << vox[i] << "\n";
        }
        return n;
    }

    dspmap_t* dspmap_handle() { return h_; }

private:
    static float& voxel_filter_res() { static float r = 0.15f; return r; }  // :132
    void sync_params() { dspmap_set_param(h_, DSPMAP_P_VOXEL_FILTER_RES, voxel_filter_res()); }
    void lazy_prefill() {
        sync_params();
        if (init_particles_ > 0) { dspmap_add_random_particles(h_, init_particles_, init_weight_); init_particles_ = 0; }
    }
    void fetch(int& obstacles_num, pcl::PointCloud<pcl::PointXYZ>& cloud, float* future_status, float threshold) {
        const int v = dspmap_local_voxel_num(h_);
        xyz_.resize((size_t)v * 3);
        int n = 0;
        // a slab's rows land at their GLOBAL voxel offsets of the caller's [VOXEL_NUM][PREDICTION_TIMES] array
        if (future_status) future_status += (size_t)dspmap_local_voxel_base(h_) * PREDICTION_TIMES;
        const int rc = future_status ? dspmap_get_occupancy_with_future(h_, threshold, xyz_.data(), v, &n, future_status)
                                     : dspmap_get_occupancy(h_, threshold, xyz_.data(), v, &n);
        obstacles_num = rc == DSPMAP_OK ? n : 0;
        for (int i = 0; i < obstacles_num; i++) {  // appended, never cleared by the map (:391,411)
            pcl::PointXYZ p;
            p.x = xyz_[3 * (size_t)i]; p.y = xyz_[3 * (size_t)i + 1]; p.z = xyz_[3 * (size_t)i + 2];
            cloud.push_back(p);
        }
    }
    dspmap_t* h_ = nullptr;
#ifdef DSPMAP_WORLD
    int world_ = 1, rank_ = 0;
    bool comm_ready_ = false;
#endif
    int init_particles_;
    float init_weight_;
    int record_flag_ = 0;        // if_record_particle_csv :479
    float record_time_ = 1.f;    // record_time :480
    int recorded_once_ = 0;      // recorded_once_flag :326
    std::vector<float> xyz_;
};

#endif /* DSP_DYNAMIC_H_MI355X */
