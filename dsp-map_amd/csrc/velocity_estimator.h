// velocity_estimator.h -- host-side initial velocity estimator that feeds the
// birth stage (reference: velocityEstimationThread, include/dsp_dynamic.h:1377-1544,
// "adjacent" to the hot path: it runs on a helper thread concurrently with
// prediction + update, :297,311).  Ground split -> Euclidean clustering ->
// Hungarian matching against the previous frame's centroids -> per-point
// velocity tags.  PCL's EuclideanClusterExtraction and munkres-cpp are not
// available; their published algorithms are implemented here (hash-grid radius
// search; Kuhn-Munkres).
#pragma once
#include <vector>
#include "../../include/dspmap.h"

class VelocityEstimator {
public:
    void configure(int half_fov_h, int half_fov_v, int angle_resolution);
    // rotate sensor-frame points into the world-aligned frame and keep those inside the FOV wedge
    // (update() :244-257): output = cloud_in_current_view_rotated (xyz packed)
    void rotate_and_filter(const float* pts_xyz, int n, const float quat[4], std::vector<float>& view);
    // velocityEstimationThread :1377-1544; `out` keeps its previous content when `view` is empty (:1379)
    void run(const std::vector<float>& view, const float cur_pos[3], float dt, float voxel_filtered_resolution,
             std::vector<dspmap_vpoint>& out);
    // clusters_feature_vector_dynamic_last (:1401,1542) as 5 floats per cluster {cx, cy, cz, point_num (int bits), intensity}: the
    // layout of the device estimator's copy (VelEst::last) -- the two implementations hand the state to each other when a frame
    // has to switch between them, so that there is ONE `last` like the reference's function static
    int export_last(float* out5, int cap) const;
    void import_last(const float* in5, int n);
    const float* planes_h() const { return ph_.data(); }
    const float* planes_v() const { return pv_.data(); }

private:
    struct Cluster {  // ClusterFeature :98-109
        float cx = 0, cy = 0, cz = 0;
        int point_num = 0;
        float vx = -10000.f, vy = -10000.f, vz = -10000.f, v = 0.f, intensity = 0.f;
    };
    int np_h_ = 0, np_v_ = 0;
    std::vector<float> ph0_, pv0_, ph_, pv_;
    std::vector<Cluster> last_;  // clusters_feature_vector_dynamic_last :1401
    bool configured_ = false;
    std::vector<int> cell_start_;   // clustering grid (kept between frames: no per-frame allocation of the big array)
    friend struct dspmap;
public:
    bool configured() const { return configured_; }
};
