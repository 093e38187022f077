// map_example.cpp -- the calling sequence of the reference's ROS node
// (g-ch/DSP-map src/map_sim_example.cpp) without ROS: a global DSPMap, the
// setters of its main() (:522-528), then per frame the body of cloudCallback():
// crop + cap the cloud (:319-336), my_map.update(...) (:345-349),
// getOccupancyMapWithFutureStatus(..., 0.2) (:378), getVoxelPositionFromIndexPublic (:409).
// Build: g++ -std=c++14 -Iinclude examples/map_example.cpp -Ldsp-map_amd/lib -ldspmap_hip -o map_example
#include "dsp_dynamic.h"

#include <chrono>
#include <cstdio>

/// Define a map object (namespace scope, constructed before main: src/map_sim_example.cpp:39)
DSPMap my_map;
const float res = 0.1;
const unsigned int MAX_POINT_NUM = 5000;
float point_clouds[MAX_POINT_NUM * 3];
float x_min = -MAP_LENGTH_VOXEL_NUM * VOXEL_RESOLUTION / 2;
float x_max = MAP_LENGTH_VOXEL_NUM * VOXEL_RESOLUTION / 2;
float y_min = -MAP_WIDTH_VOXEL_NUM * VOXEL_RESOLUTION / 2;
float y_max = MAP_WIDTH_VOXEL_NUM * VOXEL_RESOLUTION / 2;
float z_min = -MAP_HEIGHT_VOXEL_NUM * VOXEL_RESOLUTION / 2;
float z_max = MAP_HEIGHT_VOXEL_NUM * VOXEL_RESOLUTION / 2;

static bool inRange(float lo, float hi, float v) { return v > lo && v < hi; }

int main(int argc, char** argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 20;
    my_map.setPredictionVariance(0.05, 0.05);
    my_map.setObservationStdDev(0.1);
    my_map.setNewBornParticleNumberofEachPoint(20);
    my_map.setNewBornParticleWeight(0.0001);
    DSPMap::setOriginalVoxelFilterResolution(res);
    my_map.setParticleRecordFlag(0, 19.0);

    static float future_status[VOXEL_NUM][PREDICTION_TIMES];
    double total = 0;
    int occupied_last = 0;
    for (int f = 0; f < frames; f++) {
        const double t = f / 30.0;
        // synthetic cloud in the sensor frame: a wall 3 m ahead, a floor 1 m below, a box moving sideways
        int useful_point_num = 0;
        auto push = [&](float x, float y, float z) {
            if (useful_point_num < (int)MAX_POINT_NUM && inRange(x_min, x_max, x) && inRange(y_min, y_max, y) && inRange(z_min, z_max, z)) {
                point_clouds[useful_point_num * 3] = x; point_clouds[useful_point_num * 3 + 1] = y; point_clouds[useful_point_num * 3 + 2] = z;
                ++useful_point_num;
            }
        };
        for (float y = -2.5f; y < 2.5f; y += res) for (float z = -1.0f; z < 1.3f; z += res) push(3.0f - 0.5f * (float)t, y, z);
        for (float x = 0.8f; x < 3.0f; x += res) for (float y = -1.5f; y < 1.5f; y += res) push(x, y, -1.0f);
        for (float y = 0; y < 0.4f; y += res) for (float z = -0.9f; z < 0.7f; z += res) push(1.8f, -1.0f + 1.0f * (float)t + y, z);

        auto t0 = std::chrono::steady_clock::now();
        if (!my_map.update(useful_point_num, 3, point_clouds, 0.5f * (float)t, 0.f, 1.2f, t, 1.f, 0.f, 0.f, 0.f)) return 1;
        int occupied_num = 0;
        pcl::PointCloud<pcl::PointXYZ> cloud_to_publish;
        my_map.getOccupancyMapWithFutureStatus(occupied_num, cloud_to_publish, &future_status[0][0], 0.2);
        total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        occupied_last = occupied_num;
        if ((int)cloud_to_publish.size() != occupied_num) return 2;
    }
    float px, py, pz;
    my_map.getVoxelPositionFromIndexPublic(VOXEL_NUM / 2, px, py, pz);
    double fsum = 0;
    for (int i = 0; i < VOXEL_NUM; i++) fsum += future_status[i][0];
    printf("frames %d  occupied %d  future[0] mass %.3f  centre voxel (%.3f %.3f %.3f)  avg update+readout %.3f ms\n", frames,
           occupied_last, fsum, px, py, pz, 1e3 * total / frames);
    return occupied_last > 0 ? 0 : 3;
}
