// csv_driver.cpp -- test program: the drop-in `class DSPMap` (include/dsp_dynamic.h) fed from a binary frame file,
// with the particle CSV dump of update() armed (setParticleRecordFlag, reference include/dsp_dynamic.h:326-350,375-378).
// usage: csv_driver <frames.bin> <out_dir> <record_flag> <record_time>
// frames.bin: int32 n_frames, table_n, rand_n; float32 p_tab[table_n], v_tab[table_n]; int32 r_tab[rand_n];
//             per frame: int32 n; float32 pos[3]; float64 stamp; float32 quat[4]; float32 pts[n*3]
#include "dsp_dynamic.h"

#include <cstdio>

DSPMap my_map;

int main(int argc, char** argv) {
    if (argc < 5) return 64;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 65;
    particle_save_folder = argv[2];
    int hdr[3];
    if (fread(hdr, sizeof(int), 3, f) != 3) return 66;
    std::vector<float> p(hdr[1]), v(hdr[1]);
    std::vector<int> r(hdr[2]);
    if (fread(p.data(), 4, p.size(), f) != p.size() || fread(v.data(), 4, v.size(), f) != v.size() ||
        fread(r.data(), 4, r.size(), f) != r.size()) return 67;
    my_map.setPredictionVariance(0.05, 0.05);
    my_map.setObservationStdDev(0.1);
    my_map.setNewBornParticleNumberofEachPoint(20);
    my_map.setNewBornParticleWeight(0.0001);
    DSPMap::setOriginalVoxelFilterResolution(0.1f);
    my_map.setParticleRecordFlag(atoi(argv[3]), (float)atof(argv[4]));
    dspmap_set_gaussian_tables(my_map.dspmap_handle(), p.data(), v.data(), hdr[1]);
    dspmap_set_rand_table(my_map.dspmap_handle(), r.data(), hdr[2]);
    dspmap_set_param(my_map.dspmap_handle(), DSPMAP_P_VELOCITY_ESTIMATOR, 0);   // every in-view point a static birth source
    for (int k = 0; k < hdr[0]; k++) {
        int n; float pos[3], q[4]; double stamp;
        if (fread(&n, 4, 1, f) != 1 || fread(pos, 4, 3, f) != 3 || fread(&stamp, 8, 1, f) != 1 || fread(q, 4, 4, f) != 4) return 68;
        std::vector<float> pts((size_t)n * 3);
        if (n && fread(pts.data(), 4, pts.size(), f) != pts.size()) return 69;
        const int rc = my_map.update(n, 3, pts.data(), pos[0], pos[1], pos[2], stamp, q[0], q[1], q[2], q[3]);
        printf("frame %d rc %d\n", k, rc);
        my_map.clearOccupancyMapPrediction();
    }
    fclose(f);
    return 0;
}
