// gauss_tables.cpp -- ORACLE helper (test infrastructure).
// Fills the two Gaussian tables the way the reference does:
// generateGaussianRandomsVectorZeroCenter, include/dsp_dynamic.h:1150-1160
// (std::default_random_engine + normal_distribution<double>, draws interleaved
// position/velocity).  The reference seeds with time(NULL); here the seed is
// an argument so that runs are reproducible.
#include <random>
extern "C" void dspo_fill_gaussian_tables(float* p_tab, float* v_tab, int n, float p_stddev,
                                          float v_stddev, unsigned seed) {
    std::default_random_engine random(seed);
    std::normal_distribution<double> n1(0, p_stddev);
    std::normal_distribution<double> n2(0, v_stddev);
    for (int i = 0; i < n; i++) {
        p_tab[i] = (float)n1(random);
        v_tab[i] = (float)n2(random);
    }
}
