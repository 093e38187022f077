// microbenchmark: VALU issue rate on MI355X for fp32 (scalar per lane), packed fp32 (v_pk_fma_f32), fp64 and exp2.
// 8 independent dependency chains per lane, one wave per SIMD and eight waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    const float t = (float)threadIdx.x * 1e-3f;
    if (MODE == 0) {          // v_fma_f32
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = t + j;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __fmaf_rn(x[j], a, b);
        float s = 0; for (int j = 0; j < 8; ++j) s += x[j];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else if (MODE == 1) {   // v_pk_fma_f32: two fp32 FMAs per instruction
        f2 x[8];
        for (int j = 0; j < 8; ++j) x[j] = f2{t + j, t - j};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], f2{a, a}, f2{b, b});
        f2 s = 0; for (int j = 0; j < 8; ++j) s += x[j];
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else if (MODE == 2) {   // v_fma_f64
        double x[8];
        for (int j = 0; j < 8; ++j) x[j] = t + j;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __fma_rn(x[j], (double)a, (double)b);
        double s = 0; for (int j = 0; j < 8; ++j) s += x[j];
        out[blockIdx.x * 256 + threadIdx.x] = (float)s;
    } else {                  // v_exp_f32
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = t + j;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_exp2f(x[j]) - 1.0f;
        float s = 0; for (int j = 0; j < 8; ++j) s += x[j];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}
int main() {
    float* out; hipMalloc(&out, 8192 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_exp_f32 (+v_sub)"};
    for (int blocks : {256, 2048}) {   // 256 blocks x 4 waves = one wave per SIMD; 2048 = eight
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-4f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-4f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-4f);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-4f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double winstr = (double)blocks * 4 * iters * 8;   // wave instructions of the timed kind
            const double cyc = ms * 1e-3 * 2.4e9 * 1024 / winstr * (blocks >= 1024 ? 1 : (double)(blocks * 4) / 1024);
            printf("blocks %5d  %-20s %.3f ms  %.2f G wave-instr/s  ~%.1f SIMD cycles per wave instruction\n", blocks, names[mode], ms,
                   winstr / ms * 1e-6, cyc);
        }
    }
    return 0;
}
