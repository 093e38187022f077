// microbenchmark: what does a launch cost whose workgroups find nothing to do?  (sparse maps: 87 120 tiles at 264x264x80, a
// few thousand of them hold particles.)  Variants: exit at once / exit after one uniform scalar load / after a per-workgroup
// flag load / after a flag load and a 64-float zeroing store per wave (what k_predict did per empty tile in round 2).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_nop(const int* n, const int* flags, float* a) { if (n == nullptr) a[0] = 1.f; }
__global__ void __launch_bounds__(256) k_count(const int* n, const int* flags, float* a) { if ((int)blockIdx.x >= *n) return; a[blockIdx.x * 256 + threadIdx.x] = 1.f; }
__global__ void __launch_bounds__(256) k_flag(const int* n, const int* flags, float* a) { if (!flags[blockIdx.x]) return; a[blockIdx.x * 256 + threadIdx.x] = 1.f; }
__global__ void __launch_bounds__(256) k_flag_zero(const int* n, const int* flags, float* a) {
    a[(size_t)blockIdx.x * 256 + threadIdx.x] = 0.f;
    if (!flags[blockIdx.x]) return;
    a[blockIdx.x * 256 + threadIdx.x] = 1.f;
}
int main() {
    float* a; hipMalloc(&a, 256u << 20);
    int *n, *flags; hipMalloc(&n, 4); hipMalloc(&flags, 4 << 20); hipMemset(n, 0, 4); hipMemset(flags, 0, 4 << 20);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int chain = 10, reps = 100;
    const char* names[4] = {"nop", "n_active", "flag", "flag+zero"};
    for (int bs : {256, 64})
    for (int mode = 0; mode < 4; ++mode)
        for (int grid : {2723, 16335, 21780, 43560, 87120}) {
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            for (int i = 0; i < chain; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_nop, dim3(grid), dim3(bs), 0, st, n, flags, a);
                else if (mode == 1) hipLaunchKernelGGL(k_count, dim3(grid), dim3(bs), 0, st, n, flags, a);
                else if (mode == 2) hipLaunchKernelGGL(k_flag, dim3(grid), dim3(bs), 0, st, n, flags, a);
                else hipLaunchKernelGGL(k_flag_zero, dim3(grid), dim3(bs), 0, st, n, flags, a);
            }
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("block %3d %-10s grid %6d: %7.2f us per launch\n", bs, names[mode], grid, ms * 1e3 / reps / chain);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    return 0;
}
