// microbenchmark: time per node of a captured chain of dependent kernels, as a function of the grid size and of what the
// kernels do (nothing / a store per thread).  Answers: what does one dependent launch cost inside a HIP graph on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_nop(float* a, int n) { if (n < 0) a[0] = 1.f; }
__global__ void k_store(float* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = (float)i; }
__global__ void k_spin(float* a, int n, long long cycles) {   // one wave per workgroup busy for `cycles`
    long long t0 = clock64(); while (clock64() - t0 < cycles) {}
    if (n < 0) a[0] = 1.f;
}
int main() {
    float* a; hipMalloc(&a, 64u << 20);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int chain = 10, reps = 200;
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {1, 64, 1024, 4096, 16384}) {
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            for (int i = 0; i < chain; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_nop, dim3(grid), dim3(256), 0, st, a, 0);
                else if (mode == 1) hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, st, a, grid * 256);
                else hipLaunchKernelGGL(k_spin, dim3(grid), dim3(64), 0, st, a, 0, 20000ll);   // ~10 us at 100 MHz clock64? (prints tell)
            }
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, st);
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s) grid %6d: %.2f us per node\n", mode, mode == 0 ? "nop" : mode == 1 ? "store" : "spin", grid, ms * 1e3 / reps / chain);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    return 0;
}
