// microbenchmark: throughput of scattered float atomics (no return) on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned h32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(float* a, unsigned n, int iters, unsigned spread) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned base = (MODE == 1) ? (blockIdx.x * 4096u) % n : 0u;   // MODE 1: block-local window of `spread` floats
    for (int i = 0; i < iters; ++i) {
        unsigned r = h32(t * 977u + i * 0x9e3779b9u);
        unsigned idx = (MODE == 1) ? (base + r % spread) % n : r % n;
        unsafeAtomicAdd(&a[idx], 1.0f);
    }
}
__global__ void kst(float* a, unsigned n, int iters) {   // plain scattered stores for comparison
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) { unsigned r = h32(t * 977u + i * 0x9e3779b9u); a[r % n] = 1.0f; }
}
int main() {
    const int iters = 16, blocks = 8192, tpb = 256;
    for (unsigned n : {65536u, 1u << 20, 10u << 20, 100u << 20}) {
        float* a; hipMalloc(&a, (size_t)n * 4); hipMemset(a, 0, (size_t)n * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(tpb), 0, 0, a, n, iters, 0u);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(tpb), 0, 0, a, n, iters, 2048u);
                else hipLaunchKernelGGL(kst, dim3(blocks), dim3(tpb), 0, 0, a, n, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("n=%9u floats mode=%d (%s): %.3f ms  %.1f G ops/s\n", n, mode, mode == 0 ? "atomics random" : mode == 1 ? "atomics block-window 2048" : "stores random",
                   ms, (double)blocks * tpb * iters / ms * 1e-6);
        }
        hipFree(a);
    }
    return 0;
}
