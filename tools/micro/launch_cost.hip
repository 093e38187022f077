// microbenchmark: host time of one kernel launch as a function of how the arguments travel -- the cost behind DSPMAP_P_USE_GRAPH = 2
// (twelve plain launches per frame).  Empty kernels; the host issues 20000 launches and is timed over the issue loop only.
//   A  hipLaunchKernelGGL, two scalar arguments
//   B  hipLaunchKernelGGL, three by-value structs of 320 + 720 + 128 bytes and four pointers (the shape of this library's kernels)
//   C  the same kernel through hipModuleLaunchKernel with ONE pre-packed argument buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER)
//   D  a captured graph of 9 B-launches, replayed (per node)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
struct S320 { int v[80]; };
struct S720 { void* p[90]; };
struct S128 { float f[32]; };
__global__ void k_small(int* a, int n) { if (n < 0) a[0] = 1; }
__global__ void k_big(S320 d, S720 s, S128 fp, int* a, int* b, int* c, int* e, int n) { if (n < 0) a[0] = d.v[0] + (int)(size_t)s.p[0] + (int)fp.f[0] + b[0] + c[0] + e[0]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* a; CHK(hipMalloc(&a, 64));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    S320 d{}; S720 s{}; S128 fp{};
    const int N = 20000;
    for (int v = 0; v < 4; ++v) {
        CHK(hipStreamSynchronize(st));
        hipFunction_t f = nullptr;
        CHK(hipGetFuncBySymbol(&f, (const void*)k_big));
        struct __attribute__((packed, aligned(8))) Pack { S320 d; S720 s; S128 fp; int* a; int* b; int* c; int* e; int n; } pk;
        memset(&pk, 0, sizeof(pk)); pk.a = pk.b = pk.c = pk.e = a;
        size_t sz = sizeof(pk);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (v == 3) {
            CHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            for (int i = 0; i < 9; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, st, d, s, fp, a, a, a, a, 0);
            CHK(hipStreamEndCapture(st, &g));
            CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        auto one = [&]() {
            if (v == 0) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, st, a, 0);
            else if (v == 1) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, st, d, s, fp, a, a, a, a, 0);
            else if (v == 2) (void)hipModuleLaunchKernel(f, 64, 1, 1, 256, 1, 1, 0, st, nullptr, extra);
            else (void)hipGraphLaunch(ge, st);
        };
        for (int i = 0; i < 200; ++i) one();
        CHK(hipStreamSynchronize(st));
        const int n = v == 3 ? N / 9 : N;
        const double t0 = now();
        for (int i = 0; i < n; ++i) one();
        const double t1 = now();
        CHK(hipStreamSynchronize(st));
        const double t2 = now();
        printf("%c: host %.2f us per %s, %.2f us incl. drain\n", "ABCD"[v], (t1 - t0) / n * 1e6 / (v == 3 ? 9 : 1), v == 3 ? "node (9-node graph replay)" : "launch", (t2 - t0) / n * 1e6 / (v == 3 ? 9 : 1));
    }
    return 0;
}
