// microbenchmark / hardware question behind DSPMAP_P_ESTIMATOR_QUEUE: kernel C follows kernel A in ONE queue (plain launches or nodes of one
// graph).  A reads a buffer X from every XCD (its lines now sit in all eight L2s).  While A is still running, kernel B -- on ANOTHER
// queue -- rewrites X, writes its L2 back (release fence, agent scope) and publishes a flag with an agent-scope atomic.  C starts after
// A, finds the flag set at its first look and reads X with plain loads, WITHOUT an acquire fence of its own.  Does C see B's values?
// Only if the dispatch of C invalidated the L2s (the packet's acquire fence) -- stale values mean it did not, and a consumer that skips
// its own fence when the flag is already there would be wrong.
// Control: the same with the release fence of B left out (must show stale / missing values if the test is sensitive at all).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_a(const int* __restrict__ x, int n, int* sink, float spin_us) {   // read X, then stay busy
    int acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += x[i];
    if (acc == 0x7fffffff) *sink = acc;
    const long long t0 = wall_clock64(), dt = (long long)(spin_us * 100.f);
    while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_b(int* x, int n, int* flag, int val, int with_release) {          // one workgroup rewrites X and publishes
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = val;
    if (with_release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (with_release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_c(const int* x, int n, int* flag, int val, int* stale, int* late, int fence_always) {
    __shared__ int s_ready;
    if (threadIdx.x == 0) {
        int ready = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - val >= 0;
        s_ready = ready;
        if (!ready) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - val < 0 && wall_clock64() - t0 < 10000000ll) __builtin_amdgcn_s_sleep(4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else if (fence_always) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (threadIdx.x == 0 && !s_ready) atomicAdd(late, 1);
    int bad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) bad += x[i] != val ? 1 : 0;
    if (bad) atomicAdd(stale, bad);
}

// Second question (same queue): kernel K1's workgroup 0 stores a word with a plain store; K2, the next kernel of the same queue, reads it with
// plain loads from every XCD -- after K2 of the PREVIOUS round has cached the word's line in every L2.  Stale reads would mean that
// a kernel boundary inside one queue does not invalidate the L2s.
__global__ void k_w1(int* w, int val) { if (blockIdx.x == 0 && threadIdx.x == 0) w[3] = val; }
__global__ void k_r2(const int* w, int val, int* stale) { if (threadIdx.x == 0 && w[3] != val) atomicAdd(stale, 1); }

int main() {
    const int n = 256 * 1024;   // 1 MB
    int *x, *flag, *stale, *late, *sink;
    CHK(hipMalloc(&x, n * 4)); CHK(hipMalloc(&flag, 4)); CHK(hipMalloc(&stale, 4)); CHK(hipMalloc(&late, 4)); CHK(hipMalloc(&sink, 4));
    hipStream_t s1, s2;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int variant = 0; variant < 4; ++variant) {
        // 0 plain launches, 1 graph {A, C}, 2 plain launches without B's release (control), 3 graph + C fences always
        const bool graph = variant == 1 || variant == 3;
        const int with_release = variant != 2, fence_always = variant == 3;
        CHK(hipMemset(x, 0, n * 4)); CHK(hipMemset(flag, 0, 4)); CHK(hipMemset(stale, 0, 4)); CHK(hipMemset(late, 0, 4));
        CHK(hipDeviceSynchronize());
        const int trials = 200;
        for (int t = 1; t <= trials; ++t) {
            if (graph) {
                hipGraph_t g; hipGraphExec_t ge;
                CHK(hipStreamBeginCapture(s1, hipStreamCaptureModeRelaxed));
                hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, s1, x, n, sink, 150.f);
                hipLaunchKernelGGL(k_c, dim3(256), dim3(256), 0, s1, x, n, flag, t, stale, late, fence_always);
                CHK(hipStreamEndCapture(s1, &g));
                CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CHK(hipGraphLaunch(ge, s1));
                hipLaunchKernelGGL(k_b, dim3(1), dim3(1024), 0, s2, x, n, flag, t, with_release);   // runs while A spins
                CHK(hipStreamSynchronize(s1)); CHK(hipStreamSynchronize(s2));
                hipGraphExecDestroy(ge); hipGraphDestroy(g);
            } else {
                hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, s1, x, n, sink, 150.f);
                hipLaunchKernelGGL(k_c, dim3(256), dim3(256), 0, s1, x, n, flag, t, stale, late, fence_always);
                hipLaunchKernelGGL(k_b, dim3(1), dim3(1024), 0, s2, x, n, flag, t, with_release);
                CHK(hipStreamSynchronize(s1)); CHK(hipStreamSynchronize(s2));
            }
        }
        int hs = 0, hl = 0;
        CHK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hl, late, 4, hipMemcpyDeviceToHost));
        printf("variant %d (%s%s%s): %d trials, stale words seen by C %d (of %lld read), workgroups of C that had to wait %d\n", variant,
               graph ? "graph {A, C}" : "plain launches", with_release ? "" : ", NO release in B", fence_always ? ", C fences always" : "", trials, hs,
               (long long)trials * n, hl);
    }
    {
        int* w; CHK(hipMalloc(&w, 256)); CHK(hipMemset(w, 0, 256)); CHK(hipMemset(stale, 0, 4));
        for (int mode = 0; mode < 2; ++mode) {   // 0 plain launches, 1 one graph {K1, K2} per round
            CHK(hipMemset(stale, 0, 4)); CHK(hipDeviceSynchronize());
            const int rounds = 2000;
            for (int t = 1; t <= rounds; ++t) {
                if (mode == 0) {
                    hipLaunchKernelGGL(k_w1, dim3(64), dim3(64), 0, s1, w, t);
                    hipLaunchKernelGGL(k_r2, dim3(1024), dim3(64), 0, s1, w, t, stale);
                } else {
                    hipGraph_t g; hipGraphExec_t ge;
                    CHK(hipStreamBeginCapture(s1, hipStreamCaptureModeRelaxed));
                    hipLaunchKernelGGL(k_w1, dim3(64), dim3(64), 0, s1, w, t);
                    hipLaunchKernelGGL(k_r2, dim3(1024), dim3(64), 0, s1, w, t, stale);
                    CHK(hipStreamEndCapture(s1, &g));
                    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    CHK(hipGraphLaunch(ge, s1));
                    CHK(hipStreamSynchronize(s1));
                    hipGraphExecDestroy(ge); hipGraphDestroy(g);
                }
            }
            CHK(hipStreamSynchronize(s1));
            int hs = 0; CHK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost));
            printf("same queue, %s: %d rounds x 1024 workgroups, stale reads of the word the previous kernel stored: %d\n", mode ? "graph {K1, K2}" : "plain launches", rounds, hs);
        }
    }
    return 0;
}
