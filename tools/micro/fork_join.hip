// microbenchmark: what does the estimator's side branch cost the captured frame, and what would other ways of tying the two
// branches together cost?  The frame of workload B is modelled by spin kernels of its kernels' durations (main chain: 6, 28, 15,
// 8, 14, 17.5, 13.6, 13.5, 25 us; side branch: 24 us on 32 workgroups of 1024 threads, then 45 us on one workgroup).
//   V0  main chain alone
//   V1  fork after the first kernel, event join before the 7th (the frame as it is captured today)
//   V2  fork after the first kernel, NO join edge: the 7th kernel polls a flag in memory that the side branch's last kernel
//       sets; the branch is tied back in only at the end of the capture (a graph with two leaves)
//   V3  the side branch forks BEFORE the first kernel (a second root), event join
//   V4  second root + flag join
//   V5  two graphs on two streams, flag join, nothing else between them
//   V6  no graph: the kernels of V1 launched one by one on two streams tied by events
//   V7  no graph, main chain alone (V0 as plain launches)
//   V8  no graph: main chain as plain launches, side kernels on the other stream, flag join
// Prints microseconds per frame over back-to-back replays.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

struct Flags { int frame; int side_frame; int done; int timeouts; };

__device__ __forceinline__ void spin_us(float us) {
    const long long t0 = wall_clock64(), dt = (long long)(us * 100.f);
    while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(1);
}
// mode: 0 plain, 1 first kernel of the main chain (advances the frame counter), 2 first kernel of the side branch (advances its
// own counter), 3 last kernel of the side branch (publishes the counter), 4 consumer (polls the flag, bounded)
__global__ void k_spin(Flags* f, float us, int mode) {
    if (mode == 4) {
        if (threadIdx.x == 0) {
            const int want = __hip_atomic_load(&f->frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&f->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (wall_clock64() - t0 > 200000) { if (blockIdx.x == 0) atomicAdd(&f->timeouts, 1); break; }   // 2 ms
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
    }
    spin_us(us);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (mode == 1) __hip_atomic_fetch_add(&f->frame, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode == 2) __hip_atomic_fetch_add(&f->side_frame, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (mode == 3) {
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(&f->done, __hip_atomic_load(&f->side_frame, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static const float MAIN_US[9] = {6.f, 28.f, 15.f, 8.f, 14.f, 17.5f, 13.6f, 13.5f, 25.f};

static void main_kernel(hipStream_t st, Flags* f, int i, bool flag_join) {
    const int mode = i == 0 ? 1 : (i == 6 && flag_join ? 4 : 0);
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, f, MAIN_US[i], mode);
}
static void side_kernels(hipStream_t st, Flags* f) {
    hipLaunchKernelGGL(k_spin, dim3(32), dim3(1024), 0, st, f, 24.f, 2);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, st, f, 45.f, 3);
}

int main() {
    Flags* f; CHK(hipMalloc(&f, sizeof(Flags)));
    hipStream_t st, st2;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    const int reps = 400;
    for (int round = 0; round < 2; ++round)
    for (int v = 0; v <= 8; ++v) {
        CHK(hipMemset(f, 0, sizeof(Flags)));
        CHK(hipDeviceSynchronize());
        const bool side = v >= 1 && v != 7, root = v == 3 || v == 4, flagj = v == 2 || v == 4 || v == 5 || v == 8, two = v == 5 || v == 8, plain = v >= 6;
        hipGraph_t g = nullptr, g2 = nullptr; hipGraphExec_t ge = nullptr, ge2 = nullptr;
        if (two && !plain) {
            CHK(hipStreamBeginCapture(st2, hipStreamCaptureModeRelaxed));
            side_kernels(st2, f);
            CHK(hipStreamEndCapture(st2, &g2));
            CHK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        }
        if (!plain) CHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        auto body = [&]() {
        if (side && root && !two) { hipEventRecord(ef, st); hipStreamWaitEvent(st2, ef, 0); }
        main_kernel(st, f, 0, flagj);
        if (side && !root && !two) hipEventRecord(ef, st);
        main_kernel(st, f, 1, flagj);          // (queued before the side branch's first kernel, as in the library)
        if (side && !two) {
            if (!root) hipStreamWaitEvent(st2, ef, 0);
            side_kernels(st2, f);
            hipEventRecord(ej, st2);
        }
        for (int i = 2; i < 9; ++i) {
            if (i == 6 && side && !flagj) hipStreamWaitEvent(st, ej, 0);
            main_kernel(st, f, i, flagj);
        }
        if (side && flagj && !two) hipStreamWaitEvent(st, ej, 0);   // the capture has to rejoin its branches
        };
        if (!plain) {
            body();
            CHK(hipStreamEndCapture(st, &g));
            CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        auto frame = [&]() { if (plain) { if (two) side_kernels(st2, f); body(); return; } if (two) hipGraphLaunch(ge2, st2); hipGraphLaunch(ge, st); };
        for (int r = 0; r < 30; ++r) frame();
        CHK(hipStreamSynchronize(st)); CHK(hipStreamSynchronize(st2));
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) frame();
        hipEventRecord(e1, st); CHK(hipEventSynchronize(e1));
        CHK(hipStreamSynchronize(st2));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        Flags h; CHK(hipMemcpy(&h, f, sizeof(h), hipMemcpyDeviceToHost));
        printf("V%d: %.2f us per frame (sum of the main chain's spins 140.6)  frames %d side %d done %d timeouts %d\n", v, ms * 1e3 / reps, h.frame, h.side_frame, h.done, h.timeouts);
        if (!plain) { hipGraphExecDestroy(ge); hipGraphDestroy(g); }
        if (two && !plain) { hipGraphExecDestroy(ge2); hipGraphDestroy(g2); }
    }
    return 0;
}
