// microbenchmark behind k_place's store path (round 6): a tile's arrivals land in a handful of slot rows of its [slot][64] cell block.
//   mode 0  what k_place does: every arrival stores its position (12 B) and weight (4 B) straight to its cell -- one lane, one cell, the
//           lanes of a store instruction scattered over the tile's rows;
//   mode 1  the same cells written ROW BY ROW: the arrivals are staged in LDS by (row, lane), then every touched row is ONE store
//           instruction whose active lanes are consecutive cells (the others masked off).
// Same cells, same bytes; only the shape of the store instructions differs.  hipcc --offload-arch=gfx950 -O3 -o place_store_bench place_store_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned h32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
struct P3 { float x, y, z; };
#define ROWS 8
// n arrivals per tile (<= 256 * 2), rows [r0, r0 + ROWS) of a tile of `slots` rows
template <int MODE>
__global__ void __launch_bounds__(256) k(float* pos, float* w, int slots, int n, int r0) {
    __shared__ float s_p[ROWS][64][3];
    __shared__ float s_w[ROWS][64];
    __shared__ unsigned long long s_m[ROWS];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const size_t tcell = (size_t)tile * slots * 64;
    if (tid < ROWS) s_m[tid] = 0ull;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const unsigned r = h32(tile * 7919u + i * 0x9e3779b9u);
        const int row = (int)(r % ROWS), lane = (int)((r >> 8) & 63);
        const float v = (float)i;
        if (MODE == 0) {
            const size_t c = tcell + (size_t)(r0 + row) * 64 + lane;
            P3 p; p.x = v; p.y = v + 1; p.z = v + 2;
            reinterpret_cast<P3*>(pos)[c] = p;
            w[c] = v;
        } else {
            s_p[row][lane][0] = v; s_p[row][lane][1] = v + 1; s_p[row][lane][2] = v + 2; s_w[row][lane] = v;
            atomicOr(&s_m[row], 1ull << lane);
        }
    }
    if (MODE == 1) {
        __syncthreads();
        const int wave = tid >> 6, l = tid & 63;
        for (int row = wave; row < ROWS; row += 4) {
            if ((s_m[row] >> l) & 1ull) {
                const size_t c = tcell + (size_t)(r0 + row) * 64 + l;
                P3 p; p.x = s_p[row][l][0]; p.y = s_p[row][l][1]; p.z = s_p[row][l][2];
                reinterpret_cast<P3*>(pos)[c] = p;
                w[c] = s_w[row][l];
            }
        }
    }
}
int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { int tiles, slots, n; const char* name; } cfg[] = {{16335, 48, 118, "132x132x60 saturated"}, {87120, 72, 300, "264x264x80 saturated"}};
    for (auto& c : cfg) {
        const size_t cells = (size_t)c.tiles * c.slots * 64;
        float *pos, *w; hipMalloc(&pos, cells * 12); hipMalloc(&w, cells * 4);
        hipMemset(pos, 0, cells * 12); hipMemset(w, 0, cells * 4);
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(c.tiles), dim3(256), 0, 0, pos, w, c.slots, c.n, c.slots / 2);
                else hipLaunchKernelGGL(k<1>, dim3(c.tiles), dim3(256), 0, 0, pos, w, c.slots, c.n, c.slots / 2);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%s: %d tiles x %d arrivals, mode %d (%s): %.3f ms  %.1f G arrivals/s\n", c.name, c.tiles, c.n, mode, mode == 0 ? "cell by cell" : "row by row through LDS", best,
                   (double)c.tiles * c.n / best * 1e-6);
        }
        hipFree(pos); hipFree(w);
    }
    return 0;
}
