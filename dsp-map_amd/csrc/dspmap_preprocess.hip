// dspmap_preprocess.hip -- the caller-side cloud pre-processing of the reference node on the device
// (reference src/map_sim_example.cpp:309-336):
//
//   pcl::VoxelGrid(leaf) centroid filter  (:313-317)   -> k_pp_accumulate
//   camera -> map axis swap x = z, y = -x, z = -y (:321-323), crop to the open map box (:325),
//   cap at max_points in output order (:332)           -> k_pp_count, k_pp_scan, k_pp_emit
//
// pcl::VoxelGrid is third-party (PCL, version unpinned by the reference, readme.md:21-24); its published
// algorithm (pcl/filters/impl/voxel_grid.hpp, identical arithmetic in 1.8 ... 1.12) is restated:
//   min_b = floor(min_p * inv_leaf), div = max_b - min_b + 1, idx = ijk0 + ijk1*div0 + ijk2*div0*div1 with
//   ijk = (int)(floor(p * inv_leaf) - (float)min_b); one output point per occupied leaf = the mean of its
//   points; output ordered by ascending idx (the filter sorts (idx, point) pairs).
// PCL sorts and then sums every leaf's points in sorted order (std::sort: unspecified among equal keys).  Here:
//   * leaf membership floor(p * inv_leaf) does not depend on the cloud's bounding box (min_b only offsets the
//     index) and the output order is lexicographic in the lattice coordinates (k2, k1, k0) whatever the offsets;
//   * the crop keeps a centroid only if it lies in the map box, and a centroid lies inside its leaf;
//   so only the leaves that INTERSECT the map box matter: the sums are accumulated with float atomics into a dense
//   grid over exactly those leaves (size fixed by the map, e.g. 100 x 61 x 100 leaves) and the occupied leaves are
//   compacted in lattice order -- same leaves, same order, same cap, means equal up to fp32 summation order, no
//   sort, no bounding-box pass, and far returns (a corridor's vanishing point) cost nothing.
#include "dspmap_device.h"
#include "dspmap_internal.h"

#define PP_TPB 256
#define PP_MAX_CELLS (1ll << 27)   // 128 M leaves = 2 GiB of accumulators; finer lattices over the map box are refused

struct PPGrid {
    int min_b[3], div[3];
    float inv_leaf;
    long long cells;
};

// leaf sums: acc[cell] = {sum x, sum y, sum z, count}
__global__ void __launch_bounds__(PP_TPB) k_pp_accumulate(const float* __restrict__ pts, int n, int stride, PPGrid g,
                                                          float4* __restrict__ acc) {
    const int i = blockIdx.x * PP_TPB + threadIdx.x;
    if (i >= n) return;
    const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) return;
    // voxel_grid.hpp: ijk = static_cast<int>(std::floor(p * inverse_leaf_size) - static_cast<float>(min_b))
    const int i0 = (int)(floorf(x * g.inv_leaf) - (float)g.min_b[0]);
    const int i1 = (int)(floorf(y * g.inv_leaf) - (float)g.min_b[1]);
    const int i2 = (int)(floorf(z * g.inv_leaf) - (float)g.min_b[2]);
    if (i0 < 0 || i0 >= g.div[0] || i1 < 0 || i1 >= g.div[1] || i2 < 0 || i2 >= g.div[2]) return;   // leaf does not touch the map box
    const long long cell = (long long)i0 + (long long)i1 * g.div[0] + (long long)i2 * g.div[0] * g.div[1];
    float* a = reinterpret_cast<float*>(&acc[cell]);
    unsafeAtomicAdd(a, x);
    unsafeAtomicAdd(a + 1, y);
    unsafeAtomicAdd(a + 2, z);
    unsafeAtomicAdd(a + 3, 1.f);
}

// centroid of a leaf (centroid /= count, voxel_grid.hpp), axis swap (:321-323), open-box crop (:190-197,325)
__device__ __forceinline__ bool pp_point(const float4 a, int swap_axes, float hx, float hy, float hz, float& x, float& y, float& z) {
    if (!(a.w > 0.f)) return false;
    const float cx = __fdiv_rn(a.x, a.w), cy = __fdiv_rn(a.y, a.w), cz = __fdiv_rn(a.z, a.w);
    if (swap_axes) { x = cz; y = -cx; z = -cy; } else { x = cx; y = cy; z = cz; }
    return x > -hx && x < hx && y > -hy && y < hy && z > -hz && z < hz;
}
__global__ void __launch_bounds__(PP_TPB) k_pp_count(const float4* __restrict__ acc, long long cells, int swap_axes, float hx,
                                                     float hy, float hz, int* __restrict__ blk_cnt, int* __restrict__ n_leaves) {
    __shared__ int s_c[PP_TPB / 64], s_l[PP_TPB / 64];
    const long long c = (long long)blockIdx.x * PP_TPB + threadIdx.x;
    float x, y, z;
    bool leaf = false, keep = false;
    if (c < cells) {
        const float4 a = acc[c];
        leaf = a.w > 0.f;
        keep = pp_point(a, swap_axes, hx, hy, hz, x, y, z);
    }
    const u64 bk = __ballot(keep), bl = __ballot(leaf);
    if (lane_id() == 0) { s_c[threadIdx.x >> 6] = (int)__popcll(bk); s_l[threadIdx.x >> 6] = (int)__popcll(bl); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0, tl = 0;
        for (int k = 0; k < PP_TPB / 64; ++k) { t += s_c[k]; tl += s_l[k]; }
        blk_cnt[blockIdx.x] = t;
        if (tl) atomicAdd(n_leaves, tl);
    }
}
__global__ void __launch_bounds__(1024) k_pp_scan(int* __restrict__ blk_cnt, int nblk, int* __restrict__ total) {
    __shared__ int s_w[16];
    __shared__ int s_run;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + tid;
        const int v = i < nblk ? blk_cnt[i] : 0;
        const int inc = wave_incl_scan_i(v);
        if (l == 63) s_w[w] = inc;
        __syncthreads();
        int off = s_run;
        for (int k = 0; k < w; ++k) off += s_w[k];
        if (i < nblk) blk_cnt[i] = off + inc - v;
        __syncthreads();
        if (tid == 1023) s_run = off + inc;
        __syncthreads();
    }
    if (tid == 0) *total = s_run;
}
__global__ void __launch_bounds__(PP_TPB) k_pp_emit(const float4* __restrict__ acc, long long cells, int swap_axes, float hx,
                                                    float hy, float hz, const int* __restrict__ blk_off, int max_points,
                                                    float* __restrict__ out) {
    __shared__ int s_c[PP_TPB / 64];
    const long long c = (long long)blockIdx.x * PP_TPB + threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f;
    bool keep = false;
    if (c < cells) keep = pp_point(acc[c], swap_axes, hx, hy, hz, x, y, z);
    const u64 b = __ballot(keep);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) s_c[w] = (int)__popcll(b);
    __syncthreads();
    int off = blk_off[blockIdx.x];
    for (int k = 0; k < w; ++k) off += s_c[k];
    if (keep) {
        const int pos = off + (int)__popcll(b & lanemask_lt());
        if (pos < max_points) {   // :332: the loop stops once the buffer is full
            out[3 * (size_t)pos] = x; out[3 * (size_t)pos + 1] = y; out[3 * (size_t)pos + 2] = z;
        }
    }
}

extern "C" int dspmap_preprocess_cloud(dspmap_t* m, int n, const float* points_dev, int stride_floats, float leaf, int swap_axes,
                                       int max_points, float* out_dev, int* n_out, int* n_leaves_out) {
    READY(m);
    if (n < 0 || (n > 0 && !points_dev) || stride_floats < 3 || !(leaf > 0.f) || max_points < 0 || (max_points > 0 && !out_dev) || !n_out)
        return dspmap_fail(m, DSPMAP_E_ARG, "bad arguments");
    *n_out = 0;
    if (n_leaves_out) *n_leaves_out = 0;
    if (n == 0) return DSPMAP_OK;   // empty cloud -> empty cloud (pcl::VoxelGrid returns an empty output)
    const float hx = m->d.half_x, hy = m->d.half_y, hz = m->d.half_z;   // x_min .. z_max of src/map_sim_example.cpp:52-57
    // the map box in the INPUT frame: x_map = z_in, y_map = -x_in, z_map = -y_in (:321-323)
    const float hin[3] = {swap_axes ? hy : hx, swap_axes ? hz : hy, swap_axes ? hx : hz};
    PPGrid g;
    g.inv_leaf = 1.0f / leaf;   // inverse_leaf_size_ = Array4f::Ones() / leaf_size_
    long long cells = 1;
    for (int a = 0; a < 3; ++a) {
        const int lo = (int)floorf(-hin[a] * g.inv_leaf), hi = (int)floorf(hin[a] * g.inv_leaf);
        g.min_b[a] = lo;
        g.div[a] = hi - lo + 1;
        cells *= (long long)g.div[a];
        if (cells > PP_MAX_CELLS)
            return dspmap_fail(m, DSPMAP_E_ARG, "leaf size %.4g too small for the map box (more than %lld leaves)", leaf, (long long)PP_MAX_CELLS);
    }
    g.cells = cells;
    if (!m->pp_box) HIPCHK(m, hipMalloc((void**)&m->pp_box, 2 * sizeof(int)));
    if ((size_t)cells > m->pp_cells_cap) {
        if (m->pp_acc) (void)hipFree(m->pp_acc);
        if (m->pp_blk) (void)hipFree(m->pp_blk);
        m->pp_acc = nullptr; m->pp_blk = nullptr; m->pp_cells_cap = 0;
        HIPCHK(m, hipMalloc((void**)&m->pp_acc, sizeof(float4) * (size_t)cells));
        HIPCHK(m, hipMalloc((void**)&m->pp_blk, sizeof(int) * ((size_t)(cells + PP_TPB - 1) / PP_TPB + 1)));
        m->pp_cells_cap = (size_t)cells;
    }
    float4* acc = (float4*)m->pp_acc;
    int* totals = (int*)m->pp_box;
    HIPCHK(m, hipMemsetAsync(acc, 0, sizeof(float4) * (size_t)cells, m->stream));
    HIPCHK(m, hipMemsetAsync(totals, 0, 2 * sizeof(int), m->stream));
    hipLaunchKernelGGL(k_pp_accumulate, dim3((n + PP_TPB - 1) / PP_TPB), dim3(PP_TPB), 0, m->stream, points_dev, n, stride_floats, g, acc);
    const int nblk = (int)((cells + PP_TPB - 1) / PP_TPB);
    hipLaunchKernelGGL(k_pp_count, dim3(nblk), dim3(PP_TPB), 0, m->stream, acc, cells, swap_axes, hx, hy, hz, m->pp_blk, totals + 1);
    hipLaunchKernelGGL(k_pp_scan, dim3(1), dim3(1024), 0, m->stream, m->pp_blk, nblk, totals);
    if (max_points > 0)
        hipLaunchKernelGGL(k_pp_emit, dim3(nblk), dim3(PP_TPB), 0, m->stream, acc, cells, swap_axes, hx, hy, hz, m->pp_blk, max_points, out_dev);
    int h_tot[2] = {0, 0};
    HIPCHK(m, hipMemcpyAsync(h_tot, totals, 2 * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipGetLastError());
    *n_out = h_tot[0] < max_points ? h_tot[0] : max_points;
    if (n_leaves_out) *n_leaves_out = h_tot[1];
    return DSPMAP_OK;
}
