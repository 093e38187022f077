// dspmap_velest.hip -- the initial velocity estimator ON THE DEVICE.
//
// Reference: velocityEstimationThread, include/dsp_dynamic.h:1377-1544 -- the helper thread update() forks before the
// prediction and joins before the birth stage (:297,311).  It turns the frame's view (cloud_in_current_view_rotated)
// into the birth cloud input_cloud_with_velocity: ground split (:1387-1398), Euclidean clustering of the non-ground
// points (:1407-1417), per-cluster centroid and static test (:1419-1447), Hungarian matching against the previous
// frame's possibly-dynamic clusters with the distance / point-count gates (:1449-1475), velocity = centroid shift / dt
// with the 5 m/s limit (:1477-1499), per-point tags, dynamic clusters first, then the static points (:1505-1540).
//
// Third-party pieces of that function that are not under /root/reference (PCL EuclideanClusterExtraction + KdTree,
// saebyn/munkres-cpp; versions unpinned by the reference) are replaced by what they COMPUTE, not by how:
//   * PCL grows a cluster by radius search and returns its indices SORTED ascending; clusters come back sorted by size.
//     The cluster is therefore the connected component of the "closer than the tolerance" graph and the growth order
//     leaves no trace.  Here: every pair of non-ground points is tested once (tiles of 256 x 256, the squared distance
//     in the reference's fp32 arithmetic), close pairs are united in a lock-free union-find whose root is the component's
//     SMALLEST index (atomicMin hooking), so the result does not depend on the order the pairs are visited in.
//   * the assignment is the minimum-cost one; the O(n^3) Hungarian algorithm runs in one wavefront with the columns on
//     the lanes, in double precision and with first-minimum tie breaking = the sequential algorithm, step for step.
// Everything a cluster sums (centroids) is summed sequentially in ascending point index, the reference's order.
//
// Three kernels (launched by launch_velocity_estimator):
//   k_ve_view     one workgroup: compacts the view in input order, world coordinates, ground split, union-find init
//   k_ve_pairs    tiles: pair tests + unions
//   k_ve_clusters one workgroup: components -> clusters -> order -> centroids -> matching -> tags -> birth cloud
#include <hip/hip_runtime.h>
#include "dspmap_device.h"
#include "dspmap_kernels.h"

#define VE_NT 1024
#define VE_TILE 256

// exclusive prefix sum over a 1024-thread workgroup; s_tmp: 17 ints
__device__ __forceinline__ int ve_excl_scan(int v, int* s_tmp, int* total) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nw = blockDim.x >> 6;
    const int inc = wave_incl_scan_i(v);
    if (l == 63) s_tmp[w] = inc;
    __syncthreads();
    if (w == 0) {
        const int t = l < nw ? s_tmp[l] : 0;
        const int ti = wave_incl_scan_i(t);
        if (l < nw) s_tmp[l] = ti - t;
        if (l == 63) s_tmp[16] = ti;
    }
    __syncthreads();
    const int r = inc - v + s_tmp[w];
    *total = s_tmp[16];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// k_ve_view: cloud_in_current_view_rotated (:244-257) in input order -> world frame (:1389-1391), ground split (:1393)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VE_NT) k_ve_view(DevState s, VelEst ve) {
    __shared__ int s_tmp[17];
    const int n_pts = min(s.fpar->n_pts, ve.cap);
    const float cx = s.fpar->cur_pos[0], cy = s.fpar->cur_pos[1], cz = s.fpar->cur_pos[2];
    const float res_f = s.fpar->res_filter;
    const int tid = threadIdx.x;
    int base_v = 0, base_g = 0;
    for (int b0 = 0; b0 < n_pts; b0 += VE_NT) {
        const int i = b0 + tid;
        const bool in = i < n_pts && s.pt_pyr[i] >= 0;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ng = false;
        if (in) {
            const float4 r = s.pt_rot[i];
            w.x = r.x + cx; w.y = r.y + cy; w.z = r.z + cz;   // :1389-1391
            ng = w.z > res_f;                                  // :1393
        }
        int tot_v, tot_g;
        const int pv = base_v + ve_excl_scan(in ? 1 : 0, s_tmp, &tot_v);
        const int pg = base_g + ve_excl_scan(in && ng ? 1 : 0, s_tmp, &tot_g);
        if (in) {
            ve.w[pv] = w;
            ve.parent[pv] = ng ? pv : -1;     // ground points take no part in the clustering
            if (ng) ve.ng_list[pg] = pv;
        }
        base_v += tot_v; base_g += tot_g;
    }
    if (tid == 0) { ve.n[0] = base_v; ve.n[1] = base_g; }
}

// ---------------------------------------------------------------------------------------------------------------
// k_ve_pairs: every pair of non-ground points once; pairs within the cluster tolerance 2 * voxel_filtered_resolution
// (:1411) are united.  Root = smallest index of the component.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ve_find(int* __restrict__ parent, int x) {
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) { x = p; p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return x;
}
__device__ __forceinline__ void ve_union(int* __restrict__ parent, int a, int b) {
    while (true) {
        a = ve_find(parent, a); b = ve_find(parent, b);
        if (a == b) return;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int old = atomicMin(&parent[hi], lo);   // hook the larger root under the smaller one
        if (old == hi) return;                        // hi was still a root: done
        a = old; b = lo;                              // somebody re-parented hi meanwhile: unite its new parent with lo
    }
}
__global__ void __launch_bounds__(VE_TILE) k_ve_pairs(DevState s, VelEst ve, int ntile_cap) {
    __shared__ float4 s_p[VE_TILE];
    __shared__ int s_id[VE_TILE];
    const int n_ng = ve.n[1];
    const int nt = (n_ng + VE_TILE - 1) / VE_TILE;
    // block -> (bi <= bj) over the upper triangle of the capacity-sized tile grid
    int bi = 0, rem = (int)blockIdx.x;
    while (rem >= ntile_cap - bi) { rem -= ntile_cap - bi; ++bi; }
    const int bj = bi + rem;
    if (bi >= nt || bj >= nt) return;
    const float tol = 2 * s.fpar->res_filter, tol2 = tol * tol;   // :1411
    const int tid = threadIdx.x;
    const int jj = bj * VE_TILE + tid;
    if (jj < n_ng) { const int id = ve.ng_list[jj]; s_id[tid] = id; s_p[tid] = ve.w[id]; }
    __syncthreads();
    const int ii = bi * VE_TILE + tid;
    if (ii >= n_ng) return;
    const int me = ve.ng_list[ii];
    const float4 p = ve.w[me];
    const int nj = min(VE_TILE, n_ng - bj * VE_TILE);
    for (int j = (bi == bj ? tid + 1 : 0); j < nj; ++j) {
        const float4 q = s_p[j];
        const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 <= tol2) ve_union(ve.parent, me, s_id[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_ve_clusters
// ---------------------------------------------------------------------------------------------------------------
#define VE_SORT_MAX 8192     // view points the one-workgroup ordering handles (keys in LDS)
#define VE_HMAX (VE_SORT_MAX / 5 + 8)   // clusters have >= 5 points
// generateRandomFloat :1551-1553 fed from the tabulated rand() stream (the birth stage's stream, dspmap_kernels.hip)
__device__ __forceinline__ float ve_rand_float(const DevState& s, int rtab_n, int c, float lo, float hi) {
    const int r = s.r_tab[c % max(rtab_n, 1)];
    return lo + __fdiv_rn((float)r, __fdiv_rn((float)2147483647, (hi - lo)));
}

// cost / gate of (dynamic cluster r, last cluster c) :1459-1472
__device__ __forceinline__ float ve_cost(const VeCluster& a, const float* __restrict__ last, int c, bool* gate) {
    const float ex = a.cx - last[c * 5], ey = a.cy - last[c * 5 + 1], ez = a.cz - last[c * 5 + 2];
    const float d = sqrtf(ex * ex + ey * ey + ez * ez);   // clusterDistance :1369-1374
    const int dn = abs(a.point_num - __float_as_int(last[c * 5 + 3]));
    if (dn > 100 || d >= 1.5f) { *gate = false; return 1.5f * 5000.f; }
    *gate = true;
    return __fdiv_rn(d, 1.5f) * 1000.f;
}

__global__ void __launch_bounds__(VE_NT) k_ve_clusters(DevState s, VelEst ve, FilterParams fp) {
    __shared__ unsigned s_key[VE_SORT_MAX];
    __shared__ double s_hu[VE_HMAX], s_hv[VE_HMAX], s_minv[VE_HMAX];   // Hungarian: potentials, column minima
    __shared__ int s_hp[VE_HMAX], s_way[VE_HMAX], s_used[VE_HMAX];     // row of a column, predecessor column, column used
    __shared__ int s_tmp[17];
    __shared__ int s_k, s_ndyn;
    __shared__ double s_big;
    const int tid = threadIdx.x;
    const int n = ve.n[0];
    if (n == 0) return;   // :1379: an empty view leaves the previous output (and the previous clusters) untouched
    const int n_ng = ve.n[1];
    VeCluster* cl = ve.cl;
    int* root_of = ve.root;      // [cap] root of every view point, -1 = ground / not clustered
    int* size_of = ve.size;      // [cap] component size at its root
    // ---- components: flatten, count
    for (int i = tid; i < n; i += VE_NT) size_of[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += VE_NT) {
        int r = -1;
        if (ve.parent[i] >= 0) { r = ve_find(ve.parent, i); atomicAdd(&size_of[r], 1); }
        root_of[i] = r;
    }
    __syncthreads();
    // ---- clusters = components of 5 .. 10000 points (:1412-1413), listed in seed (= root index) order
    int base = 0;
    for (int b0 = 0; b0 < n; b0 += VE_NT) {
        const int i = b0 + tid;
        const bool is_c = i < n && root_of[i] == i && size_of[i] >= 5 && size_of[i] <= 10000;
        int tot;
        const int k = base + ve_excl_scan(is_c ? 1 : 0, s_tmp, &tot);
        if (is_c) {
            VeCluster c;
            c.cx = c.cy = c.cz = 0.f; c.point_num = size_of[i];
            c.vx = c.vy = c.vz = -10000.f; c.intensity = 0.f;   // :104-108
            c.root = i; c.start = 0; c.is_dyn = 0; c.dyn_idx = -1;
            cl[k] = c;
        }
        base += tot;
    }
    if (tid == 0) s_k = base;
    __syncthreads();
    const int K = s_k;
    // ---- PCL returns the clusters largest first (equal sizes: seed order); rank -> position in that order
    int* rank_of = ve.rank;      // [cap/5+1] rank of cluster k
    int* by_rank = ve.by_rank;   // inverse
    for (int k = tid; k < K; k += VE_NT) {
        const int sz = cl[k].point_num;
        int r = 0;
        for (int q = 0; q < K; ++q) { const int sq = cl[q].point_num; r += (sq > sz || (sq == sz && q < k)) ? 1 : 0; }
        rank_of[k] = r; by_rank[r] = k;
    }
    __syncthreads();
    // cluster id of a root: write the rank at the root's slot of size_of (no longer needed as a size)
    for (int k = tid; k < K; k += VE_NT) size_of[cl[k].root] = -1 - rank_of[k];   // negative = "rank follows"
    __syncthreads();
    // ---- order the clustered points by (cluster rank, point index): one bitonic sort of 32-bit keys in LDS
    int npad = 1;
    while (npad < n) npad <<= 1;
    for (int i = tid; i < npad; i += VE_NT) {
        unsigned key = 0xffffffffu;
        if (i < n) {
            const int r = root_of[i];
            if (r >= 0 && size_of[r] < 0) key = ((unsigned)(-1 - size_of[r]) << 13) | (unsigned)i;
        }
        s_key[i] = key;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= npad; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npad; i += VE_NT) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned a = s_key[i], b = s_key[p];
                    const bool up = (i & k2) == 0;
                    if ((a > b) == up) { s_key[i] = b; s_key[p] = a; }
                }
            }
            __syncthreads();
        }
    // starts of the clusters in the sorted order
    {
        int run = 0;
        for (int b0 = 0; b0 < K; b0 += VE_NT) {
            const int r = b0 + tid;
            int tot;
            const int st = run + ve_excl_scan(r < K ? cl[by_rank[r]].point_num : 0, s_tmp, &tot);
            if (r < K) cl[by_rank[r]].start = st;
            run += tot;
        }
    }
    __syncthreads();
    // ---- per cluster (in rank order): intensity draw (:1422, every cluster), centroid (:1424-1434), static test (:1436)
    const int r_cur = s.fs->r_cur;
    for (int r = tid; r < K; r += VE_NT) {
        VeCluster c = cl[by_rank[r]];
        c.intensity = ve_rand_float(s, fp.rtab_n, r_cur + r, 0.1f, 1.f);
        bool stat = c.point_num > 200;      // DYNAMIC_CLUSTER_MAX_POINT_NUM :52 (its centroid is never used)
        if (!stat) {
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int j = 0; j < c.point_num; ++j) {     // ascending point index = PCL's sorted indices
                const float4 p = ve.w[s_key[c.start + j] & 8191u];
                sx += p.x; sy += p.y; sz += p.z;
            }
            c.cx = __fdiv_rn(sx, (float)c.point_num); c.cy = __fdiv_rn(sy, (float)c.point_num); c.cz = __fdiv_rn(sz, (float)c.point_num);
            stat = c.cz > 1.5f;             // DYNAMIC_CLUSTER_MAX_CENTER_HEIGHT :53
        }
        c.is_dyn = stat ? 0 : 1;
        cl[by_rank[r]] = c;
    }
    __syncthreads();
    // possibly-dynamic clusters in rank order -> dyn_idx
    {
        int run = 0;
        for (int b0 = 0; b0 < K; b0 += VE_NT) {
            const int r = b0 + tid;
            int tot;
            const int di = run + ve_excl_scan(r < K ? cl[by_rank[r]].is_dyn : 0, s_tmp, &tot);
            if (r < K && cl[by_rank[r]].is_dyn) { cl[by_rank[r]].dyn_idx = di; ve.dyn_list[di] = by_rank[r]; }
            run += tot;
        }
        if (tid == 0) s_ndyn = run;
    }
    __syncthreads();
    const int n_dyn = s_ndyn;
    const int n_last = ve.n[2];
    const float dt = s.fpar->dt;
    // ---- Hungarian matching (:1454-1499), one wavefront, columns on the lanes
    if (n_last > 0 && n_dyn > 0 && (double)dt > 0.00001 && (double)dt < 10.0) {
        const int nr = n_dyn, nc = n_last, N = max(nr, nc);   // N <= cap / 5 < VE_HMAX
        // padding value = the largest real cost
        if (tid == 0) s_big = 0.0;
        __syncthreads();
        {
            double big = 0.0;
            for (int e = tid; e < nr * nc; e += VE_NT) {
                bool g;
                big = fmax(big, (double)ve_cost(cl[ve.dyn_list[e / nc]], ve.last, e % nc, &g));
            }
            for (int o = 32; o > 0; o >>= 1) big = fmax(big, __shfl_xor(big, o, WAVE));
            if ((tid & 63) == 0) atomicMax((unsigned long long*)&s_big, (unsigned long long)__double_as_longlong(big));   // non-negative doubles order like integers
        }
        __syncthreads();
        const double big = s_big;
        for (int j = tid; j <= N; j += VE_NT) { s_hu[j] = 0.0; s_hv[j] = 0.0; s_hp[j] = 0; s_way[j] = 0; }
        __syncthreads();
        if (tid < 64) {
            // Kuhn-Munkres with potentials (u, v), rows added one at a time; every column j belongs to lane j % 64 in
            // every loop; LDS operations of one wavefront execute in program order, so no barrier is needed inside
            const int l = tid;
            for (int i = 1; i <= N; ++i) {
                if (l == 0) s_hp[0] = i;
                int j0 = 0;
                for (int j = l; j <= N; j += 64) { s_minv[j] = 1e300; s_used[j] = 0; }
                asm volatile("" ::: "memory");
                do {
                    if (l == (j0 & 63)) s_used[j0] = 1;
                    asm volatile("" ::: "memory");
                    const int i0 = s_hp[j0];
                    const double ui0 = s_hu[i0];
                    double delta = 1e300;
                    int j1 = 0x7fffffff;
                    for (int j = l; j <= N; j += 64) {
                        if (j >= 1 && !s_used[j]) {
                            double a = big;
                            if (i0 <= nr && j <= nc) { bool g; a = (double)ve_cost(cl[ve.dyn_list[i0 - 1]], ve.last, j - 1, &g); }
                            const double cur = a - ui0 - s_hv[j];
                            double mv = s_minv[j];
                            if (cur < mv) { mv = cur; s_minv[j] = cur; s_way[j] = j0; }
                            if (mv < delta) { delta = mv; j1 = j; }
                        }
                    }
                    // first minimum over the columns: smallest value, then smallest column (the sequential loop's strict <)
                    for (int o = 32; o > 0; o >>= 1) {
                        const double od = __shfl_xor(delta, o, WAVE);
                        const int oj = __shfl_xor(j1, o, WAVE);
                        if (od < delta || (od == delta && oj < j1)) { delta = od; j1 = oj; }
                    }
                    asm volatile("" ::: "memory");
                    for (int j = l; j <= N; j += 64) {
                        if (s_used[j]) { s_hu[s_hp[j]] += delta; s_hv[j] -= delta; }   // rows p[j] of used columns are distinct
                        else s_minv[j] -= delta;
                    }
                    asm volatile("" ::: "memory");
                    j0 = j1;
                } while (s_hp[j0] != 0);
                if (l == 0) {
                    do { const int jn = s_way[j0]; s_hp[j0] = s_hp[jn]; j0 = jn; } while (j0);
                }
                asm volatile("" ::: "memory");
            }
        }
        __syncthreads();
        // matched pairs with an open gate: velocity, inherited intensity, the 5 m/s limit (:1481-1493)
        for (int j = 1 + tid; j <= N; j += VE_NT) {
            const int i = s_hp[j];
            if (i >= 1 && i <= nr && j <= nc) {
                VeCluster c = cl[ve.dyn_list[i - 1]];
                bool gate;
                (void)ve_cost(c, ve.last, j - 1, &gate);
                if (gate) {
                    c.vx = __fdiv_rn(c.cx - ve.last[(j - 1) * 5], dt);
                    c.vy = __fdiv_rn(c.cy - ve.last[(j - 1) * 5 + 1], dt);
                    c.vz = __fdiv_rn(c.cz - ve.last[(j - 1) * 5 + 2], dt);
                    const float v = sqrtf(c.vx * c.vx + c.vy * c.vy + c.vz * c.vz);
                    c.intensity = ve.last[(j - 1) * 5 + 4];
                    if (v > 5.f) c.vx = c.vy = c.vz = 0.f;
                    cl[ve.dyn_list[i - 1]] = c;
                }
            }
        }
        __syncthreads();
    }
    // ---- the birth cloud: dynamic clusters' points (:1505-1524), then the ground points in view order followed by the
    //      static clusters' points, cluster by cluster (static_points: :1396,1438-1441 -> :1529-1540)
    int n_dyn_pts = 0, n_stat_pts = 0;
    {   // output bases of the clusters (rank order), dynamic and static runs separately
        int run_d = 0, run_s = 0;
        for (int b0 = 0; b0 < K; b0 += VE_NT) {
            const int r = b0 + tid;
            const bool has = r < K;
            const int kq = has ? by_rank[r] : 0;
            const int szd = has && cl[kq].is_dyn ? cl[kq].point_num : 0;
            const int szs = has && !cl[kq].is_dyn ? cl[kq].point_num : 0;
            int td, ts;
            const int od = run_d + ve_excl_scan(szd, s_tmp, &td);
            const int os = run_s + ve_excl_scan(szs, s_tmp, &ts);
            if (has) rank_of[kq] = cl[kq].is_dyn ? od : os;     // rank_of now holds the cluster's output base inside its run
            run_d += td; run_s += ts;
        }
        n_dyn_pts = run_d; n_stat_pts = run_s;
    }
    __syncthreads();
    const int n_ground = n - n_ng;
    BirthSrc* out = s.birth;
    // clustered points, from the sorted order
    int n_clustered = n_dyn_pts + n_stat_pts;
    for (int p = tid; p < n_clustered; p += VE_NT) {
        const unsigned key = s_key[p];
        const int i = (int)(key & 8191u), r = (int)(key >> 13);
        const int kq = by_rank[r];
        const VeCluster c = cl[kq];
        const int within = p - c.start;
        const float4 w = ve.w[i];
        BirthSrc b;
        b.x = w.x; b.y = w.y; b.z = w.z;
        int pos;
        if (c.is_dyn) { b.nx = c.vx; b.ny = c.vy; b.nz = c.vz; b.intensity = c.intensity; pos = rank_of[kq] + within; }
        else { b.nx = b.ny = b.nz = 0.f; b.intensity = 0.f; pos = n_dyn_pts + n_ground + rank_of[kq] + within; }
        out[pos] = b;
    }
    // ground points in view order
    {
        int run = 0;
        for (int b0 = 0; b0 < n; b0 += VE_NT) {
            const int i = b0 + tid;
            const bool g = i < n && ve.parent[i] < 0;
            int tot;
            const int pg = run + ve_excl_scan(g ? 1 : 0, s_tmp, &tot);
            if (g) {
                const float4 w = ve.w[i];
                BirthSrc b;
                b.x = w.x; b.y = w.y; b.z = w.z; b.nx = b.ny = b.nz = 0.f; b.intensity = 0.f;
                out[n_dyn_pts + pg] = b;
            }
            run += tot;
        }
    }
    // ---- clusters_feature_vector_dynamic_last = clusters_feature_vector_dynamic (:1542); the rand() stream moved on by K
    for (int d0 = tid; d0 < n_dyn; d0 += VE_NT) {
        const VeCluster c = cl[ve.dyn_list[d0]];
        ve.last[d0 * 5] = c.cx; ve.last[d0 * 5 + 1] = c.cy; ve.last[d0 * 5 + 2] = c.cz;
        ve.last[d0 * 5 + 3] = __int_as_float(c.point_num); ve.last[d0 * 5 + 4] = c.intensity;
    }
    if (tid == 0) {
        ve.n[2] = n_dyn;
        s.fs->est_n = n_dyn_pts + n_ground + n_stat_pts;
        s.fs->r_cur = (int)(((long long)r_cur + K) % max(fp.rtab_n, 1));
    }
}

void launch_velocity_estimator(const LaunchCtx& c, int n_pts_grid) {
    const VelEst& ve = c.ve;
    hipLaunchKernelGGL(k_ve_view, dim3(1), dim3(VE_NT), 0, c.stream, c.s, ve);
    const int cap = n_pts_grid < ve.cap ? n_pts_grid : ve.cap;
    const int nt = (cap + VE_TILE - 1) / VE_TILE;
    if (nt > 0) hipLaunchKernelGGL(k_ve_pairs, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(VE_TILE), 0, c.stream, c.s, ve, nt);
    hipLaunchKernelGGL(k_ve_clusters, dim3(1), dim3(VE_NT), 0, c.stream, c.s, ve, c.fp);
}
