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
//     leaves no trace.  Here: the non-ground points are hashed into a grid of tolerance-sized cells, every point looks
//     at the 27 cells around it, close pairs (the squared distance in the reference's fp32 arithmetic) are united in a
//     lock-free union-find whose root is the component's SMALLEST index (atomicMin hooking): the result does not depend
//     on the order the pairs are visited in.
//   * the assignment is the minimum-cost one; the O(n^3) Hungarian algorithm runs in one wavefront with the columns on
//     the lanes, in double precision and with first-minimum tie breaking = the sequential algorithm, step for step.
// Everything a cluster sums (centroids) is summed sequentially in ascending point index, the reference's order.
//
// The cloud is a few thousand points: the whole job is two one-workgroup kernels that live in LDS (positions, hash grid,
// union-find: ~10^5 dependent steps that cost an LDS access each instead of an L2 round trip):
//   k_ve_components  view in input order -> world frame -> ground split -> hash grid -> unions -> root of every point
//   k_ve_clusters    components -> clusters -> PCL's order (stable radix sort) -> centroids -> matching -> tags -> birth cloud
#include <hip/hip_runtime.h>
#include "dspmap_device.h"
#include "dspmap_kernels.h"
#include "dspmap_birth.h"

#define VE_NT 1024
#ifdef VE_DEBUG
#define VE_MARK(k) do { if (threadIdx.x == 0) ((long long*)&ve.cl[VE_CAP / 5 + 2])[k] = (long long)wall_clock64(); } while (0)
#else
#define VE_MARK(k) do {} while (0)
#endif
#define VE_CAP 6144                 // view points the estimator handles (host: larger clouds go to the host stage)
#define VE_HB 4096                  // hash buckets
#define VE_NIL 0xffffu
#define VE_HMAX (VE_CAP / 5 + 8)    // clusters have >= 5 points
#define VE_KLDS 128                 // clusters whose records k_ve_clusters keeps in LDS

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

// ---- union-find in LDS.  Parents only ever DECREASE (every write is an atomicMin) and always point at a member of the
// same component, so concurrent finds, path halving and hooking cannot separate what was united.
__device__ __forceinline__ int ve_find(int* parent, int x) {
    int p = parent[x];
    while (p != x) {
        const int g = parent[p];
        if (g != p) atomicMin(&parent[x], g);   // path halving
        x = p; p = g;
    }
    return x;
}
__device__ __forceinline__ int ve_union(int* parent, int a, int b) {
    while (true) {
        a = ve_find(parent, a); b = ve_find(parent, b);
        if (a == b) return a;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int old = atomicMin(&parent[hi], lo);   // hook the larger root under the smaller one
        if (old == hi) return lo;                     // hi was still a root: done
        a = old; b = lo;                              // re-parented meanwhile: unite its new parent with lo
    }
}
__device__ __forceinline__ unsigned ve_hash(int cx, int cy, int cz) {
    return ((unsigned)cx * 73856093u ^ (unsigned)cy * 19349663u ^ (unsigned)cz * 83492791u) & (VE_HB - 1);
}

// ---------------------------------------------------------------------------------------------------------------
// k_ve_components: cloud_in_current_view_rotated (:244-257) in input order -> world frame (:1389-1391), ground split
// (:1393), and the close pairs of the non-ground points under the cluster tolerance (:1411).
// VE_NB workgroups.  Every workgroup builds the same picture in its LDS (the cloud is small: compacting it and hashing it
// costs less than handing it over through memory), then looks for the close pairs of ITS contiguous slice of the
// non-ground points, unites them in a local union-find and emits the resulting spanning forest (a few hundred edges
// instead of ~30 close pairs per point); k_ve_clusters merges the forests.  One workgroup alone walks ~10^5 dependent
// LDS steps (144 us measured); the slices cut that chain by VE_NB.
// Output: ve.w[v] world position of view point v, ve.root[v] = -1 for ground points, ve.ng_view[g] = view index of
//         non-ground point g, ve.n[0] = view points, ve.n[1] = non-ground points, ve.edges / ve.ecnt per workgroup.
// ---------------------------------------------------------------------------------------------------------------
#define VE_NB 32
#define VE_IPT ((VE_CAP + VE_NT - 1) / VE_NT)   // consecutive input points per thread
__global__ void __launch_bounds__(VE_NT) k_ve_components(DevState s, VelEst ve) {
    __shared__ float sx[VE_CAP], sy[VE_CAP], sz[VE_CAP];   // non-ground points, by non-ground rank g
    __shared__ int s_parent[VE_CAP];
    __shared__ unsigned short s_next[VE_CAP];
    __shared__ int s_head[VE_HB];
    __shared__ int s_tmp[17];
    __shared__ int s_ne;
    const int n_pts = min(s.fpar->n_pts, VE_CAP);
    const float cx = s.fpar->cur_pos[0], cy = s.fpar->cur_pos[1], cz = s.fpar->cur_pos[2];
    const float res_f = s.fpar->res_filter;
    const float tol = 2 * res_f, tol2 = tol * tol;   // :1411
    const int tid = threadIdx.x;
    const bool first = blockIdx.x == 0;
    for (int h = tid; h < VE_HB; h += VE_NT) s_head[h] = -1;
    if (tid == 0) s_ne = 0;
    // ---- the view in input order: thread t owns the input points [t * VE_IPT, (t + 1) * VE_IPT); one scan ranks both the
    //      view points (low half of the packed count) and the non-ground ones (high half)
    float wx[VE_IPT], wy[VE_IPT], wz[VE_IPT];
    int flag[VE_IPT];   // 0 not in view, 1 ground, 2 non-ground
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < VE_IPT; ++k) {
        const int i = tid * VE_IPT + k;
        flag[k] = 0; wx[k] = wy[k] = wz[k] = 0.f;
        if (i < n_pts && s.pt_pyr[i] >= 0) {
            const float4 r = s.pt_rot[i];
            wx[k] = r.x + cx; wy[k] = r.y + cy; wz[k] = r.z + cz;   // :1389-1391
            flag[k] = wz[k] > res_f ? 2 : 1;                          // :1393
            cnt += flag[k] == 2 ? 0x10001 : 1;
        }
    }
    int tot;
    int off = ve_excl_scan(cnt, s_tmp, &tot);
    const int n_view = tot & 0xffff, n_ng = tot >> 16;
    {
        int pv = off & 0xffff, pg = off >> 16;
#pragma unroll
        for (int k = 0; k < VE_IPT; ++k) {
            if (flag[k]) {
                if (first) ve.w[pv] = make_float4(wx[k], wy[k], wz[k], 0.f);
                if (flag[k] == 2) {
                    sx[pg] = wx[k]; sy[pg] = wy[k]; sz[pg] = wz[k]; s_parent[pg] = pg;
                    if (first) ve.ng_view[pg] = pv;
                    ++pg;
                } else if (first) ve.root[pv] = -1;     // ground points take no part in the clustering
                ++pv;
            }
        }
    }
    if (first && tid == 0) { ve.n[0] = n_view; ve.n[1] = n_ng; }
    __syncthreads();
    // ---- hash grid of tolerance-sized cells: bucket chains through s_next
    for (int g = tid; g < n_ng; g += VE_NT) {
        const int kx = (int)floorf(__fdiv_rn(sx[g], tol)), ky = (int)floorf(__fdiv_rn(sy[g], tol)), kz = (int)floorf(__fdiv_rn(sz[g], tol));
        const int prev = atomicExch(&s_head[ve_hash(kx, ky, kz)], g);
        s_next[g] = (unsigned short)(prev < 0 ? VE_NIL : prev);
    }
    __syncthreads();
    // ---- this workgroup's slice; work item = (point of the slice, one of the 27 cells around it): every pair (j < g)
    //      that may be closer than the tolerance sits in those cells
    const int per = (n_ng + VE_NB - 1) / VE_NB;
    const int g_lo = (int)blockIdx.x * per, g_hi = min(n_ng, g_lo + per);
    for (int it = tid; it < (g_hi - g_lo) * 27; it += VE_NT) {
        const int g = g_lo + it / 27, c = it % 27;
        const float px = sx[g], py = sy[g], pz = sz[g];
        const int kx = (int)floorf(__fdiv_rn(px, tol)) + c % 3 - 1, ky = (int)floorf(__fdiv_rn(py, tol)) + (c / 3) % 3 - 1,
                  kz = (int)floorf(__fdiv_rn(pz, tol)) + c / 9 - 1;
        // (two of the 27 cells may share a bucket: its chain is then walked twice and a pair united twice, which changes
        // nothing; a chain also holds the points of unrelated cells: the distance decides)
        for (int j = s_head[ve_hash(kx, ky, kz)]; j >= 0; j = (s_next[j] == VE_NIL ? -1 : (int)s_next[j])) {
            if (j >= g) continue;
            const float ex = sx[j] - px, ey = sy[j] - py, ez = sz[j] - pz;
            const float d2 = ex * ex + ey * ey + ez * ez;
            if (d2 <= tol2) {
                const int rg = ve_find(s_parent, g), rj = ve_find(s_parent, j);   // most close pairs already share a root
                if (rg != rj) ve_union(s_parent, rg, rj);
            }
        }
    }
    __syncthreads();
    // ---- the slice's spanning forest
    unsigned* edges = ve.edges + (size_t)blockIdx.x * VE_CAP;
    for (int g = tid; g < n_ng; g += VE_NT) {
        const int r = ve_find(s_parent, g);
        if (r != g) edges[atomicAdd(&s_ne, 1)] = ((unsigned)g << 16) | (unsigned)r;
    }
    __syncthreads();
    if (tid == 0) ve.ecnt[blockIdx.x] = s_ne;
}

// ---------------------------------------------------------------------------------------------------------------
// k_ve_clusters
// ---------------------------------------------------------------------------------------------------------------
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

// lanes of the wave holding the same 6-bit digit as this lane (all lanes active)
__device__ __forceinline__ u64 ve_match6(unsigned d) {
    u64 m = ~0ull;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const u64 bal = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return m;
}

// one stable pass of the radix sort: keys of `src` ordered by ((key >> shift) & 63), equal digits keep their order.
// The 16 waves own consecutive ranges; s_cnt: [64 digits][16 waves].
__device__ __forceinline__ void ve_radix_pass(const unsigned* src, unsigned* dst, int npad, int shift, int* s_cnt, int* s_tmp) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int per_wave = npad / 16;   // npad is a multiple of 1024
    s_cnt[tid] = 0;
    __syncthreads();
    for (int it = 0; it < per_wave; it += 64) {
        const unsigned d = (src[w * per_wave + it + l] >> shift) & 63u;
        const u64 same = ve_match6(d);
        if (l == __ffsll((long long)same) - 1) s_cnt[d * 16 + w] += (int)__popcll(same);   // this wave's own counters
    }
    __syncthreads();
    int tot;
    const int off = ve_excl_scan(s_cnt[tid], s_tmp, &tot);   // digit-major, wave-minor = the stable order
    s_cnt[tid] = off;
    __syncthreads();
    for (int it = 0; it < per_wave; it += 64) {
        const unsigned key = src[w * per_wave + it + l];
        const unsigned d = (key >> shift) & 63u;
        const u64 same = ve_match6(d);
        const int base = s_cnt[d * 16 + w];
        dst[base + (int)__popcll(same & lanemask_lt())] = key;
        if (l == __ffsll((long long)same) - 1) s_cnt[d * 16 + w] = base + (int)__popcll(same);
    }
    __syncthreads();
}

__device__ __forceinline__ void ve_clusters_block(const DevState& s, const VelEst& ve, const FilterParams& fp) {
    __shared__ unsigned s_key[2][VE_CAP];   // sort buffers; while the components are counted: s_key[1] = size at the root
    __shared__ double s_hu[VE_HMAX], s_hv[VE_HMAX], s_minv[VE_HMAX];   // Hungarian: potentials, column minima
    __shared__ int s_hp[VE_HMAX], s_way[VE_HMAX], s_used[VE_HMAX];     // row of a column, predecessor column, column used
    __shared__ int s_cnt[1024];
    __shared__ int s_tmp[17];
    __shared__ int s_root[VE_CAP];              // root (view index) of every view point, -1 = ground
    __shared__ VeCluster s_cl[VE_KLDS];         // the usual few dozen clusters live in LDS (more: the global arrays)
    __shared__ int s_rank[VE_KLDS], s_byrank[VE_KLDS], s_dyn[VE_KLDS];
    __shared__ int s_k, s_ndyn;
    __shared__ double s_big;
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int n = ve.n[0];
    const int r_cur = s.fs->r_cur;   // (requested now: the per-cluster draws below need it)
    if (n == 0) return;   // :1379: an empty view leaves the previous output (and the previous clusters) untouched
    const int n_ng = ve.n[1];
    VE_MARK(0);
    // ---- merge the spanning forests of k_ve_components' slices: the union-find over the non-ground points lives in
    //      s_key[0] for the moment; root of a component = its smallest index
    {
        int* parent = (int*)s_key[0];
        for (int g = tid; g < n_ng; g += VE_NT) parent[g] = g;
        __syncthreads();
        // all slices' edge counts in one round trip, then one flat loop over the edges (independent loads)
        if (tid < VE_NB) s_cnt[tid] = ve.ecnt[tid];
        __syncthreads();
        if (tid == 0) { int run = 0; for (int b = 0; b < VE_NB; ++b) { const int c = s_cnt[b]; s_cnt[b] = run; run += c; } s_cnt[VE_NB] = run; }
        __syncthreads();
        const int n_edges = s_cnt[VE_NB];
        for (int f = tid; f < n_edges; f += VE_NT) {
            int b = 0;
            while (b + 1 < VE_NB && s_cnt[b + 1] <= f) ++b;
            const unsigned pr = ve.edges[(size_t)b * VE_CAP + (f - s_cnt[b])];
            ve_union(parent, (int)(pr >> 16), (int)(pr & 0xffffu));
        }
        __syncthreads();
        for (int i = tid; i < n; i += VE_NT) s_root[i] = -1;   // ground points take no part in the clustering
        __syncthreads();
        for (int g = tid; g < n_ng; g += VE_NT) s_root[ve.ng_view[g]] = ve.ng_view[ve_find(parent, g)];
        __syncthreads();
    }
    VE_MARK(1);
    const int* root_of = s_root;     // view index of the component's first point, -1 = ground
    int* size_of = (int*)s_key[1];   // component size at its root; later -1 - (rank of the cluster)
    // ---- component sizes (one LDS atomic per distinct root of a wavefront's 64 points)
    for (int i = tid; i < VE_CAP; i += VE_NT) size_of[i] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n; b0 += VE_NT) {
        const int i = b0 + tid;
        const int r = i < n ? root_of[i] : -1;
        u64 todo = __ballot(r >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int rk = __shfl(r, leader, WAVE);
            const u64 grp = __ballot(r == rk);
            if (l == leader) atomicAdd(&size_of[rk], (int)__popcll(grp));
            todo &= ~grp;
        }
    }
    __syncthreads();
    VE_MARK(2);
    // ---- clusters = components of 5 .. 10000 points (:1412-1413), listed in seed (= root index) order.
    //      Thread t owns the points [t * VE_IPT, (t + 1) * VE_IPT): one scan for the whole cloud.
    {
        int mine = 0;
        bool is_c[VE_IPT];
#pragma unroll
        for (int q = 0; q < VE_IPT; ++q) {
            const int i = tid * VE_IPT + q;
            is_c[q] = i < n && root_of[i] == i && size_of[i] >= 5 && size_of[i] <= 10000;
            mine += is_c[q] ? 1 : 0;
        }
        int tot;
        int k = ve_excl_scan(mine, s_tmp, &tot);
#pragma unroll
        for (int q = 0; q < VE_IPT; ++q) {
            if (is_c[q]) {
                const int i = tid * VE_IPT + q;
                VeCluster c;
                c.cx = c.cy = c.cz = 0.f; c.point_num = size_of[i];
                c.vx = c.vy = c.vz = -10000.f; c.intensity = 0.f;   // :104-108
                c.root = i; c.start = 0; c.is_dyn = 0; c.dyn_idx = -1;
                ve.cl[k++] = c;
            }
        }
        if (tid == 0) s_k = tot;
    }
    __syncthreads();
    const int K = s_k;
    // the usual few dozen clusters: their records and index arrays move to LDS (the phases below are chains of
    // dependent reads of these small arrays -- an LDS access each instead of an L2 round trip)
    const bool k_lds = K <= VE_KLDS;
    if (k_lds) for (int k = tid; k < K; k += VE_NT) s_cl[k] = ve.cl[k];
    __syncthreads();
    VeCluster* cl = k_lds ? s_cl : ve.cl;
    int* dyn_list = k_lds ? s_dyn : ve.dyn_list;
    // the clusters' display intensities (:1422: one rand() per cluster, in cluster order) are requested now -- a random
    // access into the 16 MB table, i.e. an HBM round trip that the ranking and the sort below hide
    float my_int[(VE_HMAX + VE_NT - 1) / VE_NT];
#pragma unroll
    for (int q = 0; q < (VE_HMAX + VE_NT - 1) / VE_NT; ++q) {
        const int r = q * VE_NT + tid;
        my_int[q] = r < K ? ve_rand_float(s, fp.rtab_n, r_cur + r, 0.1f, 1.f) : 0.f;
    }
    VE_MARK(3);
    // ---- PCL returns the clusters largest first (equal sizes: seed order); rank -> position in that order
    int* rank_of = k_lds ? s_rank : ve.rank;        // rank of cluster k; later its output base
    int* by_rank = k_lds ? s_byrank : ve.by_rank;   // inverse
    for (int k = tid; k < K; k += VE_NT) {
        const int sz = cl[k].point_num;
        int r = 0;
        for (int q = 0; q < K; ++q) { const int sq = cl[q].point_num; r += (sq > sz || (sq == sz && q < k)) ? 1 : 0; }
        rank_of[k] = r; by_rank[r] = k;
    }
    __syncthreads();
    for (int k = tid; k < K; k += VE_NT) size_of[cl[k].root] = -1 - rank_of[k];   // negative = "the cluster's rank follows"
#pragma unroll
    for (int q = 0; q < (VE_HMAX + VE_NT - 1) / VE_NT; ++q) {
        const int r = q * VE_NT + tid;
        if (r < K) cl[by_rank[r]].intensity = my_int[q];
    }
    __syncthreads();
    VE_MARK(4);
    // ---- order the clustered points by (cluster rank, point index): the keys start in index order, so a STABLE sort by
    //      the 12-bit rank (two 6-bit passes) is enough; points outside every cluster carry the largest rank and go last
    int npad = ((n + 1023) >> 10) << 10;
    for (int i = tid; i < npad; i += VE_NT) {
        unsigned key = 0xffffffffu;
        if (i < n) {
            const int r = root_of[i];
            if (r >= 0 && size_of[r] < 0) key = ((unsigned)(-1 - size_of[r]) << 13) | (unsigned)i;
        }
        s_key[0][i] = key;
    }
    __syncthreads();
    ve_radix_pass(s_key[0], s_key[1], npad, 13, s_cnt, s_tmp);   // (size_of lived in s_key[1]: consumed above)
    const bool one_pass = K <= 63;   // ranks below 63 and the "no cluster" digit 63: the low digit orders everything
    if (!one_pass) ve_radix_pass(s_key[1], s_key[0], npad, 19, s_cnt, s_tmp);
    const unsigned* sorted = one_pass ? s_key[1] : s_key[0];
    float* s_cost = (float*)(one_pass ? s_key[0] : s_key[1]);   // the free buffer: the Hungarian's cost matrix
    VE_MARK(5);
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
    VE_MARK(6);
    // ---- per cluster (in rank order): intensity draw (:1422, every cluster), centroid (:1424-1434), static test (:1436).
    //      One wavefront per cluster: 64 members are fetched at once, then summed one after another in ascending index
    //      (the reference's fp32 order) by every lane redundantly.
    for (int r = wave; r < K; r += VE_NT / 64) {
        VeCluster c = cl[by_rank[r]];
        bool stat = c.point_num > 200;      // DYNAMIC_CLUSTER_MAX_POINT_NUM :52 (such a cluster's centroid is never used)
        if (!stat) {
            float ax = 0.f, ay = 0.f, az = 0.f;
            float4 p[4];   // <= 200 members: all four chunks are requested before the first sum (one memory round trip)
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const int j = ch * 64 + l;
                p[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < c.point_num) p[ch] = ve.w[sorted[c.start + j] & 8191u];
            }
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const int m = min(64, c.point_num - ch * 64);
                for (int q0 = 0; q0 < m; q0 += 8) {   // lane indices are wave-uniform: v_readlane, no LDS crossbar; eight members are
                    float xs[8], ys[8], zs[8];        // fetched ahead of the three dependent chains of additions
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        xs[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[ch].x), q0 + u));
                        ys[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[ch].y), q0 + u));
                        zs[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[ch].z), q0 + u));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (q0 + u < m) { ax += xs[u]; ay += ys[u]; az += zs[u]; }
                }
            }
            c.cx = __fdiv_rn(ax, (float)c.point_num); c.cy = __fdiv_rn(ay, (float)c.point_num); c.cz = __fdiv_rn(az, (float)c.point_num);
            stat = c.cz > 1.5f;             // DYNAMIC_CLUSTER_MAX_CENTER_HEIGHT :53
        }
        c.is_dyn = stat ? 0 : 1;
        if (l == 0) cl[by_rank[r]] = c;
    }
    __syncthreads();
    VE_MARK(7);
    // possibly-dynamic clusters in rank order -> dyn_idx
    {
        int run = 0;
        for (int b0 = 0; b0 < K; b0 += VE_NT) {
            const int r = b0 + tid;
            int tot;
            const int di = run + ve_excl_scan(r < K ? cl[by_rank[r]].is_dyn : 0, s_tmp, &tot);
            if (r < K && cl[by_rank[r]].is_dyn) { cl[by_rank[r]].dyn_idx = di; dyn_list[di] = by_rank[r]; }
            run += tot;
        }
        if (tid == 0) s_ndyn = run;
    }
    __syncthreads();
    const int n_dyn = s_ndyn;
    const int n_last = ve.n[2];
    const float dt = s.fpar->dt;
    VE_MARK(8);
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
                big = fmax(big, (double)ve_cost(cl[dyn_list[e / nc]], ve.last, e % nc, &g));
            }
            for (int o = 32; o > 0; o >>= 1) big = fmax(big, __shfl_xor(big, o, WAVE));
            if (l == 0) atomicMax((unsigned long long*)&s_big, (unsigned long long)__double_as_longlong(big));   // non-negative doubles order like integers
        }
        __syncthreads();
        const double big = s_big;
        for (int j = tid; j <= N; j += VE_NT) { s_hu[j] = 0.0; s_hv[j] = 0.0; s_hp[j] = 0; s_way[j] = 0; }
        // the usual case (a few dozen clusters): the padded cost matrix is laid out in LDS (s_key[1] is free by now)
        const bool in_lds = N <= 64;
        if (in_lds)
            for (int e = tid; e < N * N; e += VE_NT) {
                const int i0 = e / N, j = e % N;
                float a = (float)big;
                if (i0 < nr && j < nc) { bool g; a = ve_cost(cl[dyn_list[i0]], ve.last, j, &g); }
                s_cost[e] = a;
            }
        __syncthreads();
        if (tid < 64) {
            // Kuhn-Munkres with potentials (u, v), rows added one at a time; every column j belongs to lane j % 64 in
            // every loop; LDS operations of one wavefront execute in program order, so no barrier is needed inside
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
                            if (in_lds) a = (double)s_cost[(i0 - 1) * N + (j - 1)];
                            else if (i0 <= nr && j <= nc) { bool g; a = (double)ve_cost(cl[dyn_list[i0 - 1]], ve.last, j - 1, &g); }
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
                VeCluster c = cl[dyn_list[i - 1]];
                bool gate;
                (void)ve_cost(c, ve.last, j - 1, &gate);
                if (gate) {
                    c.vx = __fdiv_rn(c.cx - ve.last[(j - 1) * 5], dt);
                    c.vy = __fdiv_rn(c.cy - ve.last[(j - 1) * 5 + 1], dt);
                    c.vz = __fdiv_rn(c.cz - ve.last[(j - 1) * 5 + 2], dt);
                    const float v = sqrtf(c.vx * c.vx + c.vy * c.vy + c.vz * c.vz);
                    c.intensity = ve.last[(j - 1) * 5 + 4];
                    if (v > 5.f) c.vx = c.vy = c.vz = 0.f;
                    cl[dyn_list[i - 1]] = c;
                }
            }
        }
        __syncthreads();
    }
    VE_MARK(9);
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
    const int n_clustered = n_dyn_pts + n_stat_pts;
    for (int p = tid; p < n_clustered; p += VE_NT) {
        const unsigned key = sorted[p];
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
    // ground points in view order (thread t owns the points [t * VE_IPT, (t + 1) * VE_IPT))
    {
        int mine = 0;
#pragma unroll
        for (int q = 0; q < VE_IPT; ++q) { const int i = tid * VE_IPT + q; mine += (i < n && root_of[i] < 0) ? 1 : 0; }
        int tot;
        int pg = ve_excl_scan(mine, s_tmp, &tot);
#pragma unroll
        for (int q = 0; q < VE_IPT; ++q) {
            const int i = tid * VE_IPT + q;
            if (i < n && root_of[i] < 0) {
                const float4 w = ve.w[i];
                BirthSrc b;
                b.x = w.x; b.y = w.y; b.z = w.z; b.nx = b.ny = b.nz = 0.f; b.intensity = 0.f;
                out[n_dyn_pts + pg++] = b;
            }
        }
    }
    VE_MARK(10);
    // ---- clusters_feature_vector_dynamic_last = clusters_feature_vector_dynamic (:1542); the rand() stream moved on by K
    for (int d0 = tid; d0 < n_dyn; d0 += VE_NT) {
        const VeCluster c = cl[dyn_list[d0]];
        ve.last[d0 * 5] = c.cx; ve.last[d0 * 5 + 1] = c.cy; ve.last[d0 * 5 + 2] = c.cz;
        ve.last[d0 * 5 + 3] = __int_as_float(c.point_num); ve.last[d0 * 5 + 4] = c.intensity;
    }
    VE_MARK(11);
    if (tid == 0) {
#ifdef VE_DEBUG
        ((long long*)&ve.cl[VE_CAP / 5 + 2])[12] = K; ((long long*)&ve.cl[VE_CAP / 5 + 2])[13] = n_dyn; ((long long*)&ve.cl[VE_CAP / 5 + 2])[14] = n_last; ((long long*)&ve.cl[VE_CAP / 5 + 2])[15] = n;
#endif
        ve.n[2] = n_dyn;
        s.fs->est_n = n_dyn_pts + n_ground + n_stat_pts;
        s.fs->r_cur = (int)(((long long)r_cur + K) % max(fp.rtab_n, 1));
    }
}

// with_rank: the birth stage's rank (k_birth_rank's workgroup job: it needs nothing but the finished birth cloud) follows in
// the same workgroup, so that the estimator's branch of the frame hands over a cloud whose table cursors are assigned
__global__ void __launch_bounds__(VE_NT) k_ve_clusters(MapDims d, DevState s, VelEst ve, FilterParams fp, int with_rank, int* xq, int xq_seq) {
    ve_clusters_block(s, ve, fp);
    if (with_rank) {
        // (the cloud was written by THIS workgroup: its stores have to be complete, not written back across the chip --
        // an agent-scope fence flushes the XCD's L2 on this part and costs microseconds on the frame's longer branch)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
        birth_rank_block(d, s, fp, xq != nullptr);
    }
    if (xq) {
        // a queue of its own (DSPMAP_P_ESTIMATOR_QUEUE): the frame's first birth kernel, on the other queue, waits for this word.  Everything
        // this workgroup wrote (the birth cloud, the clusters, the table cursors) has to be in memory first: every wave waits for its own
        // stores, the barrier collects the waves, the fence writes this XCD's L2 back -- once per frame, on the branch with the slack
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            // (the write-back must have been acknowledged before the word goes out: the compiler may drop the wait behind buffer_wbl2 when it
            // can prove this wave's own counter empty -- MI355X guide, "compiler hazard" -- so it is spelled out)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            xq_publish(xq + 1, xq_seq);
        }
    }
}

// k_ve_view (DSPMAP_P_ESTIMATOR_QUEUE): the estimator's kernels run on a queue of their own and may start before the frame's first kernel
// has -- the estimator makes its own picture of the frame.  Every workgroup first waits (one lane) until the PREVIOUS frame's birth stage
// has ended (xq[0]; `want` = 0: the host has ordered this launch behind the handle's stream with an event instead): from then on the
// rand() cursor, the birth cloud and the rank's arrays belong to this frame's estimator.  Then the frame's parameter block is taken from
// its slot of the pinned ring (workgroup 0 keeps a copy in HBM for the two kernels that follow: `s` of those has fpar = that copy and
// pt_rot / pt_pyr = the arrays written here), the field of view's boundary planes are rotated (:226-232) and the points rotated and
// binned (:244-263) exactly as k_obs_points does it -- the same device functions, the same bits.
__global__ void __launch_bounds__(256) k_ve_view(MapDims d, DevState s, VelEst ve, const FrameParams* __restrict__ slot, int* xq, int* gave_up, int want) {
    __shared__ float s_ph[DSP_MAX_PLANES_H * 3];
    __shared__ float s_pv[DSP_MAX_PLANES_V * 3];
    if (threadIdx.x == 0) xq_wait(xq, want, gave_up);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < sizeof(FrameParams) / 4)
        reinterpret_cast<int*>(ve.v_fpar)[threadIdx.x] = reinterpret_cast<const int*>(slot)[threadIdx.x];
    const int n_pts = min(slot->n_pts, VE_CAP);
    const float* __restrict__ pts = slot->pts;
    const float q[4] = {slot->quat[0], slot->quat[1], slot->quat[2], slot->quat[3]};
    const int nh = d.np_h + 1, nv = d.np_v + 1;
    for (int i = threadIdx.x; i < nh + nv; i += blockDim.x) {
        float o[3];
        if (i < nh) {
            rotate_by_quat(s.planes_h0[3 * i], s.planes_h0[3 * i + 1], s.planes_h0[3 * i + 2], q, o);
            s_ph[3 * i] = o[0]; s_ph[3 * i + 1] = o[1]; s_ph[3 * i + 2] = o[2];
        } else {
            const int j = i - nh;
            rotate_by_quat(s.planes_v0[3 * j], s.planes_v0[3 * j + 1], s.planes_v0[3 * j + 2], q, o);
            s_pv[3 * j] = o[0]; s_pv[3 * j + 1] = o[1]; s_pv[3 * j + 2] = o[2];
        }
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pts) {
        float r[3];
        rotate_by_quat(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], q, r);
        const int pyr = pyramid_of(d, s_ph, s_pv, r[0], r[1], r[2]);
        const float len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        ve.v_rot[i] = make_float4(r[0], r[1], r[2], len);
        ve.v_pyr[i] = pyr;
    }
}

void launch_velocity_estimator(const LaunchCtx& c, bool with_rank) {
    hipLaunchKernelGGL(k_ve_components, dim3(VE_NB), dim3(VE_NT), 0, c.stream, c.s, c.ve);
    hipLaunchKernelGGL(k_ve_clusters, dim3(1), dim3(VE_NT), 0, c.stream, c.d, c.s, c.ve, c.fp, with_rank ? 1 : 0, (int*)nullptr, 0);
}
void launch_velocity_estimator_xq(const LaunchCtx& c, bool with_rank, const FrameParams* slot, int* xq, int* gave_up, int want, int xq_seq) {
    hipLaunchKernelGGL(k_ve_view, dim3((VE_CAP + 255) / 256), dim3(256), 0, c.stream, c.d, c.s, c.ve, slot, xq, gave_up, want);
    DevState s2 = c.s;   // the estimator's own picture of the frame
    s2.fpar = c.ve.v_fpar; s2.pt_rot = c.ve.v_rot; s2.pt_pyr = c.ve.v_pyr;
    hipLaunchKernelGGL(k_ve_components, dim3(VE_NB), dim3(VE_NT), 0, c.stream, s2, c.ve);
    hipLaunchKernelGGL(k_ve_clusters, dim3(1), dim3(VE_NT), 0, c.stream, c.d, s2, c.ve, c.fp, with_rank ? 1 : 0, xq, xq_seq);
}
int velocity_estimator_capacity() { return VE_CAP; }
int velocity_estimator_slices() { return VE_NB; }

#ifdef VE_DEBUG
extern "C" int dspmap_debug_ve(const VelEst* ve, long long* out) {
    return (int)hipMemcpy(out, &ve->cl[VE_CAP / 5 + 2], 16 * sizeof(long long), hipMemcpyDeviceToHost);
}
#endif
