// dspmap_birth.h -- the parts of the birth stage (mapAddNewBornParticlesByObservation :796-921) that depend on nothing
// but the frame's birth cloud, as device functions: the whole-frame path runs them as extra workgroups of k_predict
// (rank) and k_place (children) so that they leave the frame's critical path; the stage API and the split-phase
// multi-GPU path launch them as kernels of their own (dspmap_kernels.hip).
#pragma once
#include "dspmap_device.h"

// BK independent exclusive prefix sums over the workgroup at once (two barriers in total).
// Element order: v[0] of all threads, then v[1] of all threads, ... -- i.e. the order of the
// coalesced index i = j * blockDim + tid.  Returns the grand total; v[j] becomes the exclusive prefix.
// s_tmp: BK * 16 + 1 ints.
template <int NB>
__device__ __forceinline__ int block_excl_scan_multi(int (&v)[NB], int* s_tmp) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nw = blockDim.x >> 6;
    int inc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        inc[j] = wave_incl_scan_i(v[j]);
        if (l == 63) s_tmp[j * 16 + w] = inc[j];
    }
    __syncthreads();
    if (w == 0) {
        int run = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int t = l < nw ? s_tmp[j * 16 + l] : 0;
            const int ti = wave_incl_scan_i(t);
            if (l < nw) s_tmp[j * 16 + l] = run + ti - t;
            run += __builtin_amdgcn_readlane(ti, 63);
        }
        if (l == 0) s_tmp[NB * 16] = run;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] = inc[j] - v[j] + s_tmp[j * 16 + w];
    const int total = s_tmp[NB * 16];
    __syncthreads();
    return total;
}


#define BK 8
// The frame's birth sources (input_cloud_with_velocity :134).
//  * caller-supplied / estimator cloud: fpar->birth[0 .. fpar->n_birth)
//  * synthesised cloud (fpar->static_birth): what velocityEstimationThread emits when it tags nothing dynamic, and what
//    dsp_static.h:1285-1308 always emits -- every point of the frame's view, world position = rotated + current
//    position, zero velocity tag, intensity 0.  It is not stored: the readers rebuild entry i from k_obs_points'
//    output (pt_rot, pt_pyr; intensity -2 marks a point outside the field of view = not a source).  When the view is
//    EMPTY the reference returns before clearing its previous output (:1379-1381, dsp_static.h:1288-1290) and the
//    birth stage re-uses the last non-empty view's cloud: that one is kept in DevState::birth (written by
//    k_birth_insert of every frame with a non-empty view, length in FrameScalars::stale_n).
struct BirthView {
    const BirthSrc* stored;
    const float4* rot;
    const int* pyr;
    float cx, cy, cz;
    int n;
    bool live;      // rebuild from the frame's view
};
__device__ __forceinline__ BirthView birth_view(const DevState& s) {
    BirthView v;
    const FrameParams* fp = s.fpar;
    const int mode = fp->static_birth;
    v.live = mode == 1 && s.fs->view_epoch == fp->epoch;
    v.stored = fp->birth; v.rot = s.pt_rot; v.pyr = s.pt_pyr;
    v.cx = fp->cur_pos[0]; v.cy = fp->cur_pos[1]; v.cz = fp->cur_pos[2];
    // mode 2: the device velocity estimator wrote the cloud (and its length) -- or left the previous one (empty view)
    v.n = mode == 1 ? (v.live ? fp->n_pts : s.fs->stale_n) : (mode == 2 ? s.fs->est_n : fp->n_birth);
    return v;
}
__device__ __forceinline__ BirthSrc birth_at(const BirthView& v, int i) {
    if (!v.live) return v.stored[i];
    const float4 r = v.rot[i];
    BirthSrc b;
    b.x = r.x + v.cx; b.y = r.y + v.cy; b.z = r.z + v.cz;   // :1389-1391
    b.nx = b.ny = b.nz = 0.f;
    b.intensity = v.pyr[i] >= 0 ? 0.f : -2.f;
    return b;
}
// is source point i a birth source, and in which voxel (:818-820, :827 / :847).  dsp_static.h:797-825 has no voxel
// lookup for the source: a point outside the map still draws and may place children inside it (gv = 0 then: unused).
// cp: the frame's sensor position where FrameScalars::cur_pos may still be the previous frame's (the estimator on a queue of its own runs its
// rank before the frame's first kernel has copied it there); nullptr = FrameScalars::cur_pos
__device__ __forceinline__ bool birth_src_voxel(const MapDims& d, const DevState& s, const BirthSrc& src, float& cx, float& cy, float& cz, int& gv,
                                                const float* cp = nullptr) {
    if (!cp) cp = s.fs->cur_pos;
    cx = src.x - cp[0];  // :818-820
    cy = src.y - cp[1];
    cz = src.z - cp[2];
    if (!(src.intensity > -1.5f)) return false;
    if (voxel_of(d, cx, cy, cz, gv)) return true;  // :827 / :847
    gv = 0;
    return d.static_model != 0;
}

// k_birth_rank's workgroup (any blockDim that is a multiple of 64): rank of every valid source point among the valid
// ones -> first position-table cursor of the point (3 draws per child, always consumed, :871-873).  Validity comes
// straight from the source point, so the rank needs nothing but the frame's birth cloud: in a whole frame it rides on
// k_predict's launch.  Also clears the points' "child inside the map" words for k_birth_children.
__device__ __forceinline__ void birth_rank_block(const MapDims& d, const DevState& s, const FilterParams& fp, bool fpar_pos = false) {
    const BirthView bv = birth_view(s);
    const float* cp = fpar_pos ? s.fpar->cur_pos : nullptr;
    const int n_birth = bv.n;
    __shared__ int s_tmp[BK * 16 + 1];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int p_cur = s.fs->p_cur;
    const int nb = fp.nb_num;
    int run = 0;
    for (int base = 0; base < n_birth; base += nt * BK) {
        int v[BK];
        bool ok[BK];
#pragma unroll
        for (int j = 0; j < BK; ++j) {   // coalesced index, all loads in flight together
            const int i = base + j * nt + tid;
            float cx, cy, cz; int gv;
            ok[j] = i < n_birth && birth_src_voxel(d, s, birth_at(bv, i), cx, cy, cz, gv, cp);
            v[j] = ok[j] ? 1 : 0;
            if (i < n_birth) s.plan_inside[i] = 0u;
        }
        const int tot = block_excl_scan_multi<BK>(v, s_tmp);
#pragma unroll
        for (int j = 0; j < BK; ++j)
            if (ok[j]) s.plan_pbase[base + j * nt + tid] = (int)(((long long)p_cur + 3ll * nb * (long long)(run + v[j])) % fp.tab_n);
        run += tot;
    }
    if (tid == 0) { s.fs->p_cur = (int)(((long long)p_cur + 3ll * nb * run) % fp.tab_n); s.fs->n_birth_ovf = 0; }
}

// One thread per (source point, child): the child's position (:871-873) and destination voxel, "inside the map" (:875),
// and its birth index (point * n_nb + child) in the per-voxel bucket that k_birth_insert ranks.  Needs the birth cloud
// and the rank only: in a whole frame it rides on k_place's launch.
#define BIRTH_BUCKET_CAP 128
// inside_out: the caller collects the "inside the map" bits itself (a wave that generates all children of one point: a ballot)
__device__ __forceinline__ void birth_child_thread(const MapDims& d, const DevState& s, const FilterParams& fp, float4* __restrict__ child,
                                                   int* __restrict__ vb_cnt, int* __restrict__ vb_idx, const int t, bool* inside_out = nullptr) {
    const BirthView bv = birth_view(s);
    const int n_birth = bv.n;
    const int nb = fp.nb_num;
    const int i = t / nb, k = t - i * nb;
    if (i >= n_birth) return;
    float cx, cy, cz; int gsrc;
    if (!birth_src_voxel(d, s, birth_at(bv, i), cx, cy, cz, gsrc)) return;
    const int c = (int)(((long long)s.plan_pbase[i] + 3 * k) % fp.tab_n);
    const float x = cx + s.p_tab[c];                         // :871-873
    const float y = cy + s.p_tab[(c + 1) % fp.tab_n];
    const float z = cz + s.p_tab[(c + 2) % fp.tab_n];
    int gv = 0;
    int lv = -1;
    if (voxel_of_lv(d, x, y, z, gv, lv)) {                   // :875
        if (inside_out) *inside_out = true; else atomicOr(&s.plan_inside[i], 1u << k);
        if (lv >= 0) {                                       // children landing in another slab are inserted by their owner
            const int pos = atomicAdd(&vb_cnt[lv], 1);
            if (pos < BIRTH_BUCKET_CAP) vb_idx[(size_t)lv * BIRTH_BUCKET_CAP + pos] = t;
            else s.birth_ovf[atomicAdd(&s.fs->n_birth_ovf, 1)] = t;   // bucket full: WHICH children land in it depends on the arrival
                                                                      // order, so the others are kept too (k_birth_insert ranks over both)
        } else {
            lv = -1;
        }
    }
    child[t] = make_float4(x, y, z, __int_as_float(lv));
}
