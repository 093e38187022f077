// dspmap_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the
// per-frame loop of the particle-based dynamic occupancy map.
//
// Reference behaviour being reproduced: include/dsp_dynamic.h of g-ch/DSP-map
//   update() preamble      :220-293   -> k_reset, k_obs_points, k_obs_gather
//   mapPrediction          :627-701   -> k_predict, k_claim (moveParticle :1206-1274)
//   mapUpdate              :704-793   -> k_ck_partial, k_ck_finalize, k_weight
//   mapAddNewBorn...       :796-921   -> k_birth_split, k_birth_plan, k_birth_insert
//   mapOccupancy...Resample:924-1057  -> k_resample
//   getOccupancyMap*       :385-438   -> k_occ_count, k_occ_scan, k_occ_emit, k_clear_future
// No MFMA: there is no dense contraction on this path; the kernels are
// HBM-streaming (predict / claim / resample) or LDS+VALU pair loops (update).
#include <hip/hip_runtime.h>
#include "dspmap_device.h"
#include "dspmap_kernels.h"

#define RESET_PLANES 1
#define RESET_OBS 2
#define RESET_PRED 4

// --------------------------------------------------------------------------
// k_reset: per-frame housekeeping.
//  RESET_PLANES: rotate the 29+17 boundary-plane normals by the sensor attitude (:226-232)
//  RESET_OBS   : zero per-pyramid observation counters, max range = -1 (:235-238), Ck = 0
//  RESET_PRED  : zero the per-pyramid particle counters (pyramids are rebuilt by prediction, :638-642)
// --------------------------------------------------------------------------
__global__ void k_reset(MapDims d, DevState s, int flags, float qw, float qx, float qy, float qz,
                        float cx, float cy, float cz) {
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int gn = gridDim.x * blockDim.x;
    if (flags & RESET_PLANES) {
        const float q[4] = {qw, qx, qy, qz};
        const int nh = d.np_h + 1, nv = d.np_v + 1;
        for (int i = gt; i < nh + nv; i += gn) {
            float o[3];
            if (i < nh) {
                rotate_by_quat(s.planes_h0[3 * i], s.planes_h0[3 * i + 1], s.planes_h0[3 * i + 2], q, o);
                s.planes_h[3 * i] = o[0]; s.planes_h[3 * i + 1] = o[1]; s.planes_h[3 * i + 2] = o[2];
            } else {
                const int j = i - nh;
                rotate_by_quat(s.planes_v0[3 * j], s.planes_v0[3 * j + 1], s.planes_v0[3 * j + 2], q, o);
                s.planes_v[3 * j] = o[0]; s.planes_v[3 * j + 1] = o[1]; s.planes_v[3 * j + 2] = o[2];
            }
        }
        if (gt == 0) { s.fs->cur_pos[0] = cx; s.fs->cur_pos[1] = cy; s.fs->cur_pos[2] = cz; }
    }
    if (flags & RESET_OBS) {
        for (int i = gt; i < d.np; i += gn) { s.obs_cnt[i] = 0; s.obs_maxlen[i] = -1.f; }
        for (int i = gt; i < d.np * DSP_OBS_CAP; i += gn) s.obs_ck[i] = 0.f;
        if (gt == 0) { s.fs->n_valid = 0; s.fs->n_obs = 0; s.fs->has_expected_override = 0; }
    }
    if (flags & RESET_PRED) {
        for (int i = gt; i < d.np; i += gn) s.pyr_cnt[i] = 0;
        if (gt == 0) {
            s.fs->n_born = 0; s.fs->n_born_dropped = 0; s.fs->n_exp_up = 0; s.fs->n_exp_down = 0;
            s.fs->mover_count = 0;
        }
    }
}

// --------------------------------------------------------------------------
// Observation binning, update() :244-290.
// k_obs_points: one thread per input point: rotate into the world-aligned
// sensor-centred frame (:247), FOV test (:250), pyramid cell (:260-263), range (:266).
// --------------------------------------------------------------------------
__global__ void k_obs_points(MapDims d, DevState s, int n_pts, const float* __restrict__ pts,
                             float qw, float qx, float qy, float qz, int make_static_birth) {
    __shared__ float s_ph[DSP_MAX_PLANES_H * 3];
    __shared__ float s_pv[DSP_MAX_PLANES_V * 3];
    for (int i = threadIdx.x; i < (d.np_h + 1) * 3; i += blockDim.x) s_ph[i] = s.planes_h[i];
    for (int i = threadIdx.x; i < (d.np_v + 1) * 3; i += blockDim.x) s_pv[i] = s.planes_v[i];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = false;
    if (i < n_pts) {
        const float q[4] = {qw, qx, qy, qz};
        float r[3];
        rotate_by_quat(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], q, r);
        const int pyr = pyramid_of(d, s_ph, s_pv, r[0], r[1], r[2]);
        const float len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        s.pt_rot[i] = make_float4(r[0], r[1], r[2], len);
        s.pt_pyr[i] = pyr;
        valid = pyr >= 0;
        if (make_static_birth) {
            // what velocityEstimationThread emits for a static point (:1389-1391,1529-1540):
            // world position = rotated + current_position, zero velocity tag, intensity 0
            BirthSrc b;
            b.x = r[0] + s.fs->cur_pos[0]; b.y = r[1] + s.fs->cur_pos[1]; b.z = r[2] + s.fs->cur_pos[2];
            b.nx = b.ny = b.nz = 0.f;
            b.intensity = valid ? 0.f : -2.f;  // -2 = not a source (point outside the FOV)
            s.birth[i] = b;
        }
    }
    wave_count_add(&s.fs->n_valid, valid);  // valid_points :286
}

// k_obs_gather: one wave per pyramid.  Appends matching points in INPUT order
// (stable, ballot + prefix popcount) to the pyramid's bin, keeps the first 99
// (count saturates, :279-284), tracks the max range over ALL matches (:275-277).
__global__ void k_obs_gather(MapDims d, DevState s, int n_pts) {
    const int b = blockIdx.x;
    const int l = lane_id();
    int count = 0;
    float maxlen = -1.f;
    for (int base = 0; base < n_pts; base += WAVE) {
        const int i = base + l;
        const bool match = i < n_pts && s.pt_pyr[i] == b;
        const u64 m = __ballot(match);
        if (match) {
            const int pos = count + (int)__popcll(m & lanemask_lt());
            const float4 p = s.pt_rot[i];
            if (pos < DSP_OBS_CAP - 1) s.obs[b * DSP_OBS_CAP + pos] = p;
            maxlen = fmaxf(maxlen, p.w);
        }
        count += (int)__popcll(m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxlen = fmaxf(maxlen, __shfl_xor(maxlen, o, WAVE));
    if (l == 0) {
        const int c = min(count, DSP_OBS_CAP - 1);
        s.obs_cnt[b] = c;
        s.obs_maxlen[b] = maxlen;
        if (c) atomicAdd(&s.fs->n_obs, c);
    }
}

// --------------------------------------------------------------------------
// k_predict: mapPrediction :645-694, one lane per particle slot.
//   constant-velocity advance + ego-motion shift (:665-667), vz := 0 (:661-663),
//   out-of-map removal (:688), same-voxel particles are registered in their
//   pyramid (:1233-1259); particles whose voxel changed are only MARKED
//   (mvmask) -- k_claim moves them, which also guarantees every particle is
//   advanced exactly once (the role of flag 7, :649,1219).
// Occupancy words of the block's voxels are staged in LDS, edited with LDS
// atomics and written back once.
// part[blockIdx*4 + {0,1,2,3}] = {live in, left the map, pyramid full, moved}
// --------------------------------------------------------------------------
#define ACT_STAY 0
#define ACT_OUT 1
#define ACT_MOVE 2
#define ACT_PYRFULL 3
#define ACT_EXP_UP 4
#define ACT_EXP_DOWN 5

__global__ void k_predict(MapDims d, DevState s, FilterParams fp, float odx, float ody, float odz, float dt,
                          int vpw, int has_vz, int* __restrict__ part, u64* __restrict__ mvmask,
                          u64* __restrict__ expmask) {
    __shared__ u64 s_mask[512];
    __shared__ u64 s_mv[512];
    __shared__ u64 s_ex[512];
    __shared__ float s_ph[DSP_MAX_PLANES_H * 3];
    __shared__ float s_pv[DSP_MAX_PLANES_V * 3];
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x;
    for (int i = tid; i < (d.np_h + 1) * 3; i += blockDim.x) s_ph[i] = s.planes_h[i];
    for (int i = tid; i < (d.np_v + 1) * 3; i += blockDim.x) s_pv[i] = s.planes_v[i];
    if (tid < 4) s_cnt[tid] = 0;
    const int lv0 = blockIdx.x * vpw;
    const int nwords = vpw * d.mw;
    for (int i = tid; i < nwords; i += blockDim.x) {
        const int lv = lv0 + i / d.mw;
        u64 m = 0;
        if (lv < d.v_loc) m = s.mask[(size_t)lv0 * d.mw + i] & ~s.nbmask[(size_t)lv0 * d.mw + i];
        s_mask[i] = m;  // particles born/seeded this frame (flag 15) are not predicted (:649)
        s_mv[i] = 0;
        s_ex[i] = 0;
    }
    __syncthreads();
    const int vl = tid / d.slots;
    const int sl = tid - vl * d.slots;
    const int lv = lv0 + vl;
    const bool inrange = vl < vpw && lv < d.v_loc;
    const int wi = vl * d.mw + (sl >> 6);
    const u64 bit = 1ull << (sl & 63);
    const bool live = inrange && (s_mask[wi] & bit);
    const size_t idx = (size_t)lv * d.slots + sl;
    float px = 0.f, py = 0.f, pz = 0.f;
    int action = ACT_STAY;
    if (live) {
        float vx = s.vx[idx], vy = s.vy[idx];
        px = s.px[idx]; py = s.py[idx]; pz = s.pz[idx];
        if (has_vz) {
            // velocity process noise only when |vx*vy*vz| >= 1e-6 (:653-659): reachable only for
            // constructor-seeded particles on their first step
            const float vz = s.vz0[idx];
            if (!(fabs((double)(vx * vy * vz)) < 1e-6)) {
                const int c = (int)(((long long)s.fs->v_cur + 3ll * (long long)((size_t)(lv + d.v_base) * d.slots + sl)) % fp.tab_n);
                vx += s.v_tab[c];
                vy += s.v_tab[(c + 1) % fp.tab_n];
                s.vx[idx] = vx; s.vy[idx] = vy;
            }
            s.vz0[idx] = 0.f;
        }
        px += dt * vx + odx;        // :665
        py += dt * vy + ody;        // :666
        pz += dt * 0.f + odz;       // :667 with vz forced to 0 (:662)
        int gv;
        if (!voxel_of(d, px, py, pz, gv)) {
            action = ACT_OUT;
        } else {
            const int nlv = gv - d.v_base;
            if (nlv == lv) action = ACT_STAY;
            else if (nlv < 0) action = ACT_EXP_DOWN;
            else if (nlv >= d.v_loc) action = ACT_EXP_UP;
            else action = ACT_MOVE;
        }
        if (action != ACT_OUT) { s.px[idx] = px; s.py[idx] = py; s.pz[idx] = pz; }
    }
    // pyramid registration of particles that stay in their voxel
    int pyr = -1;
    if (live && action == ACT_STAY) pyr = pyramid_of(d, s_ph, s_pv, px, py, pz);
    const int pos = wave_agg_inc(s.pyr_cnt, pyr, pyr >= 0);
    if (pyr >= 0) {
        if (pos < d.capp) {
            const size_t o = (size_t)pyr * d.capp + pos;
            s.fov_rec[o] = make_float4(px, py, pz, s.w[idx]);
            s.fov_slot[o] = (int)idx;
        } else {
            action = ACT_PYRFULL;  // pyramid list full: the particle vanishes (-2, :1256-1259)
        }
    }
    if (live) {
        if (action == ACT_OUT || action == ACT_PYRFULL) atomicAnd(&s_mask[wi], ~bit);
        else if (action == ACT_MOVE) atomicOr(&s_mv[wi], bit);
        else if (action == ACT_EXP_UP || action == ACT_EXP_DOWN) atomicOr(&s_ex[wi], bit);
    }
    // per-block statistics (reduced lazily by the host; no global atomics here)
    {
        const u64 b0 = __ballot(live), b1 = __ballot(live && action == ACT_OUT);
        const u64 b2 = __ballot(live && action == ACT_PYRFULL), b3 = __ballot(live && action == ACT_MOVE);
        if (lane_id() == 0) {
            if (b0) atomicAdd(&s_cnt[0], (int)__popcll(b0));
            if (b1) atomicAdd(&s_cnt[1], (int)__popcll(b1));
            if (b2) atomicAdd(&s_cnt[2], (int)__popcll(b2));
            if (b3) atomicAdd(&s_cnt[3], (int)__popcll(b3));
        }
    }
    __syncthreads();
    for (int i = tid; i < nwords; i += blockDim.x) {
        const int lvw = lv0 + i / d.mw;
        if (lvw < d.v_loc) {
            const size_t g = (size_t)lv0 * d.mw + i;
            // movers / exports keep their live bit until k_claim / the export pass has copied them out
            const u64 keep = s_mask[i] | (s.mask[g] & s.nbmask[g]);
            s.mask[g] = keep;
            mvmask[g] = s_mv[i];
            if (expmask) expmask[g] = s_ex[i];
        }
    }
    if (tid < 4) part[blockIdx.x * 4 + tid] = s_cnt[tid];
}

// --------------------------------------------------------------------------
// k_claim: the voxel-changing half of moveParticle (:1209-1230) for the
// particles k_predict marked.  One lane per slot; a mover claims the lowest
// free slot of its destination voxel with one atomic OR (first-free-slot rule
// :1214-1215), copies its record, registers in its pyramid (:1233-1259) and
// only then releases its source slot.  Destination full -> the particle
// vanishes (-1, :1227-1229).
// part2[blockIdx*2 + {0,1}] = {voxel full, pyramid full}
// --------------------------------------------------------------------------
__global__ void k_claim(MapDims d, DevState s, int vpw, u64* __restrict__ mvmask, int* __restrict__ part2) {
    __shared__ u64 s_mv[512];
    __shared__ float s_ph[DSP_MAX_PLANES_H * 3];
    __shared__ float s_pv[DSP_MAX_PLANES_V * 3];
    __shared__ int s_cnt[2];
    __shared__ int s_any;
    const int tid = threadIdx.x;
    const int lv0 = blockIdx.x * vpw;
    const int nwords = vpw * d.mw;
    if (tid == 0) s_any = 0;
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nwords; i += blockDim.x) {
        const int lvw = lv0 + i / d.mw;
        const u64 m = lvw < d.v_loc ? mvmask[(size_t)lv0 * d.mw + i] : 0ull;
        s_mv[i] = m;
        if (m) s_any = 1;
    }
    __syncthreads();
    if (!s_any) {
        if (tid < 2) part2[blockIdx.x * 2 + tid] = 0;
        return;
    }
    for (int i = tid; i < (d.np_h + 1) * 3; i += blockDim.x) s_ph[i] = s.planes_h[i];
    for (int i = tid; i < (d.np_v + 1) * 3; i += blockDim.x) s_pv[i] = s.planes_v[i];
    __syncthreads();
    const int vl = tid / d.slots;
    const int sl = tid - vl * d.slots;
    const int lv = lv0 + vl;
    const bool inrange = vl < vpw && lv < d.v_loc;
    const int wi = vl * d.mw + (sl >> 6);
    const u64 bit = 1ull << (sl & 63);
    const bool mover = inrange && (s_mv[wi] & bit);
    const size_t idx = (size_t)lv * d.slots + sl;
    int pyr = -1;
    size_t nidx = 0;
    int nlv = -1, nsl = -1;
    float px = 0, py = 0, pz = 0, w = 0;
    bool vfull = false;
    if (mover) {
        px = s.px[idx]; py = s.py[idx]; pz = s.pz[idx];
        const float vx = s.vx[idx], vy = s.vy[idx];
        w = s.w[idx];
        int gv = 0;
        voxel_of(d, px, py, pz, gv);  // in-map and in-slab by construction (k_predict)
        nlv = gv - d.v_base;
        nsl = claim_slot(s.mask, nlv, d);
        if (nsl >= 0) {
            nidx = (size_t)nlv * d.slots + nsl;
            s.px[nidx] = px; s.py[nidx] = py; s.pz[nidx] = pz;
            s.vx[nidx] = vx; s.vy[nidx] = vy; s.w[nidx] = w;
            pyr = pyramid_of(d, s_ph, s_pv, px, py, pz);
        } else {
            vfull = true;
        }
        // the record is in registers now: release the source slot.  The wait makes sure the loads
        // above have returned before another mover can see the slot free and overwrite it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAnd(&s.mask[(size_t)lv * d.mw + (sl >> 6)], ~bit);
    }
    const int pos = wave_agg_inc(s.pyr_cnt, pyr, pyr >= 0);
    bool pfull = false;
    if (pyr >= 0) {
        if (pos < d.capp) {
            const size_t o = (size_t)pyr * d.capp + pos;
            s.fov_rec[o] = make_float4(px, py, pz, w);
            s.fov_slot[o] = (int)nidx;
        } else {
            pfull = true;  // :1256-1259
            atomicAnd(&s.mask[(size_t)nlv * d.mw + (nsl >> 6)], ~(1ull << (nsl & 63)));
        }
    }
    {
        const u64 b0 = __ballot(vfull), b1 = __ballot(pfull);
        if (lane_id() == 0) {
            if (b0) atomicAdd(&s_cnt[0], (int)__popcll(b0));
            if (b1) atomicAdd(&s_cnt[1], (int)__popcll(b1));
        }
    }
    __syncthreads();
    for (int i = tid; i < nwords; i += blockDim.x) {
        const int lvw = lv0 + i / d.mw;
        if (lvw < d.v_loc && s_mv[i]) mvmask[(size_t)lv0 * d.mw + i] = 0ull;
    }
    if (tid < 2) part2[blockIdx.x * 2 + tid] = s_cnt[tid];
}

// --------------------------------------------------------------------------
// mapUpdate pass 1, :709-739:  Ck[k] = sum over particles i in the 3x3
// pyramid neighbourhood of obs k of P_d * w_i * g(x)g(y)g(z).
// Work item = (pyramid b, chunk of its particles).  The chunk is staged in LDS
// (float4 {x,y,z,P_d*w}); LANES ARE OBSERVATIONS of the neighbourhood N(b), the
// particle is broadcast from LDS, so every lane accumulates its own Ck
// privately (no cross-lane reduction) and issues one float atomic at the end.
// (N(b) of a particle bin == the set of obs bins whose N() contains b: the
// 3x3 clipped neighbourhood relation is symmetric, :1128-1147.)
// blockIdx -> (b, chunk) is XCD-aware: all chunks of one pyramid land on the
// same XCD (blockIdx % 8) so its obs tile and Ck lines stay in one L2.
// --------------------------------------------------------------------------
#define CK_TPB 128
#define CK_PCH 128

__device__ __forceinline__ void decode_pyr_block(int bid, int nchunk, int np, int& b, int& chunk) {
    const int xcd = bid & 7;
    const int j = bid >> 3;
    b = (j / nchunk) * 8 + xcd;
    chunk = j % nchunk;
    (void)np;
}
__device__ __forceinline__ int neighbor_bins(const MapDims& d, int b, int* bins) {
    // findPyramidNeighborIndexInFOV :1128-1147 (h-major order, clipped at the FOV edge)
    const int h0 = b / d.np_v, v0 = b % d.np_v;
    int n = 0;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j) {
            const int h = h0 + i, v = v0 + j;
            if (h >= 0 && h < d.np_h && v >= 0 && v < d.np_v) bins[n++] = h * d.np_v + v;
        }
    return n;
}

__global__ void __launch_bounds__(CK_TPB) k_ck_partial(MapDims d, DevState s, FilterParams fp, int nchunk) {
    __shared__ float4 s_p[CK_PCH];
    __shared__ int s_bin[9];
    __shared__ int s_off[10];
    int b, chunk;
    decode_pyr_block(blockIdx.x, nchunk, d.np, b, chunk);
    if (b >= d.np) return;
    const int P = min(s.pyr_cnt[b], d.capp);
    const int start = chunk * CK_PCH;
    if (start >= P) return;
    const int npart = min(CK_PCH, P - start);
    const int tid = threadIdx.x;
    if (tid == 0) {
        int bins[9];
        const int n = neighbor_bins(d, b, bins);
        int off = 0;
        for (int k = 0; k < 9; ++k) {
            s_off[k] = off;
            if (k < n) { s_bin[k] = bins[k]; off += s.obs_cnt[bins[k]]; } else s_bin[k] = -1;
        }
        s_off[9] = off;
    }
    __syncthreads();
    const int O = s_off[9];
    if (O == 0) return;
    for (int i = tid; i < npart; i += CK_TPB) {
        float4 r = s.fov_rec[(size_t)b * d.capp + start + i];
        r.w = fp.p_det * r.w;  // P_detection * weight (pre-update weights), :732
        s_p[i] = r;
    }
    __syncthreads();
    for (int o = tid; o < O; o += CK_TPB) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < 9; ++q) k += (o >= s_off[q]) ? 1 : 0;
        const int oi = s_bin[k] * DSP_OBS_CAP + (o - s_off[k]);
        const float4 z = s.obs[oi];
        float acc = 0.f;
        for (int i = 0; i < npart; ++i) {
            const float4 p = s_p[i];
            acc += p.w * pair_gk(p.x, p.y, p.z, z.x, z.y, z.z, fp.sigma_ob, fp.inv_sigma_ob, fp.pdf_c3);
        }
        unsafeAtomicAdd(&s.obs_ck[oi], acc);
    }
}

// k_ck_finalize: Ck += expected_new_born_objects + kappa (:737) and the birth
// normaliser  w_nb * sum_k 1/Ck  (:799-805).  One workgroup, deterministic tree reduction.
__global__ void __launch_bounds__(1024) k_ck_finalize(MapDims d, DevState s, FilterParams fp) {
    __shared__ float s_red[1024];
    const int tid = threadIdx.x;
    float lambda;
    if (s.fs->has_expected_override) lambda = s.fs->expected_newborn;
    else lambda = fp.nb_weight * (float)s.fs->n_valid * (float)fp.nb_num;  // :292
    const float add = lambda + fp.kappa;
    float acc = 0.f;
    for (int i = tid; i < d.np * DSP_OBS_CAP; i += 1024) {
        const int b = i / DSP_OBS_CAP, j = i - b * DSP_OBS_CAP;
        if (j < s.obs_cnt[b]) {
            const float ck = s.obs_ck[i] + add;
            s.obs_ck[i] = ck;
            acc += __fdiv_rn(1.f, ck);
        }
    }
    s_red[tid] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        s.fs->expected_newborn = lambda;
        s.fs->newborn_w = fp.nb_weight * s_red[0];  // :805
    }
}

// --------------------------------------------------------------------------
// mapUpdate pass 2, :743-790: for every particle inside the FOV
//   skip if occluded: |p| > max_range[b] + 0.3 and the pyramid has observations (:759-765)
//   w *= (1-P_d) + sum over obs k of N(b) of P_d*g/Ck                         (:768-786)
// Lanes are particles; the neighbourhood's observations {x,y,z,P_d/Ck} are
// staged in LDS and broadcast.  The new weight is scattered back to the slot.
// --------------------------------------------------------------------------
#define WU_TPB 256

__global__ void __launch_bounds__(WU_TPB) k_weight(MapDims d, DevState s, FilterParams fp, int nchunk) {
    __shared__ float4 s_o[9 * DSP_OBS_CAP];
    __shared__ int s_bin[9];
    __shared__ int s_off[10];
    int b, chunk;
    decode_pyr_block(blockIdx.x, nchunk, d.np, b, chunk);
    if (b >= d.np) return;
    const int P = min(s.pyr_cnt[b], d.capp);
    const int start = chunk * WU_TPB;
    if (start >= P) return;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int bins[9];
        const int n = neighbor_bins(d, b, bins);
        int off = 0;
        for (int k = 0; k < 9; ++k) {
            s_off[k] = off;
            if (k < n) { s_bin[k] = bins[k]; off += s.obs_cnt[bins[k]]; } else s_bin[k] = -1;
        }
        s_off[9] = off;
    }
    __syncthreads();
    const int O = s_off[9];
    for (int o = tid; o < O; o += WU_TPB) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < 9; ++q) k += (o >= s_off[q]) ? 1 : 0;
        const int oi = s_bin[k] * DSP_OBS_CAP + (o - s_off[k]);
        float4 z = s.obs[oi];
        z.w = __fdiv_rn(fp.p_det, s.obs_ck[oi]);
        s_o[o] = z;
    }
    __syncthreads();
    const int i = start + tid;
    if (i >= P) return;
    const size_t ri = (size_t)b * d.capp + i;
    const float4 p = s.fov_rec[ri];
    const float maxlen = s.obs_maxlen[b];
    const float dist = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
    if (maxlen > 0.f && dist > maxlen + fp.occl_margin) return;  // occluded :761-765
    float sum = 0.f;
    for (int o = 0; o < O; ++o) {
        const float4 z = s_o[o];
        sum += pair_gk(p.x, p.y, p.z, z.x, z.y, z.z, fp.sigma_ob, fp.inv_sigma_ob, fp.pdf_c3) * z.w;
    }
    s.w[s.fov_slot[ri]] = p.w * ((1.f - fp.p_det) + sum);  // :786
}

// --------------------------------------------------------------------------
// Birth, mapAddNewBornParticlesByObservation :796-921.
// k_birth_split: one wave per source point.  Dempster-Shafer static/dynamic
// split from the mass already in the point's voxel (:827-866), lanes = slots.
// --------------------------------------------------------------------------
__global__ void k_birth_split(MapDims d, DevState s, FilterParams fp, int n_birth) {
    const int wpb = blockDim.x / WAVE;
    const int i = blockIdx.x * wpb + threadIdx.x / WAVE;
    if (i >= n_birth) return;
    const int l = lane_id();
    const BirthSrc src = s.birth[i];
    BirthPlan pl;
    pl.cx = src.x - s.fs->cur_pos[0];  // :818-820
    pl.cy = src.y - s.fs->cur_pos[1];
    pl.cz = src.z - s.fs->cur_pos[2];
    pl.gvox = -1; pl.n_static = 0; pl.inside = 0; pl.pbase = pl.vbase = pl.rbase = 0;
    int gv;
    const bool ok = src.intensity > -1.5f && voxel_of(d, pl.cx, pl.cy, pl.cz, gv);  // :827 / :847
    int n_static = 0;
    if (ok) {
        pl.gvox = gv;
        const int lv = gv - d.v_base;
        if (lv >= 0 && lv < d.v_loc) {
            float ws = 0.f, wsd = 0.f, wd = 0.f;
            for (int e = 0; e < d.mw; ++e) {
                const int sl = e * 64 + l;
                const u64 m = s.mask[(size_t)lv * d.mw + e] & ~s.nbmask[(size_t)lv * d.mw + e];  // 0.9<flag<14 :830
                if (sl < d.slots && ((m >> l) & 1ull)) {
                    const size_t idx = (size_t)lv * d.slots + sl;
                    const float vabs = fabsf(s.vx[idx]) + fabsf(s.vy[idx]) + 0.f;  // vz == 0
                    const float w = s.w[idx];
                    if (vabs < 0.1f) ws += w; else if (vabs < 0.5f) wsd += w; else wd += w;
                }
            }
            ws = wave_sum(ws); wsd = wave_sum(wsd); wd = wave_sum(wd);
            // Dempster-Shafer :850-866
            const float total = ws + wd + wsd;
            const float m_s = __fdiv_rn(ws, total), m_d = __fdiv_rn(wd, total), m_sd = __fdiv_rn(wsd, total);
            const float p_s = (m_s + m_s + m_sd) * 0.5f;
            const float p_d = (m_d + m_d + m_sd) * 0.5f;
            const float p_s_n = __fdiv_rn(p_s, p_s + p_d);
            const float f = (float)fp.model_nb * p_s_n;
            int ns = (f != f) ? 0 : (int)f;  // empty voxel -> NaN -> minimum applies (Appendix A-8)
            n_static = max(fp.min_static_nb, ns);
        }
        // else: the source voxel belongs to another slab; that rank supplies n_static (all-reduce max)
    }
    pl.n_static = n_static;
    if (l == 0) { s.plan[i] = pl; s.nstatic[i] = n_static; }
}

// The sequential consumption order of the three random streams (:871-873 position table,
// :884-886 velocity table, :895-897 rand()) is reproduced with block-wide prefix sums over the
// source points: draws of point i start at cursor + (draws of all earlier points).
__device__ __forceinline__ int block_excl_scan_1024(int v, int* s_tmp, int* total) {
    // s_tmp: 17 ints.  blockDim.x == 1024
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int inc = wave_incl_scan_i(v);
    if (l == 63) s_tmp[w] = inc;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k < 16; ++k) { const int t = s_tmp[k]; s_tmp[k] = run; run += t; }
        s_tmp[16] = run;
    }
    __syncthreads();
    const int r = inc - v + s_tmp[w];
    *total = s_tmp[16];
    __syncthreads();
    return r;
}

// k_birth_rank (one workgroup): rank of every valid source point among the valid ones ->
// first position-table cursor of the point (3 draws per child, always consumed, :871-873).
__global__ void __launch_bounds__(1024) k_birth_rank(MapDims d, DevState s, FilterParams fp, int n_birth) {
    __shared__ int s_tmp[17];
    __shared__ int s_run;
    const int tid = threadIdx.x;
    if (tid == 0) s_run = 0;
    __syncthreads();
    const int p_cur = s.fs->p_cur;
    const int nb = fp.nb_num;
    for (int base = 0; base < n_birth; base += 1024) {
        const int i = base + tid;
        const bool ok = i < n_birth && s.plan[i].gvox >= 0;
        int tot;
        const int r = block_excl_scan_1024(ok ? 1 : 0, s_tmp, &tot);
        if (ok) s.plan[i].pbase = (int)(((long long)p_cur + 3ll * nb * (long long)(s_run + r)) % fp.tab_n);
        __syncthreads();
        if (tid == 0) s_run += tot;
        __syncthreads();
    }
    if (tid == 0) s.fs->p_cur = (int)(((long long)p_cur + 3ll * nb * s_run) % fp.tab_n);
}

// Children are inserted in the reference's sequential order WITHOUT a sort:
// k_birth_children computes every child's position (:871-873) and destination voxel, marks it
// "inside the map" (:875) and records its birth index (point*n_nb+child) in the per-voxel bucket;
// k_birth_insert ranks each child among its voxel's children by birth index and takes the
// rank-th free slot of the PRE-birth occupancy word -- exactly the slot addAParticle's
// first-free scan (:1183-1201) would hand out when children arrive one after another; children
// whose rank exceeds the free slots are dropped, as in the reference (:1198-1200).
// New particles only set their bit in nbmask (live = mask | nbmask), so the pre-birth word
// stays stable while the kernel runs.
#define BIRTH_BUCKET_CAP 128

__global__ void k_birth_children(MapDims d, DevState s, FilterParams fp, int n_birth, float4* __restrict__ child,
                                 int* __restrict__ vb_cnt, int* __restrict__ vb_idx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = fp.nb_num;
    const int i = t / nb, k = t - i * nb;
    if (i >= n_birth) return;
    const BirthPlan pl = s.plan[i];
    if (pl.gvox < 0) return;
    const int c = (int)(((long long)pl.pbase + 3 * k) % fp.tab_n);
    const float x = pl.cx + s.p_tab[c];                      // :871-873
    const float y = pl.cy + s.p_tab[(c + 1) % fp.tab_n];
    const float z = pl.cz + s.p_tab[(c + 2) % fp.tab_n];
    int gv = 0;
    int lv = -1;
    if (voxel_of(d, x, y, z, gv)) {                          // :875
        atomicOr(&s.plan[i].inside, 1u << k);
        lv = gv - d.v_base;
        if (lv >= 0 && lv < d.v_loc) {                       // children landing in another slab are inserted by their owner
            const int pos = atomicAdd(&vb_cnt[lv], 1);
            if (pos < BIRTH_BUCKET_CAP) vb_idx[(size_t)lv * BIRTH_BUCKET_CAP + pos] = t;
        } else {
            lv = -1;
        }
    }
    child[t] = make_float4(x, y, z, __int_as_float(lv));
}

// k_birth_cursors (one workgroup): velocity-table and rand() cursors per source point (:884-886,:895-897)
__global__ void __launch_bounds__(1024) k_birth_cursors(MapDims d, DevState s, FilterParams fp, int n_birth) {
    __shared__ int s_tmp[17];
    __shared__ int s_run[2];
    const int tid = threadIdx.x;
    if (tid < 2) s_run[tid] = 0;
    __syncthreads();
    const int v_cur = s.fs->v_cur, r_cur = s.fs->r_cur;
    const int nb = fp.nb_num;
    for (int base = 0; base < n_birth; base += 1024) {
        const int i = base + tid;
        int cv = 0, cr = 0, n_static = 0;
        bool ok = false;
        if (i < n_birth) {
            const BirthPlan pl = s.plan[i];
            ok = pl.gvox >= 0;
            if (ok) {
                n_static = s.nstatic[i];
                const BirthSrc src = s.birth[i];
                if (src.intensity > 0.01f) {
                    const int model_end = src.nx > -100.f ? fp.model_nb : n_static;  // :881
                    for (int k = n_static; k < nb; ++k) {
                        if (!((pl.inside >> k) & 1u)) continue;
                        if (k < model_end) cv += 3; else cr += 3;
                    }
                }
            }
        }
        int totv, totr;
        const int ev = block_excl_scan_1024(cv, s_tmp, &totv);
        const int er = block_excl_scan_1024(cr, s_tmp, &totr);
        if (ok) {
            s.plan[i].n_static = n_static;
            s.plan[i].vbase = (int)(((long long)v_cur + s_run[0] + ev) % fp.tab_n);
            s.plan[i].rbase = (int)(((long long)r_cur + s_run[1] + er) % max(fp.rtab_n, 1));
        }
        __syncthreads();
        if (tid == 0) { s_run[0] += totv; s_run[1] += totr; }
        __syncthreads();
    }
    if (tid == 0) {
        s.fs->v_cur = (int)(((long long)v_cur + s_run[0]) % fp.tab_n);
        s.fs->r_cur = (int)(((long long)r_cur + s_run[1]) % max(fp.rtab_n, 1));
    }
}

// generateRandomFloat :1551-1553 fed from the rand() table
__device__ __forceinline__ float rand_float(const DevState& s, const FilterParams& fp, int c, float lo, float hi) {
    const int r = s.r_tab[c % max(fp.rtab_n, 1)];
    return lo + __fdiv_rn((float)r, __fdiv_rn((float)2147483647, (hi - lo)));
}

// k_birth_insert: one thread per (source point, child): velocity by branch (:877-903); vz = 0
// (:905-907); weight = the global newborn weight (:909); newborn flag (= nbmask bit).
__global__ void k_birth_insert(MapDims d, DevState s, FilterParams fp, int n_birth, const float4* __restrict__ child,
                               const int* __restrict__ vb_cnt, const int* __restrict__ vb_idx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = fp.nb_num;
    const int i = t / nb, k = t - i * nb;
    bool born = false, dropped = false;
    if (i < n_birth) {
        const BirthPlan pl = s.plan[i];
        if (pl.gvox >= 0 && ((pl.inside >> k) & 1u)) {
            const float4 ch = child[t];
            const int lv = __float_as_int(ch.w);
            if (lv >= 0) {
                const BirthSrc src = s.birth[i];
                float vx = 0.f, vy = 0.f;
                if (k >= pl.n_static && src.intensity > 0.01f) {
                    const int model_end = src.nx > -100.f ? fp.model_nb : pl.n_static;
                    const unsigned lo_mask = (k >= 32 ? ~0u : ((1u << k) - 1u)) & ~((pl.n_static >= 32) ? ~0u : ((1u << pl.n_static) - 1u));
                    const unsigned before = pl.inside & lo_mask;  // inside children in [n_static, k)
                    if (k < model_end) {
                        const int rank = __popc(before);
                        const int cv = (int)(((long long)pl.vbase + 3 * rank) % fp.tab_n);
                        vx = src.nx + 4 * s.v_tab[cv];                        // :884
                        vy = src.ny + 4 * s.v_tab[(cv + 1) % fp.tab_n];      // :885
                    } else {
                        const unsigned model_bits = (model_end >= 32) ? ~0u : ((1u << model_end) - 1u);
                        const int rank = __popc(before & ~model_bits);
                        const int cr = pl.rbase + 3 * rank;
                        vx = rand_float(s, fp, cr, -1.5f, 1.5f);              // :895
                        vy = rand_float(s, fp, cr + 1, -1.5f, 1.5f);          // :896
                    }
                }
                // rank among this voxel's children, by birth index
                const int n = min(vb_cnt[lv], BIRTH_BUCKET_CAP);
                int rank = 0;
                bool recorded = false;
                const int* bl = vb_idx + (size_t)lv * BIRTH_BUCKET_CAP;
                for (int j = 0; j < n; ++j) {
                    const int o = bl[j];
                    rank += (o < t) ? 1 : 0;
                    recorded |= (o == t);
                }
                int sl = -1;
                if (recorded) {  // rank-th free slot of the pre-birth occupancy
                    int r = rank;
                    for (int e = 0; e < d.mw && sl < 0; ++e) {
                        const int nbits = min(64, d.slots - e * 64);
                        const u64 valid = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
                        u64 fr = ~s.mask[(size_t)lv * d.mw + e] & valid;
                        const int nf = (int)__popcll(fr);
                        if (r >= nf) { r -= nf; continue; }
                        for (int q = 0; q < r; ++q) fr &= fr - 1ull;
                        sl = e * 64 + (__ffsll((long long)fr) - 1);
                    }
                }
                if (sl >= 0) {
                    const size_t idx = (size_t)lv * d.slots + sl;
                    s.px[idx] = ch.x; s.py[idx] = ch.y; s.pz[idx] = ch.z;
                    s.vx[idx] = vx; s.vy[idx] = vy;
                    s.w[idx] = s.fs->newborn_w;
                    atomicOr(&s.nbmask[(size_t)lv * d.mw + (sl >> 6)], 1ull << (sl & 63));  // flag 15
                    born = true;
                } else {
                    dropped = true;
                }
            }
        }
    }
    wave_count_add(&s.fs->n_born, born);
    wave_count_add(&s.fs->n_born_dropped, dropped);
}

// --------------------------------------------------------------------------
// mapOccupancyCalculationAndResample :924-1057, two kernels:
//  k_resample_scan : one lane per voxel looks at the occupancy words; empty voxels get their
//                    (zero) result written right there, non-empty ones are appended to a work
//                    list (one atomic per 1024-voxel block) -> perfect load balance whatever the
//                    spatial clustering of the particles.
//  k_resample_work : persistent waves stride over the work list; ONE WAVE PER VOXEL, LANES =
//                    SLOTS.  All six field loads of a voxel are issued unconditionally up front
//                    and the next voxel's loads are in flight while the current one is processed.
//     cull w < 1e-3 (:941), mass = wavefront reduction (:970-974), mean velocity
//     (:944-948,976-984), constant-velocity future rollout scattered with float atomics
//     (:950-964; a voxel whose particles are all static adds its mass to itself for every
//     horizon without any index math), systematic resampling driven by a wavefront (DPP) prefix
//     scan of the weights (:1005-1053) incl. lowest-free-slot copies and the "no free slot ->
//     fold the weight back" rule (:1037-1041).
// --------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_resample_scan(MapDims d, DevState s, int* __restrict__ work_list,
                                                        int* __restrict__ work_count) {
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int lv = blockIdx.x * 1024 + threadIdx.x;
    bool nonempty = false;
    if (lv < d.v_loc) {
        for (int e = 0; e < d.mw; ++e)
            nonempty |= (s.mask[(size_t)lv * d.mw + e] | s.nbmask[(size_t)lv * d.mw + e]) != 0ull;
        if (!nonempty) s.res4[lv] = make_float4(0.f, 0.f, 0.f, 0.f);  // :974-984 for an empty voxel
    }
    const u64 b = __ballot(nonempty);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) s_w[w] = (int)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < 16; ++k) { const int t = s_w[k]; s_w[k] = tot; tot += t; }
        s_base = tot ? atomicAdd(work_count, tot) : 0;
    }
    __syncthreads();
    if (nonempty) work_list[s_base + s_w[w] + (int)__popcll(b & lanemask_lt())] = lv;
}

template <int EPL>
struct VoxRegs {
    u64 m[EPL], nb[EPL];
    float w[EPL], vx[EPL], vy[EPL], px[EPL], py[EPL], pz[EPL];
    int lv;
};

template <int EPL>
__device__ __forceinline__ void load_voxel(const MapDims& d, const DevState& s, int lv, VoxRegs<EPL>& r) {
    const int l = lane_id();
    r.lv = lv;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        r.nb[e] = s.nbmask[(size_t)lv * EPL + e];
        r.m[e] = s.mask[(size_t)lv * EPL + e] | r.nb[e];  // newborns live only in nbmask until now
        const int sl = e * 64 + l;
        const size_t idx = (size_t)lv * d.slots + (sl < d.slots ? sl : 0);
        r.w[e] = s.w[idx]; r.vx[e] = s.vx[idx]; r.vy[e] = s.vy[idx];
        r.px[e] = s.px[idx]; r.py[e] = s.py[idx]; r.pz[e] = s.pz[idx];
    }
}

template <int EPL>
__device__ __forceinline__ int resample_voxel(const MapDims& d, const DevState& s, const VoxRegs<EPL>& r) {
    const int l = lane_id();
    const int lv = r.lv;
    float w[EPL];
    bool alive[EPL];
    u64 alive_m[EPL], old_m[EPL];
    int n = 0, n_old = 0;
    float occ = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const bool live = (r.m[e] >> l) & 1ull;
        w[e] = live ? r.w[e] : 0.f;
        alive[e] = live && !(w[e] < 1e-3f);  // :941
        alive_m[e] = __ballot(alive[e]);
        old_m[e] = alive_m[e] & ~r.nb[e];    // flag < 10 :944
        n += (int)__popcll(alive_m[e]);
        n_old += (int)__popcll(old_m[e]);
        if (!alive[e]) w[e] = 0.f;
        occ += w[e];
    }
    occ = wave_sum(occ);  // :970,974
    float4 res = make_float4(occ, 0.f, 0.f, 0.f);
    if (n_old > 0) {
        float vxs = 0.f, vys = 0.f, wold = 0.f;
        u64 moving = 0ull;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const bool old = (old_m[e] >> l) & 1ull;
            const float vx = old ? r.vx[e] : 0.f, vy = old ? r.vy[e] : 0.f;
            vxs += vx; vys += vy;
            wold += old ? w[e] : 0.f;
            moving |= __ballot(old && (vx != 0.f || vy != 0.f));
        }
        vxs = wave_sum(vxs); vys = wave_sum(vys);
        res.y = __fdiv_rn(vxs, (float)n_old);
        res.z = __fdiv_rn(vys, (float)n_old);
        // future rollout :950-964
        if (!moving) {
            // every survivor is static: p + 0*t stays in this voxel for every horizon
            const float sown = wave_sum(wold);
            if (l < d.T && sown != 0.f) unsafeAtomicAdd(&s.fut[(size_t)lv * d.T + l], sown);
        } else {
            const int gz = (lv + d.v_base) / (d.ny * d.nx);  // z layer is unchanged (vz == 0)
            for (int t = 0; t < d.T; ++t) {
                const float pt = d.pred_t[t];
                float sown = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const bool old = (old_m[e] >> l) & 1ull;
                    const float fx = r.px[e] + r.vx[e] * pt;
                    const float fy = r.py[e] + r.vy[e] * pt;
                    const bool in = old && !(fx >= d.half_x || fx <= -d.half_x || fy >= d.half_y || fy <= -d.half_y);
                    const int xi = (int)__fdiv_rn(fx + d.half_x, d.res);
                    const int yi = (int)__fdiv_rn(fy + d.half_y, d.res);
                    const int dl = gz * d.ny * d.nx + yi * d.nx + xi - d.v_base;
                    const bool own = in && dl == lv;
                    sown += own ? w[e] : 0.f;
                    if (in && !own && dl >= 0 && dl < d.v_loc) unsafeAtomicAdd(&s.fut[(size_t)dl * d.T + t], w[e]);
                }
                sown = wave_sum(sown);
                if (l == 0 && sown != 0.f) unsafeAtomicAdd(&s.fut[(size_t)lv * d.T + t], sown);
            }
        }
    }
    if (l == 0) s.res4[lv] = res;  // voxels_objects_number[v][0..3] :974-984
    u64 newmask[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) newmask[e] = alive_m[e];
    if (n >= 5) {  // :986
        const int n_after = n > d.M ? d.M : n;               // :992-997
        const float w_after = __fdiv_rn(occ, (float)n_after);  // :1000
        // inclusive prefix sum of the surviving weights in slot order (acc_ori_weight :1011).
        // (fp32 prefix scan; the reference accumulates sequentially, so the two differ in the last
        // bits -- this only matters when a running sum sits exactly on a threshold, e.g. a voxel that
        // holds nothing but equal-weight newborns with n > M; see DESIGN.md "threshold ties")
        float A[EPL];
        float carry = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            A[e] = wave_incl_scan(w[e]) + carry;
            carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A[e]), 63));
        }
        // K(a) = number of thresholds tau_m < a, tau_0 = 0.5 w', tau_{m+1} = tau_m + w' (fp32, :1006,1015,1043)
        int K[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) K[e] = 0;
        {
            float tau = w_after * 0.5f;
            for (int q = 0; q <= n_after; ++q) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) K[e] += (A[e] > tau) ? 1 : 0;
                tau += w_after;
            }
        }
        int extra[EPL];
        u64 kept_m[EPL], removed_m[EPL], copy_m[EPL];
        int prev_last = 0;
        bool any_copy = false;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            int kp = __builtin_amdgcn_update_dpp(0, K[e], 0x138, 0xf, 0xf, false);  // wave_shr:1
            if (l == 0) kp = prev_last;
            prev_last = __builtin_amdgcn_readlane(K[e], 63);
            const int want = alive[e] ? K[e] - kp : 0;
            extra[e] = want > 1 ? want - 1 : 0;
            kept_m[e] = __ballot(alive[e] && want >= 1);
            removed_m[e] = alive_m[e] & ~kept_m[e];  // :1046-1049
            copy_m[e] = __ballot(extra[e] > 0);
            any_copy |= copy_m[e] != 0;
        }
        int fold[EPL], src_of[EPL];
        u64 taken[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) { fold[e] = 0; src_of[e] = -1; taken[e] = 0; }
        if (any_copy) {
            // sequential part of the reference loop, wave-uniform: copies go to the LOWEST free slot at
            // the time the sweep reaches the heavy particle (:1019-1035); free = empty after the cull,
            // or freed by a removal earlier in the sweep, and not yet taken by a copy.
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                u64 cm = copy_m[e];
                while (cm) {
                    const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)cm) - 1);
                    cm &= cm - 1;
                    const int dj = __builtin_amdgcn_readlane(extra[e], j);
                    int nfold = 0;
                    for (int c = 0; c < dj; ++c) {
                        int fe = -1, fq = -1;
#pragma unroll
                        for (int e2 = 0; e2 < EPL; ++e2) {
                            if (fe >= 0) continue;
                            const int nbits = min(64, d.slots - e2 * 64);
                            const u64 valid = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
                            u64 below;  // slots of word e2 already passed by the sweep
                            if (e2 < e) below = ~0ull; else if (e2 == e) below = (1ull << j) - 1ull; else below = 0ull;
                            const u64 occupied = (alive_m[e2] & ~(removed_m[e2] & below)) | taken[e2];
                            const u64 fr = ~occupied & valid;
                            if (fr) { fe = e2; fq = __ffsll((long long)fr) - 1; }
                        }
                        if (fe < 0) { nfold = dj - c; break; }  // full: fold the rest back (:1037-1041)
#pragma unroll
                        for (int e2 = 0; e2 < EPL; ++e2)
                            if (e2 == fe) {
                                taken[e2] |= 1ull << fq;
                                src_of[e2] = (l == fq) ? (e * 64 + j) : src_of[e2];
                            }
                    }
                    if (nfold) fold[e] = (l == j) ? nfold : fold[e];
                }
            }
        }
        // write back: kept particles get w' (+ folded copies, :1014,1039); copies replicate their source
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const size_t idx = (size_t)lv * d.slots + e * 64 + l;
            if ((kept_m[e] >> l) & 1ull) {
                float wn = w_after;
                for (int f = 0; f < fold[e]; ++f) wn += w_after;
                s.w[idx] = wn;
            }
            if (any_copy) {
                // fetch the source record from its lane (the whole voxel is in registers)
                const int so = src_of[e] < 0 ? 0 : src_of[e];
                float cpx = 0, cpy = 0, cpz = 0, cvx = 0, cvy = 0;
#pragma unroll
                for (int e2 = 0; e2 < EPL; ++e2) {
                    const int sl = so - e2 * 64;
                    const bool pick = sl >= 0 && sl < 64;
                    const int srcl = pick ? sl : 0;
                    const float tx = __shfl(r.px[e2], srcl, WAVE), ty = __shfl(r.py[e2], srcl, WAVE), tz = __shfl(r.pz[e2], srcl, WAVE);
                    const float tvx = __shfl(r.vx[e2], srcl, WAVE), tvy = __shfl(r.vy[e2], srcl, WAVE);
                    if (pick) { cpx = tx; cpy = ty; cpz = tz; cvx = tvx; cvy = tvy; }
                }
                if ((taken[e] >> l) & 1ull) {
                    s.px[idx] = cpx; s.py[idx] = cpy; s.pz[idx] = cpz; s.vx[idx] = cvx; s.vy[idx] = cvy;
                    if (s.vz0) s.vz0[idx] = s.vz0[(size_t)lv * d.slots + so];
                    s.w[idx] = w_after;
                }
            }
            newmask[e] = kept_m[e] | taken[e];
        }
    }
    int live_out = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        live_out += (int)__popcll(newmask[e]);
        if (l == 0) {
            s.mask[(size_t)lv * EPL + e] = newmask[e];
            if (r.nb[e]) s.nbmask[(size_t)lv * EPL + e] = 0ull;  // newborn flag -> 1 (:968)
        }
    }
    return live_out;
}

template <int EPL>
__global__ void __launch_bounds__(256) k_resample_work(MapDims d, DevState s, const int* __restrict__ work_list,
                                                       const int* __restrict__ work_count, int* __restrict__ part_live) {
    const int wpb = blockDim.x >> 6;
    const int wave = blockIdx.x * wpb + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * wpb;
    const int count = *work_count;
    int live_out = 0;
    int i = wave;
    VoxRegs<EPL> cur, nxt;
    if (i < count) load_voxel<EPL>(d, s, __builtin_amdgcn_readfirstlane(work_list[i]), cur);
    while (i < count) {
        const int inext = i + nwaves;
        if (inext < count) load_voxel<EPL>(d, s, __builtin_amdgcn_readfirstlane(work_list[inext]), nxt);
        live_out += resample_voxel<EPL>(d, s, cur);
        cur = nxt;
        i = inext;
    }
    if (lane_id() == 0) part_live[wave] = live_out;
}

// --------------------------------------------------------------------------
// Readout, :385-438.  Occupied voxels in ascending index order (stable compaction).
// --------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_occ_count(MapDims d, DevState s, float thr) {
    __shared__ int s_c[4];
    const int lv = blockIdx.x * 256 + threadIdx.x;
    const bool occ = lv < d.v_loc && s.res4[lv].x > thr;
    const u64 b = __ballot(occ);
    if (lane_id() == 0) s_c[threadIdx.x >> 6] = (int)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) s.blk_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
__global__ void __launch_bounds__(1024) k_occ_scan(DevState s, int nblk) {
    __shared__ int s_tmp[17];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? s.blk_cnt[i] : 0;
        int tot;
        const int ex = block_excl_scan_1024(v, s_tmp, &tot);
        if (i < nblk) s.blk_cnt[i] = s_run + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) s.fs->occupied_count = s_run;
}
__global__ void __launch_bounds__(256) k_occ_emit(MapDims d, DevState s, float thr, int cap) {
    __shared__ int s_c[4];
    const int lv = blockIdx.x * 256 + threadIdx.x;
    const bool occ = lv < d.v_loc && s.res4[lv].x > thr;
    const u64 b = __ballot(occ);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) s_c[w] = (int)__popcll(b);
    __syncthreads();
    int off = s.blk_cnt[blockIdx.x];
    for (int k = 0; k < w; ++k) off += s_c[k];
    if (occ) {
        const int pos = off + (int)__popcll(b & lanemask_lt());
        if (pos < cap) {
            // getVoxelPositionFromIndex :1090-1107 on the GLOBAL index
            const int index = lv + d.v_base;
            const int zc = d.ny * d.nx;
            const int zi = index / zc;
            const int rest = index - zi * zc;
            const int yi = rest / d.nx;
            const int xi = rest - yi * d.nx;
            const float cx = -d.half_x + d.res * 0.5f, cy = -d.half_y + d.res * 0.5f, cz = -d.half_z + d.res * 0.5f;
            s.occ_xyz[3 * pos] = (float)xi * d.res + cx;
            s.occ_xyz[3 * pos + 1] = (float)yi * d.res + cy;
            s.occ_xyz[3 * pos + 2] = (float)zi * d.res + cz;
        }
    }
}

// --------------------------------------------------------------------------
// state helpers
// --------------------------------------------------------------------------
__device__ __forceinline__ unsigned hash_u32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// benchmark fill (SURVEY 8d "saturated"): per_voxel zero-velocity particles per voxel,
// uniform in-voxel positions (kept 2% away from the faces), slots 0..per_voxel-1.
__global__ void k_seed_uniform(MapDims d, DevState s, int per_voxel, float weight, unsigned seed) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)d.v_loc * d.slots;
    if (t >= total) return;
    const int lv = (int)(t / d.slots), sl = (int)(t - (size_t)lv * d.slots);
    if (sl == 0) {
        for (int e = 0; e < d.mw; ++e) {
            const int nbits = max(0, min(64, per_voxel - e * 64));
            s.mask[(size_t)lv * d.mw + e] = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
            s.nbmask[(size_t)lv * d.mw + e] = 0ull;
        }
    }
    if (sl >= per_voxel) return;
    const int index = lv + d.v_base;
    const int zc = d.ny * d.nx;
    const int zi = index / zc, rest = index - zi * zc, yi = rest / d.nx, xi = rest - yi * d.nx;
    const unsigned h0 = hash_u32(seed ^ hash_u32((unsigned)index * 73u + (unsigned)sl));
    const unsigned h1 = hash_u32(h0 + 0x9e3779b9U), h2 = hash_u32(h1 + 0x9e3779b9U);
    const float u0 = 0.02f + 0.96f * (float)(h0 >> 8) * (1.f / 16777216.f);
    const float u1 = 0.02f + 0.96f * (float)(h1 >> 8) * (1.f / 16777216.f);
    const float u2 = 0.02f + 0.96f * (float)(h2 >> 8) * (1.f / 16777216.f);
    s.px[t] = ((float)xi + u0) * d.res - d.half_x;
    s.py[t] = ((float)yi + u1) * d.res - d.half_y;
    s.pz[t] = ((float)zi + u2) * d.res - d.half_z;
    s.vx[t] = 0.f; s.vy[t] = 0.f; s.w[t] = weight;
}

// import sparse records {flag,vx,vy,vz,px,py,pz,w} at (global voxel, slot); slot < 0 = first free
__global__ void k_import(MapDims d, DevState s, int n, const int* __restrict__ voxel, const int* __restrict__ slot,
                         const float* __restrict__ rec, int* __restrict__ n_failed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lv = voxel[i] - d.v_base;
    bool ok = lv >= 0 && lv < d.v_loc;
    int sl = -1;
    if (ok) {
        sl = slot ? slot[i] : -1;
        if (sl >= d.slots) ok = false;
        else if (sl < 0) { sl = claim_slot(s.mask, lv, d); ok = sl >= 0; }
        else {
            const u64 bit = 1ull << (sl & 63);
            const u64 prev = atomicOr(&s.mask[(size_t)lv * d.mw + (sl >> 6)], bit);
            ok = !(prev & bit);
        }
    }
    if (!ok) { atomicAdd(n_failed, 1); return; }
    const float* r = rec + 8 * (size_t)i;
    const size_t idx = (size_t)lv * d.slots + sl;
    s.vx[idx] = r[1]; s.vy[idx] = r[2];
    if (s.vz0) s.vz0[idx] = r[3];
    s.px[idx] = r[4]; s.py[idx] = r[5]; s.pz[idx] = r[6]; s.w[idx] = r[7];
    if (r[0] > 10.f) atomicOr(&s.nbmask[(size_t)lv * d.mw + (sl >> 6)], 1ull << (sl & 63));
}

__global__ void k_export(MapDims d, DevState s, int* __restrict__ voxel, int* __restrict__ slot,
                         float* __restrict__ rec, int* __restrict__ count, int cap) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)d.v_loc * d.slots;
    bool live = false;
    int lv = 0, sl = 0;
    if (t < total) {
        lv = (int)(t / d.slots); sl = (int)(t - (size_t)lv * d.slots);
        live = ((s.mask[(size_t)lv * d.mw + (sl >> 6)] | s.nbmask[(size_t)lv * d.mw + (sl >> 6)]) >> (sl & 63)) & 1ull;
    }
    const int pos = wave_agg_inc1(count, live);
    if (live && pos < cap) {
        const bool nbf = (s.nbmask[(size_t)lv * d.mw + (sl >> 6)] >> (sl & 63)) & 1ull;
        voxel[pos] = lv + d.v_base;
        slot[pos] = sl;
        float* r = rec + 8 * (size_t)pos;
        r[0] = nbf ? 15.f : 1.f;
        r[1] = s.vx[t]; r[2] = s.vy[t]; r[3] = s.vz0 ? s.vz0[t] : 0.f;
        r[4] = s.px[t]; r[5] = s.py[t]; r[6] = s.pz[t]; r[7] = s.w[t];
    }
}

// addRandomParticles :594-624 from the rand() table: 6 draws per particle, newborn flag (addAParticle)
__global__ void k_add_random(MapDims d, DevState s, FilterParams fp, int n, float weight) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = s.fs->r_cur + 6 * i;
    const float px = rand_float(s, fp, c, -d.half_x, d.half_x);
    const float py = rand_float(s, fp, c + 1, -d.half_y, d.half_y);
    const float pz = rand_float(s, fp, c + 2, -d.half_z, d.half_z);
    const float vx = rand_float(s, fp, c + 3, -1.f, 1.f);
    const float vy = rand_float(s, fp, c + 4, -1.f, 1.f);
    const float vz = rand_float(s, fp, c + 5, -1.f, 1.f);
    int gv;
    if (!voxel_of(d, px, py, pz, gv)) return;
    const int lv = gv - d.v_base;
    if (lv < 0 || lv >= d.v_loc) return;
    const int sl = claim_slot(s.mask, lv, d);
    if (sl < 0) return;
    const size_t idx = (size_t)lv * d.slots + sl;
    s.px[idx] = px; s.py[idx] = py; s.pz[idx] = pz; s.vx[idx] = vx; s.vy[idx] = vy; s.w[idx] = weight;
    if (s.vz0) s.vz0[idx] = vz;
    atomicOr(&s.nbmask[(size_t)lv * d.mw + (sl >> 6)], 1ull << (sl & 63));
}
__global__ void k_advance_rcur(DevState s, FilterParams fp, int by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) s.fs->r_cur = (int)(((long long)s.fs->r_cur + by) % max(fp.rtab_n, 1));
}

// ==========================================================================
// launchers
// ==========================================================================
int sweep_geometry(int slots, int* vpw_out) {
    // threads per block: multiple of 64, covering whole voxels, least idle lanes
    int best = 256, best_vpw = 256 / slots > 0 ? 256 / slots : 1;
    double best_eff = 0.0;
    for (int tpb = 128; tpb <= 512; tpb += 64) {
        if (slots > tpb) continue;
        const int vpw = tpb / slots;
        const double eff = (double)(vpw * slots) / tpb;
        if (eff > best_eff + 1e-9) { best_eff = eff; best = tpb; best_vpw = vpw; }
    }
    *vpw_out = best_vpw;
    return best;
}


void launch_frame_setup(const LaunchCtx& c, const float quat[4], const float cur_pos[3], bool reset_obs) {
    const int flags = RESET_PLANES | RESET_PRED | (reset_obs ? RESET_OBS : 0);
    const int n = c.d.np * DSP_OBS_CAP;
    const int grid = reset_obs ? (n + 1023) / 1024 : 1;
    hipLaunchKernelGGL(k_reset, dim3(grid), dim3(1024), 0, c.stream, c.d, c.s, flags, quat[0], quat[1], quat[2], quat[3],
                       cur_pos[0], cur_pos[1], cur_pos[2]);
}

void launch_obs_bin(const LaunchCtx& c, int n_pts, const float* pts_dev, const float quat[4], bool make_static_birth) {
    if (n_pts > 0)
        hipLaunchKernelGGL(k_obs_points, dim3((n_pts + 255) / 256), dim3(256), 0, c.stream, c.d, c.s, n_pts, pts_dev,
                           quat[0], quat[1], quat[2], quat[3], make_static_birth ? 1 : 0);
    hipLaunchKernelGGL(k_obs_gather, dim3(c.d.np), dim3(WAVE), 0, c.stream, c.d, c.s, n_pts);
}

void launch_predict_only(const LaunchCtx& c, float odx, float ody, float odz, float dt) {
    const KernelScratch* k = &c.k;
    hipLaunchKernelGGL(k_predict, dim3(k->nblk_sweep), dim3(k->tpb_sweep), 0, c.stream, c.d, c.s, c.fp, odx, ody, odz, dt,
                       k->vpw_sweep, c.s.vz0 ? 1 : 0, k->part_predict, k->mvmask, k->expmask);
}
void launch_claim(const LaunchCtx& c) {
    const KernelScratch* k = &c.k;
    hipLaunchKernelGGL(k_claim, dim3(k->nblk_sweep), dim3(k->tpb_sweep), 0, c.stream, c.d, c.s, k->vpw_sweep, k->mvmask,
                       k->part_claim);
}
void launch_predict(const LaunchCtx& c, float odx, float ody, float odz, float dt) {
    launch_predict_only(c, odx, ody, odz, dt);
    launch_claim(c);
}

void launch_ck_partial(const LaunchCtx& c) {
    const int nchunk = (c.d.capp + CK_PCH - 1) / CK_PCH;
    const int np8 = (c.d.np + 7) / 8 * 8;
    hipLaunchKernelGGL(k_ck_partial, dim3(np8 * nchunk), dim3(CK_TPB), 0, c.stream, c.d, c.s, c.fp, nchunk);
}
void launch_ck_finalize(const LaunchCtx& c) {
    hipLaunchKernelGGL(k_ck_finalize, dim3(1), dim3(1024), 0, c.stream, c.d, c.s, c.fp);
}
void launch_weight_update(const LaunchCtx& c) {
    const int nchunk = (c.d.capp + WU_TPB - 1) / WU_TPB;
    const int np8 = (c.d.np + 7) / 8 * 8;
    hipLaunchKernelGGL(k_weight, dim3(np8 * nchunk), dim3(WU_TPB), 0, c.stream, c.d, c.s, c.fp, nchunk);
}

void launch_birth_split(const LaunchCtx& c, int n_birth) {
    if (n_birth <= 0) return;
    hipLaunchKernelGGL(k_birth_split, dim3((n_birth + 3) / 4), dim3(256), 0, c.stream, c.d, c.s, c.fp, n_birth);
}
void launch_birth_plan_insert(const LaunchCtx& c, int n_birth) {
    if (n_birth <= 0) return;
    const long long total = (long long)n_birth * c.fp.nb_num;
    const unsigned gb = (unsigned)((total + 255) / 256);
    (void)hipMemsetAsync(c.k.vb_cnt, 0, sizeof(int) * (size_t)c.d.v_loc, c.stream);
    hipLaunchKernelGGL(k_birth_rank, dim3(1), dim3(1024), 0, c.stream, c.d, c.s, c.fp, n_birth);
    hipLaunchKernelGGL(k_birth_children, dim3(gb), dim3(256), 0, c.stream, c.d, c.s, c.fp, n_birth, c.k.child, c.k.vb_cnt, c.k.vb_idx);
    hipLaunchKernelGGL(k_birth_cursors, dim3(1), dim3(1024), 0, c.stream, c.d, c.s, c.fp, n_birth);
    hipLaunchKernelGGL(k_birth_insert, dim3(gb), dim3(256), 0, c.stream, c.d, c.s, c.fp, n_birth, c.k.child, c.k.vb_cnt, c.k.vb_idx);
}
void launch_birth(const LaunchCtx& c, int n_birth, bool) {
    launch_birth_split(c, n_birth);
    launch_birth_plan_insert(c, n_birth);
}

void launch_resample(const LaunchCtx& c) {
    const KernelScratch* k = &c.k;
    (void)hipMemsetAsync(k->work_count, 0, sizeof(int), c.stream);
    hipLaunchKernelGGL(k_resample_scan, dim3((c.d.v_loc + 1023) / 1024), dim3(1024), 0, c.stream, c.d, c.s, k->work_list, k->work_count);
    const int nblk = k->nblk_resample;  // persistent: nblk*4 waves stride over the work list
    if (c.d.mw == 1) hipLaunchKernelGGL(k_resample_work<1>, dim3(nblk), dim3(256), 0, c.stream, c.d, c.s, k->work_list, k->work_count, k->part_resample);
    else hipLaunchKernelGGL(k_resample_work<2>, dim3(nblk), dim3(256), 0, c.stream, c.d, c.s, k->work_list, k->work_count, k->part_resample);
}

void launch_occupied_compact(const LaunchCtx& c, float thr) {
    const int nblk = (c.d.v_loc + 255) / 256;
    hipLaunchKernelGGL(k_occ_count, dim3(nblk), dim3(256), 0, c.stream, c.d, c.s, thr);
    hipLaunchKernelGGL(k_occ_scan, dim3(1), dim3(1024), 0, c.stream, c.s, nblk);
    hipLaunchKernelGGL(k_occ_emit, dim3(nblk), dim3(256), 0, c.stream, c.d, c.s, thr, c.d.v_loc);
}
void launch_clear_future(const LaunchCtx& c) {
    (void)hipMemsetAsync(c.s.fut, 0, sizeof(float) * (size_t)c.d.v_loc * c.d.T, c.stream);
}

void launch_seed_uniform(const LaunchCtx& c, int per_voxel, float weight, unsigned seed) {
    const size_t total = (size_t)c.d.v_loc * c.d.slots;
    hipLaunchKernelGGL(k_seed_uniform, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, c.d, c.s, per_voxel, weight, seed);
}
void launch_import(const LaunchCtx& c, int n, const int* voxel_dev, const int* slot_dev, const float* rec8_dev, int* n_failed_dev) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_import, dim3((n + 255) / 256), dim3(256), 0, c.stream, c.d, c.s, n, voxel_dev, slot_dev, rec8_dev, n_failed_dev);
}
void launch_export(const LaunchCtx& c, int* voxel_out, int* slot_out, float* rec8_out, int* count_dev, int cap) {
    const size_t total = (size_t)c.d.v_loc * c.d.slots;
    hipLaunchKernelGGL(k_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, c.d, c.s, voxel_out, slot_out, rec8_out, count_dev, cap);
}
void launch_add_random(const LaunchCtx& c, int n, float weight) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_add_random, dim3((n + 255) / 256), dim3(256), 0, c.stream, c.d, c.s, c.fp, n, weight);
    hipLaunchKernelGGL(k_advance_rcur, dim3(1), dim3(64), 0, c.stream, c.s, c.fp, 6 * n);
}

// fold per-block partial counters (written without global atomics by the sweeps) into FrameScalars
__global__ void __launch_bounds__(1024) k_reduce_counters(DevState s, KernelScratch k, MapDims d) {
    __shared__ int s_red[1024];
    const int tid = threadIdx.x;
    int acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < k.nblk_sweep; i += 1024) {
        acc[0] += k.part_predict[i * 4]; acc[1] += k.part_predict[i * 4 + 1];
        acc[2] += k.part_predict[i * 4 + 2]; acc[3] += k.part_predict[i * 4 + 3];
        acc[4] += k.part_claim[i * 2]; acc[5] += k.part_claim[i * 2 + 1];
    }
    for (int i = tid; i < k.nblk_resample * 4; i += 1024) acc[6] += k.part_resample[i];
    int out[7];
    for (int c = 0; c < 7; ++c) {
        s_red[tid] = acc[c];
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        out[c] = s_red[0];
        __syncthreads();
    }
    if (tid == 0) {
        s.fs->n_live_in = out[0]; s.fs->n_out_of_map = out[1];
        s.fs->n_pyramid_full = out[2] + out[5]; s.fs->n_moved = out[3];
        s.fs->n_voxel_full = out[4]; s.fs->n_live_out = out[6];
        int nf = 0;
        for (int b = 0; b < d.np; ++b) nf += min(s.pyr_cnt[b], d.capp);
        s.fs->n_fov = nf;
    }
}
void launch_reduce_counters(const LaunchCtx& c) {
    hipLaunchKernelGGL(k_reduce_counters, dim3(1), dim3(1024), 0, c.stream, c.s, c.k, c.d);
}
