// dspmap_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the
// per-frame loop of the particle-based dynamic occupancy map.
//
// Reference behaviour being reproduced: include/dsp_dynamic.h of g-ch/DSP-map
//   update() preamble      :220-293   -> k_obs_points (+ k_reset's per-frame resets), k_obs_gather
//   mapPrediction          :627-701   -> k_predict, k_place (dspmap_sweep.hip; moveParticle :1206-1274)
//   mapUpdate              :704-793   -> k_pyr_prepare (range sort + work items), k_ck_partial, k_weight, k_ck_sum
//   mapAddNewBorn...       :796-921   -> k_birth_split, k_birth_rank, k_birth_children (dspmap_birth.h), k_birth_cursors,
//                                         k_birth_insert
//   mapOccupancy...Resample:924-1057  -> k_resample (dspmap_sweep.hip)
//   getOccupancyMap*       :385-438   -> k_occ_count, k_occ_scan, k_occ_emit, k_future_combine
// In a whole frame independent jobs share a launch (k_predict carries the gather and the birth rank, k_place the
// children, k_birth_split_cksum the 1/Ck reduction): DESIGN.md section 4.
// No MFMA: there is no dense contraction on this path; the kernels are
// HBM-streaming (predict / claim / resample) or LDS+VALU pair loops (update).
#include <hip/hip_runtime.h>
#include "dspmap_device.h"
#include "dspmap_kernels.h"
#include "dspmap_birth.h"

#define RESET_PLANES 1
#define RESET_OBS 2
#define RESET_PRED 4

// --------------------------------------------------------------------------
// k_reset: per-frame housekeeping.
//  RESET_PLANES: rotate the 29+17 boundary-plane normals by the sensor attitude (:226-232)
//  RESET_OBS   : zero per-pyramid observation counters, max range = -1 (:235-238), Ck = 0
//  RESET_PRED  : zero the per-pyramid particle counters (pyramids are rebuilt by prediction, :638-642)
// --------------------------------------------------------------------------
__global__ void k_reset(MapDims d, DevState s, int flags) {
    const float qw = s.fpar->quat[0], qx = s.fpar->quat[1], qy = s.fpar->quat[2], qz = s.fpar->quat[3];
    const float cx = s.fpar->cur_pos[0], cy = s.fpar->cur_pos[1], cz = s.fpar->cur_pos[2];
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
        for (int i = gt; i < d.np * DSP_OBS_CAP; i += gn) s.obs_ck[i] = 0;
        if (gt == 0) { s.fs->n_valid = 0; s.fs->n_obs = 0; s.fs->has_expected_override = 0; s.fs->bits_on = 0; }
    }
    if (flags & RESET_PRED) {
        for (int i = gt; i < d.np; i += gn) s.pyr_cnt[i] = 0;
        if (gt == 0) {
            s.fs->n_voxel_full_import = 0; s.fs->n_exp_up = 0; s.fs->n_exp_down = 0; s.fs->n_pyr_removed = 0; s.fs->n_dirty = 0; s.fs->n_overflow_inexact = 0; s.fs->n_place_vf = 0; s.fs->n_place_pf = 0; s.fs->n_view_tiles = 0; s.fs->pred_epoch = s.fs->pred_epoch + 1; s.fs->live_hint = s.fs->live_acc; s.hint_out[0] = s.fs->live_acc; s.fs->live_acc = 0; s.hint_out[1] = s.fs->mv_acc; s.fs->mv_acc = 0;
        }
    }
}

// --------------------------------------------------------------------------
// Observation binning, update() :244-290.
// k_obs_points: one thread per input point: rotate into the world-aligned
// sensor-centred frame (:247), FOV test (:250), pyramid cell (:260-263), range (:266).
// --------------------------------------------------------------------------
template <bool FUSED>
__global__ void __launch_bounds__(256) k_obs_points(MapDims d, DevState s, const FrameParams* __restrict__ ring, int ring_mask, int bits_blk0) {
    // bits_blk0 >= 0 (sparse whole frames): the workgroups from that index on rebuild the tile bitmaps from the per-tile flags, 256 tiles
    // each (DevState::vis_bits): whatever wrote the flags since the last frame -- a frame, an import, a restored checkpoint -- is in them
    if (FUSED && bits_blk0 >= 0 && (int)blockIdx.x >= bits_blk0) {
        const int ntl = (d.v_loc + 63) >> 6;
        const int t = ((int)blockIdx.x - bits_blk0) * 256 + (int)threadIdx.x;
        const bool on = t < ntl && (s.tile_live[t] != 0 || s.fut_dirty[t] != 0);
        const u64 b = __ballot(on);
        const int w0 = t >> 5;   // (lane 0's tile: a multiple of 64)
        if (lane_id() == 0 && t < ntl) {
            s.vis_bits[w0] = (unsigned)b; s.vis_bits[w0 + 1] = (unsigned)(b >> 32);
            s.pred_bits[w0] = (unsigned)b; s.pred_bits[w0 + 1] = (unsigned)(b >> 32);
            s.arr_bits[w0] = 0u; s.arr_bits[w0 + 1] = 0u;
        }
        return;
    }
    // A captured frame takes its parameter block straight from the pinned host ring the caller filled (no copy node in
    // front of the graph): every workgroup of this -- the frame's first -- kernel reads the slot over the bus, workgroup 0
    // also stores it in HBM for the kernels that follow.  The ring's read position is advanced by k_predict, after every
    // workgroup here has used it.
    const FrameParams* __restrict__ fpp = s.fpar;
    if (FUSED && ring) {
        fpp = ring + (*s.ring_seq & ring_mask);
        if (blockIdx.x == 0 && threadIdx.x < sizeof(FrameParams) / 4)
            reinterpret_cast<int*>(s.fpar)[threadIdx.x] = reinterpret_cast<const int*>(fpp)[threadIdx.x];
    }
    const int n_pts = fpp->n_pts;
    const float* __restrict__ pts = fpp->pts;
    const float qw = fpp->quat[0], qx = fpp->quat[1], qy = fpp->quat[2], qz = fpp->quat[3];
    const float cpx = fpp->cur_pos[0], cpy = fpp->cur_pos[1], cpz = fpp->cur_pos[2];
    (void)cpx; (void)cpy; (void)cpz;
    const int make_static_birth = fpp->static_birth;
    const float q[4] = {qw, qx, qy, qz};
    __shared__ float s_ph[DSP_MAX_PLANES_H * 3];
    __shared__ float s_pv[DSP_MAX_PLANES_V * 3];
    if (FUSED) {
        // whole-frame variant: k_reset's work rides on this kernel (one launch less per frame).  Every workgroup
        // rotates the boundary planes for itself (:226-232); workgroup 0 also publishes them for the later kernels.
        const int nh = d.np_h + 1, nv = d.np_v + 1;
        for (int i = threadIdx.x; i < nh + nv; i += blockDim.x) {
            float o[3];
            if (i < nh) {
                rotate_by_quat(s.planes_h0[3 * i], s.planes_h0[3 * i + 1], s.planes_h0[3 * i + 2], q, o);
                s_ph[3 * i] = o[0]; s_ph[3 * i + 1] = o[1]; s_ph[3 * i + 2] = o[2];
                if (blockIdx.x == 0) { s.planes_h[3 * i] = o[0]; s.planes_h[3 * i + 1] = o[1]; s.planes_h[3 * i + 2] = o[2]; }
            } else {
                const int j = i - nh;
                rotate_by_quat(s.planes_v0[3 * j], s.planes_v0[3 * j + 1], s.planes_v0[3 * j + 2], q, o);
                s_pv[3 * j] = o[0]; s_pv[3 * j + 1] = o[1]; s_pv[3 * j + 2] = o[2];
                if (blockIdx.x == 0) { s.planes_v[3 * j] = o[0]; s.planes_v[3 * j + 1] = o[1]; s.planes_v[3 * j + 2] = o[2]; }
            }
        }
        const int gt = blockIdx.x * blockDim.x + threadIdx.x, gn = (bits_blk0 >= 0 ? bits_blk0 : (int)gridDim.x) * blockDim.x;   // (the bitmap workgroups have left)
        for (int i = gt; i < d.np * DSP_OBS_CAP; i += gn) s.obs_ck[i] = 0;         // Ck = 0 (:235-238)
        for (int i = gt; i < d.np; i += gn) s.pyr_cnt[i] = 0;                       // pyramids are rebuilt by prediction (:638-642)
        if (gt == 0) {
            s.fs->cur_pos[0] = cpx; s.fs->cur_pos[1] = cpy; s.fs->cur_pos[2] = cpz;
            s.fs->n_valid = 0; s.fs->n_obs = 0; s.fs->has_expected_override = 0;   // k_obs_gather accumulates the first two
            s.fs->bits_on = bits_blk0 >= 0 ? 1 : 0;
            s.fs->n_voxel_full_import = 0; s.fs->n_exp_up = 0; s.fs->n_exp_down = 0; s.fs->n_pyr_removed = 0; s.fs->n_dirty = 0; s.fs->n_overflow_inexact = 0; s.fs->n_place_vf = 0; s.fs->n_place_pf = 0; s.fs->n_view_tiles = 0; s.fs->pred_epoch = s.fs->pred_epoch + 1; s.fs->live_hint = s.fs->live_acc; s.hint_out[0] = s.fs->live_acc; s.fs->live_acc = 0; s.hint_out[1] = s.fs->mv_acc; s.fs->mv_acc = 0;
        }
    } else {
        for (int i = threadIdx.x; i < (d.np_h + 1) * 3; i += blockDim.x) s_ph[i] = s.planes_h[i];
        for (int i = threadIdx.x; i < (d.np_v + 1) * 3; i += blockDim.x) s_pv[i] = s.planes_v[i];
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pts) {
        float r[3];
        rotate_by_quat(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], q, r);
        const int pyr = pyramid_of(d, s_ph, s_pv, r[0], r[1], r[2]);
        const float len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        s.pt_rot[i] = make_float4(r[0], r[1], r[2], len);
        s.pt_pyr[i] = pyr;
        // the view is not empty: this frame's synthesised birth cloud is the live one (dspmap_birth.h, BirthView)
        if (make_static_birth == 1 && pyr >= 0) s.fs->view_epoch = fpp->epoch;
    }
}

__global__ void k_obs_gather(MapDims d, DevState s) { obs_gather_wave(d, s, (int)blockIdx.x); }

// --------------------------------------------------------------------------
// k_pyr_sort: one workgroup per pyramid orders its particle list by RANGE bucket (counting sort in LDS).
// Why: the pair kernels' cost is (particles of a pyramid) x (observations of its neighbourhood), and in maps larger
// than a few metres most of those pairs are metres apart -- their pdf product is < 1e-19 (9 sigma) and adds nothing.
// |range(p) - range(o)| <= |p - o|, so with a chunk of range-sorted particles the pair kernels drop every observation
// whose range is farther than 9 sigma from the chunk's range interval: exact for Ck (such terms are zero on the
// fixed-point grid) and below the rounding of the weight update.  Measured on the corridor scene: 28 % / 72 % / 83 %
// of the pairs at 66x66x40 / 132x132x60 / 264x264x80.
// The order inside a bucket is arbitrary; nothing downstream depends on it (Ck sums are order-free, the weight of a
// particle is its own sum over the observations in their fixed order).
// --------------------------------------------------------------------------
__device__ __forceinline__ int range_bucket(const MapDims& d, const float4& r) {
    const float len = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z);
    return min(PS_NBK - 1, (int)(len * d.rng_inv_bw));
}
__device__ __forceinline__ int block_excl_scan_1024(int v, int* s_tmp, int* total);
#ifndef PSU
#define PSU 4   // list entries per thread and step in k_pyr_prepare's passes
#endif
#ifdef PYR_PROF
__device__ long long g_pprof[8 * 1024];
extern "C" int dspmap_debug_pyr_prof(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pprof), sizeof(long long) * (size_t)n); }
#define PSTAMP(k) do { if (threadIdx.x == 0 && b < 1024) g_pprof[b * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
__device__ __forceinline__ void pyr_sort_block(const MapDims& d, const DevState& s, int b) {
    __shared__ int s_hist[PS_NBK], s_base[PS_NBK];
    __shared__ int s_sel[8192];
    __shared__ int s_pick[2];
    __shared__ int s_scan[17];
    const int tid = threadIdx.x;
    PSTAMP(0);
    const int P_all = min(s.pyr_cnt[b], d.capa);
#ifdef PYR_PROF
    if (tid == 0 && b < 1024) g_pprof[b * 8 + 7] = P_all;
#endif
    if (tid == 0 && s.pyr_gcnt) s.pyr_gcnt[b] = P_all;   // (a sharded map: the ranks' list lengths are summed by the Ck all-reduce)
    if (P_all == 0) return;
    const float4* __restrict__ src = s.fov_rec + (size_t)b * d.capa;
    const int* __restrict__ src_slot = s.fov_slot + (size_t)b * d.capa;
    int* __restrict__ src_key = s.fov_key + (size_t)b * d.capa;
    // A full list (:1245-1259): the reference registers a pyramid's particles in the order of its voxel / slot sweep and
    // turns away what comes after SAFE_PARTICLE_NUM_PYRAMID entries.  The list here was filled in arrival order, but
    // every entry carries its sweep key: the capp SMALLEST keys stay (radix select of the capp-th key, 4 x 8 bits),
    // the others lose their slot -- the same particles as in the reference, independent of the arrival order.
    int kstar = 0x7fffffff;
    if (s.pyr_kstar) {
        // a sharded map: SAFE_PARTICLE_NUM_PYRAMID bounds the list over ALL ranks -- the threshold was selected over the union of
        // the ranks' entries (k_pyr_hist / k_pyr_pick + an all-reduce per digit, dspmap_dist.hip); this rank may have to turn
        // entries away although its own share is short of the capacity, and keeps fewer than CAPP of them
        kstar = s.pyr_kstar[b];
    } else if (P_all > d.capp) {
        // keys are below v_glob * slots: 13 bits per pass (8192 bins), highest digits first
        const unsigned kmax = (unsigned)d.v_glob * (unsigned)d.slots;
        const int nbits = 32 - __clz((int)max(kmax, 2u) - 1);
        const int npass = (nbits + 12) / 13;
        unsigned prefix = 0;
        int want = d.capp;
        for (int ps = npass - 1; ps >= 0; --ps) {
            const int shift = ps * 13;
            for (int q = tid; q < 8192; q += 1024) s_sel[q] = 0;
            __syncthreads();
            for (int i00 = 0; i00 < P_all; i00 += 1024 * PSU) {
                unsigned k_u[PSU];
#pragma unroll
                for (int u = 0; u < PSU; ++u) k_u[u] = (unsigned)src_key[min(i00 + u * 1024 + tid, P_all - 1)];   // (the step's loads together)
#pragma unroll
                for (int u = 0; u < PSU; ++u) {
                const int i = i00 + u * 1024 + tid;
                int bin = -1;
                if (i < P_all) {
                    const unsigned k = k_u[u];
                    if (ps == npass - 1 || (k >> (shift + 13)) == (prefix >> (shift + 13))) bin = (int)((k >> shift) & 8191u);
                }
                // the keys of a pyramid share their HIGH bits: in the first pass one LDS atomic per distinct bin of a wavefront, not one
                // per lane; the lower digits scatter over the 8192 bins -- there the grouping loop would run once per lane (round 6)
                if (ps == npass - 1) {
                    u64 todo = __ballot(bin >= 0);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const int bk = __shfl(bin, leader, WAVE);
                        const u64 grp = __ballot(bin == bk);
                        if (lane_id() == leader) atomicAdd(&s_sel[bk], (int)__popcll(grp));
                        todo &= ~grp;
                    }
                } else if (bin >= 0) atomicAdd(&s_sel[bin], 1);
                }
            }
            __syncthreads();
            {   // the bin that holds the want-th key: thread t owns bins [8t, 8t + 8); exclusive prefix over the threads
                int c8[8], mine = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) { c8[q] = s_sel[tid * 8 + q]; mine += c8[q]; }
                int tot;
                int ex = block_excl_scan_1024(mine, s_scan, &tot);
                if (ex < want && want <= ex + mine) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (ex < want && want <= ex + c8[q]) { s_pick[0] = tid * 8 + q; s_pick[1] = want - ex; }
                        ex += c8[q];
                    }
                }
            }
            __syncthreads();
            prefix |= (unsigned)s_pick[0] << shift;
            want = s_pick[1];
            __syncthreads();
        }
        kstar = (int)prefix;
    }
    PSTAMP(1);
    if (tid < PS_NBK) s_hist[tid] = 0;
    __syncthreads();
    int removed = 0;
    // (PSU entries per thread and step, their loads issued together: a step is one memory round trip, ~1.5 us, and a list of config E
    // has 55 k entries -- 54 steps per pass at one entry per thread: 75 + 80 us for the two passes of the longest lists, round 5)
    for (int i0 = tid; i0 < P_all; i0 += 1024 * PSU) {
        int key_u[PSU];
        float4 r_u[PSU];
#pragma unroll
        for (int u = 0; u < PSU; ++u) {   // unconditional loads (clamped index): a predicated load would be waited for before the next is issued
            const int ic = min(i0 + u * 1024, P_all - 1);
            key_u[u] = src_key[ic];
            r_u[u] = src[ic];
        }
#pragma unroll
        for (int u = 0; u < PSU; ++u) {
        const int i = i0 + u * 1024;
        if (i >= P_all) continue;
        const int key_i = key_u[u];
        const float4 r_i = r_u[u];
        if (key_i <= kstar) atomicAdd(&s_hist[range_bucket(d, r_i)], 1);
        else if (key_i != 0x7fffffff) {
            // turned away: the particle vanishes (-2): its cell -> (voxel, slot) -> occupancy bit.  The entry is marked so
            // that a second preparation of the same lists (stage API: after the prediction and again before the update)
            // changes nothing.
            src_key[i] = 0x7fffffff;
            const int c = src_slot[i];
            const int tile = c / (64 * d.slots), rem = c - tile * 64 * d.slots, slot = rem >> 6, lv = tile * 64 + (rem & 63);
            atomicAnd(&s.mask[(size_t)lv * d.mw + (slot >> 6)], ~(1ull << (slot & 63)));
            // the reference hands the slot back AT ONCE (:1256-1259): the arrivals that the sweep serves after this particle may
            // take it.  k_place ran before the lists were cut, so the voxel is noted for k_place_fix, which re-slots its arrivals
            atomicOr(&s.ta[(size_t)lv * d.mw + (slot >> 6)], 1ull << (slot & 63));
            if (atomicExch(&s.dflag[lv], 1) == 0) {
                const int q = atomicAdd(&s.fs->n_dirty, 1);
                if (q < DSP_DIRTY_CAP) s.dirty[q] = lv; else atomicAdd(&s.fs->n_overflow_inexact, 1);
            }
            ++removed;
        }
        }
    }
    if (__ballot(removed != 0)) { removed = wave_sum_i(removed); if (lane_id() == 0 && removed) atomicAdd(&s.fs->n_pyr_removed, removed); }
    __syncthreads();
    PSTAMP(2);
    if (tid < PS_NBK) {   // exclusive scan over the 128 buckets: two waves
        const int c = s_hist[tid];
        const int inc = wave_incl_scan_i(c);
        s_base[tid] = inc - c;
        if (tid == 63) s_hist[0] = inc;   // total of the first wave (s_hist is re-zeroed below)
    }
    __syncthreads();
    const int first = s_hist[0];
    __syncthreads();
    if (tid < PS_NBK) {
        if (tid >= 64) s_base[tid] += first;
        s_hist[tid] = 0;
    }
    __syncthreads();
    {
        // software-pipelined: the NEXT step's loads are issued before this step's (scattered) stores -- loads and stores share one
        // in-order counter on this part, so a step that waits for its loads also waits for every store issued before them
        int key_u[PSU], sl_u[PSU], key_n[PSU], sl_n[PSU];
        float4 r_u[PSU], r_n[PSU];
        auto fetch = [&](int i0, int (&kk)[PSU], int (&ss)[PSU], float4 (&rr)[PSU]) {
#pragma unroll
            for (int u = 0; u < PSU; ++u) {
                const int ic = min(i0 + u * 1024, P_all - 1);
                kk[u] = src_key[ic]; rr[u] = src[ic]; ss[u] = src_slot[ic];
            }
        };
        fetch(tid, key_u, sl_u, r_u);
        for (int i0 = tid; i0 < P_all; i0 += 1024 * PSU) {
            const bool more = i0 + 1024 * PSU < P_all;
            if (more) fetch(i0 + 1024 * PSU, key_n, sl_n, r_n);
#pragma unroll
            for (int u = 0; u < PSU; ++u) {
                const int i = i0 + u * 1024;
                if (i >= P_all || key_u[u] > kstar) continue;
                const float4 r = r_u[u];
                const int k = range_bucket(d, r);
                const int pos = s_base[k] + atomicAdd(&s_hist[k], 1);
                s.fov_rec_s[(size_t)b * d.capp + pos] = r;
                s.fov_slot_s[(size_t)b * d.capp + pos] = sl_u[u];
                // where the entry went: k_place_fix re-points the entries of the ARRIVALS it moves -- only theirs is noted (a stayer's sweep
                // key is its own cell; an arrival carries its source's): one scattered store less for nine entries in ten
                const int c = sl_u[u];
                const int tile = c / (64 * d.slots), rem = c - tile * 64 * d.slots;
                if (key_u[u] != g_of_lv(d, tile * 64 + (rem & 63)) * d.slots + (rem >> 6)) s.fov_spos[(size_t)b * d.capa + i] = pos;
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < PSU; ++u) { key_u[u] = key_n[u]; sl_u[u] = sl_n[u]; r_u[u] = r_n[u]; }
            }
        }
    }
    __syncthreads();
    PSTAMP(3);
}
__device__ __forceinline__ void pyr_items_block(const MapDims& d, const DevState& s, int* __restrict__ ck_items, int* __restrict__ wu_items,
                                                int* __restrict__ n_items, int* __restrict__ nb_tab);
// k_pyr_prepare: what the pair kernels need once per frame, in one launch: workgroups 0 .. np-1 order the pyramid lists,
// the last workgroup expands the work-item lists and neighbourhood tables (the two are independent of each other).
// (two workgroups per CU -- 64 registers: the 448 pyramids + 1 are ONE round of workgroups on 256 CUs; measured -2.4 us at 132x132x60
// saturated, round 6.  Also tried in round 6 and dropped: the list's keys and range buckets held in registers across the passes -- one
// read of the list instead of one per pass -- needs 2 x 14 registers per thread at that size on top of the passes' own: one workgroup
// per CU and spills, 37.8 against 13 us on lists that need no selection, LOG.md)
__global__ void __launch_bounds__(1024, 8) k_pyr_prepare(MapDims d, DevState s, int* __restrict__ ck_items, int* __restrict__ wu_items,
                                                      int* __restrict__ n_items, int* __restrict__ nb_tab) {
    if ((int)blockIdx.x == d.np) pyr_items_block(d, s, ck_items, wu_items, n_items, nb_tab);
    else pyr_sort_block(d, s, (int)blockIdx.x);
}

// --------------------------------------------------------------------------
// Distributed selection of the CAPP-th smallest sweep key of every pyramid over the ranks of a sharded map (:1256-1259 with the
// reference's GLOBAL capacity): radix select, 8 bits per pass, most significant digit first.  Per pass every rank counts its
// entries (those that match the digits chosen so far) per pyramid and digit -- k_pyr_hist --, the [np][256] tables are summed
// over the ranks (ncclAllReduce / the group driver's reduction kernel), and every rank picks the digit in which the cumulative
// count reaches what is still wanted -- k_pyr_pick: the same table, hence the same choice, everywhere.  sel[b] = {digits so
// far, entries still wanted among them; -1: the pyramid's list is not overfull}.  After the last pass kstar[b] is the key itself
// (sweep keys are unique: one per particle).
// --------------------------------------------------------------------------
#define PSEL_PASSES 4   // keys are below 2^31
__global__ void __launch_bounds__(256) k_pyr_hist(MapDims d, DevState s, int pass, const int2* __restrict__ sel, int* __restrict__ hist) {
    __shared__ int s_h[256];
    const int b = (int)blockIdx.x, tid = threadIdx.x;
    s_h[tid] = 0;
    __syncthreads();
    const int2 st = pass > 0 ? sel[b] : make_int2(0, 0);
    if (pass == 0 || st.y >= 0) {
        const int shift = 24 - 8 * pass;
        const int P_all = min(s.pyr_cnt[b], d.capa);
        const int* __restrict__ key = s.fov_key + (size_t)b * d.capa;
        for (int i = tid; i < P_all; i += 256) {
            const unsigned k = (unsigned)key[i];
            if (k == 0x7fffffffu) continue;   // (turned away by an earlier preparation of the same lists)
            if (pass > 0 && (k >> (shift + 8)) != ((unsigned)st.x >> (shift + 8))) continue;
            atomicAdd(&s_h[(k >> shift) & 255u], 1);
        }
    }
    __syncthreads();
    hist[b * 256 + tid] = s_h[tid];
}
__global__ void __launch_bounds__(64) k_pyr_pick(MapDims d, int pass, const int* __restrict__ hist, int2* __restrict__ sel, int* __restrict__ kstar) {
    const int b = (int)blockIdx.x, l = threadIdx.x;
    int2 st = pass > 0 ? sel[b] : make_int2(0, d.capp);
    int c4[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { c4[q] = hist[b * 256 + l * 4 + q]; mine += c4[q]; }
    const int inc = wave_incl_scan_i(mine);
    if (pass == 0) {
        const int total = __shfl(inc, 63, WAVE);
        if (total <= d.capp) st.y = -1;   // the list fits: nobody is turned away
    }
    if (st.y >= 0) {
        int ex = inc - mine;
        const bool here = ex < st.y && st.y <= inc;
        int digit = 0, left = 0;
        if (here) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ex < st.y && st.y <= ex + c4[q]) { digit = l * 4 + q; left = st.y - ex; }
                ex += c4[q];
            }
        }
        const u64 who = __ballot(here);
        const int src = __ffsll((long long)who) - 1;    // (exactly one lane: the counts are non-negative and their sum >= want)
        digit = __shfl(digit, src, WAVE); left = __shfl(left, src, WAVE);
        st.x |= digit << (24 - 8 * pass);
        st.y = left;
    }
    if (l == 0) {
        sel[b] = st;
        if (pass == PSEL_PASSES - 1) kstar[b] = st.y >= 0 ? st.x : 0x7fffffff;
    }
}
// ... and what this rank keeps of every list under the selected thresholds (the pair kernels' list lengths)
__global__ void __launch_bounds__(256) k_pyr_kept(MapDims d, DevState s, const int* __restrict__ kstar, int* __restrict__ kept) {
    __shared__ int s_n;
    const int b = (int)blockIdx.x, tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int P_all = min(s.pyr_cnt[b], d.capa), ks = kstar[b];
    const int* __restrict__ key = s.fov_key + (size_t)b * d.capa;
    int n = 0;
    for (int i = tid; i < P_all; i += 256) { const int k = key[i]; n += (k <= ks && k != 0x7fffffff) ? 1 : 0; }
    n = wave_sum_i(n);
    if (lane_id() == 0 && n) atomicAdd(&s_n, n);
    __syncthreads();
    if (tid == 0) kept[b] = s_n;
}
void launch_pyr_kept(const LaunchCtx& c, const int* kstar, int* kept) {
    hipLaunchKernelGGL(k_pyr_kept, dim3(c.d.np), dim3(256), 0, c.stream, c.d, c.s, kstar, kept);
}
void launch_pyr_hist(const LaunchCtx& c, int pass, const int2* sel, int* hist) {
    hipLaunchKernelGGL(k_pyr_hist, dim3(c.d.np), dim3(256), 0, c.stream, c.d, c.s, pass, sel, hist);
}
void launch_pyr_pick(const LaunchCtx& c, int pass, const int* hist, int2* sel, int* kstar) {
    hipLaunchKernelGGL(k_pyr_pick, dim3(c.d.np), dim3(64), 0, c.stream, c.d, pass, hist, sel, kstar);
}
int pyr_select_passes() { return PSEL_PASSES; }

// --------------------------------------------------------------------------
// k_place_fix: a particle that its pyramid's full list turns away gives its slot back AT ONCE (:1256-1259), so the arrivals
// that the reference's sweep serves after it may take that slot.  k_place gives out the slots before the lists are cut
// (k_pyr_prepare), i.e. as if nobody were turned away; this pass re-slots the arrivals of the (few) voxels in which that
// made a difference -- the voxels the cut noted in DevState::dirty:
//   * the voxel's arrivals are collected from its tile's inbox and put in sweep order (source key);
//   * the reference's walk is replayed twice, lane 0, a few dozen steps: as k_place did it (which slot holds which
//     arrival now), and with the turned-away particles handing their slots back -- arrivals from lower voxel indices
//     against the occupancy before the prediction, then the voxel's own turned-away particles leave, then the arrivals
//     from higher indices against the occupancy after it;
//   * an arrival whose slot differs is moved (position, velocity, weight), the occupancy word follows, and its entry in
//     the range-sorted pyramid list is pointed to the new cell (the weight update writes through that entry).
// One wave per dirty voxel.  Runs between k_pyr_prepare and the weight update's write-back: as the first workgroups of
// k_ck_partial's launch in a frame (that kernel reads the sorted COPIES of the particles, not their cells), or on its own
// (stage API).  Idempotent: a second run finds nothing turned away.
// Not treated (counted in n_overflow_inexact): an arrival that found its voxel FULL in k_place and would fit now.
// --------------------------------------------------------------------------
#define PF_MAXA 128   // arrivals of one voxel the pass orders (a voxel has at most 72 slots)
__device__ __forceinline__ int first_free(const u64* occ, int mw, int slots) {
    for (int e = 0; e < mw; ++e) {
        const int nbits = min(64, slots - e * 64);
        const u64 valid = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
        const u64 fr = ~occ[e] & valid;
        if (fr) return e * 64 + (__ffsll((long long)fr) - 1);
    }
    return -1;
}
__device__ __forceinline__ void place_fix_wave(const MapDims& d, const DevState& s, const float4* __restrict__ in_rec, const u64* __restrict__ omask,
                                               const int* __restrict__ refs, const int lv, int* s_key, int* s_idx, int* s_s1, int* s_s2) {
    const int l = lane_id();
    const int tile = lv >> 6, cap = 64 * d.slots;
    // the tile's inbox, its arrival count and its pmask words are THIS prediction's only if k_place served the tile in it (a tile
    // without arrivals keeps an earlier frame's: its voxels can still be dirty -- a stayer turned away by a full list -- and then
    // there is nothing to re-slot)
    const int n = s.in_n[2 * tile + 1] == s.fs->pred_epoch ? min(s.in_n[2 * tile], cap) : 0;
    const size_t base = (size_t)tile * cap;
    const int gD = g_of_lv(d, lv);   // the reference's index of the voxel: sweep keys
    int m = 0;   // (wave-uniform)
    // lanes hand data to each other through LDS below: the wave runs in lockstep, but the COMPILER must not move a lane's
    // load above another lane's store -- a wavefront-scope fence between the steps
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // the voxel's arrivals
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + l;
        bool mine = false;
        int key = 0;
        if (i < n) {
            mine = __float_as_int(in_rec[(base + i) * 2].x) == lv;   // (.x: the destination's storage index)
            key = __float_as_int(in_rec[(base + i) * 2 + 1].w);
        }
        const u64 g = __ballot(mine);
        const int k = m + (int)__popcll(g & lanemask_lt());
        if (mine && k < PF_MAXA) { s_key[k] = key; s_idx[k] = i; }
        m += (int)__popcll(g);
    }
    wave_sync();
    u64 tw[2] = {0ull, 0ull};
    for (int e = 0; e < d.mw; ++e) tw[e] = s.ta[(size_t)lv * d.mw + e];
    if (m > PF_MAXA) {
        if (l == 0) atomicAdd(&s.fs->n_overflow_inexact, 1);
    } else if (m > 0 && (tw[0] | tw[1])) {
        // sweep order: rank of every arrival among the voxel's arrivals (keys are distinct)
        int rk[2] = {0, 0};
        for (int j = 0; j < 2; ++j) {
            const int a = l + 64 * j;
            if (a < m) { const int ka = s_key[a]; for (int x = 0; x < m; ++x) rk[j] += s_key[x] < ka ? 1 : 0; }
        }
        int ka[2], ia[2];
        for (int j = 0; j < 2; ++j) { const int a = l + 64 * j; ka[j] = a < m ? s_key[a] : 0; ia[j] = a < m ? s_idx[a] : 0; }
        wave_sync();
        for (int j = 0; j < 2; ++j) { const int a = l + 64 * j; if (a < m) { s_key[rk[j]] = ka[j]; s_idx[rk[j]] = ia[j]; } }   // (one wave: reads above are done)
        wave_sync();
        // the two walks (lane 0)
        if (l == 0) {
            u64 org[2] = {0ull, 0ull}, cur0[2] = {0ull, 0ull};
            for (int e = 0; e < d.mw; ++e) { org[e] = omask[(size_t)lv * d.mw + e]; cur0[e] = s.pmask[(size_t)lv * d.mw + e]; }
            const long long dkey = (long long)gD * d.slots;   // key of (D, slot 0): arrivals below it come before D's own sweep
            // as placed: nobody hands a slot back
            {
                u64 occ[2] = {org[0], org[1]}, took[2] = {0ull, 0ull};
                int a = 0;
                for (; a < m && (long long)s_key[a] < dkey; ++a) {
                    const int sl = first_free(occ, d.mw, d.slots);
                    s_s1[a] = sl;
                    if (sl >= 0) { occ[sl >> 6] |= 1ull << (sl & 63); took[sl >> 6] |= 1ull << (sl & 63); }
                }
                occ[0] = cur0[0] | took[0]; occ[1] = cur0[1] | took[1];
                for (; a < m; ++a) {
                    const int sl = first_free(occ, d.mw, d.slots);
                    s_s1[a] = sl;
                    if (sl >= 0) occ[sl >> 6] |= 1ull << (sl & 63);
                }
            }
            // the reference: a turned-away particle takes its slot and hands it back at once
            {
                u64 occ[2] = {org[0], org[1]}, held[2] = {0ull, 0ull};
                int a = 0;
                for (; a < m && (long long)s_key[a] < dkey; ++a) {
                    const int s1 = s_s1[a];
                    const bool away = s1 >= 0 && ((tw[s1 >> 6] >> (s1 & 63)) & 1ull);
                    const int sl = s1 >= 0 ? first_free(occ, d.mw, d.slots) : -1;   // (an arrival k_place found no slot for stays dropped)
                    s_s2[a] = away ? -2 : sl;
                    if (!away && sl >= 0) { occ[sl >> 6] |= 1ull << (sl & 63); held[sl >> 6] |= 1ull << (sl & 63); }
                    if (s1 < 0 && first_free(occ, d.mw, d.slots) >= 0) atomicAdd(&s.fs->n_overflow_inexact, 1);
                }
                // the voxel's own sweep: its turned-away particles (cells occupied after the prediction) are gone
                occ[0] = (cur0[0] & ~tw[0]) | held[0]; occ[1] = (cur0[1] & ~tw[1]) | held[1];
                for (; a < m; ++a) {
                    const int s1 = s_s1[a];
                    const bool away = s1 >= 0 && ((tw[s1 >> 6] >> (s1 & 63)) & 1ull);
                    const int sl = s1 >= 0 ? first_free(occ, d.mw, d.slots) : -1;
                    s_s2[a] = away ? -2 : sl;
                    if (!away && sl >= 0) occ[sl >> 6] |= 1ull << (sl & 63);
                    if (s1 < 0 && first_free(occ, d.mw, d.slots) >= 0) atomicAdd(&s.fs->n_overflow_inexact, 1);
                }
            }
        }
        wave_sync();
        // moves: every source is read before any destination is written (a destination may be another mover's source)
        const size_t tcell = (size_t)tile * cap;
        const int ln = lv & 63;
        P3 mp[2]; V2 mvv[2]; float mw_[2], mz[2];
        bool mvd[2];
        int from[2], to[2];
        for (int j = 0; j < 2; ++j) {
            const int a = l + 64 * j;
            mvd[j] = false; from[j] = to[j] = -1;
            if (a < m) { from[j] = s_s1[a]; to[j] = s_s2[a]; mvd[j] = from[j] >= 0 && to[j] >= 0 && to[j] != from[j]; }
            if (mvd[j]) {
                const size_t sidx = tcell + (size_t)from[j] * 64 + ln;
                mp[j] = ld_pos(s, sidx); mvv[j] = ld_vel(s, sidx); mw_[j] = s.w[sidx]; mz[j] = s.vz0 ? s.vz0[sidx] : 0.f;
            }
        }
        u64 clr[2] = {0ull, 0ull}, setb[2] = {0ull, 0ull};
        for (int j = 0; j < 2; ++j) {
            if (mvd[j]) {
                const size_t didx = tcell + (size_t)to[j] * 64 + ln;
                st_pos(s, didx, mp[j].x, mp[j].y, mp[j].z); st_vel(s, didx, mvv[j].x, mvv[j].y); s.w[didx] = mw_[j];
                if (s.vz0) s.vz0[didx] = mz[j];
                clr[from[j] >> 6] |= 1ull << (from[j] & 63);
                setb[to[j] >> 6] |= 1ull << (to[j] & 63);
            }
        }
        for (int e = 0; e < d.mw; ++e) {
            const u64 c = wave_or_u64(clr[e]), st = wave_or_u64(setb[e]);
            if (l == 0 && (c | st)) s.mask[(size_t)lv * d.mw + e] = (s.mask[(size_t)lv * d.mw + e] & ~c) | st;   // (arrivals live in `mask`; this wave owns the voxel now)
        }
        // the moved arrivals' entries in their pyramids' lists (k_place noted the entry beside the inbox record; the range sort
        // noted where it put it)
        for (int j = 0; j < 2; ++j) {
            if (mvd[j]) {
                const int ref = refs[(size_t)tile * cap * 8 + cap + s_idx[l + 64 * j]];
                if (ref >= 0 && s.fov_key[ref] != 0x7fffffff) {
                    const int didx = (int)(tcell + (size_t)to[j] * 64 + ln);
                    s.fov_slot[ref] = didx;
                    const int b = ref / d.capa, sp = s.fov_spos[ref];
                    if (sp >= 0 && sp < d.capp) s.fov_slot_s[(size_t)b * d.capp + sp] = didx;
                }
            }
        }
    }
    // the voxel is clean again
    if (l == 0) {
        for (int e = 0; e < d.mw; ++e) s.ta[(size_t)lv * d.mw + e] = 0ull;
        s.dflag[lv] = 0;
    }
}
// the pass as a workgroup function: wave w of workgroup g (of ng) takes the dirty voxels w + 4 g, + 4 ng, ...
// lds: 4 x 4 x PF_MAXA ints (the caller's dynamic LDS)
#define PF_LDS_BYTES (4 * 4 * PF_MAXA * 4)
__device__ __forceinline__ void place_fix_block(const MapDims& d, const DevState& s, const float4* __restrict__ in_rec, const u64* __restrict__ omask,
                                                const int* __restrict__ refs, int* lds, int g, int ng) {
    const int nd = min(s.fs->n_dirty, DSP_DIRTY_CAP);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int* mine = lds + w * 4 * PF_MAXA;
    for (int q = g * 4 + w; q < nd; q += ng * 4)
        place_fix_wave(d, s, in_rec, omask, refs, s.dirty[q], mine, mine + PF_MAXA, mine + 2 * PF_MAXA, mine + 3 * PF_MAXA);
}
#define PF_WG 128   // workgroups of the pass (the first ones of k_ck_partial's launch in a frame): one dirty voxel per wave
__global__ void __launch_bounds__(256) k_place_fix(MapDims d, DevState s, const float4* __restrict__ in_rec, const u64* __restrict__ omask,
                                                   const int* __restrict__ refs) {
    extern __shared__ int s_pf[];
    place_fix_block(d, s, in_rec, omask, refs, s_pf, (int)blockIdx.x, (int)gridDim.x);
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
#define CK_TPB 256
#define CK_PCH 64
#define CK_SMALL_FOV 60000   // particles in the field of view up to which k_ck_partial takes half-size chunks
#define WU_TPB 256

// Work items of the two pair kernels: (pyramid, chunk of its particle list).  The list lengths are only
// known on the device, so a one-workgroup kernel expands them into a compact item list each frame and the
// pair kernels run a fixed grid that strides over it -- no empty workgroups, and the items of a heavy
// pyramid spread over all XCDs.  item = (pyramid << 12) | chunk.
// exclusive prefix sum over the workgroup
__device__ __forceinline__ int block_excl_scan_1024(int v, int* s_tmp, int* total) {
    // s_tmp: 17 ints.  Any blockDim.x that is a multiple of 64, up to 1024; all threads must call.
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nw = blockDim.x >> 6;
    const int inc = wave_incl_scan_i(v);
    if (l == 63) s_tmp[w] = inc;
    __syncthreads();
    if (w == 0) {  // scan of the wave totals by the first wave (no serial loop)
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

// lanes sharing one particle in k_weight (sized for the full neighbourhood also where the range cull applies: larger
// items -- split sized for the culled count -- widen the item's range window and were measured slower)
__device__ __forceinline__ int wu_split(int O) { return O <= 64 ? 1 : (O <= 128 ? 2 : (O <= 256 ? 4 : 8)); }

__device__ __forceinline__ void pyr_items_block(const MapDims& d, const DevState& s, int* __restrict__ ck_items, int* __restrict__ wu_items,
                                                int* __restrict__ n_items, int* __restrict__ nb_tab) {
    __shared__ int s_tmp[17];
    const int tid = threadIdx.x;
    // chunk size of k_ck_partial's items: with few particles in the field of view the kernel is one round of workgroups
    // whose run time is the pair loop of ONE item, so the items are halved (twice as many workgroups, half the loop)
    int mine = 0;
    for (int b = tid; b < d.np; b += (int)blockDim.x) mine += pyr_len(d, s, b);
    int tot_fov;
    (void)block_excl_scan_1024(mine, s_tmp, &tot_fov);
    const int pch = tot_fov <= CK_SMALL_FOV ? CK_PCH / 2 : CK_PCH;
    int base_ck = 0, base_wu = 0;
    for (int b0 = 0; b0 < d.np; b0 += (int)blockDim.x) {
        const int b = b0 + tid;
        int nck = 0, nwu = 0;
        if (b < d.np) {
            const int P = pyr_len(d, s, b);
            // observations in the 3x3 neighbourhood: decides how many lanes share one particle in k_weight
            const int h0 = b / d.np_v, v0 = b % d.np_v;
            // ... and the neighbourhood table the items of this pyramid read instead of rebuilding it: valid
            // neighbours compacted in h-major order (findPyramidNeighborIndexInFOV :1128-1147) + offsets
            int O = 0, nv = 0;
            int* tab = nb_tab + b;   // entry e of pyramid b at [e * np + b]: lanes = pyramids, coalesced stores
            for (int i = -d.nn; i <= d.nn; ++i)
                for (int j = -d.nn; j <= d.nn; ++j) {
                    const int h = h0 + i, v = v0 + j;
                    if (h >= 0 && h < d.np_h && v >= 0 && v < d.np_v) {
                        tab[(size_t)nv * d.np] = h * d.np_v + v;
                        tab[(size_t)(DSP_MAX_NBINS + nv) * d.np] = O;
                        O += s.obs_cnt[h * d.np_v + v];
                        ++nv;
                    }
                }
            for (; nv < d.nbins; ++nv) { tab[(size_t)nv * d.np] = -1; tab[(size_t)(DSP_MAX_NBINS + nv) * d.np] = O; }
            tab[(size_t)(DSP_MAX_NBINS + d.nbins) * d.np] = O;
            const int pw = WU_TPB / wu_split(O);      // particles per k_weight item
            nck = O > 0 ? (P + pch - 1) / pch : 0;
            nwu = max(1, (P + pw - 1) / pw);          // chunk 0 always exists: it owns the bin's 1/Ck sum
        }
        int tot_ck, tot_wu;
        const int o_ck = base_ck + block_excl_scan_1024(nck, s_tmp, &tot_ck);
        const int o_wu = base_wu + block_excl_scan_1024(nwu, s_tmp, &tot_wu);
        for (int c = 0; c < nck; ++c) ck_items[o_ck + c] = (b << 12) | c;
        for (int c = 0; c < nwu; ++c) wu_items[o_wu + c] = (b << 12) | c;
        base_ck += tot_ck; base_wu += tot_wu;
    }
    if (tid == 0) { n_items[0] = base_ck; n_items[1] = base_wu; n_items[2] = pch; }
}

// neighbourhood table of a pyramid in LDS: s_bin[nbins] bins, s_off[nbins+1] exclusive offsets of their
// observation counts (nbins = (2*nn+1)^2 <= 25), copied from k_pyr_prepare's table: one load, one barrier.
__device__ __forceinline__ void neighbor_load(const MapDims& d, const int* __restrict__ nb_tab, int b, int* s_bin, int* s_off) {
    const int tid = threadIdx.x;
    const int* tab = nb_tab + b;   // entry e of pyramid b at [e * np + b]
    if (tid < d.nbins) s_bin[tid] = tab[(size_t)tid * d.np];
    else if (tid >= 32 && tid - 32 <= d.nbins) s_off[tid - 32] = tab[(size_t)(DSP_MAX_NBINS + tid - 32) * d.np];
}

__global__ void __launch_bounds__(CK_TPB) k_ck_partial(MapDims d, DevState s, FilterParams fp, const int* __restrict__ items,
                                                       const int* __restrict__ n_items, const int* __restrict__ nb_tab,
                                                       const float4* __restrict__ in_rec, const u64* __restrict__ omask, const int* __restrict__ refs, int with_fix) {
    // the first PF_WG workgroups re-slot the arrivals of the voxels in which a full pyramid list turned a particle away
    // (k_place_fix: nothing to do in most frames); the pair items do not depend on it
    extern __shared__ float4 s_z[];
    if ((int)blockIdx.x < PF_WG) { if (with_fix) place_fix_block(d, s, in_rec, omask, refs, reinterpret_cast<int*>(s_z), (int)blockIdx.x, PF_WG); return; }
    const int BX = (int)blockIdx.x - PF_WG, GX = (int)gridDim.x - PF_WG;   // [nbins * DSP_OBS_CAP] the neighbourhood's observations within range of the chunk ...
    int* s_oi = reinterpret_cast<int*>(s_z + (size_t)d.nbins * DSP_OBS_CAP);   // ... and their global indices
    __shared__ float4 s_p[CK_PCH];
    __shared__ int s_bin[DSP_MAX_NBINS];
    __shared__ int s_off[DSP_MAX_NBINS + 1];
    __shared__ float s_rng[2];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int total = n_items[0], pch = n_items[2];
    int item_next = BX < total ? items[BX] : 0;
    for (int it = BX; it < total; it += GX) {
        // the dependent-load chain of an item is what bounds this kernel (few pairs per lane): everything that only
        // needs the item id is requested at once -- particle count, neighbourhood table, the chunk's particles (read
        // unmasked, rows always exist) and the next item's id
        const int item = item_next;
        const int b = item >> 12, chunk = item & 0xfff;
        const int start = chunk * pch;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < pch && start + tid < d.capp) r = s.fov_rec_s[(size_t)b * d.capp + start + tid];
        const int P = pyr_len(d, s, b);
        if (it + GX < total) item_next = items[it + GX];
        const int npart = min(pch, P - start);
        __syncthreads();  // LDS reuse across items
        neighbor_load(d, nb_tab, b, s_bin, s_off);
        if (tid < CK_PCH) {   // the first wave holds the chunk: its range interval decides which observations matter
            const float len = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z);
            float lo = tid < npart ? len : 3.0e38f, hi = tid < npart ? len : -3.0e38f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, WAVE)); hi = fmaxf(hi, __shfl_xor(hi, o, WAVE)); }
            r.w = fp.p_det * r.w;  // P_detection * weight (pre-update weights), :732
            // The radius beyond which this chunk's terms are EXACTLY zero: a term is snapped to the 2^-34 grid before it is added
            // (ck_snap), so c3 * w * exp2(-K s) < 2^-35 adds nothing; ranges R apart mean a distance >= R, i.e.
            // s >= (1000 R / sigma - sqrt(3))^2 for the truncated table indices (u in (1000 z - 1, 1000 z]).  With the chunk's
            // largest weight: s > (log2(c3 w) + 35) / K  <=>  R > sigma * (sqrt(.) / 1000 + 0.0018); 0.01 sigma of margin covers
            // the roundings of the two ranges.  Never wider than the handle's cull radius (which also bounds the +-9.9 clamp).
            float wm = tid < npart ? r.w : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o, WAVE));
            const float need = (__log2f(fp.pdf_c3 * wm) + 35.f) * (1.f / 7.213475204444817e-07f);
            const float rad = fminf(fp.cull_r, fp.sigma_ob * (sqrtf(fmaxf(need, 0.f)) * 0.001f + 0.01f));
            if (tid == 0) { s_rng[0] = lo - rad; s_rng[1] = hi + rad; s_n = 0; }
            s_p[tid] = r;
        }
        __syncthreads();
        const int O_all = s_off[d.nbins];
        if (O_all == 0) continue;
        {   // keep the observations whose range lies within 9 sigma of the chunk's (compacted, any order: Ck is order-free)
            const float lo = s_rng[0], hi = s_rng[1];
            for (int base = 0; base < O_all; base += CK_TPB) {
                const int o = base + tid;
                bool keep = false;
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                int oi = 0;
                if (o < O_all) {
                    int k = 0;
                    for (int q = 1; q < d.nbins; ++q) k += (o >= s_off[q]) ? 1 : 0;
                    oi = s_bin[k] * DSP_OBS_CAP + (o - s_off[k]);
                    z = s.obs[oi];
                    keep = z.w >= lo && z.w <= hi;
                }
                const u64 bal = __ballot(keep);
                int wbase = 0;
                if (lane_id() == 0 && bal) wbase = atomicAdd(&s_n, (int)__popcll(bal));
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                if (keep) {
                    const int pos = wbase + (int)__popcll(bal & lanemask_lt());
                    s_z[pos] = z;
                    s_oi[pos] = oi;
                }
            }
        }
        __syncthreads();
        const int O = s_n;
        if (O == 0) continue;
        // lanes = (observation, particle group): with few observations the 256 lanes split the particle
        // chunk G ways so that every lane is busy and the loop is short; partial sums meet in the (fixed-point, order-independent) atomic
        const int opad = O <= 64 ? 64 : (O <= 128 ? 128 : 256);
        const int G = CK_TPB / opad;
        const int g = tid / opad;
        for (int o = tid % opad; o < O; o += opad) {
            const int oi = s_oi[o];
            const float4 z = s_z[o];
            // every term is snapped to the 2^-34 grid before it is added, so the (double) partial sum is exact and the
            // total does not depend on the order of the particles in the pyramid's list either (built with atomics)
            double acc = 0.0;
            int i = g;
            for (; i + G < npart; i += 2 * G) {   // two particles per iteration: packed fp32
                const float4 p = s_p[i], p2 = s_p[i + G];
                const f2v gk = pair_gk2(f2v{p.x, p2.x}, f2v{p.y, p2.y}, f2v{p.z, p2.z}, f2v{z.x, z.x}, f2v{z.y, z.y}, f2v{z.z, z.z},
                                        fp.sigma_ob, fp.inv_sigma_ob, fp.pdf_c3);
                acc = __dadd_rn(acc, ck_snap(p.w * gk.x));
                acc = __dadd_rn(acc, ck_snap(p2.w * gk.y));
            }
            if (i < npart) {
                const float4 p = s_p[i];
                acc = __dadd_rn(acc, ck_snap(p.w * pair_gk(p.x, p.y, p.z, z.x, z.y, z.z, fp.sigma_ob, fp.inv_sigma_ob, fp.pdf_c3)));
            }
            atomicAdd(reinterpret_cast<unsigned long long*>(&s.obs_ck[oi]), (unsigned long long)__double2ll_rn(acc * CK_FIX_SCALE));
        }
    }
}

// Ck += expected_new_born_objects + kappa (:737) is applied on the fly by k_weight (each workgroup
// adds the frame constant when it stages its observation tile); the chunk-0 workgroup of every
// pyramid also writes the final Ck of its own bin and the bin's sum of 1/Ck.  k_ck_sum then
// reduces the 448 partial sums deterministically into the birth normaliser w_nb * sum_k 1/Ck (:799-805).
__device__ __forceinline__ float frame_lambda(const DevState& s, const FilterParams& fp) {
    if (s.fs->has_expected_override) return s.fs->expected_newborn;
    return fp.nb_weight * (float)s.fs->n_valid * (float)fp.nb_num;  // :292
}
__device__ __forceinline__ void ck_sum_block(const MapDims& d, const DevState& s, const FilterParams& fp, float* s_red);
__global__ void __launch_bounds__(512) k_ck_sum(MapDims d, DevState s, FilterParams fp) {
    __shared__ float s_red[512];
    ck_sum_block(d, s, fp, s_red);
}

// --------------------------------------------------------------------------
// mapUpdate pass 2, :743-790: for every particle inside the FOV
//   skip if occluded: |p| > max_range[b] + 0.3 and the pyramid has observations (:759-765)
//   w *= (1-P_d) + sum over obs k of N(b) of P_d*g/Ck                         (:768-786)
// Lanes are particles; the neighbourhood's observations {x,y,z,P_d/Ck} are
// staged in LDS and broadcast.  The new weight is scattered back to the slot.
// --------------------------------------------------------------------------
// SKIP: far pairs are branched over (maps much larger than 9 sigma: most pairs are far and a wave's range-sorted
// particles agree about it); otherwise every pair is evaluated two at a time with packed fp32 and far ones are
// masked to zero.  Both variants add exactly the same terms.
template <bool SKIP>
__global__ void __launch_bounds__(WU_TPB) k_weight(MapDims d, DevState s, FilterParams fp, const int* __restrict__ items,
                                                   const int* __restrict__ n_items, const int* __restrict__ nb_tab) {
    extern __shared__ float4 s_o[];   // [nbins * DSP_OBS_CAP]: the neighbourhood's observations {x, y, z, P_d/Ck} ...
    float* s_len = reinterpret_cast<float*>(s_o + (size_t)d.nbins * DSP_OBS_CAP);   // ... and their ranges
    __shared__ int s_bin[DSP_MAX_NBINS];
    __shared__ int s_off[DSP_MAX_NBINS + 1];
    __shared__ float s_inv[WU_TPB / 64];
    __shared__ float s_mm[2 * WU_TPB / 64];
    __shared__ int s_wc[64];   // [2][4 waves][8 classes] per-wave kept counts of a staging round (double-buffered)
    const int tid = threadIdx.x;
    const int total = n_items[1];
    const float add = frame_lambda(s, fp) + fp.kappa;  // :737
    for (int it = blockIdx.x; it < total; it += gridDim.x) {
        const int item = items[it];
        const int b = item >> 12, chunk = item & 0xfff;
        const int P = pyr_len(d, s, b);
        __syncthreads();  // LDS reuse across items
        if (chunk == 0) {
            // final Ck of this pyramid's own observations + their sum of 1/Ck (:799-804)
            const int nob = s.obs_cnt[b];
            float inv = 0.f;
            if (tid < nob) {
                const float ck = ck_from_fix(s.obs_ck[b * DSP_OBS_CAP + tid]) + add;
                s.obs_ckf[b * DSP_OBS_CAP + tid] = ck;
                inv = __fdiv_rn(1.f, ck);
            }
            inv = wave_sum(inv);
            if ((tid & 63) == 0) s_inv[tid >> 6] = inv;
            __syncthreads();
            if (tid == 0) s.part_inv[b] = (s_inv[0] + s_inv[1]) + (s_inv[2] + s_inv[3]);
            if (P == 0) continue;
        }
        neighbor_load(d, nb_tab, b, s_bin, s_off);
        __syncthreads();
        const int O = s_off[d.nbins];
        // SPL adjacent lanes share one particle and split the observation loop (short critical path when
        // the neighbourhood holds hundreds of observations); their partial sums are combined with shuffles.
        // Lane `sub` owns the observations o with o % spl == sub, in their original order.
        const int spl = wu_split(O);
        const int pw = WU_TPB / spl;
        const int i = chunk * pw + tid / spl;
        const int sub = tid % spl;
        const bool valid = i < P;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        size_t ri = 0;
        bool occluded = false;
        float dist = 0.f;
        if (valid) {
            ri = (size_t)b * d.capp + i;
            p = s.fov_rec_s[ri];
            const float maxlen = s.obs_maxlen[b];
            dist = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
            occluded = maxlen > 0.f && dist > maxlen + fp.occl_margin;  // :761-765
        }
        int n_mine = 0;   // observations lane `sub` iterates over: entry j at s_o[j * spl + sub]
        if (!SKIP) {
            for (int o = tid; o < O; o += WU_TPB) {
                int k = 0;
                for (int q = 1; q < d.nbins; ++q) k += (o >= s_off[q]) ? 1 : 0;
                const int oi = s_bin[k] * DSP_OBS_CAP + (o - s_off[k]);
                float4 z = s.obs[oi];
                s_len[o] = z.w;
                z.w = __fdiv_rn(fp.p_det, ck_from_fix(s.obs_ck[oi]) + add);
                s_o[o] = z;
            }
            n_mine = (O - sub + spl - 1) / spl;
        } else {
            // stage only the observations within 9 sigma of the range interval of this item's (range-sorted) particles.
            // The compaction is done per residue class o % spl and keeps the original order inside a class, so every
            // lane sums the same terms in the same order as it would over the full list.
            float lo = (valid && !occluded) ? dist : 3.0e38f, hi = (valid && !occluded) ? dist : -3.0e38f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, WAVE)); hi = fmaxf(hi, __shfl_xor(hi, o, WAVE)); }
            if ((tid & 63) == 0) { s_mm[(tid >> 6) * 2] = lo; s_mm[(tid >> 6) * 2 + 1] = hi; }
            __syncthreads();
            lo = fminf(fminf(s_mm[0], s_mm[2]), fminf(s_mm[4], s_mm[6])) - fp.cull_r;
            hi = fmaxf(fmaxf(s_mm[1], s_mm[3]), fmaxf(s_mm[5], s_mm[7])) + fp.cull_r;
            const int wave = tid >> 6, l = tid & 63;
            // lanes of my residue class (WU_TPB and 64 are multiples of spl: class of o == class of the lane); spl is 1, 2, 4 or 8
            const u64 cmask = (spl == 1 ? ~0ull : spl == 2 ? 0x5555555555555555ull : spl == 4 ? 0x1111111111111111ull : 0x0101010101010101ull) << sub;
            int run = 0, round = 0;
            for (int base = 0; base < O; base += WU_TPB, ++round) {
                const int o = base + tid;
                bool keep = false;
                float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                int oi = 0;
                if (o < O) {
                    int k = 0;
                    for (int q = 1; q < d.nbins; ++q) k += (o >= s_off[q]) ? 1 : 0;
                    oi = s_bin[k] * DSP_OBS_CAP + (o - s_off[k]);
                    z = s.obs[oi];
                    keep = z.w >= lo && z.w <= hi;
                }
                const u64 bal = __ballot(keep) & cmask;
                int* wc = s_wc + (round & 1) * 32;
                if (l < spl) wc[wave * 8 + l] = (int)__popcll(bal);   // lane c (< spl) counts class c: its cmask is class c's
                __syncthreads();
                int off = run, tot = 0;
                for (int w = 0; w < WU_TPB / 64; ++w) { const int c = wc[w * 8 + sub]; tot += c; off += w < wave ? c : 0; }
                if (keep) {
                    const int pos = (off + (int)__popcll(bal & lanemask_lt())) * spl + sub;
                    s_len[pos] = z.w;
                    z.w = __fdiv_rn(fp.p_det, ck_from_fix(s.obs_ck[oi]) + add);
                    s_o[pos] = z;
                }
                run += tot;
            }
            n_mine = run;
        }
        __syncthreads();
        float sum = 0.f;
        // a pair is evaluated iff the two ranges are within 9 sigma of each other (a per-pair rule: the result does
        // not depend on which particles share a workgroup).  The lanes of a wave hold range-sorted neighbours, so the
        // test goes the same way for (nearly) all of them and a far observation costs a compare, not a pdf.
        if (valid && !occluded) {
            if (SKIP) {
                for (int j = 0; j < n_mine; ++j) {
                    const int o = j * spl + sub;
                    if (fabsf(s_len[o] - dist) > fp.cull_r) continue;
                    const float4 z = s_o[o];
                    sum += pair_gk(p.x, p.y, p.z, z.x, z.y, z.z, fp.sigma_ob, fp.inv_sigma_ob, fp.pdf_c3) * z.w;
                }
            } else {
                int o = sub;
                for (; o + spl < O; o += 2 * spl) {   // two observations per iteration, same summation order
                    const float4 z = s_o[o], z2 = s_o[o + spl];
                    const f2v gk = pair_gk2(f2v{p.x, p.x}, f2v{p.y, p.y}, f2v{p.z, p.z}, f2v{z.x, z2.x}, f2v{z.y, z2.y}, f2v{z.z, z2.z},
                                            fp.sigma_ob, fp.inv_sigma_ob, fp.pdf_c3);
                    if (fabsf(s_len[o] - dist) <= fp.cull_r) sum += gk.x * z.w;
                    if (fabsf(s_len[o + spl] - dist) <= fp.cull_r) sum += gk.y * z2.w;
                }
                if (o < O && fabsf(s_len[o] - dist) <= fp.cull_r) {
                    const float4 z = s_o[o];
                    sum += pair_gk(p.x, p.y, p.z, z.x, z.y, z.z, fp.sigma_ob, fp.inv_sigma_ob, fp.pdf_c3) * z.w;
                }
            }
        }
        if (spl >= 2) sum += __shfl_xor(sum, 1, WAVE);
        if (spl >= 4) sum += __shfl_xor(sum, 2, WAVE);
        if (spl >= 8) sum += __shfl_xor(sum, 4, WAVE);
        if (valid && !occluded && sub == 0) s.w[s.fov_slot_s[ri]] = p.w * ((1.f - fp.p_det) + sum);  // :786
    }
}

// --------------------------------------------------------------------------
// Birth, mapAddNewBornParticlesByObservation :796-921.
// k_birth_split: one wave per source point.  Dempster-Shafer static/dynamic
// split from the mass already in the point's voxel (:827-866), lanes = slots.
// --------------------------------------------------------------------------
// cvr (whole frame, the children are done): also the point's draws from the velocity table / the rand() stream (:884-886,
// :895-897: three per child inside the map beyond the static ones, by branch) -- what k_birth_cursors would count, for the
// insertion kernel that computes its cursors itself
__device__ __forceinline__ void birth_split_wave(const MapDims& d, const DevState& s, const FilterParams& fp, int i, int2* cvr = nullptr,
                                                 const unsigned* inside_in = nullptr) {   // inside_in: the point's "inside the map" bits, if the caller has them
    const BirthView bv = birth_view(s);
    if (cvr) *cvr = make_int2(0, 0);
    if (i >= bv.n) return;
    const int l = lane_id();
    const BirthSrc src = birth_at(bv, i);
    BirthPlan pl;
    pl.gvox = -1; pl.n_static = 0; pl.inside = 0; pl.pbase = pl.vbase = pl.rbase = 0;
    int gv;
    const bool ok = birth_src_voxel(d, s, src, pl.cx, pl.cy, pl.cz, gv);
    int n_static = 0;
    if (ok) {
        pl.gvox = gv;
        const int lv = lv_of_g(d, gv);
        if (lv >= 0) {
            float ws = 0.f, wsd = 0.f, wd = 0.f;
            for (int e = 0; e < d.mw; ++e) {
                const int sl = e * 64 + l;
                const u64 m = s.mask[(size_t)lv * d.mw + e] & ~s.nbmask[(size_t)lv * d.mw + e];  // 0.9<flag<14 :830
                if (sl < d.slots && ((m >> l) & 1ull)) {
                    const size_t idx = pidx(d, lv, sl);
                    const V2 pv = ld_vel(s, idx);
                    const float vabs = fabsf(pv.x) + fabsf(pv.y) + 0.f;  // vz == 0
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
    if (cvr && ok && src.intensity > 0.01f) {
        const unsigned inside = inside_in ? *inside_in : s.plan_inside[i];
        const int nb = fp.nb_num;
        const int model_end = src.nx > -100.f ? fp.model_nb : n_static;  // :881
        auto below = [](int k) { return k >= 32 ? ~0u : ((1u << k) - 1u); };   // bits [0, k)
        const int lo = min(n_static, nb), mid = min(max(model_end, lo), nb);
        *cvr = make_int2(3 * __popc(inside & below(mid) & ~below(lo)), 3 * __popc(inside & below(nb) & ~below(mid)));
    }
}
__global__ void k_birth_split(MapDims d, DevState s, FilterParams fp) {
    birth_split_wave(d, s, fp, (int)(blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE));
}

// One wave = one source point of a frame whose children were not generated yet: the children (birth_child_thread's job, lanes =
// children, "inside the map" bits by ballot) AND the split (birth_split_wave's job, lanes = slots), the same arithmetic in the same
// order, with their loads in common stages -- {source point, position cursor} -> {table draws, occupancy words} -> {the slot's
// particle, the child's bucket}: three round trips instead of five on a kernel that is nothing but its round trips.
__device__ __forceinline__ void birth_point_wave(const MapDims& d, const DevState& s, const FilterParams& fp, int i, float4* __restrict__ child,
                                                 int* __restrict__ vb_cnt, int* __restrict__ vb_idx, int2* cvr) {
    const BirthView bv = birth_view(s);
    *cvr = make_int2(0, 0);
    if (i >= bv.n) return;
    const int l = lane_id(), nb = fp.nb_num;
    // stage 1
    const BirthSrc src = birth_at(bv, i);
    const int pb = s.plan_pbase[i];
    BirthPlan pl;
    pl.gvox = -1; pl.n_static = 0; pl.inside = 0; pl.pbase = pl.vbase = pl.rbase = 0;
    int gv;
    const bool ok = birth_src_voxel(d, s, src, pl.cx, pl.cy, pl.cz, gv);
    if (!ok) {   // not a birth source: no children, an empty plan
        if (l == 0) { s.plan[i] = pl; s.nstatic[i] = 0; s.plan_inside[i] = 0u; }
        return;
    }
    pl.gvox = gv;
    const int lvs = lv_of_g(d, gv);
    const bool own = lvs >= 0;   // (else: the source voxel belongs to another slab; that rank supplies n_static)
    const int lvq = own ? lvs : 0;
    // stage 2
    const int c = (int)(((long long)pb + 3 * min(l, nb - 1)) % fp.tab_n);
    const float t0 = s.p_tab[c], t1 = s.p_tab[(c + 1) % fp.tab_n], t2 = s.p_tab[(c + 2) % fp.tab_n];
    u64 mwd[2] = {0ull, 0ull};
    for (int e = 0; e < d.mw; ++e) mwd[e] = s.mask[(size_t)lvq * d.mw + e] & ~s.nbmask[(size_t)lvq * d.mw + e];  // 0.9<flag<14 :830
    // stage 3: this lane's slot of the source voxel (requested whether or not it is live: the address is valid) ...
    V2 pv[2];
    float pw[2];
    bool on[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int sl = e * 64 + l;
        on[e] = own && e < d.mw && sl < d.slots && ((mwd[e] >> l) & 1ull);
        const size_t idx = pidx(d, lvq, on[e] ? sl : 0);
        pv[e] = ld_vel(s, idx);
        pw[e] = s.w[idx];
    }
    // ... and this lane's child (:871-875)
    bool in = false;
    if (l < nb) {
        const int t = i * nb + l;
        const float x = pl.cx + t0, y = pl.cy + t1, z = pl.cz + t2;
        int gvc = 0, lvc = -1;
        if (voxel_of_lv(d, x, y, z, gvc, lvc)) {
            in = true;
            if (lvc >= 0) {                                      // children landing in another slab are inserted by their owner
                const int pos = atomicAdd(&vb_cnt[lvc], 1);
                if (pos < BIRTH_BUCKET_CAP) vb_idx[(size_t)lvc * BIRTH_BUCKET_CAP + pos] = t;
                else s.birth_ovf[atomicAdd(&s.fs->n_birth_ovf, 1)] = t;
            } else {
                lvc = -1;
            }
        }
        child[t] = make_float4(x, y, z, __int_as_float(lvc));
    }
    const unsigned inside = (unsigned)__ballot(in);
    // the split (:827-866)
    int n_static = 0;
    if (own) {
        float ws = 0.f, wsd = 0.f, wd = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (on[e]) {
                const float vabs = fabsf(pv[e].x) + fabsf(pv[e].y) + 0.f;  // vz == 0
                if (vabs < 0.1f) ws += pw[e]; else if (vabs < 0.5f) wsd += pw[e]; else wd += pw[e];
            }
        }
        ws = wave_sum(ws); wsd = wave_sum(wsd); wd = wave_sum(wd);
        const float total = ws + wd + wsd;
        const float m_s = __fdiv_rn(ws, total), m_d = __fdiv_rn(wd, total), m_sd = __fdiv_rn(wsd, total);
        const float p_s = (m_s + m_s + m_sd) * 0.5f;
        const float p_d = (m_d + m_d + m_sd) * 0.5f;
        const float p_s_n = __fdiv_rn(p_s, p_s + p_d);
        const float f = (float)fp.model_nb * p_s_n;
        const int ns = (f != f) ? 0 : (int)f;  // empty voxel -> NaN -> minimum applies (Appendix A-8)
        n_static = max(fp.min_static_nb, ns);
    }
    pl.n_static = n_static;
    if (l == 0) { s.plan[i] = pl; s.nstatic[i] = n_static; s.plan_inside[i] = inside; }
    if (src.intensity > 0.01f) {
        const int model_end = src.nx > -100.f ? fp.model_nb : n_static;  // :881
        auto below = [](int k) { return k >= 32 ? ~0u : ((1u << k) - 1u); };   // bits [0, k)
        const int lo = min(n_static, nb), mid = min(max(model_end, lo), nb);
        *cvr = make_int2(3 * __popc(inside & below(mid) & ~below(lo)), 3 * __popc(inside & below(nb) & ~below(mid)));
    }
}

// The sequential consumption order of the three random streams (:871-873 position table,
// :884-886 velocity table, :895-897 rand()) is reproduced with block-wide prefix sums over the
// source points: draws of point i start at cursor + (draws of all earlier points).
// k_birth_rank (one workgroup): rank of every valid source point among the valid ones ->
// first position-table cursor of the point (3 draws per child, always consumed, :871-873).
// Every thread owns BK consecutive points, so all loads are in flight together and one scan suffices.
// with_ck_sum: also reduce the per-pyramid 1/Ck sums into the birth normaliser (k_ck_sum's job),
// which saves a launch per frame.
__device__ __forceinline__ void ck_sum_block(const MapDims& d, const DevState& s, const FilterParams& fp, float* s_red) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    if (tid < 512) {
        for (int i = tid; i < d.np; i += 512) acc += s.part_inv[i];
        s_red[tid] = acc;
    }
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        s.fs->expected_newborn = frame_lambda(s, fp);
        s.fs->newborn_w = fp.nb_weight * s_red[0];  // :805
    }
}
__global__ void __launch_bounds__(1024) k_birth_rank(MapDims d, DevState s, FilterParams fp, int with_ck_sum) {
    __shared__ float s_red[512];
    if (with_ck_sum) ck_sum_block(d, s, fp, s_red);
    birth_rank_block(d, s, fp);
}
// whole frame: the split (one wave per source point) and the rank (one workgroup, independent of the split) in ONE launch
__global__ void __launch_bounds__(1024) k_birth_split_rank(MapDims d, DevState s, FilterParams fp, int with_ck_sum) {
    if (blockIdx.x == gridDim.x - 1) {
        __shared__ float s_red[512];
        if (with_ck_sum) ck_sum_block(d, s, fp, s_red);
        birth_rank_block(d, s, fp);
    } else birth_split_wave(d, s, fp, (int)(blockIdx.x * (1024 / WAVE) + threadIdx.x / WAVE));
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

__global__ void k_birth_children(MapDims d, DevState s, FilterParams fp, float4* __restrict__ child,
                                 int* __restrict__ vb_cnt, int* __restrict__ vb_idx) {
    birth_child_thread(d, s, fp, child, vb_cnt, vb_idx, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// k_birth_cursors (one workgroup): velocity-table and rand() cursors per source point (:884-886,:895-897)
__global__ void __launch_bounds__(1024) k_birth_cursors(MapDims d, DevState s, FilterParams fp) {
    const BirthView bv = birth_view(s);
    const int n_birth = bv.n;
    __shared__ int s_tmp[BK * 16 + 1];
    const int tid = threadIdx.x;
    const int v_cur = s.fs->v_cur, r_cur = s.fs->r_cur;
    const int nb = fp.nb_num;
    int run_v = 0, run_r = 0;
    for (int base = 0; base < n_birth; base += 1024 * BK) {
        int gvox[BK], nst[BK];
        unsigned inside[BK];
        float inten[BK], snx[BK];
#pragma unroll
        for (int j = 0; j < BK; ++j) {   // coalesced index, every load independent: one memory round trip
            const int i = base + j * 1024 + tid;
            gvox[j] = -1; nst[j] = 0; inside[j] = 0u; inten[j] = 0.f; snx[j] = 0.f;
            if (i < n_birth) {
                gvox[j] = s.plan[i].gvox; inside[j] = s.plan_inside[i];
                nst[j] = s.nstatic[i];
                const BirthSrc b = birth_at(bv, i);
                inten[j] = b.intensity; snx[j] = b.nx;
            }
        }
        int cv[BK], cr[BK];
#pragma unroll
        for (int j = 0; j < BK; ++j) {
            cv[j] = 0; cr[j] = 0;
            if (gvox[j] >= 0 && inten[j] > 0.01f) {
                const int model_end = snx[j] > -100.f ? fp.model_nb : nst[j];  // :881
                for (int k = nst[j]; k < nb; ++k) {
                    if (!((inside[j] >> k) & 1u)) continue;
                    if (k < model_end) cv[j] += 3; else cr[j] += 3;
                }
            }
        }
        const int totv = block_excl_scan_multi<BK>(cv, s_tmp);
        const int totr = block_excl_scan_multi<BK>(cr, s_tmp);
#pragma unroll
        for (int j = 0; j < BK; ++j) {
            if (gvox[j] >= 0) {
                BirthPlan* pl = &s.plan[base + j * 1024 + tid];
                pl->n_static = nst[j];
                pl->vbase = (int)(((long long)v_cur + run_v + cv[j]) % fp.tab_n);
                pl->rbase = (int)(((long long)r_cur + run_r + cr[j]) % max(fp.rtab_n, 1));
            }
        }
        run_v += totv; run_r += totr;
    }
    if (tid == 0) {
        s.fs->v_cur = (int)(((long long)v_cur + run_v) % fp.tab_n);
        s.fs->r_cur = (int)(((long long)r_cur + run_r) % max(fp.rtab_n, 1));
    }
}

// generateRandomFloat :1551-1553 fed from the rand() table
__device__ __forceinline__ float rand_float(const DevState& s, const FilterParams& fp, int c, float lo, float hi) {
    const int r = s.r_tab[c % max(fp.rtab_n, 1)];
    return lo + __fdiv_rn((float)r, __fdiv_rn((float)2147483647, (hi - lo)));
}

// k_birth_insert: one thread per (source point, child): velocity by branch (:877-903); vz = 0
// (:905-907); weight = the global newborn weight (:909); newborn flag (= nbmask bit).
// FUSED: the cursors of the point's draws (k_birth_cursors' job) are computed here, from the draw counts k_birth_split_cksum_cvr
// left: the sums of the split's workgroups before this block's first point (a block-wide reduction over at most a few hundred
// entries) plus the points in between.  Block 0 writes the new cursors; everybody reads the copies taken before the births.
template <bool FUSED>
__global__ void __launch_bounds__(256) k_birth_insert(MapDims d, DevState s, FilterParams fp, const float4* __restrict__ child,
                               const int* __restrict__ vb_cnt, const int* __restrict__ vb_idx, int* __restrict__ part_birth,
                               const u64* __restrict__ nbsnap, int wg_off) {
    // (DSPMAP_P_ESTIMATOR_QUEUE) the frame's first birth kernel gave up waiting for the estimator's queue: the birth cloud is not complete
    // and nothing of it is inserted -- the frame ends without a birth stage, the host fails its next call (dspmap_check_estimator_queue)
    if (s.xq && s.fpar->from_ring && __hip_atomic_load(s.xq + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)(s.fpar->ring_pos + 1u)) {
        if (threadIdx.x < 2) part_birth[blockIdx.x * 2 + threadIdx.x] = 0;
        return;
    }
    const BirthView bv = birth_view(s);
    const int n_birth = bv.n;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = fp.nb_num;
    const int i = t / nb, k = t - i * nb;
    bool born = false, dropped = false;
    int f_vbase = 0, f_rbase = 0;
    if (FUSED) {
        __shared__ int s_part[4][4];
        const int tid = (int)threadIdx.x;
        const int i_first = (int)((blockIdx.x * blockDim.x) / (unsigned)nb);
        const int G = min(i_first, n_birth) >> 4;     // workgroups of the split (16 points each) wholly before this block's first point
        const int ngr = (n_birth + 15) >> 4;
        int acc[4] = {0, 0, 0, 0};                    // {velocity draws before, rand() draws before, all velocity draws, all rand() draws}
        for (int g = tid; g < ngr; g += 256) {
            const int2 w = s.birth_cvr[wg_off + g];
            acc[2] += w.x; acc[3] += w.y;
            if (g < G) { acc[0] += w.x; acc[1] += w.y; }
        }
        int2 loc = make_int2(0, 0);
        for (int j = G << 4; j < min(i, n_birth); ++j) { const int2 c = s.birth_cvr[j]; loc.x += c.x; loc.y += c.y; }   // (at most 15 + 256 / nb points)
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q] = wave_sum_i(acc[q]); if (lane_id() == 0) s_part[tid >> 6][q] = acc[q]; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = (s_part[0][q] + s_part[1][q]) + (s_part[2][q] + s_part[3][q]);
        const int v0 = s.fs->v_cur_in, r0 = s.fs->r_cur_in;
        f_vbase = (int)(((long long)v0 + acc[0] + loc.x) % fp.tab_n);
        f_rbase = (int)(((long long)r0 + acc[1] + loc.y) % max(fp.rtab_n, 1));
        if (t == 0) {
            s.fs->v_cur = (int)(((long long)v0 + acc[2]) % fp.tab_n);
            s.fs->r_cur = (int)(((long long)r0 + acc[3]) % max(fp.rtab_n, 1));
        }
    }
    if (bv.live && t == 0) s.fs->stale_n = n_birth;
    if (i < n_birth) {
        // this kernel is a chain of dependent loads: everything that only needs (i, t) is requested at once, then
        // everything that only needs the destination voxel
        const BirthPlan pl = s.plan[i];
        const unsigned pl_inside = s.plan_inside[i];
        const float4 ch = child[t];          // (garbage for children that were not generated: only used under the tests below)
        const BirthSrc src = birth_at(bv, i);
        // a non-empty view's synthesised cloud is kept for the frames whose view is empty (:1379-1381)
        if (bv.live && k == 0) const_cast<BirthSrc*>(bv.stored)[i] = src;
        const float newborn_w = s.fs->newborn_w;
        if (pl.gvox >= 0 && ((pl_inside >> k) & 1u)) {
            const int lv = __float_as_int(ch.w);
            if (lv >= 0) {
                const int n = min(vb_cnt[lv], BIRTH_BUCKET_CAP);
                u64 occ[2];
                // free = not live and not a newborn of an EARLIER call (flag 15, :1184-1185); this call's own newborns go to
                // nbmask while the kernel runs, so the pre-birth word is taken from the snapshot
                for (int e = 0; e < d.mw; ++e) occ[e] = s.mask[(size_t)lv * d.mw + e] | (nbsnap ? nbsnap[(size_t)lv * d.mw + e] : 0ull);
                float vx = 0.f, vy = 0.f;
                if (k >= pl.n_static && src.intensity > 0.01f) {
                    const int model_end = src.nx > -100.f ? fp.model_nb : pl.n_static;
                    const unsigned lo_mask = (k >= 32 ? ~0u : ((1u << k) - 1u)) & ~((pl.n_static >= 32) ? ~0u : ((1u << pl.n_static) - 1u));
                    const unsigned before = pl_inside & lo_mask;  // inside children in [n_static, k)
                    if (k < model_end) {
                        const int rank = __popc(before);
                        const int cv = (int)(((long long)(FUSED ? f_vbase : pl.vbase) + 3 * rank) % fp.tab_n);
                        vx = src.nx + 4 * s.v_tab[cv];                        // :884
                        vy = src.ny + 4 * s.v_tab[(cv + 1) % fp.tab_n];      // :885
                    } else {
                        const unsigned model_bits = (model_end >= 32) ? ~0u : ((1u << model_end) - 1u);
                        const int rank = __popc(before & ~model_bits);
                        const int cr = (FUSED ? f_rbase : pl.rbase) + 3 * rank;
                        vx = rand_float(s, fp, cr, -1.5f, 1.5f);              // :895
                        vy = rand_float(s, fp, cr + 1, -1.5f, 1.5f);          // :896
                    }
                }
                // rank among this voxel's children, by birth index
                int rank = 0;
                const int n_all = vb_cnt[lv];
                bool recorded = n_all > BIRTH_BUCKET_CAP;   // (then every child of the voxel is in the bucket or in the overflow list)
                const int4* bl4 = reinterpret_cast<const int4*>(vb_idx + (size_t)lv * BIRTH_BUCKET_CAP);
                for (int j = 0; j < n; j += 16) {  // 4 x 16-byte loads in flight per step (the bucket row is 512 B, always readable); entries beyond n ignored
                    int4 v4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v4[u] = bl4[(j >> 2) + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int o[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool valid = j + u * 4 + q < n;
                            rank += (valid && o[q] < t) ? 1 : 0;
                            recorded |= valid && (o[q] == t);
                        }
                    }
                }
                if (n_all > BIRTH_BUCKET_CAP) {
                    // more children than the bucket holds (a dense cloud into one voxel): the bucket kept whichever arrived first,
                    // the rest are in the overflow list -- ranking over both gives the reference's order whatever the arrival order
                    const int n_ovf = s.fs->n_birth_ovf;
                    for (int j = 0; j < n_ovf; ++j) {
                        const int o = s.birth_ovf[j];
                        rank += (o < t && __float_as_int(child[o].w) == lv) ? 1 : 0;
                    }
                }
                int sl = -1;
                if (recorded) {  // rank-th free slot of the pre-birth occupancy
                    int r = rank;
                    for (int e = 0; e < d.mw && sl < 0; ++e) {
                        const int nbits = min(64, d.slots - e * 64);
                        const u64 valid = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
                        u64 fr = ~occ[e] & valid;
                        const int nf = (int)__popcll(fr);
                        if (r >= nf) { r -= nf; continue; }
                        for (int q = 0; q < r; ++q) fr &= fr - 1ull;
                        sl = e * 64 + (__ffsll((long long)fr) - 1);
                    }
                }
                if (sl >= 0) {
                    const size_t idx = pidx(d, lv, sl);
                    st_pos(s, idx, ch.x, ch.y, ch.z);
                    st_vel(s, idx, vx, vy);
                    note_speed(s, vx, vy);
                    if (vx != 0.f || vy != 0.f) s.tile_moving[lv >> 6] = 1;   // (a newborn of a matched cluster: the tile's velocity rows count again)
                    s.w[idx] = newborn_w;
                    atomicOr(&s.nbmask[(size_t)lv * d.mw + (sl >> 6)], 1ull << (sl & 63));  // flag 15
                    s.tile_live[lv >> 6] = 1;   // (the tile may have been empty: the sweeps must visit it again)
                    if (s.vis_bits) {   // ... this frame's resampling too (one atomic per tile and frame, not per newborn: the others find the bit)
                        const int tl = lv >> 6;
                        if (!((__hip_atomic_load(&s.vis_bits[tl >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (tl & 31)) & 1u)) atomicOr(&s.vis_bits[tl >> 5], 1u << (tl & 31));
                    }
                    born = true;
                } else {
                    dropped = true;
                }
            }
        }
    }
    // per-block partial counts (a same-address global atomic per wave would serialise, ~12 ns each)
    __shared__ int s_bc[2];
    if (threadIdx.x < 2) s_bc[threadIdx.x] = 0;
    __syncthreads();
    const u64 bb = __ballot(born), bd = __ballot(dropped);
    if (lane_id() == 0) {
        if (bb) atomicAdd(&s_bc[0], (int)__popcll(bb));
        if (bd) atomicAdd(&s_bc[1], (int)__popcll(bd));
    }
    __syncthreads();
    if (threadIdx.x < 2) part_birth[blockIdx.x * 2 + threadIdx.x] = s_bc[threadIdx.x];
}

// --------------------------------------------------------------------------
// Readout, :385-438.  Occupied voxels in ascending index order (stable compaction).
// --------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_occ_count(MapDims d, DevState s, float thr) {
    __shared__ int s_c[4];
    const int tv = blockIdx.x * 256 + threadIdx.x;   // the slab's voxels in the reference's index order, whatever the storage order
    const bool occ = tv < d.v_true && s.res4[lv_of_true(d, tv)].x > thr;
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
    const int tv = blockIdx.x * 256 + threadIdx.x;
    const bool occ = tv < d.v_true && s.res4[lv_of_true(d, tv)].x > thr;
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
            const int index = tv + d.v_base;
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

// ==========================================================================
// launchers
// ==========================================================================


void launch_frame_setup(const LaunchCtx& c, bool reset_obs) {
    const int flags = RESET_PLANES | RESET_PRED | (reset_obs ? RESET_OBS : 0);
    const int n = c.d.np * DSP_OBS_CAP;
    const int grid = reset_obs ? (n + 1023) / 1024 : 1;
    hipLaunchKernelGGL(k_reset, dim3(grid), dim3(1024), 0, c.stream, c.d, c.s, flags);
}

void launch_setup_and_bin(const LaunchCtx& c, int n_pts_grid, bool gather, const FrameParams* ring, int ring_mask) {   // whole frame: launch_frame_setup(c, true) + launch_obs_bin
    const int grid = n_pts_grid > 0 ? (n_pts_grid + 255) / 256 : 1;
    const int nbits = c.tile_bits ? (c.k.ntiles + 255) / 256 : 0;   // the bitmap rebuild rides along (sparse whole frames)
    hipLaunchKernelGGL(k_obs_points<true>, dim3(grid + nbits), dim3(256), 0, c.stream, c.d, c.s, ring, ring_mask, c.tile_bits ? grid : -1);
    if (gather) hipLaunchKernelGGL(k_obs_gather, dim3(c.d.np), dim3(WAVE), 0, c.stream, c.d, c.s);
}
void launch_obs_bin(const LaunchCtx& c, int n_pts_grid) {
    if (n_pts_grid > 0) hipLaunchKernelGGL(k_obs_points<false>, dim3((n_pts_grid + 255) / 256), dim3(256), 0, c.stream, c.d, c.s, (const FrameParams*)nullptr, 0, -1);
    hipLaunchKernelGGL(k_obs_gather, dim3(c.d.np), dim3(WAVE), 0, c.stream, c.d, c.s);
}

// map corner farther than 12 cull radii: most pairs are far (28 % at 8 radii, 72 % at 16, measured) -> k_weight<true>
static bool weight_culls(const LaunchCtx& c) { return (float)PS_NBK / c.d.rng_inv_bw > 12.f * c.fp.cull_r; }
void launch_place_fix(const LaunchCtx& c) {   // stage API: after launch_pyr_prepare (a frame's k_ck_partial launch carries the pass)
    hipLaunchKernelGGL(k_place_fix, dim3(PF_WG), dim3(256), PF_LDS_BYTES, c.stream, c.d, c.s, c.k.in_rec, c.k.omask, reinterpret_cast<const int*>(c.k.mv_rec));
}
void launch_pyr_prepare(const LaunchCtx& c) {
    hipLaunchKernelGGL(k_pyr_prepare, dim3(c.d.np + 1), dim3(1024), 0, c.stream, c.d, c.s, c.k.ck_items, c.k.wu_items, c.k.n_items, c.k.nb_tab);
}
void launch_ck_partial(const LaunchCtx& c, bool prepared, bool with_fix) {
    if (!prepared) launch_pyr_prepare(c);
    hipLaunchKernelGGL(k_ck_partial, dim3(4096 + PF_WG), dim3(CK_TPB), (sizeof(float4) + sizeof(int)) * (size_t)c.d.nbins * DSP_OBS_CAP, c.stream, c.d, c.s, c.fp, c.k.ck_items, c.k.n_items, c.k.nb_tab,
                       c.k.in_rec, c.k.omask, reinterpret_cast<const int*>(c.k.mv_rec), with_fix ? 1 : 0);
}
void launch_ck_finalize(const LaunchCtx& c) {  // after launch_weight_update: reduces the per-pyramid 1/Ck sums
    hipLaunchKernelGGL(k_ck_sum, dim3(1), dim3(512), 0, c.stream, c.d, c.s, c.fp);
}
void launch_weight_update(const LaunchCtx& c) {  // after launch_ck_partial (which also builds the item lists)
    const bool skip = weight_culls(c);
    const size_t lds = (sizeof(float4) + sizeof(float)) * (size_t)c.d.nbins * DSP_OBS_CAP;
    if (skip) hipLaunchKernelGGL(k_weight<true>, dim3(4096), dim3(WU_TPB), lds, c.stream, c.d, c.s, c.fp, c.k.wu_items, c.k.n_items, c.k.nb_tab);
    else hipLaunchKernelGGL(k_weight<false>, dim3(4096), dim3(WU_TPB), lds, c.stream, c.d, c.s, c.fp, c.k.wu_items, c.k.n_items, c.k.nb_tab);
}

// n_birth_grid sizes the launches (>= the frame's n_birth, which the kernels read from FrameParams)
void launch_birth_split(const LaunchCtx& c, int n_birth_grid) {
    if (n_birth_grid <= 0) return;
    hipLaunchKernelGGL(k_birth_split, dim3((n_birth_grid + 3) / 4), dim3(256), 0, c.stream, c.d, c.s, c.fp);
}
// Zeroing kernel instead of hipMemsetAsync: a memset node inside the captured frame graph
// faulted on ROCm 7.2 whenever other streams were busy between replays; kernel nodes do not.
__global__ void k_zero_i32(int* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
__global__ void k_copy_u64(const u64* __restrict__ src, u64* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// k_birth_insert, preceded by the snapshot of the newborn bits when earlier newborns may exist (c.k.nbsnap != nullptr)
static void launch_insert(const LaunchCtx& c, unsigned gb, bool fused = false) {
    if (c.k.nbsnap) {
        const size_t W = (size_t)c.d.v_loc * c.d.mw;
        hipLaunchKernelGGL(k_copy_u64, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, c.stream, c.s.nbmask, c.k.nbsnap, W);
    }
    if (fused) hipLaunchKernelGGL(k_birth_insert<true>, dim3(gb), dim3(256), 0, c.stream, c.d, c.s, c.fp, c.k.child, c.k.vb_cnt, c.k.vb_idx, c.k.part_birth, c.k.nbsnap, c.birth_cap);
    else hipLaunchKernelGGL(k_birth_insert<false>, dim3(gb), dim3(256), 0, c.stream, c.d, c.s, c.fp, c.k.child, c.k.vb_cnt, c.k.vb_idx, c.k.part_birth, c.k.nbsnap, c.birth_cap);
}
static void birth_children_insert(const LaunchCtx& c, int n_birth_grid, bool in_frame, bool all_static);
void launch_birth_plan_insert(const LaunchCtx& c, int n_birth_grid, bool in_frame, bool all_static) {
    if (n_birth_grid <= 0) return;
    // in_frame: k_resample, which follows, zeroes the buckets again, and k_birth_rank also does k_ck_sum's job.
    hipLaunchKernelGGL(k_birth_rank, dim3(1), dim3(1024), 0, c.stream, c.d, c.s, c.fp, in_frame ? 1 : 0);
    birth_children_insert(c, n_birth_grid, in_frame, all_static);
}
static void birth_children_insert(const LaunchCtx& c, int n_birth_grid, bool in_frame, bool all_static) {
    const long long total = (long long)n_birth_grid * c.fp.nb_num;
    const unsigned gb = (unsigned)((total + 255) / 256);
    // invariant: the per-voxel buckets (vb_cnt) are all zero whenever no birth stage is in progress.
    hipLaunchKernelGGL(k_birth_children, dim3(gb), dim3(256), 0, c.stream, c.d, c.s, c.fp, c.k.child, c.k.vb_cnt, c.k.vb_idx);
    // all_static (every birth source has intensity 0, the synthesized cloud): no child draws from the velocity or
    // rand() streams (:877-903), so the cursor kernel has nothing to compute and k_birth_insert never reads its output
    if (!all_static) hipLaunchKernelGGL(k_birth_cursors, dim3(1), dim3(1024), 0, c.stream, c.d, c.s, c.fp);
    launch_insert(c, gb);
    if (!in_frame) hipLaunchKernelGGL(k_zero_i32, dim3((c.d.v_loc + 255) / 256), dim3(256), 0, c.stream, c.k.vb_cnt, c.d.v_loc);
}
// split (one wave per source point) and the reduction of the 1/Ck sums (one workgroup) in one launch
__global__ void __launch_bounds__(1024) k_birth_split_cksum(MapDims d, DevState s, FilterParams fp) {
    if (blockIdx.x == gridDim.x - 1) {
        __shared__ float s_red[512];
        ck_sum_block(d, s, fp, s_red);
    } else birth_split_wave(d, s, fp, (int)(blockIdx.x * (1024 / WAVE) + threadIdx.x / WAVE));
}
// ... and, for the insertion kernel that computes the velocity-table / rand() cursors itself (one launch less: a kernel of this
// chain costs ~5 us however little it does): every point's draw counts, their sums per workgroup (16 points), and a copy of the
// two cursors as they stand before the frame's births
// CHILDREN (frame with the device estimator on a map without the split placement): the point's wave also generates its newborn
// children first (k_birth_children's job; the "inside the map" bits are a ballot) -- the estimator's branch of the frame, the
// longer one at the metric's size, ends with k_ve_clusters instead of a third kernel
// one workgroup's share (16 source points, or -- the last workgroup -- the 1/Ck reduction and the cursor copies); bx = the workgroup's index
template <bool CHILDREN>
__device__ __forceinline__ void birth_split_cvr_block(const MapDims& d, const DevState& s, const FilterParams& fp, int wg_off, float4* __restrict__ child,
                                                      int* __restrict__ vb_cnt, int* __restrict__ vb_idx, int bx, int2* s_c, float* s_red) {
    if (bx == (int)gridDim.x - 1) {
        ck_sum_block(d, s, fp, s_red);
        if (threadIdx.x == 0) { s.fs->v_cur_in = s.fs->v_cur; s.fs->r_cur_in = s.fs->r_cur; }
        return;
    }
    const int wave = (int)threadIdx.x / WAVE;
    const int i = bx * (1024 / WAVE) + wave;
    int2 c;
    if (CHILDREN) birth_point_wave(d, s, fp, i, child, vb_cnt, vb_idx, &c);
    else birth_split_wave(d, s, fp, i, &c);
    if (lane_id() == 0) { s_c[wave] = c; if (i < birth_view(s).n) s.birth_cvr[i] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int2 t = make_int2(0, 0);
        for (int w = 0; w < 1024 / WAVE; ++w) { t.x += s_c[w].x; t.y += s_c[w].y; }
        s.birth_cvr[wg_off + bx] = t;
    }
}
// DSPMAP_P_ESTIMATOR_QUEUE (s.xq set): the estimator ran on a queue of its own; its birth cloud, rank and cursors are complete once xq[1] has
// reached this frame's number.  Normally that was tens of microseconds ago and every workgroup goes ahead at its first look.  When it is
// NOT there yet, the machine must not fill up with waiting workgroups (the estimator's own kernels need room to be scheduled -- several
// maps on one GPU did deadlock that way): only workgroup 0 waits; every other workgroup that finds the word missing notes its index in a
// list and leaves, and workgroup 0 does the listed shares itself once the word has arrived and every workgroup has decided.  The shares
// are independent of who runs them (a point's children land in per-voxel buckets that the insertion ranks), so the result is the same.
// Control words behind the two hand-over words, zeroed by this frame's k_predict: xq[5] listed workgroups, XQ_NDEC counters of the
// workgroups that have decided (spread over as many 256-byte blocks: 385 atomics on ONE address serialise in memory, 17 us on the
// metric's frame), then the list.
template <bool CHILDREN>
__global__ void __launch_bounds__(1024) k_birth_split_cksum_cvr(MapDims d, DevState s, FilterParams fp, int wg_off, float4* __restrict__ child,
                                                                int* __restrict__ vb_cnt, int* __restrict__ vb_idx) {
    __shared__ int2 s_c[1024 / WAVE];
    __shared__ float s_red[512];
    __shared__ int s_flag;
    if (s.xq) {
        const int want = (int)(s.fpar->ring_pos + 1u);
        // (xq[10], test hook DSPMAP_XQ_TEST_DELAY_US: in every third frame every workgroup takes the cloud for unfinished at this first look --
        // the deferral path below then runs in a known set of frames, whatever the two streams' timing; workgroup 0's wait finds the truth)
        if (threadIdx.x == 0) s_flag = (__hip_atomic_load(s.xq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want >= 0 && !(s.xq[10] != 0 && want % 3 == 2)) ? 1 : 0;
        __syncthreads();
        const bool ready = s_flag != 0;
        if (blockIdx.x != 0) {
            if (threadIdx.x == 0) {
                // (relaxed atomics at agent scope are performed in memory, in this lane's program order: the list entry is there before the
                // count says so -- a release would write the XCD's L2 back in every one of the kernel's workgroups, 25 us on the metric's frame)
                if (!ready) {
                    const int k = __hip_atomic_fetch_add(s.xq + 5, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(s.xq + XQ_LIST + k, (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (acknowledged by memory before the count below is sent)
                }
                __hip_atomic_fetch_add(s.xq + XQ_DEC + ((int)blockIdx.x & (XQ_NDEC - 1)) * 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!ready) return;
            // found ready at the first look: the word may have arrived after this kernel's launch-time invalidate -- one lane's acquire
            // (L1 invalidate, no write-back) before anybody of the workgroup reads what the other queue wrote
            if (threadIdx.x == 0) xq_acquire();
            __syncthreads();
        } else {
            // workgroup 0: waits if it has to, does its own share, and then -- ALWAYS: another workgroup may have looked before the word
            // arrived although this one looked after -- makes sure every workgroup has decided and does the listed shares
            __syncthreads();
            if (threadIdx.x == 0 && !ready) {
                xq_wait(s.xq + 1, want, s.hint_out + 3);
                // gave up: the birth cloud is NOT complete -- this frame's insertion is called off (k_birth_insert looks at the word), the
                // host fails its next call and goes on without the estimator's queue (dspmap_check_estimator_queue)
                if (__hip_atomic_load(s.xq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want < 0)
                    __hip_atomic_store(s.xq + 8, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            birth_split_cvr_block<CHILDREN>(d, s, fp, wg_off, child, vb_cnt, vb_idx, 0, s_c, s_red);
            __syncthreads();
            if (threadIdx.x < XQ_NDEC) {   // (one wave, the counters' loads in flight together)
                const long long t0 = wall_clock64();
                for (;;) {
                    int dec = __hip_atomic_load(s.xq + XQ_DEC + (int)threadIdx.x * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int o = 32; o > 0; o >>= 1) dec += __shfl_xor(dec, o, WAVE);
                    if (dec >= (int)gridDim.x - 1) break;
                    if (wall_clock64() - t0 > 20000000ll) { if (threadIdx.x == 0) s.hint_out[3] = want; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
                if (threadIdx.x == 0) {
                    s_flag = __hip_atomic_load(s.xq + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!ready || s_flag) { s.xq[6] += 1; s.xq[7] += s_flag; }   // diagnostics (dspmap_debug_estimator_queue): frames in which somebody had to wait, shares done for others
                }
            }
            __syncthreads();
            const int n_listed = s_flag;
            for (int k = 0; k < n_listed; ++k) {
                __syncthreads();
                const int bx = __hip_atomic_load(s.xq + XQ_LIST + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                birth_split_cvr_block<CHILDREN>(d, s, fp, wg_off, child, vb_cnt, vb_idx, bx, s_c, s_red);
            }
            return;
        }
    }
    birth_split_cvr_block<CHILDREN>(d, s, fp, wg_off, child, vb_cnt, vb_idx, (int)blockIdx.x, s_c, s_red);
}
// split-phase (multi-GPU) frame: rank + children as soon as the prediction is queued (they only need the birth cloud; the
// driver's host synchronisation for the neighbour exchange leaves the GPU idle right there), cursors + insert at the end
void launch_birth_early(const LaunchCtx& c, int n_birth_grid, bool with_rank) {   // with_rank = false: it rode on k_predict's launch
    if (n_birth_grid <= 0) return;
    const unsigned gb = (unsigned)(((long long)n_birth_grid * c.fp.nb_num + 255) / 256);
    if (with_rank) hipLaunchKernelGGL(k_birth_rank, dim3(1), dim3(1024), 0, c.stream, c.d, c.s, c.fp, 0);
    hipLaunchKernelGGL(k_birth_children, dim3(gb), dim3(256), 0, c.stream, c.d, c.s, c.fp, c.k.child, c.k.vb_cnt, c.k.vb_idx);
}
void launch_birth_finish(const LaunchCtx& c, int n_birth_grid, bool all_static) {
    if (n_birth_grid <= 0) return;
    const unsigned gb = (unsigned)(((long long)n_birth_grid * c.fp.nb_num + 255) / 256);
    if (!all_static) hipLaunchKernelGGL(k_birth_cursors, dim3(1), dim3(1024), 0, c.stream, c.d, c.s, c.fp);
    launch_insert(c, gb);
}
void launch_birth_split_cksum(const LaunchCtx& c, int n_birth_grid) {
    hipLaunchKernelGGL(k_birth_split_cksum, dim3((n_birth_grid + 15) / 16 + 1), dim3(1024), 0, c.stream, c.d, c.s, c.fp);
}
void launch_birth_late(const LaunchCtx& c, int n_birth_grid, bool all_static, bool with_children) {
    if (n_birth_grid <= 0) return;
    const unsigned gb = (unsigned)(((long long)n_birth_grid * c.fp.nb_num + 255) / 256);
    if (all_static) {   // (no child draws from the velocity or rand() streams: no cursors)
        launch_birth_split_cksum(c, n_birth_grid);
        launch_insert(c, gb);
    } else {            // the children are done (they rode on earlier launches): the insertion computes its cursors itself
        if (with_children) hipLaunchKernelGGL(k_birth_split_cksum_cvr<true>, dim3((n_birth_grid + 15) / 16 + 1), dim3(1024), 0, c.stream, c.d, c.s, c.fp, c.birth_cap, c.k.child, c.k.vb_cnt, c.k.vb_idx);
        else hipLaunchKernelGGL(k_birth_split_cksum_cvr<false>, dim3((n_birth_grid + 15) / 16 + 1), dim3(1024), 0, c.stream, c.d, c.s, c.fp, c.birth_cap, c.k.child, c.k.vb_cnt, c.k.vb_idx);
        launch_insert(c, gb, true);
    }
}
void launch_birth(const LaunchCtx& c, int n_birth_grid, bool in_frame, bool all_static) {
    if (n_birth_grid <= 0) return;
    hipLaunchKernelGGL(k_birth_split_rank, dim3((n_birth_grid + 15) / 16 + 1), dim3(1024), 0, c.stream, c.d, c.s, c.fp, in_frame ? 1 : 0);
    birth_children_insert(c, n_birth_grid, in_frame, all_static);
}

// host readback of the synthesised birth cloud (dspmap_get_birth_cloud): entry i of the frame's BirthView
__global__ void k_birth_materialize(DevState s, BirthSrc* __restrict__ out, int cap, int* __restrict__ n_out) {
    const BirthView bv = birth_view(s);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_out = bv.n;
    if (i < bv.n && i < cap) out[i] = birth_at(bv, i);
}
void launch_birth_materialize(const LaunchCtx& c, BirthSrc* out, int cap, int* n_out) {
    hipLaunchKernelGGL(k_birth_materialize, dim3((cap + 255) / 256), dim3(256), 0, c.stream, c.s, out, cap, n_out);
}

void launch_scan_blocks(const LaunchCtx& c, int nblk) {
    hipLaunchKernelGGL(k_occ_scan, dim3(1), dim3(1024), 0, c.stream, c.s, nblk);
}
void launch_occupied_compact(const LaunchCtx& c, float thr) {
    const int nblk = (c.d.v_true + 255) / 256;
    hipLaunchKernelGGL(k_occ_count, dim3(nblk), dim3(256), 0, c.stream, c.d, c.s, thr);
    hipLaunchKernelGGL(k_occ_scan, dim3(1), dim3(1024), 0, c.stream, c.s, nblk);
    hipLaunchKernelGGL(k_occ_emit, dim3(nblk), dim3(256), 0, c.stream, c.d, c.s, thr, c.d.v_true);
}
// fut_out[v][t] = fut[t][v] + fut_stat[v]: the caller's [V][T] layout from the horizon-major accumulators and the
// static-particle mass (the same for every horizon).  Pure function of the accumulators: callable any number of times.
// (fut_out and res_out are in the reference's voxel order, the accumulators in storage order)
__global__ void k_future_combine(MapDims d, DevState s) {
    const int tv = blockIdx.x * blockDim.x + threadIdx.x;
    if (tv >= d.v_true) return;
    const int lv = lv_of_true(d, tv);
    const float st = s.fut_stat[lv];
    for (int t = 0; t < d.T; ++t) s.fut_out[(size_t)tv * d.T + t] = fut_value(s.fut[(size_t)t * d.v_loc + lv]) + st;
}
void launch_future_combine(const LaunchCtx& c) {
    hipLaunchKernelGGL(k_future_combine, dim3((c.d.v_true + 255) / 256), dim3(256), 0, c.stream, c.d, c.s);
}
// voxels_objects_number[v][0..3] (:118-120) in the reference's voxel order (cube storage only: otherwise res4 IS in that order)
__global__ void k_results_true(MapDims d, DevState s, float4* __restrict__ out) {
    const int tv = blockIdx.x * blockDim.x + threadIdx.x;
    if (tv < d.v_true) out[tv] = s.res4[lv_of_true(d, tv)];
}
void launch_results_true(const LaunchCtx& c, float4* out) {
    hipLaunchKernelGGL(k_results_true, dim3((c.d.v_true + 255) / 256), dim3(256), 0, c.stream, c.d, c.s, out);
}
void launch_clear_future(const LaunchCtx& c) {
    (void)hipMemsetAsync(c.s.fut, 0, sizeof(u64) * (size_t)c.d.v_loc * c.d.T, c.stream);
    (void)hipMemsetAsync(c.s.fut_stat, 0, sizeof(float) * (size_t)c.d.v_loc, c.stream);
}
