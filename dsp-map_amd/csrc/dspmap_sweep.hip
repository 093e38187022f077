// dspmap_sweep.hip -- the three HBM-streaming sweeps over the particle store
// (prediction, mover re-binning, occupancy/rollout/resampling) + state helpers.
//
// Layout: voxels are grouped in TILES of 64; inside a tile the
// particle fields are stored slot-major, index = ((lv>>6)*SLOTS + slot)*64 + (lv&63).
// ONE LANE OWNS ONE VOXEL, one wave owns one tile, and the wave walks the slot
// rows that are live anywhere in the tile:
//   * every row access is one fully coalesced 256-byte wave transaction;
//   * slots fill from the bottom (first-free-slot rule), so rows above the
//     tile's occupancy are never touched: HBM traffic follows the LIVE particle
//     count, not the 2x capacity the reference's dense sweeps pay for
//     (include/dsp_dynamic.h:645-647,929-938);
//   * all per-voxel reductions (mass, mean velocity, resampling thresholds) are
//     sequential per lane in slot order -- the reference's own operation order
//     (:938-1053), so sums, thresholds and copy placement match it bit for bit,
//     and at saturation every lane is busy (a wave-per-voxel mapping keeps 24 of
//     64 lanes busy at 24 particles/voxel).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <climits>
#include <cmath>
#include <cstring>
#include "dspmap_device.h"
#include "dspmap_kernels.h"
#include "dspmap_birth.h"

__device__ __forceinline__ u64 valid_bits(const MapDims& d, int e) {
    const int nbits = min(64, d.slots - e * 64);
    return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
}

// rows of a tile are processed in batches of RB: all loads of a batch are issued before any is
// consumed, so a wave pays one memory round trip per batch instead of one per row
#ifndef RB
#define RB 2
#endif
#define TB 2      // records per thread and step in the tails of k_predict / k_place
#ifndef DENSE_MAX
#define DENSE_MAX 2048
#endif
// DENSE_MAX: live cells up to which k_predict may take the dense-lane path
#define HIST_NP 1024   // pyramids up to which k_predict ranks its stayers with an LDS histogram
#ifndef LSTG
#define LSTG 96
#endif
// LSTG: records of each kind a workgroup of k_predict notes in LDS before spilling to HBM

// Deferred wave-aggregated append of up to N items per lane into counted lists
// (pyramids_in_fov registration :1245-1254, mover routing): one global atomic per DISTINCT key of
// the batch, all atomics in flight together, return values collected afterwards.
// key[r] < 0 = no item.  pos[r] receives the list position.  Wave-uniform control flow only.
template <int N>
__device__ __forceinline__ void batch_append(int* cnt, const int (&key)[N], int (&pos)[N]) {
    const int l = lane_id();
    u64 todo[N];
    bool any = false;
#pragma unroll
    for (int r = 0; r < N; ++r) { todo[r] = __ballot(key[r] >= 0); any |= todo[r] != 0ull; pos[r] = -1; }
    while (any) {
        int nkeys = 0, mykey = -1, mybase = 0;
        while (any && nkeys < 64) {
            int k = -1;
#pragma unroll
            for (int r = 0; r < N; ++r)
                if (k < 0 && todo[r]) k = __builtin_amdgcn_readlane(key[r], __ffsll((long long)todo[r]) - 1);
            int c = 0;
            any = false;
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const u64 g = __ballot(key[r] == k);
                c += (int)__popcll(g);
                todo[r] &= ~g;
                any |= todo[r] != 0ull;
            }
            if (l == nkeys) { mykey = k; mybase = atomicAdd(&cnt[k], c); }
            ++nkeys;
        }
        for (int j = 0; j < nkeys; ++j) {
            const int k = __builtin_amdgcn_readlane(mykey, j);
            int run = __builtin_amdgcn_readlane(mybase, j);
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const u64 g = __ballot(key[r] == k);
                if (key[r] == k) pos[r] = run + (int)__popcll(g & lanemask_lt());
                run += (int)__popcll(g);
            }
        }
    }
}

// wave-aggregated reservation of one entry per flagged lane in an LDS counter (returns -1 when not flagged)
__device__ __forceinline__ int lds_agg_inc(int* cnt, bool flag) {
    const u64 g = __ballot(flag);
    if (!g) return -1;
    const int leader = __ffsll((long long)g) - 1;
    int base = 0;
    if (lane_id() == leader) base = atomicAdd(cnt, (int)__popcll(g));
    base = __builtin_amdgcn_readlane(base, leader);
    return flag ? base + (int)__popcll(g & lanemask_lt()) : -1;
}

// every NW-th live row of the tile, starting at `wave` (the NW waves of a workgroup share one tile)
template <int NW>
__device__ __forceinline__ u64 rows_of_wave(u64 tor, int wave) {
    u64 mine = 0ull;
    int k = 0;
    while (tor) {
        const u64 low = tor & (~tor + 1ull);
        if ((k & (NW - 1)) == wave) mine |= low;
        tor ^= low;
        ++k;
    }
    return mine;
}

// --------------------------------------------------------------------------
// k_predict: mapPrediction :645-694.
//   constant-velocity advance + ego-motion shift (:665-667), vz := 0 (:661-663),
//   out-of-map removal (:688).  The streaming loop issues no global atomics: a particle that must
//   be registered in a pyramid (:1233-1259) or that changed voxel (moveParticle :1206) is noted in
//   the tile's staging area (movers from the bottom, in-FOV stayers from the top; both together
//   never exceed the tile's capacity).  The workgroup's tail then
//     * registers the stayers in their pyramids (one aggregated atomic per distinct pyramid), and
//     * routes the movers to the inbox of their DESTINATION tile (one atomic per distinct tile);
//   k_place, which owns a destination tile exclusively, gives them slots.  Every particle is
//   advanced exactly once (the role of flag 7, :649,1219).
// part[blockIdx*4 + {0,1,2,3}] = {live in, left the map, pyramid full, moved}
// --------------------------------------------------------------------------
// First prediction of constructor-seeded particles: the reference draws the velocity noise (:653-659) from the
// table in SWEEP order (voxel-major, slot-minor), 3 values per particle whose |vx*vy*vz| >= 1e-6.  k_vz_count ranks
// those particles: per-voxel counts -> exclusive prefix inside a 256-voxel block (vz_pre) + block totals (blk_cnt,
// scanned by k_occ_scan).  Only runs in the rare frames in which a vz array exists.
template <int MW>
__global__ void __launch_bounds__(256) k_vz_count(MapDims d, DevState s, int* __restrict__ vz_pre, u64* __restrict__ vz_q) {
    __shared__ int s_w[4];
    const int tv = blockIdx.x * 256 + threadIdx.x;   // the slab's voxels in the reference's SWEEP order (= index order), whatever the storage order
    const bool in = tv < d.v_true;
    const int lv = in ? lv_of_true(d, tv) : 0;
    int q = 0;
    if (in) {
#pragma unroll
        for (int e = 0; e < MW; ++e) {
            u64 live = s.mask[(size_t)lv * MW + e] & ~s.nbmask[(size_t)lv * MW + e];
            u64 qual = 0ull;
            while (live) {
                const int b = __ffsll((long long)live) - 1;
                live &= live - 1ull;
                const size_t idx = pidx(d, lv, e * 64 + b);
                const V2 v = ld_vel(s, idx);
                if (!(fabs((double)(v.x * v.y * s.vz0[idx])) < 1e-6)) { ++q; qual |= 1ull << b; }
            }
            vz_q[(size_t)lv * MW + e] = qual;   // which slots draw noise: k_predict ranks with popcounts, no re-reads
        }
    }
    const int inc = wave_incl_scan_i(q);
    if (lane_id() == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) off += s_w[k];
    if (in) vz_pre[lv] = off + inc - q;
    if (threadIdx.x == 255) s.blk_cnt[blockIdx.x] = off + inc;
}

// one particle of mapPrediction: advance (:665-667, vz forced to 0 :662), classify.
// returns 0 = left the map (:688), 1 = stays in its voxel (pyr = its pyramid or -1), 2 = changed voxel (gv = the
// new voxel's STORAGE index inside the slab), 3 = left this rank's slab (multi-GPU).  zadd = dt * 0.f + odz (:667, the same for every particle).
// VIEW = false: no particle of the tile can lie in the field of view (the tile's box test), a stayer needs no pyramid.
template <bool VIEW>
__device__ __forceinline__ int advance_one(const MapDims& d, const float* s_ph, const float* s_pv, float dt, float odx, float ody,
                                           float zadd, float vx, float vy, float& px, float& py, float& pz, int lvp, int& pyr, int& gv) {
    px += dt * vx + odx;   // :665
    py += dt * vy + ody;   // :666
    pz += zadd;            // :667
    pyr = -1;
    int gtrue, nlv;
    if (!voxel_of_lv(d, px, py, pz, gtrue, nlv)) return 0;
    gv = nlv;
    if (nlv == lvp) { if (VIEW) pyr = pyramid_of(d, s_ph, s_pv, px, py, pz); return 1; }
    if (nlv < 0) return 3;
    return 2;
}

// velocity process noise of constructor-seeded particles on their first prediction (:653-659): only
// when |vx*vy*vz| >= 1e-6, drawn from the table in SWEEP order.  Rare: only the HASVZ instantiation of k_predict holds it.
__device__ __forceinline__ void vz_noise(const MapDims& d, const DevState& s, const FilterParams& fp, const int* __restrict__ vz_pre,
                                      const u64* __restrict__ vz_q, int mw, int lv, int e, int row, size_t idx, float* vxy) {
    const float vz = s.vz0[idx];
    if (!(fabs((double)(vxy[0] * vxy[1] * vz)) < 1e-6)) {
        // rank in the reference's sweep order: qualifying particles of earlier voxels (k_vz_count + k_occ_scan) + those in
        // lower slots of this voxel
        int rank = s.blk_cnt[(g_of_lv(d, lv) - d.v_base) >> 8] + vz_pre[lv];
        for (int e2 = 0; e2 <= e; ++e2) {
            u64 qm = vz_q[(size_t)lv * mw + e2];
            if (e2 == e) qm &= (1ull << row) - 1ull;
            rank += (int)__popcll(qm);
        }
        const int c = (int)(((long long)s.fs->v_cur + 3ll * (long long)rank) % fp.tab_n);
        vxy[0] += s.v_tab[c];
        vxy[1] += s.v_tab[(c + 1) % fp.tab_n];
        st_vel(s, idx, vxy[0], vxy[1]);
        note_speed(s, vxy[0], vxy[1]);
    }
    s.vz0[idx] = 0.f;
}

// Can a particle inside tile BX lie in the field of view?  (one wave; lanes 0-15 hold the corners.)  The tile's voxels fill one
// box (a run of x inside a row) or two (the tail of one row and the head of the next; lanes 0-7 / 8-15 hold the corners); every
// plane test of pyramid_of is a dot product that is monotone in each coordinate even after rounding, so its extreme over a box
// sits at a corner: if all 8 corners fail the same boundary plane, no particle of the box passes ifInPyramidsArea (:1329-1339).
// k_place of such tiles registers nothing and is free to run beside the pair kernels.
__device__ __forceinline__ int tile_view_test(const MapDims& d, const DevState& s, const int BX, const int l) {
    const float* __restrict__ gph = s.planes_h;   // (rotated by k_obs_points; uniform addresses: scalar loads)
    const float* __restrict__ gpv = s.planes_v;
    int x0, x1, y0, y1, z0, z1;
    bool two = false;
    if (d.tiling) {   // a cube of 4 x 4 x 4 voxels: one box
        cube_box(d, BX, x0, y0, z0);
        x1 = x0 + 3; y1 = y0 + 3; z1 = z0 + 3;
    } else {
    const int g0 = d.v_base + BX * 64, g1 = d.v_base + min(BX * 64 + 63, d.v_loc - 1);
    const int zc = d.nx * d.ny;
    const int r0 = g0 / d.nx, r1 = g1 / d.nx;          // first and last x-row the tile touches
    const int bx = (l >> 3) & 1;
    x0 = g0 % d.nx; x1 = g1 % d.nx; y0 = (g0 % zc) / d.nx; y1 = (g1 % zc) / d.nx; z0 = g0 / zc; z1 = g1 / zc;
    if (r1 - r0 == 1) {
        two = true;
        if (bx == 0) { x1 = d.nx - 1; y1 = y0; z1 = z0; } else { x0 = 0; y0 = y1; z0 = z1; }
    } else if (r1 - r0 > 1) {                          // nx < 64: whole rows (layers)
        x0 = 0; x1 = d.nx - 1;
        if (z0 != z1) { y0 = 0; y1 = d.ny - 1; }
    }
    }
    const float mg = d.res * 0.01f;   // a particle of voxel x has (int)((p + half) / res) == x: p may sit a rounding below the face
    const float cx = (l & 1) ? (float)(x1 + 1) * d.res - d.half_x + mg : (float)x0 * d.res - d.half_x - mg;
    const float cy = (l & 2) ? (float)(y1 + 1) * d.res - d.half_y + mg : (float)y0 * d.res - d.half_y - mg;
    const float cz = (l & 4) ? (float)(z1 + 1) * d.res - d.half_z + mg : (float)z0 * d.res - d.half_z - mg;
    const bool c8 = l < (two ? 16 : 8);
    const u64 b0 = __ballot(c8 && dot3(cx, cy, cz, gph) >= 0.f);
    const u64 b1 = __ballot(c8 && dot3(cx, cy, cz, gph + 3 * d.np_h) <= 0.f);
    const u64 b2 = __ballot(c8 && dot3(cx, cy, cz, gpv) <= 0.f);
    const u64 b3 = __ballot(c8 && dot3(cx, cy, cz, gpv + 3 * d.np_v) >= 0.f);
    auto box_in = [&](int sh) { return ((b0 >> sh) & 0xffull) && ((b1 >> sh) & 0xffull) && ((b2 >> sh) & 0xffull) && ((b3 >> sh) & 0xffull); };
    return (box_in(0) || box_in(8)) ? 1 : 0;
}
// Two-branch frame: is tile BX this launch's business?  cls > 0: the tiles whose class has a bit of cls; < 0: those that have none of -cls.
__device__ __forceinline__ bool cls_mine(int tc, int cls) { return cls > 0 ? (tc & cls) != 0 : (tc & -cls) == 0; }

// --------------------------------------------------------------------------
// k_tile_class: the tile classes of a TWO-BRANCH frame (KernelScratch::tile_cls), one thread per tile, right after the binning.
// Cube storage only (MapDims::tiling == 1: a tile is a box of 4 x 4 x 4 voxels).
// The frame (reference :300-322: prediction -> update -> births -> resampling, each a sweep over ALL voxels) touches most of a large
// map only to move it and to resample it; weights, births and pyramid lists live in the part the sensor sees.  The two parts run as
// two branches; what ties them together is (a) a particle that changes voxel ACROSS the border and (b) a newborn landing beyond it.
// Both have a bounded reach, and the classes are the border grown by those reaches:
//   Q   the tile's box, grown by the reach of a newborn (the position table's largest value, FrameParams::birth_reach), passes the
//       four boundary planes of the field of view (the corner argument of tile_view_test: a plane's dot product is monotone in every
//       coordinate, so its extreme over a box sits at the corner the normal's signs select).  Every observation lies in the wedge,
//       so every newborn lands in a Q tile, and every tile with a view (tile_view_test: the same box, not grown) is one;
//   P   a Q tile lies within the frame's largest displacement -- |od| + dt * (the largest speed the map has ever seen,
//       FrameScalars::vmax_bits), in voxels per axis, rounded up to cubes -- of the tile: nothing outside P can send a particle into Q.
// So: predict(P) -> place(Q) sees every arrival of Q; place(not Q) waits for both predictions; births stay inside Q.
// Conservative on purpose, exact never matters.
// --------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tile_class(MapDims d, DevState s, int* __restrict__ tile_cls) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int ntl = (d.v_loc + 63) >> 6;
    if (t >= ntl) return;
    const float* __restrict__ gph = s.planes_h;
    const float* __restrict__ gpv = s.planes_v;
    const float n0[3] = {gph[0], gph[1], gph[2]}, n1[3] = {gph[3 * d.np_h], gph[3 * d.np_h + 1], gph[3 * d.np_h + 2]};
    const float n2[3] = {gpv[0], gpv[1], gpv[2]}, n3[3] = {gpv[3 * d.np_v], gpv[3 * d.np_v + 1], gpv[3 * d.np_v + 2]};
    const float reach = s.fpar->birth_reach;
    const float mg = d.res * 0.01f + (reach > 0.f ? reach * 1.0001f : 0.f);   // (res * 0.01: tile_view_test's own margin)
    const float vmax = __int_as_float(s.fs->vmax_bits);
    const float dt = fabsf(s.fpar->dt);
    const float mx = (fabsf(s.fpar->od[0]) + dt * vmax) * 1.0001f, my = (fabsf(s.fpar->od[1]) + dt * vmax) * 1.0001f, mz = fabsf(s.fpar->od[2]) * 1.0001f;
    // voxels a particle can cross along an axis: floor(m / res) + 1 (it may sit right at the face), none if it cannot move that way at
    // all; in cubes: rounded up
    auto cubes = [&](float m, int n) { const int k = m > 0.f ? (m < d.res * (float)n ? (int)(m / d.res) + 1 : n) : 0; return (k + 3) >> 2; };
    const int rcx = cubes(mx, d.nx), rcy = cubes(my, d.ny), rcz = cubes(mz, d.nz);
    // does the box of voxels [x0, x1) x [y0, y1) x [z0, z1) (global z), grown by g, pass the four planes?
    auto ext = [&](const float* n, bool mxm, float ax, float bx, float ay, float by, float az, float bz) {
        const float cx = ((n[0] >= 0.f) == mxm) ? bx : ax, cy = ((n[1] >= 0.f) == mxm) ? by : ay, cz = ((n[2] >= 0.f) == mxm) ? bz : az;
        return dot3(cx, cy, cz, n);
    };
    auto box_in = [&](int x0, int x1, int y0, int y1, int z0, int z1, float g) {
        const float ax = (float)x0 * d.res - d.half_x - g, bx = (float)x1 * d.res - d.half_x + g;
        const float ay = (float)y0 * d.res - d.half_y - g, by = (float)y1 * d.res - d.half_y + g;
        const float az = (float)z0 * d.res - d.half_z - g, bz = (float)z1 * d.res - d.half_z + g;
        return ext(n0, true, ax, bx, ay, by, az, bz) >= 0.f && ext(n1, false, ax, bx, ay, by, az, bz) <= 0.f &&
               ext(n2, false, ax, bx, ay, by, az, bz) <= 0.f && ext(n3, true, ax, bx, ay, by, az, bz) >= 0.f;
    };
    const int cx = t % d.ncx, r = t / d.ncx, cy = r % d.ncy, cz = r / d.ncy;
    auto q_of = [&](int ux, int uy, int uz) { return box_in(ux * 4, ux * 4 + 4, uy * 4, uy * 4 + 4, d.z_lo + uz * 4, d.z_lo + uz * 4 + 4, mg); };
    const bool q = q_of(cx, cy, cz);
    bool p = q;
    // (one test rules most tiles out: the box of every cube within reach, grown like theirs)
    if (!p && box_in((cx - rcx) * 4, (cx + rcx) * 4 + 4, (cy - rcy) * 4, (cy + rcy) * 4 + 4, d.z_lo + (cz - rcz) * 4, d.z_lo + (cz + rcz) * 4 + 4, mg)) {
        for (int uz = max(0, cz - rcz); uz <= min(d.ncz - 1, cz + rcz) && !p; ++uz)
            for (int uy = max(0, cy - rcy); uy <= min(d.ncy - 1, cy + rcy) && !p; ++uy)
                for (int ux = max(0, cx - rcx); ux <= min(d.ncx - 1, cx + rcx); ++ux)
                    if (q_of(ux, uy, uz)) { p = true; break; }
    }
    tile_cls[t] = (q ? TILE_Q : 0) | (p ? TILE_P : 0);
}
void launch_tile_class(const LaunchCtx& c) {
    hipLaunchKernelGGL(k_tile_class, dim3((c.k.ntiles + 255) / 256), dim3(256), 0, c.stream, c.d, c.s, c.k.tile_cls);
}

// eight workgroups per CU: the sweep is a chain of phases (occupancy words, rows, tails) and only the workgroups that are in
// their row phase keep the memory system busy -- residency, not per-wave batch depth, is what moved this kernel (measured:
// 5 -> 7 -> 8 resident workgroups 0.274 -> 0.239 -> 0.224 ms at 132x132x60 saturated; 2 / 3 / 4 / 6 rows per batch all alike)
#ifndef PRED_LB
#define PRED_LB 8
#endif
template <int MW, int NW, bool HASVZ, bool SPARSE>
__global__ void __launch_bounds__(NW * 64, PRED_LB) k_predict(MapDims d, DevState s, FilterParams fp, int has_vz, int* __restrict__ part,
                                                 float4* __restrict__ mv_rec, float4* __restrict__ in_rec, int* __restrict__ in_cnt,
                                                 u64* __restrict__ expmask, const int* __restrict__ vz_pre, const u64* __restrict__ vz_q,
                                                 u64* __restrict__ omask, int extra, int* __restrict__ tile_fov, int rev, int* __restrict__ view_list,
                                                 const int* __restrict__ tcls, int cls) {
    // tcls / cls (two-branch frame): this launch sweeps only the tiles cls_mine() selects; the riders (extra) go with the in-view branch
    // rev: the tiles are walked from the last one down (workgroup -> tile mapping only).  k_place always walks AGAINST the k_predict
    // before it: a large map's live rows are several times the 256 MB Infinity Cache, and a sweep that starts where the last one
    // ENDED finds its first tiles (rows, occupancy words, inbox records) there instead of in HBM -- 132x132x60 saturated:
    // placement 0.186 -> 0.159 ms in three interleaved A/B rounds on one box.  DSPMAP_P_SWEEP_ALTERNATE = 1 flips all three
    // sweeps from frame to frame on top of that (no further gain measured).  Same results: no stage depends on the tile order.
    __shared__ float s_ph[DSP_MAX_PLANES_H * 3];
    __shared__ float s_pv[DSP_MAX_PLANES_V * 3];
    __shared__ u64 s_keep[MW * 64], s_ex[MW * 64];
    __shared__ int s_nmv, s_nst;
    __shared__ int s_cnt[4];
    __shared__ int s_any, s_view, s_mvany;
    // the first LSTG movers / stayers of the tile are noted in LDS (free: registers, not LDS, bound the
    // occupancy of this kernel), the rest in the tile's staging area in HBM
    __shared__ float4 s_mv[LSTG * 2], s_st[LSTG * 2];
    __shared__ int s_hist[HIST_NP];   // stayers per pyramid of this tile, then the base of the tile's run in each list
    __shared__ unsigned short s_cells[DENSE_MAX];   // sparse tiles: compact list of live cells ((slot << 6) | lane)
    __shared__ int s_ncell;
    const int tid = threadIdx.x;
    // SPARSE: a tile that holds nothing (k_resample saw it empty; no arrival, birth or import since) and whose future
    // accumulators need no zeroing is left at once: a sparse map is mostly such tiles, and what they cost is the time a resident
    // workgroup stays -- one scalar round trip here, before anything else of the kernel's arguments is looked at.
    // (The caller launches this variant while the map IS sparse -- FrameScalars::live_hint: on a saturated map the same flags
    // arrive with the tile's occupancy words, and asking first costs every tile a round trip and the kernel some registers:
    // +2-3 % at 132x132x60 saturated.)
    int tflags = -1;   // bit 0: the tile holds particles, bit 1: its future accumulators are to be zeroed; -1: not looked at yet
    if (SPARSE) {
        const int nextra0 = ((extra & 1) ? (d.np + NW - 1) / NW : 0) + ((extra & 2) ? 1 : 0) + ((extra & 4) ? (((d.v_loc + 63) >> 6) + 64 * NW - 1) / (64 * NW) : 0);
        if ((int)blockIdx.x >= nextra0) {
            const int bq0 = (int)blockIdx.x - nextra0;
            const int bx0 = rev ? (int)gridDim.x - nextra0 - 1 - bq0 : bq0;
            // (tile bitmaps: one cached word says that the tile holds nothing and has nothing to zero -- no per-tile count is written,
            // k_reduce_counters adds up the visited tiles' only)
            if (s.vis_bits && !((sload_i(reinterpret_cast<const int*>(s.vis_bits) + (bx0 >> 5)) >> (bx0 & 31)) & 1)) return;
            int t_live, f_dirty, f_clear;
            sload_i3(s.tile_live + bx0, s.fut_dirty + bx0, &s.fpar->clear_fut, t_live, f_dirty, f_clear);
            tflags = (t_live ? 1 : 0) | ((f_clear && f_dirty) ? 2 : 0);
            if (!tflags) {
                if (tid < 4) part[bx0 * 4 + tid] = 0;
                return;
            }
        }
    }
    const float odx = s.fpar->od[0], ody = s.fpar->od[1], odz = s.fpar->od[2], dt = s.fpar->dt;
    const int l = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and known to be: the row loops below become scalar control flow)
    // whole frame: the observation gather (448 independent waves, a chain of L2 round trips over the frame's points)
    // rides on this launch as extra workgroups behind the tiles -- it only needs k_obs_points' output, like the tiles
    // (extra & 1).  The birth rank -- one workgroup that needs nothing but the frame's birth cloud -- is the last one
    // (extra & 2).
    // They come FIRST in the grid so that they run beside the tiles instead of after them.
    if (blockIdx.x == 0 && tid == 0 && s.fpar->from_ring && (unsigned)*s.ring_seq == s.fpar->ring_pos) {
        *s.ring_seq = (int)(s.fpar->ring_pos + 1u);   // k_obs_points is done with the ring slot ...
        s.hint_out[2] = (int)(s.fpar->ring_pos + 1u); // ... and with the cloud it read over the bus (dspmap_update refills that slot 64 frames on)
    }
    // (DSPMAP_P_ESTIMATOR_QUEUE) the control words of this frame's first birth kernel: nobody has decided, nobody is listed
    if (blockIdx.x == 0 && tid < XQ_NDEC && s.xq) {
        __hip_atomic_store(s.xq + XQ_DEC + tid * 64, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(s.xq + 5, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (extra & 4) a frame that splits its placement: further workgroups list the tiles whose box can intersect the field of view --
    // every tile of the map, 64 per wave, one atomic per wave --, so that the placement that precedes the pair kernels walks THOSE
    // (a few per cent of a large map's tiles) instead of launching a workgroup per tile that finds out it has nothing to do: on the
    // saturated 132x132x60 map that launch took 65 us for 5 % of the arrivals (round 5).  The test needs the frame's planes only.
    const int ntl = (d.v_loc + 63) >> 6;
    const int ngather = (extra & 1) ? (d.np + NW - 1) / NW : 0, nrank = (extra & 2) ? 1 : 0;
    const int nviewb = (extra & 4) ? (ntl + 64 * NW - 1) / (64 * NW) : 0, nextra = ngather + nrank + nviewb;
    if ((int)blockIdx.x < nextra) {
        const int x = (int)blockIdx.x;
        if (x < ngather) {
            const int b = x * NW + wave;
            if (b < d.np) obs_gather_wave(d, s, b);
        } else if (x < ngather + nrank) {
            birth_rank_block(d, s, fp);
        } else {
            const int t0 = ((x - ngather - nrank) * NW + wave) * 64;   // this wave's 64 tiles
            const int ep = s.fpar->epoch;
            u64 bits = 0ull;
            for (int i = 0; i < 64 && t0 + i < ntl; ++i)
                if (tile_view_test(d, s, t0 + i, l)) bits |= 1ull << i;   // (wave-uniform result)
            if (t0 + l < ntl) tile_fov[t0 + l] = (ep << 1) | (int)((bits >> l) & 1ull);   // every tile gets this frame's tag, visited by the sweep or not
            const int nv = (int)__popcll(bits);
            int base = 0;
            if (nv && l == 0) base = atomicAdd(&s.fs->n_view_tiles, nv);
            base = __builtin_amdgcn_readfirstlane(base);
            if ((bits >> l) & 1ull) view_list[base + (int)__popcll(bits & lanemask_lt())] = t0 + l;
        }
        return;
    }
    const int BX = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x - nextra;   // tile index
    const int lv = BX * 64 + l;   // all four waves of the block look at the same tile
    int tflag_early = 1;   // (!SPARSE) the tile's tile_moving flag, fetched with its other flags
    if (!SPARSE) {
        // (round 5: fut_dirty is looked at here too -- it arrives in the same scalar round trip.  Zeroing EVERY tile's accumulators on
        // every pending clear was 8.4 of the 14.1 MB this kernel wrote per launch at the metric's size, 3.46 x its algorithmic bytes,
        // VERDICT r4 -- most tiles of a map never see a moving particle)
        int t_live, f_clear, f_dirty, tc;
        sload_i5(s.tile_live + BX, s.tile_moving + BX, &s.fpar->clear_fut, s.fut_dirty + BX, tcls ? tcls + BX : s.tile_live + BX, t_live, tflag_early, f_clear, f_dirty, tc);   // one scalar round trip
        if (tcls && !cls_mine(tc, cls)) return;   // the other branch's tile
        tflags = (t_live ? 1 : 0) | ((f_clear && f_dirty) ? 2 : 0);
    } else if (tcls && !cls_mine(sload_i(tcls + BX), cls)) return;
    if (tflags & 2) {
        // clearOccupancyMapPrediction (:431-438) was requested since the last frame: this tile's share of the
        // future accumulators is zeroed here instead of by two extra memset launches per frame -- if anything was added to
        // it since it was zeroed last (fut_dirty: set by whoever adds)
        // EVERY wave of the workgroup reads the tile's flags for itself: all of them must have done so before the flag is reset -- a wave
        // that found it reset already skipped its share of the horizons (round 5: three identical maps differed in horizons 1, 2, 3, 5,
        // never 0 or 4 -- wave 0's --, a cell keeping last frame's mass; the race was there since round 3 in the sparse variant and
        // showed once the dense variant looked at the flag too).  Nobody else writes these flags during the launch: the condition is
        // uniform over the workgroup
        __syncthreads();
        if (tid == 0) s.fut_dirty[BX] = 0;
        const int v0 = BX * 64, nv = min(64, d.v_loc - v0);
        for (int t = wave; t < d.T; t += NW) if (l < nv) s.fut[(size_t)t * d.v_loc + v0 + l] = 0ull;   // [T][V]: one row of 64 per wave and horizon
        if (tid < nv) s.fut_stat[v0 + tid] = 0.f;
    }
    if (!(tflags & 1)) {   // (empty, but its accumulators had to be zeroed)
        if (tid < 4) part[BX * 4 + tid] = 0;
        return;
    }
    if (wave == NW - 1) {
        // the tile's view on the field of view, tagged with the frame: a tile that was skipped here and receives arrivals is
        // tested by k_place itself
        const int v = tile_view_test(d, s, BX, l);
        if (l == 0) { if (!(extra & 4)) tile_fov[BX] = (s.fpar->epoch << 1) | v; s_view = v; }   // (extra & 4: the listing waves tag every tile)
    }
    const bool inr = lv < d.v_loc;
    const int lvs = inr ? lv : 0;
    const int tgb = __builtin_amdgcn_readfirstlane(tile_gbase(d, BX));   // the reference's voxel index of (tile, lane) = tgb + lane_goff(lane): sweep keys
    u64 mword[MW], live[MW];
    bool any = false, nb_any = false;
#pragma unroll
    for (int e = 0; e < MW; ++e) {
        mword[e] = 0ull; u64 nbword = 0ull;
        if (inr) { mword[e] = s.mask[(size_t)lv * MW + e]; nbword = s.nbmask[(size_t)lv * MW + e]; }
        live[e] = mword[e] & ~nbword;  // particles born/seeded this frame (flag 15) are not predicted (:649)
        any |= live[e] != 0ull;
        nb_any |= nbword != 0ull;      // ... and their velocities are not looked at: the tile cannot be called static (below)
        if (wave == 0) {
            s_keep[e * 64 + l] = live[e]; s_ex[e * 64 + l] = 0ull;
            if (inr) omask[(size_t)lv * MW + e] = mword[e] | nbword;   // occupancy before this prediction (k_place: arrivals from lower voxels)
        }
    }
    if (tid == 0) { s_any = 0; s_nmv = 0; s_nst = 0; s_ncell = 0; s_mvany = 0; }
    if (tid < 4) s_cnt[tid] = 0;
    if (d.np <= HIST_NP) for (int b = tid; b < d.np; b += NW * 64) s_hist[b] = 0;
    // the rotated planes are requested together with the occupancy words (one round trip less for the tiles that have work)
    for (int i = tid; i < (d.np_h + 1) * 3; i += blockDim.x) s_ph[i] = s.planes_h[i];
    for (int i = tid; i < (d.np_v + 1) * 3; i += blockDim.x) s_pv[i] = s.planes_v[i];
    __syncthreads();
    if (wave == 0 && __ballot(any)) {
        // sparse tile (few live cells per live row): the heavy per-particle work runs on DENSE lanes over a compact
        // cell list instead of row by row with mostly idle lanes -- what a realistic map (particles near surfaces
        // only) consists of; saturated tiles keep the coalesced row sweep
        int cl = 0, nrows = 0;
#pragma unroll
        for (int e = 0; e < MW; ++e) { cl += (int)__popcll(live[e]); nrows += (int)__popcll(wave_or_u64(live[e])); }
        const int nlive = wave_sum_i(cl);
        if (l == 0) s_any = (!HASVZ && nlive <= DENSE_MAX && nlive * 5 < nrows * 64 * 3) ? 2 : 1;
    }
    __syncthreads();
    if (!s_any) {  // empty tile
        if (tid < 4) part[BX * 4 + tid] = 0;
        return;   // (tile_moving keeps its value: 0 promises zeroed velocity cells, which only a sweep that read the rows can give)
    }
    const bool dense = s_any == 2;
    const int cap = 64 * d.slots;                       // records per staging area / inbox
    const size_t mv_base = (size_t)BX * cap;    // this tile's staging area (2 float4 per record)
    int c_live = 0, c_out = 0, c_pf = 0, c_mv = 0;   // c_live / c_out / c_mv: wave-uniform (sums of ballots), c_pf: per lane
    const bool view = s_view != 0;                    // can a particle of this tile lie in the field of view at all?
    // every live particle of the tile has velocity (0, 0) -- k_predict's own finding of the last frame, plus whatever arrived or was
    // born since -- : the velocity rows are not fetched (a third of what this sweep reads)
    const int tflag = SPARSE ? __builtin_amdgcn_readfirstlane(s.tile_moving[BX]) : tflag_early;   // (nobody writes a tile's flag between k_resample and this sweep)
    const bool tmov = HASVZ || tflag != 0 || !d.tile_skip;
    // a live particle with a velocity (this lane).  Particles that are not predicted this frame (flag 15: a constructor pre-fill on a
    // non-empty map, imported newborn records, a stage-API birth without a resampling) keep whatever velocity they have and this
    // sweep does not read it: a tile that holds one counts as moving -- its velocity cells are NOT zeroed (the reference never
    // touches an unpredicted particle, :649)
    bool mv_seen = nb_any;
    const float zadd = dt * 0.f + odz;                // :667, the same for every particle
    // buffer descriptors of this tile's share of the three field arrays (the tile's cells are contiguous: [slot][64]): a
    // lane's byte offset is ONE register whatever the row, the row enters as a scalar offset
    const size_t tcell = (size_t)BX * d.slots * 64;
    const int tcells = d.slots * 64;
    const brsrc rs_pos = __builtin_amdgcn_make_buffer_rsrc((void*)(s.pos + 3 * tcell), 0, tcells * 12, 0x00020000);
    const brsrc rs_vel = __builtin_amdgcn_make_buffer_rsrc((void*)(s.vel + 2 * tcell), 0, tcells * 8, 0x00020000);
    const brsrc rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(s.w + tcell), 0, tcells * 4, 0x00020000);
    // notes a stayer that needs a pyramid entry (top of the staging area, downwards) / a mover (bottom, upwards)
    auto note_stayer = [&](bool on, int pyr, int cell, float px, float py, float pz, float w) {
        const int ks = lds_agg_inc(&s_nst, on);
        if (ks >= 0) {
            const float4 a = make_float4(__int_as_float(pyr), __int_as_float(cell), px, py);
            const float4 b = make_float4(pz, w, 0.f, 0.f);
            if (ks < LSTG) { s_st[ks * 2] = a; s_st[ks * 2 + 1] = b; }
            else { const size_t o = (mv_base + cap - 1 - ks) * 2; mv_rec[o] = a; mv_rec[o + 1] = b; }
        }
    };
    auto note_mover = [&](bool on, int gv, int cell, float vx, float vy, float px, float py, float pz, float w) {
        const int km = lds_agg_inc(&s_nmv, on);
        if (km >= 0) {
            const float4 a = make_float4(__int_as_float(gv), vx, vy, px);
            // .w: source key = sweep position (global voxel, slot) of the particle: k_place serves arrivals in this order
            const float4 b = make_float4(py, pz, w, __int_as_float((tgb + lane_goff(d, cell & 63)) * d.slots + (cell >> 6)));
            if (km < LSTG) { s_mv[km * 2] = a; s_mv[km * 2 + 1] = b; }
            else { const size_t o = (mv_base + km) * 2; mv_rec[o] = a; mv_rec[o + 1] = b; }
        }
    };
    if (dense) {
        // cell list: every wave compacts its share of the live rows
#pragma unroll
        for (int e = 0; e < MW; ++e) {
            u64 tor = rows_of_wave<NW>(wave_or_u64(live[e]), wave);
            while (tor) {
                const int row = __ffsll((long long)tor) - 1;
                tor &= tor - 1ull;
                const bool on = (live[e] >> row) & 1ull;
                const int k = lds_agg_inc(&s_ncell, on);
                if (on) s_cells[k] = (unsigned short)(((e * 64 + row) << 6) | l);
            }
        }
        __syncthreads();
        const int ncell = s_ncell;
        for (int c0 = 0; c0 < ncell; c0 += NW * 64) {
            const int c = c0 + tid;
            const bool act = c < ncell;
            const int cell = act ? (int)s_cells[c] : l;   // (a valid cell of the tile for the idle lanes too)
            const int slot = cell >> 6, ln = cell & 63;
            const int coff = slot * 64 + ln;              // the cell inside the tile
            V2 v2; v2.x = 0.f; v2.y = 0.f;
            if (tmov) v2 = bl_vel(rs_vel, coff, 0);
            const P3 p3 = bl_pos(rs_pos, coff, 0);
            const float w = bl_w(rs_w, coff, 0);
            float px = p3.x, py = p3.y, pz = p3.z;
            int pyr = -1, gv = -1, kind = -1;
            if (d.static_model) {   // dsp_static.h:640-646
                if (act && (v2.x != 0.f || v2.y != 0.f)) st_vel(s, tcell + coff, 0.f, 0.f);
                v2.x = 0.f; v2.y = 0.f;
            }
            mv_seen |= act && (v2.x != 0.f || v2.y != 0.f);
            if (act) {
                kind = view ? advance_one<true>(d, s_ph, s_pv, dt, odx, ody, zadd, v2.x, v2.y, px, py, pz, BX * 64 + ln, pyr, gv)
                            : advance_one<false>(d, s_ph, s_pv, dt, odx, ody, zadd, v2.x, v2.y, px, py, pz, BX * 64 + ln, pyr, gv);
                const u64 bit = 1ull << (slot & 63);
                if (kind == 1 || kind == 3) bs_pos(rs_pos, coff, 0, px, py, pz);   // (a mover's cell is dead: its record carries the position)
                if (kind == 0 || kind == 2) atomicAnd(&s_keep[(slot >> 6) * 64 + ln], ~bit);
                else if (kind == 3) atomicOr(&s_ex[(slot >> 6) * 64 + ln], bit);
            }
            c_live += (int)__popcll(__ballot(act));
            c_out += (int)__popcll(__ballot(kind == 0));
            c_mv += (int)__popcll(__ballot(kind == 2));
            note_stayer(kind == 1 && pyr >= 0, pyr, cell, px, py, pz, w);
            note_mover(kind == 2, gv, cell, v2.x, v2.y, px, py, pz, w);
        }
    } else {
#pragma unroll
    for (int e = 0; e < MW; ++e) {
        unsigned kc_lo = 0u, kc_hi = 0u, ex_lo = 0u, ex_hi = 0u;   // slots to free / to export, this lane's voxel
        const unsigned live_lo = (unsigned)live[e], live_hi = (unsigned)(live[e] >> 32);
        u64 tor = rows_of_wave<NW>(wave_or_u64(live[e]), wave);   // wave-uniform: scalar control flow from here on
        while (tor) {
            int row[RB];
            P3 pp[RB];
            V2 vv[RB];
            float w[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {  // issue every load of the batch
                row[r] = tor ? __ffsll((long long)tor) - 1 : -1;
                if (tor) tor &= tor - 1ull;
                // unconditional loads (always a valid cell of this lane's voxel; dead cells share the row's cache lines): a
                // predicated vector load makes the compiler wait for it before issuing the next row
                const int srow = (e * 64 + (row[r] < 0 ? 0 : row[r])) * 64;   // cells before this row inside the tile (wave-uniform)
                vv[r].x = 0.f; vv[r].y = 0.f;
                if (tmov) vv[r] = bl_vel(rs_vel, l, srow);   // (wave-uniform: no predicated load)
                pp[r] = bl_pos(rs_pos, l, srow);
                w[r] = bl_w(rs_w, l, srow);
            }
            // every row of the batch is advanced, stored and noted before the next one is touched
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (row[r] < 0) continue;   // (scalar)
                const int rw = row[r];
                const int srow = (e * 64 + rw) * 64;
                const bool act = (((rw < 32 ? live_lo : live_hi) >> (rw & 31)) & 1u) != 0u;
                float vx = vv[r].x, vy = vv[r].y, px = pp[r].x, py = pp[r].y, pz = pp[r].z;
                if (d.static_model) {   // dsp_static.h:640-646: velocities forced to zero, positions follow the ego-motion only
                    if (act && (vx != 0.f || vy != 0.f)) st_vel(s, tcell + srow + l, 0.f, 0.f);
                    vx = 0.f; vy = 0.f;
                }
                if (HASVZ) {
                    if (act) { float vxy[2] = {vx, vy}; vz_noise(d, s, fp, vz_pre, vz_q, MW, lvs, e, rw, tcell + srow + l, vxy); vx = vxy[0]; vy = vxy[1]; }
                }
                mv_seen |= act && (vx != 0.f || vy != 0.f);
                int pyr = -1, gv = -1;
                int kind = view ? advance_one<true>(d, s_ph, s_pv, dt, odx, ody, zadd, vx, vy, px, py, pz, lv, pyr, gv)
                                : advance_one<false>(d, s_ph, s_pv, dt, odx, ody, zadd, vx, vy, px, py, pz, lv, pyr, gv);
                if (!act) kind = -1;
                if (kind == 1 || kind == 3) bs_pos(rs_pos, l, srow, px, py, pz);   // (a mover's cell is dead: its record carries the position)
                const unsigned bit = 1u << (rw & 31);
                const unsigned fr = (kind == 0 || kind == 2) ? bit : 0u, xp = kind == 3 ? bit : 0u;   // left the map :688 / changed voxel; left the slab
                if (rw < 32) { kc_lo |= fr; ex_lo |= xp; } else { kc_hi |= fr; ex_hi |= xp; }
                c_live += (int)__popcll(__ballot(act));
                c_out += (int)__popcll(__ballot(kind == 0));
                c_mv += (int)__popcll(__ballot(kind == 2));
                const int cell = ((e * 64 + rw) << 6) | l;
                if (view) note_stayer(kind == 1 && pyr >= 0, pyr, cell, px, py, pz, w[r]);
                note_mover(kind == 2, gv, cell, vx, vy, px, py, pz, w[r]);
            }
        }
        const u64 keep_clr = ((u64)kc_hi << 32) | kc_lo, ex = ((u64)ex_hi << 32) | ex_lo;
        if (keep_clr) atomicAnd(&s_keep[e * 64 + l], ~keep_clr);
        if (ex) atomicOr(&s_ex[e * 64 + l], ex);
    }
    }
    if (__ballot(mv_seen) && l == 0) s_mvany = 1;
    // workgroup-scope ordering is enough for the read-back below: the waves of a workgroup share the
    // CU's write-through L1 (an agent-scope fence would write back the XCD's whole L2)
    __syncthreads();
    // ---- tail 1: pyramid registration of the stayers (:1245-1259); TB records per thread so that all
    // the atomics of up to TB * blockDim records are in flight together
    const int nst = s_nst, nmv = s_nmv;
    auto st_rec = [&](int i, float4& a, float4& b) {
        if (i < LSTG) { a = s_st[i * 2]; b = s_st[i * 2 + 1]; }
        else { const size_t o = (mv_base + cap - 1 - i) * 2; a = mv_rec[o]; b = mv_rec[o + 1]; }
    };
    auto mv_get = [&](int i, float4& a, float4& b) {
        if (i < LSTG) { a = s_mv[i * 2]; b = s_mv[i * 2 + 1]; }
        else { a = mv_rec[(mv_base + i) * 2]; b = mv_rec[(mv_base + i) * 2 + 1]; }
    };
    if (d.np <= HIST_NP) {
        // rank every stayer inside its pyramid's run with an LDS atomic, reserve the runs with ONE global atomic per
        // non-empty pyramid (all in flight together), then place the records: one memory round trip per tile
        for (int i = tid; i < nst; i += NW * 64) {
            const int pyr = __float_as_int(i < LSTG ? s_st[i * 2].x : mv_rec[(mv_base + cap - 1 - i) * 2].x);
            const float r = __int_as_float(atomicAdd(&s_hist[pyr], 1));
            if (i < LSTG) s_st[i * 2 + 1].z = r;
            else reinterpret_cast<float*>(&mv_rec[(mv_base + cap - 1 - i) * 2 + 1])[2] = r;
        }
        __syncthreads();
        for (int b = tid; b < d.np; b += NW * 64) {
            const int c = s_hist[b];
            if (c) s_hist[b] = atomicAdd(&s.pyr_cnt[b], c);
        }
        __syncthreads();
        for (int i = tid; i < nst; i += NW * 64) {
            float4 a, b;
            st_rec(i, a, b);
            const int pyr = __float_as_int(a.x), sl = __float_as_int(a.y);   // sl = (slot << 6) | lane
            const int pos = s_hist[pyr] + __float_as_int(b.z);
            if (pos < d.capa) {
                const size_t o = (size_t)pyr * d.capa + pos;
                s.fov_rec[o] = make_float4(a.z, a.w, b.x, b.y);
                s.fov_slot[o] = (int)(((size_t)BX * d.slots + (sl >> 6)) * 64 + (sl & 63));
                s.fov_key[o] = (tgb + lane_goff(d, sl & 63)) * d.slots + (sl >> 6);   // a stayer's sweep key is its own cell
            } else {
                // pyramid list full: the particle vanishes (-2, :1256-1259)
                atomicAnd(&s_keep[((sl >> 12) & 1) * 64 + (sl & 63)], ~(1ull << ((sl >> 6) & 63)));
                ++c_pf;
            }
        }
    } else
    for (int i0 = 0; i0 < nst; i0 += NW * 64 * TB) {
        int key[TB], pos[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int i = i0 + j * NW * 64 + tid;
            key[j] = -1;
            if (i < nst) key[j] = __float_as_int(i < LSTG ? s_st[i * 2].x : mv_rec[(mv_base + cap - 1 - i) * 2].x);
        }
        batch_append<TB>(s.pyr_cnt, key, pos);
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            if (key[j] < 0) continue;
            float4 a, b;
            st_rec(i0 + j * NW * 64 + tid, a, b);
            const int sl = __float_as_int(a.y);   // (slot << 6) | lane
            if (pos[j] < d.capa) {
                const size_t o = (size_t)key[j] * d.capa + pos[j];
                s.fov_rec[o] = make_float4(a.z, a.w, b.x, b.y);
                s.fov_slot[o] = (int)(((size_t)BX * d.slots + (sl >> 6)) * 64 + (sl & 63));
                s.fov_key[o] = (tgb + lane_goff(d, sl & 63)) * d.slots + (sl >> 6);
            } else {
                // pyramid list full: the particle vanishes (-2, :1256-1259)
                atomicAnd(&s_keep[((sl >> 12) & 1) * 64 + (sl & 63)], ~(1ull << ((sl >> 6) & 63)));
                ++c_pf;
            }
        }
    }
    // ---- tail 2: route the movers to the inbox of their destination tile
    for (int i0 = 0; i0 < nmv; i0 += NW * 64 * TB) {
        int key[TB], pos[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int i = i0 + j * NW * 64 + tid;
            key[j] = -1;
            if (i < nmv) key[j] = __float_as_int(i < LSTG ? s_mv[i * 2].x : mv_rec[(mv_base + i) * 2].x) >> 6;   // (.x: the new voxel's storage index)
        }
        batch_append<TB>(in_cnt, key, pos);
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            if (SPARSE && s.arr_bits && key[j] >= 0 && pos[j] == 0) atomicOr(&s.arr_bits[key[j] >> 5], 1u << (key[j] & 31));   // the inbox's first record this frame
            // beyond the inbox: more arrivals than the tile has slots; k_place counts them as dropped
            if (key[j] < 0 || pos[j] >= cap) continue;
            float4 ra, rb;
            mv_get(i0 + j * NW * 64 + tid, ra, rb);
            const size_t o = ((size_t)key[j] * cap + pos[j]) * 2;
            in_rec[o] = ra; in_rec[o + 1] = rb;
        }
    }
    // per-block statistics (reduced lazily by the host; no global atomics here)
    c_pf = wave_sum_i(c_pf);   // (the other three are sums of ballots already)
    if (l == 0) {
        if (c_live) atomicAdd(&s_cnt[0], c_live);
        if (c_out) atomicAdd(&s_cnt[1], c_out);
        if (c_pf) atomicAdd(&s_cnt[2], c_pf);
        if (c_mv) atomicAdd(&s_cnt[3], c_mv);
    }
    __syncthreads();
    if (wave == 0 && inr) {
#pragma unroll
        for (int e = 0; e < MW; ++e) {
            // exports keep their live bit until the export pass has copied them out
            const u64 nm = s_keep[e * 64 + l] | (mword[e] & ~live[e]);
            if (nm != mword[e]) s.mask[(size_t)lv * MW + e] = nm;
            if (expmask && s_ex[e * 64 + l]) expmask[(size_t)lv * MW + e] = s_ex[e * 64 + l];
        }
    }
    if (tid < 4) part[BX * 4 + tid] = s_cnt[tid];
    // (a tile that was static stays so until somebody brings a velocity.)  A tile that BECOMES static has its velocity cells zeroed,
    // all of them: the flag promises that every cell of the tile -- live, dead, rows this sweep never loaded -- holds (0, 0), so
    // that whoever puts a static particle there (k_place: two scattered stores per arrival instead of three) need not write one.
    if (tmov && tflag != 0 && d.tile_skip) {
        if (!s_mvany) {
            float4* const vz4 = reinterpret_cast<float4*>(s.vel + 2 * tcell);
            for (int i = tid; i < tcells / 2; i += NW * 64) vz4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid == 0) s.tile_moving[BX] = s_mvany;
    }
}

// --------------------------------------------------------------------------
// k_place: the voxel-changing half of moveParticle (:1209-1230), in the reference's ORDER.
// The reference sweeps voxels in index order and slots inside a voxel in slot order; a particle that changes voxel
// takes the first free slot of its destination AT THAT MOMENT (:1214-1215).  For a destination voxel D that means:
//   * arrivals are served in the order of their source (voxel, slot);
//   * an arrival from a LOWER voxel index comes before D itself is swept: D's own leavers still hold their slots
//     -> first free slot of (occupancy before the prediction | slots given to earlier arrivals);
//   * an arrival from a HIGHER index comes after D's sweep: the slots D's leavers (and out-of-map particles) freed
//     are available -> first free slot of (occupancy after the prediction | slots given to earlier arrivals);
//   * no free slot -> the particle vanishes (-1, :1227-1229).
// One workgroup OWNS one destination tile: the inbox records carry their source key; they are bucketed per
// destination voxel in LDS and every arrival computes its slot in closed form from its rank among the voxel's
// arrivals -- the same particles end up in the same slots as in the sequential reference, with no global atomic
// on the occupancy words and no sequential loop.  Pyramid registration :1233-1259.
// FrameScalars::n_place_vf / n_place_pf += {voxel full, pyramid full}
// --------------------------------------------------------------------------
#ifdef PLACE_PROF
// build with DSPMAP_EXTRA_FLAGS=-DPLACE_PROF: cycle stamps of a tile's phases in k_place (tools/prof/place_prof.py)
__device__ long long g_plprof[8 * 131072];
extern "C" int dspmap_debug_place_prof(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_plprof), sizeof(long long) * (size_t)n); }
#define PLSTAMP(k) do { if (threadIdx.x == 0 && BX < 131072) g_plprof[BX * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PLSTAMP(k) do { } while (0)
#endif
#define PLACE_MAX 1024   // arrivals of one tile whose bucketed keys fit the LDS table; a tile that receives more (up to its
                         // capacity of 64 * slots records) keeps them in its own staging area of k_predict, which is dead by now
template <int MW>
__device__ __forceinline__ void place_tile(const MapDims& d, const DevState& s, const float4* __restrict__ in_rec, int* __restrict__ in_cnt,
                                           const u64* __restrict__ omask,
                                           float4* __restrict__ stage, const int BX, const int n_all, const bool was_live, const bool t_moving_in,
                                           const FilterParams& fp) {   // n_all = in_cnt[BX] > 0
    __shared__ float s_ph[DSP_MAX_PLANES_H * 3];
    __shared__ float s_pv[DSP_MAX_PLANES_V * 3];
    __shared__ u64 s_cur[MW * 64], s_org[MW * 64], s_new[MW * 64], s_own[MW * 64];
    __shared__ int s_lcnt[64], s_loff[65];
    __shared__ int s_bk[PLACE_MAX];              // source keys of the arrivals, bucketed by destination lane
    __shared__ int s_cnt[2];
    const int tid = threadIdx.x, NT = (int)blockDim.x;   // (64 threads per workgroup on sparse maps: a tile receives a handful of arrivals and
                                                         // what bounds the launch is how many tiles' latency chains run at once, launch_claim)
    // was_live: an empty tile was skipped by k_predict: its omask words are stale (and zero in truth)
    const bool t_moving = t_moving_in || !d.tile_skip;   // tile_moving as k_predict left it (this workgroup is the only one that raises it during the placement)
    const int cap = 64 * d.slots;
    const int n = min(n_all, cap);
    const bool in_lds = n <= PLACE_MAX;
    int* const gbk = reinterpret_cast<int*>(stage + (size_t)BX * cap * 2);   // (cap * 8 ints; cap are used)
    const size_t base = (size_t)BX * cap;
    // the first 256 records stay in registers across the phases (most tiles receive fewer): requested together with the
    // occupancy words, one memory round trip
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), b0 = a0;
    PLSTAMP(0);
    if (tid < n) { a0 = in_rec[(base + tid) * 2]; b0 = in_rec[(base + tid) * 2 + 1]; }
    for (int i = tid; i < (d.np_h + 1) * 3; i += blockDim.x) s_ph[i] = s.planes_h[i];
    for (int i = tid; i < (d.np_v + 1) * 3; i += blockDim.x) s_pv[i] = s.planes_v[i];
    if (tid < 64) {
        const int lv = BX * 64 + tid;
        s_lcnt[tid] = 0;
#pragma unroll
        for (int e = 0; e < MW; ++e) {
            const bool in = lv < d.v_loc;
            const u64 own = in ? s.mask[(size_t)lv * MW + e] : 0ull;
            s_own[e * 64 + tid] = own;   // the voxel's occupancy word as the placement found it: nobody else writes it meanwhile
            s_cur[e * 64 + tid] = in ? (own | s.nbmask[(size_t)lv * MW + e]) : ~0ull;
            const u64 org = in ? omask[(size_t)lv * MW + e] : 0ull;   // (unconditional: the flag and the word arrive together)
            s_org[e * 64 + tid] = in ? (was_live ? org : 0ull) : ~0ull;
            s_new[e * 64 + tid] = 0ull;
            if (in) {   // what k_place_fix needs should a pyramid list turn arrivals of this tile away: both occupancies as used here
                s.pmask[(size_t)lv * MW + e] = s_cur[e * 64 + tid];
                if (!was_live) const_cast<u64*>(omask)[(size_t)lv * MW + e] = 0ull;
            }
        }
    }
    if (tid < 2) s_cnt[tid] = 0;
    int c_vf = tid == 0 ? n_all - n : 0, c_pf = 0;
    __syncthreads();
    PLSTAMP(1);
    // bucket the arrivals' source keys by destination lane: counts, offsets, then every key into its lane's run
    for (int i = tid; i < n; i += NT) {
        const int gv = __float_as_int(i == tid ? a0.x : in_rec[(base + i) * 2].x);   // (the destination voxel's storage index)
        atomicAdd(&s_lcnt[gv & 63], 1);
    }
    __syncthreads();
    if (tid < 64) {
        const int c = s_lcnt[tid];
        const int inc = wave_incl_scan_i(c);
        s_loff[tid] = inc - c;
        if (tid == 63) s_loff[64] = inc;
        s_lcnt[tid] = 0;
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const int gv = __float_as_int(i == tid ? a0.x : in_rec[(base + i) * 2].x);
        const int key = __float_as_int(i == tid ? b0.w : in_rec[(base + i) * 2 + 1].w);
        const int ln = gv & 63;
        const int o = s_loff[ln] + atomicAdd(&s_lcnt[ln], 1);
        if (in_lds) s_bk[o] = key;
        else __hip_atomic_store(&gbk[o], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    PLSTAMP(2);
    for (int i0 = 0; i0 < n; i0 += NT) {
        const int i = i0 + tid;
        int key[1] = {-1}, pos[1];
        int ln = 0, nsl = -1, skey = 0;
        size_t nidx = 0;
        float px = 0, py = 0, pz = 0, w = 0, avx = 0, avy = 0;
        if (i < n) {
            const float4 a = i == tid ? a0 : in_rec[(base + i) * 2];
            const float4 b = i == tid ? b0 : in_rec[(base + i) * 2 + 1];
            px = a.w; py = b.x; pz = b.y; w = b.z; avx = a.y; avy = a.z;
            skey = __float_as_int(b.w);
            ln = __float_as_int(a.x) & 63;
            // position of this arrival in the reference's service order of its destination voxel, in closed form:
            // the nF arrivals from lower voxel indices take, in key order, the first free slots of the occupancy
            // BEFORE the prediction; the others then take the first free slots of the occupancy AFTER it that are
            // still left
            const int o0 = s_loff[ln], mm = s_loff[ln + 1] - o0;
            const long long dkey = (long long)(tile_gbase(d, BX) + lane_goff(d, ln)) * d.slots;   // key of (D, slot 0)
            int r = 0, nF = 0;
            if (in_lds) {
                for (int x = 0; x < mm; ++x) {
                    const int kx = s_bk[o0 + x];
                    r += kx < skey ? 1 : 0;
                    nF += (long long)kx < dkey ? 1 : 0;
                }
            } else {
                for (int x = 0; x < mm; ++x) {   // (agent-scope loads: served by the L2 the stores above went to)
                    const int kx = __hip_atomic_load(&gbk[o0 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    r += kx < skey ? 1 : 0;
                    nF += (long long)kx < dkey ? 1 : 0;
                }
            }
            u64 frO[MW], frC[MW];
            int avail = 0;
#pragma unroll
            for (int e = 0; e < MW; ++e) { frO[e] = ~s_org[e * 64 + ln] & valid_bits(d, e); avail += (int)__popcll(frO[e]); }
            if (r < nF) {            // forward arrival: r-th free slot of the old occupancy
                int q = r;
#pragma unroll
                for (int e = 0; e < MW; ++e) {
                    const int c = (int)__popcll(frO[e]);
                    if (nsl < 0) {
                        if (q < c) { u64 f = frO[e]; for (; q > 0; --q) f &= f - 1ull; nsl = e * 64 + (__ffsll((long long)f) - 1); }
                        else q -= c;
                    }
                }
            } else {                 // backward arrival: skip what the forward ones took
                int tk = min(nF, avail);   // forward arrivals that found a slot: the first tk free bits of the old occupancy
#pragma unroll
                for (int e = 0; e < MW; ++e) {
                    u64 f = frO[e], took = 0ull;
                    for (; tk > 0 && f; --tk) { took |= f & (~f + 1ull); f &= f - 1ull; }
                    frC[e] = ~s_cur[e * 64 + ln] & valid_bits(d, e) & ~took;
                }
                int q = r - nF;
#pragma unroll
                for (int e = 0; e < MW; ++e) {
                    const int c = (int)__popcll(frC[e]);
                    if (nsl < 0) {
                        if (q < c) { u64 f = frC[e]; for (; q > 0; --q) f &= f - 1ull; nsl = e * 64 + (__ffsll((long long)f) - 1); }
                        else q -= c;
                    }
                }
            }
            if (nsl >= 0) {
                nidx = pidx(d, BX * 64 + ln, nsl);
                st_pos(s, nidx, px, py, pz);
                // a static arrival in a tile of static particles finds (0, 0) in its cell already (tile_moving = 0 promises it)
                if (avx != 0.f || avy != 0.f) { st_vel(s, nidx, avx, avy); s.tile_moving[BX] = 1; if (d.v_true != d.v_glob) note_speed(s, avx, avy); }   // (a slab: the arrival may come from another rank's map)   // (k_predict wrote the tile's flag before any arrival)
                else if (t_moving) st_vel(s, nidx, 0.f, 0.f);
                s.w[nidx] = w;
                key[0] = pyramid_of(d, s_ph, s_pv, px, py, pz);
            } else {
                ++c_vf;
            }
        }
        batch_append<1>(s.pyr_cnt, key, pos);
        if (nsl >= 0) {
            bool keep = true;
            int ref = -1;
            if (key[0] >= 0) {
                if (pos[0] < d.capa) {
                    const size_t o = (size_t)key[0] * d.capa + pos[0];
                    s.fov_rec[o] = make_float4(px, py, pz, w);
                    s.fov_slot[o] = (int)nidx;
                    s.fov_key[o] = skey;   // a mover is registered when the sweep reaches its SOURCE cell
                    ref = (int)o;
                } else {
                    ++c_pf;  // :1256-1259
                    keep = false;
                }
            }
            gbk[cap + i] = ref;   // the arrival's list entry, beside its inbox record (k_place_fix re-points it if the arrival is moved)
            if (keep) atomicOr(&s_new[(nsl >> 6) * 64 + ln], 1ull << (nsl & 63));
        }
    }
    PLSTAMP(3);
    c_vf = wave_sum_i(c_vf); c_pf = wave_sum_i(c_pf);
    if (lane_id() == 0) {
        if (c_vf) atomicAdd(&s_cnt[0], c_vf);
        if (c_pf) atomicAdd(&s_cnt[1], c_pf);
    }
    __syncthreads();
    PLSTAMP(4);
    if (tid < 64) {
        const int lv = BX * 64 + tid;
#pragma unroll
        for (int e = 0; e < MW; ++e)
            if (s_new[e * 64 + tid]) s.mask[(size_t)lv * MW + e] = s_own[e * 64 + tid] | s_new[e * 64 + tid];
    }
    if (tid == 0) {
        in_cnt[BX] = 0; s.in_n[2 * BX] = n; s.in_n[2 * BX + 1] = s.fs->pred_epoch; s.tile_live[BX] = 1;   // ready for the next frame; the tile holds particles now
        if (s.vis_bits && !was_live) atomicOr(&s.vis_bits[BX >> 5], 1u << (BX & 31));   // (tile bitmaps) ... this frame's resampling has to visit it
        if (s_cnt[0]) atomicAdd(&s.fs->n_place_vf, s_cnt[0]);   // (rare events: the frame's counts, reset with the pyramid lists)
        if (s_cnt[1]) atomicAdd(&s.fs->n_place_pf, s_cnt[1]);
    }
    PLSTAMP(5);
#ifdef PLACE_PROF
    if (threadIdx.x == 0 && BX < 131072) g_plprof[BX * 8 + 6] = n;
#endif
}

#ifndef PLACE_LB
#define PLACE_LB 7
#endif
template <int MW>
__global__ void __launch_bounds__(256, PLACE_LB) k_place(MapDims d, DevState s, const float4* __restrict__ in_rec,
                                               int* __restrict__ in_cnt, int has_vz, int tab_n,
                                               const u64* __restrict__ omask, FilterParams fp, float4* __restrict__ child,
                                               int* __restrict__ vb_cnt, int* __restrict__ vb_idx, int nchild, int t0, int n0, int t1, int n1,
                                               const int* __restrict__ tile_fov, int sel, float4* __restrict__ stage, int rev,
                                               const int* __restrict__ view_list, const int* __restrict__ tcls, int cls) {
    // tcls / cls (two-branch frame): only the tiles cls_mine() selects
    // whole frame: workgroups behind the tiles generate the frame's newborn children (k_birth_children's job; needs the
    // birth cloud and the rank only, both done before this launch)
    // (the first `nchild` workgroups: they run beside the tiles, not after them).
    // Tiles of this launch: [t0, t0 + n0) followed by [t1, t1 + n1) -- all of them, or (split-phase multi-GPU frame) the slab's
    // interior before the neighbour exchange and its boundary layers after it.  A launch with fewer workgroups than tiles
    // walks them with the grid's stride (the side-stream placement keeps a small footprint that way).
    if ((int)blockIdx.x < nchild) {
        birth_child_thread(d, s, fp, child, vb_cnt, vb_idx, (int)(blockIdx.x * 256 + threadIdx.x));
        return;
    }
    // What a workgroup needs to know to decide whether a tile is its business -- arrivals, and in a split placement the tile's view
    // tag -- comes in ONE scalar round trip; the tile's live / moving flags follow in a second one for the tiles it works on (round 5;
    // there were three dependent ones before the decision.  Fetching all four words at once costs the EMPTY tiles of a sparse map --
    // 75 of 87 k workgroups at 264x264x80 -- three more cache lines each: 48 -> 87 us for that launch).
    const int nt = n0 + n1, bq0 = (int)blockIdx.x - nchild, stride = (int)gridDim.x - nchild;
    if (sel == 1 && view_list) {
        // the tiles with a view, from k_predict's list (a frame that splits its placement): a few per cent of a large map's tiles, so the
        // launch is a few hundred workgroups that all have work instead of one per tile
        if (has_vz && bq0 == 0 && threadIdx.x == 0)   // k_predict drew 3 table values per ranked particle (:655-657)
            s.fs->v_cur = (int)(((long long)s.fs->v_cur + 3ll * (long long)s.fs->occupied_count) % tab_n);
        const int nv = sload_i(&s.fs->n_view_tiles);
        for (int p = bq0; p < nv; p += stride) {
            const int BX = sload_i(view_list + p);
            int n_in, tf, t_live, t_mov;
            sload_i4(in_cnt + BX, tile_fov + BX, s.tile_live + BX, s.tile_moving + BX, n_in, tf, t_live, t_mov);
            if (n_in == 0) continue;
            place_tile<MW>(d, s, in_rec, in_cnt, omask, stage, BX, n_in, t_live != 0, t_mov != 0, fp);
            __syncthreads();
        }
        return;
    }
    if (bq0 >= nt) return;
    const int epoch = sel >= 0 ? sload_i(&s.fpar->epoch) : 0;
    // (tile bitmaps) has the tile arrivals at all?  One cached word instead of a round trip to the tile's own count
    auto has_arrivals = [&](int BX) -> bool { return !s.arr_bits || ((sload_i(reinterpret_cast<const int*>(s.arr_bits) + (BX >> 5)) >> (BX & 31)) & 1); };
    int n_in, tf = 0;
    {   // the first tile's words; a workgroup whose ONLY tile has nothing for it leaves HERE, before the loop below is set up: the
        // invariants the compiler hoists in front of it (~100 vector instructions with the scalar registers it parks in lanes) were
        // what a sparse map's placement spent its time on -- 87 120 workgroups, 12 k with arrivals
        const int bqr = rev ? nt - 1 - bq0 : bq0;
        const int BX = bqr < n0 ? t0 + bqr : t1 + (bqr - n0);
        if (!has_arrivals(BX)) { if (bq0 + stride >= nt && !(has_vz && BX == 0)) return; n_in = 0; }
        else
        if (sel >= 0) sload_i2(in_cnt + BX, tile_fov + BX, n_in, tf); else if (tcls) sload_i2(in_cnt + BX, tcls + BX, n_in, tf); else n_in = sload_i(in_cnt + BX);
        if (bq0 + stride >= nt && !(has_vz && BX == 0)) {
            if (n_in == 0) return;
            if (tcls && !cls_mine(tf, cls)) return;
            if (sel >= 0 && (tf >> 1) == epoch && ((tf & 1) != 0) != (sel != 0)) return;
        }
    }
    for (int bq = bq0; bq < nt; bq += stride) {
        const int bqr = rev ? nt - 1 - bq : bq;                 // (the launch's tiles from the last one down: see k_predict)
        const int BX = bqr < n0 ? t0 + bqr : t1 + (bqr - n0);   // tile index
        if (has_vz && BX == 0 && sel != 0 && threadIdx.x == 0)   // k_predict drew 3 table values per ranked particle (:655-657)
            s.fs->v_cur = (int)(((long long)s.fs->v_cur + 3ll * (long long)s.fs->occupied_count) % tab_n);
        if (bq != bq0) { if (!has_arrivals(BX)) n_in = 0; else if (sel >= 0) sload_i2(in_cnt + BX, tile_fov + BX, n_in, tf); else if (tcls) sload_i2(in_cnt + BX, tcls + BX, n_in, tf); else n_in = sload_i(in_cnt + BX); }
        if (n_in == 0) continue;   // (no arrivals -- or the tile's owner is done with them)
        if (tcls && !cls_mine(tf, cls)) continue;   // (sel < 0 with tcls: tf holds the tile's class)
        if (sel >= 0) {   // a split placement: the other launch owns the tiles of the other kind
            int fv = tf & 1;
            // k_predict skipped the tile (empty) and particles arrive in it: its view is tested here, by both launches alike
            if ((tf >> 1) != epoch) fv = tile_view_test(d, s, BX, lane_id());
            if ((fv != 0) != (sel != 0)) continue;
        }
        int t_live, t_mov;
        sload_i2(s.tile_live + BX, s.tile_moving + BX, t_live, t_mov);
        place_tile<MW>(d, s, in_rec, in_cnt, omask, stage, BX, n_in, t_live != 0, t_mov != 0, fp);   // (the owner's view of in_cnt is stable: only the owner resets it)
        __syncthreads();   // the tile's LDS tables are re-used by the next one
    }
}

#define RO_INLINE_MAX 384   // moving particles of a tile beyond which the tile counts as "heavy" for the choice inline rollout / k_rollout
// rows per batch of the loads in k_resample (two batches in flight) = its template parameter RBK_: 8 on sparse maps (the launch is as long as
// its fullest tiles: fewer round trips per tile), 4 on dense ones (81 / 91 registers instead of 146 / 156: five waves per SIMD instead of
// three -- 264x264x80 saturated 0.713 -> 0.623 ms, 132x132x60 0.104 -> 0.100; the realistic fills lose 4 % with 4 and 30 - 45 % with 12 / 16)
#define CPB 8   // deferred copies per step

// --------------------------------------------------------------------------
// k_resample: mapOccupancyCalculationAndResample :924-1057.  One wave per tile, one lane per voxel.
//   * the first pass (cull / mass :938-984) streams weights and velocities through registers in batches of RBK_ rows, the next
//     batch requested before this one is consumed; the second pass (systematic resampling :986-1053) re-reads the weight rows
//     wave-uniformly, a batch ahead (no LDS weight panel since round 4: 16 instead of 10 / 6 resident tiles per CU);
//   * positions are read only for moving particles (noted for k_rollout :950-964) and for copies; the velocity rows of a tile
//     whose particles are all static are not fetched (DevState::tile_moving);
//   * all per-voxel sums run sequentially per lane in slot order = the reference's operation order.
// dynamic LDS per wave: [64][M] u16 (source slot, destination slot) of deferred copies
// --------------------------------------------------------------------------
template <int MW, int RBK_>
__global__ void __launch_bounds__(256, RBK_ >= 8 ? 3 : 5) k_resample(MapDims d, DevState s, int* __restrict__ part_live, int* __restrict__ vb_cnt,
                                                  float4* __restrict__ ro_rec, int* __restrict__ ro_cnt, int rev, const int* __restrict__ tcls, int cls,
                                                  int* __restrict__ ro_sub) {
    // ro_sub (cube storage): the tile's rollout records are written in FOUR runs, one per layer of the cube (lanes 16 s .. 16 s + 15 = layer
    // s; run s starts at record s * 16 * slots), ro_sub[4 * tile + s] = its length: k_rollout's workgroup of layer s reads its run only
    extern __shared__ float s_dyn[];
    // (DSPMAP_P_ESTIMATOR_QUEUE) the frame's birth stage ended where this launch began: the next frame's estimator may have the rand() cursor
    // and the birth buffers
    if (blockIdx.x == 0 && threadIdx.x == 0 && s.xq && s.fpar->from_ring) xq_publish(s.xq, (int)(s.fpar->ring_pos + 1u));
    const int l = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cpmax = d.M;   // a voxel makes at most M copies
    unsigned short* s_cp = (unsigned short*)(s_dyn + (size_t)wave * ((64 * cpmax + 1) / 2));
    const int wq = blockIdx.x * (blockDim.x >> 6) + wave;
    const int wave_g = (rev & 1) ? ((d.v_loc + 63) >> 6) - 1 - wq : wq;   // (tiles from the last one down: see k_predict)
    const bool cubes = d.tiling != 0;
    const int lv = wave_g * 64 + l;
    if (wave_g < 0 || wave_g * 64 >= d.v_loc) return;
    if (s.vis_bits && !((sload_i(reinterpret_cast<const int*>(s.vis_bits) + (wave_g >> 5)) >> (wave_g & 31)) & 1)) return;   // (tile bitmaps) empty, and nothing arrived or was born
    int t_live, t_mov, tc;
    if (rev & 6) {
        // (DSPMAP_P_RESAMPLE_SPLIT) the stage in two launches: the tiles no newborn can reach (not Q) BESIDE the weight update and the births
        // (rev & 2), the others behind the births (rev & 4).  Q is cut around THIS frame's field of view; a frame whose view is empty
        // re-uses the birth cloud of the last non-empty one (:1379-1381), whose newborns may land anywhere: the early launch leaves
        // such a frame alone and the late one takes every tile.  (n_valid: final since k_predict's gather.)
        int nval;
        sload_i4(s.tile_live + wave_g, s.tile_moving + wave_g, tcls + wave_g, &s.fs->n_valid, t_live, t_mov, tc, nval);
        if (nval == 0) { if (rev & 2) return; }
        else if (!cls_mine(tc, cls)) return;
    } else {
    sload_i3(s.tile_live + wave_g, s.tile_moving + wave_g, tcls ? tcls + wave_g : s.tile_live + wave_g, t_live, t_mov, tc);   // (one scalar round trip)
    if (tcls && !cls_mine(tc, cls)) return;   // (two-branch frame) the other branch's tile
    }
    if (!t_live) return;   // empty since its last visit: result, buckets and lists are already zero
    if (!d.tile_skip) t_mov = 1;
    const bool inr = lv < d.v_loc;
    const int lvs = inr ? lv : 0;
    u64 m[MW], nb[MW];
    bool nonempty = false;
#pragma unroll
    for (int e = 0; e < MW; ++e) {
        m[e] = 0ull; nb[e] = 0ull;
        if (inr) {
            nb[e] = s.nbmask[(size_t)lv * MW + e];
            m[e] = s.mask[(size_t)lv * MW + e] | nb[e];  // newborns live only in nbmask until now
        }
        nonempty |= m[e] != 0ull;
    }
    if (inr) vb_cnt[lv] = 0;    // birth buckets of this frame are consumed: leave them empty for the next one
    if (!__ballot(nonempty)) {  // whole tile empty
        if (inr) s.res4[lv] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l == 0 && wave_g * 64 < d.v_loc + 63) { part_live[wave_g] = 0; ro_cnt[wave_g] = 0; s.tile_live[wave_g] = 0; }
        if (cubes && l < 4) ro_sub[4 * wave_g + l] = 0;
        return;
    }
    int nmv = 0;   // moving old particles of the tile noted for k_rollout so far (wave-uniform)
    int nmv_s[4] = {0, 0, 0, 0};   // (cube storage) ... per layer of the cube
    const int sub_l = l >> 4, run_len = 16 * d.slots;
    float mvw = 0.f;   // ... and this lane's share of their weight (k_rollout scales its fixed-point windows with the total)
    const size_t ro_base = (size_t)wave_g * 64 * d.slots;
    int n = 0, n_old = 0;
    float wsum = 0.f, vxs = 0.f, vys = 0.f, stat_w = 0.f;
    // rows stream through registers in batches of RBK_; the loads of the NEXT batch are issued before this one is consumed
    // (two register sets of RBK_ rows each: the wave's memory round trip hides behind the sequential per-voxel sums)
    struct RowBatch { int row[RBK_]; V2 vv[RBK_]; float wr[RBK_]; };
#pragma unroll
    for (int e = 0; e < MW; ++e) {
        u64 tor = wave_or_u64(m[e]);
        const brsrc rs_vel = __builtin_amdgcn_make_buffer_rsrc((void*)(s.vel + 2 * ((size_t)wave_g * d.slots * 64)), 0, d.slots * 64 * 8, 0x00020000);
        const brsrc rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(s.w + (size_t)wave_g * d.slots * 64), 0, d.slots * 64 * 4, 0x00020000);
        auto issue = [&](RowBatch& B) {
#pragma unroll
            for (int r = 0; r < RBK_; ++r) {
                B.row[r] = tor ? __ffsll((long long)tor) - 1 : -1;
                if (tor) tor &= tor - 1ull;
                // unconditional loads, see k_predict (the row enters as a wave-uniform scalar offset)
                const int srow = (e * 64 + (B.row[r] < 0 ? 0 : B.row[r])) * 64;
                B.wr[r] = bl_w(rs_w, l, srow);
                B.vv[r].x = 0.f; B.vv[r].y = 0.f;
                if (t_mov) B.vv[r] = bl_vel(rs_vel, l, srow);   // (a tile of static particles: its velocity rows are not fetched)
            }
        };
        auto consume = [&](const RowBatch& B) {
            bool act[RBK_];
            float vx[RBK_], vy[RBK_];
            bool any_mv = false;
#pragma unroll
            for (int r = 0; r < RBK_; ++r) {
                act[r] = B.row[r] >= 0 && ((m[e] >> (B.row[r] & 63)) & 1ull);
                const bool old = act[r] && !((nb[e] >> (B.row[r] & 63)) & 1ull);
                vx[r] = old ? B.vv[r].x : 0.f; vy[r] = old ? B.vv[r].y : 0.f;
                any_mv |= vx[r] != 0.f || vy[r] != 0.f;
            }
            // the rollout (:950-964) needs the position of the MOVING old particles only: their (x, y) are requested for the
            // whole batch at once -- one extra memory round trip per batch that holds a moving particle instead of one per
            // moving particle inside the sequential loop below
            float mpx[RBK_], mpy[RBK_];
            const bool batch_mv = __ballot(any_mv) != 0ull;
            if (batch_mv) {
#pragma unroll
                for (int r = 0; r < RBK_; ++r) {
                    mpx[r] = 0.f; mpy[r] = 0.f;
                    if (vx[r] != 0.f || vy[r] != 0.f) {
                        const float2 q = *reinterpret_cast<const float2*>(s.pos + 3 * pidx(d, lvs, e * 64 + B.row[r]));
                        mpx[r] = q.x; mpy[r] = q.y;
                    }
                }
            }
            bool mv_now[RBK_];
#pragma unroll
            for (int r = 0; r < RBK_; ++r) mv_now[r] = false;
#pragma unroll
            for (int r = 0; r < RBK_; ++r) {
                if (!act[r]) continue;
                const u64 bit = 1ull << B.row[r];
                const float w = B.wr[r];
                if (w < 1e-3f) {                  // :941
                    m[e] &= ~bit;
                } else {
                    if (!(nb[e] & bit)) {         // flag < 10 :944
                        ++n_old;
                        vxs += vx[r]; vys += vy[r];
                        if (vx[r] == 0.f && vy[r] == 0.f) stat_w += w;   // p + 0*t stays in this voxel for every horizon
                        else mv_now[r] = true;                           // the T future positions are k_rollout's job
                    }
                    ++n;
                    wsum += w;                    // :970
                }
            }
            // the batch's moving old particles join the tile's rollout list (stable wave-level append, no atomics: one
            // wave owns the tile)
            if (batch_mv) {
#pragma unroll
                for (int r = 0; r < RBK_; ++r) {
                    const u64 mb = __ballot(mv_now[r]);
                    if (mv_now[r]) {
                        int at = nmv + (int)__popcll(mb & lanemask_lt());
                        if (cubes) {
                            const u64 ms = 0xffffull << (16 * sub_l);
                            const int before = sub_l == 0 ? nmv_s[0] : (sub_l == 1 ? nmv_s[1] : (sub_l == 2 ? nmv_s[2] : nmv_s[3]));
                            at = sub_l * run_len + before + (int)__popcll(mb & ms & lanemask_lt());
                        }
                        const size_t o = (ro_base + at) * 2;
                        ro_rec[o] = make_float4(mpx[r], mpy[r], vx[r], vy[r]);
                        ro_rec[o + 1] = make_float4(B.wr[r], __int_as_float(lv), 0.f, 0.f);
                        mvw += B.wr[r];
                    }
                    nmv += (int)__popcll(mb);
                    if (cubes) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) nmv_s[q] += (int)__popcll(mb & (0xffffull << (16 * q)));
                    }
                }
            }
        };
        if (!tor) continue;
        RowBatch A, B;
        issue(A);
        for (;;) {   // (tor is wave-uniform: scalar control flow)
            const bool more_b = tor != 0ull;
            if (more_b) issue(B);
            consume(A);
            if (!more_b) break;
            const bool more_a = tor != 0ull;
            if (more_a) issue(A);
            consume(B);
            if (!more_a) break;
        }
    }
    if (nmv) mvw = wave_sum_f(mvw);
    if (l == 0) {
        ro_cnt[wave_g] = nmv; ro_cnt[((d.v_loc + 63) >> 6) + wave_g] = __float_as_int(mvw);
        if (cubes) { ro_sub[4 * wave_g] = nmv_s[0]; ro_sub[4 * wave_g + 1] = nmv_s[1]; ro_sub[4 * wave_g + 2] = nmv_s[2]; ro_sub[4 * wave_g + 3] = nmv_s[3]; }
        if (nmv > RO_INLINE_MAX && (wave_g & 15) == 0) atomicAdd(&s.fs->mv_acc, 16);   // (the caller's hint, from every 16th tile: with many such tiles k_rollout's LDS windows pay)
    }
    if (inr) {
        float4 res = make_float4(wsum, 0.f, 0.f, 0.f);  // voxels_objects_number[v][0..3] :974-984
        if (n_old > 0) { res.y = __fdiv_rn(vxs, (float)n_old); res.z = __fdiv_rn(vys, (float)n_old); }
        s.res4[lv] = res;
        if (stat_w != 0.f) s.fut_stat[lv] += stat_w;  // only this lane ever writes fut_stat[lv]
    }
    if (__ballot(inr && stat_w != 0.f) && l == 0) s.fut_dirty[wave_g] = 1;   // (k_predict zeroes the tile's accumulators on the next clear)
    // ---- systematic resampling :986-1053.  The walk needs every weight a second time, in slot order: the rows are re-read (they
    // left this CU's caches microseconds ago at worst to the Infinity Cache), wave-uniformly and a batch ahead of the walk, instead
    // of being kept in an LDS panel of [slots][64] floats -- that panel (12 - 18 kB per one-wave workgroup) was what bounded the
    // residency of this kernel at 10 / 6 tiles per CU; without it the registers do (16)
    int ncp = 0;
    float w_copy = 0.f;
    const bool resample = n >= 5;
    {
        const int n_after = n > d.M ? d.M : n;                                       // :992-997
        const float w_after = resample ? __fdiv_rn(wsum, (float)n_after) : 0.f;      // :1000
        w_copy = w_after;
        float acc_ori = 0.f, acc_new = w_after * 0.5f;                               // :1005-1006
        // survivors before ANY copy is placed: copies carry flag 0.6 and are not revisited (:1009), also those that
        // land in a later occupancy word than the one being walked
        u64 surv[MW];
#pragma unroll
        for (int e = 0; e < MW; ++e) surv[e] = resample ? m[e] : 0ull;
        struct WBatch { int row[RBK_]; float wr[RBK_]; };
#pragma unroll
        for (int e = 0; e < MW; ++e) {
            u64 tor = wave_or_u64(surv[e]);   // rows in which ANY voxel of the tile walks a particle (wave-uniform)
            if (!tor) continue;
            const brsrc rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(s.w + (size_t)wave_g * d.slots * 64), 0, d.slots * 64 * 4, 0x00020000);
            auto issue = [&](WBatch& B) {
#pragma unroll
                for (int r = 0; r < RBK_; ++r) {
                    B.row[r] = tor ? __ffsll((long long)tor) - 1 : -1;
                    if (tor) tor &= tor - 1ull;
                    B.wr[r] = bl_w(rs_w, l, (e * 64 + (B.row[r] < 0 ? 0 : B.row[r])) * 64);
                }
            };
            auto consume = [&](const WBatch& B) {
#pragma unroll
                for (int r = 0; r < RBK_; ++r) {
                    if (B.row[r] < 0) continue;                     // (scalar)
                    const int row = B.row[r];
                    const u64 bit = 1ull << row;
                    if (!(surv[e] & bit)) continue;                 // this voxel holds no survivor in the row
                    acc_ori += B.wr[r];                             // :1011
                    if (acc_ori > acc_new) {
                        float wn = w_after;                         // keep, new weight :1014
                        acc_new += w_after;
                        bool full = false;
                        while (acc_ori > acc_new) {                 // copy heavy particles :1021
                            int fslot = -1;
                            if (!full) {
#pragma unroll
                                for (int e2 = 0; e2 < MW; ++e2) {
                                    const u64 fr = ~m[e2] & valid_bits(d, e2);
                                    if (fslot < 0 && fr) { fslot = e2 * 64 + (__ffsll((long long)fr) - 1); m[e2] |= fr & (~fr + 1ull); }
                                }
                            }
                            if (fslot >= 0) {
                                // the copy itself (5 loads + 6 stores) is deferred: only (source, destination)
                                // is noted here so that no global access sits in this sequential loop
                                if (ncp < cpmax) s_cp[l * cpmax + ncp] = (unsigned short)(((e * 64 + row) << 8) | fslot);
                                ++ncp;
                            } else {
                                wn += w_after;                      // no free slot: fold the weight back :1037-1041
                                full = true;
                            }
                            acc_new += w_after;
                        }
                        s.w[pidx(d, lvs, e * 64 + row)] = wn;
                    } else {
                        m[e] &= ~bit;                               // remove :1046-1049
                    }
                }
            };
            WBatch A, B;
            issue(A);
            for (;;) {
                const bool more_b = tor != 0ull;
                if (more_b) issue(B);
                consume(A);
                if (!more_b) break;
                const bool more_a = tor != 0ull;
                if (more_a) issue(A);
                consume(B);
                if (!more_a) break;
            }
        }
    }
    // deferred copies :1026-1031, CPB at a time so that their loads overlap
    if (ncp > cpmax) ncp = cpmax;  // cannot happen: a voxel makes at most M copies
    for (int k0 = 0; __ballot(k0 < ncp); k0 += CPB) {
        P3 cp[CPB];
        V2 cv[CPB];
        float cvz[CPB];
        unsigned didx[CPB];
#pragma unroll
        for (int j = 0; j < CPB; ++j) {
            didx[j] = 0;
            if (k0 + j < ncp) {
                const unsigned pr = s_cp[l * cpmax + k0 + j];
                const size_t sidx = pidx(d, lvs, (int)(pr >> 8));
                didx[j] = (unsigned)pidx(d, lvs, (int)(pr & 0xff));
                cp[j] = ld_pos(s, sidx);
                cv[j].x = 0.f; cv[j].y = 0.f;
                if (t_mov) cv[j] = ld_vel(s, sidx);   // (static tile: every velocity cell is (0, 0) already)
                cvz[j] = s.vz0 ? s.vz0[sidx] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < CPB; ++j) {
            if (k0 + j < ncp) {
                st_pos(s, didx[j], cp[j].x, cp[j].y, cp[j].z);
                if (t_mov) st_vel(s, didx[j], cv[j].x, cv[j].y);
                if (s.vz0) s.vz0[didx[j]] = cvz[j];
                s.w[didx[j]] = w_copy;
            }
        }
    }
    int live_out = 0;
    if (inr) {
#pragma unroll
        for (int e = 0; e < MW; ++e) {
            live_out += (int)__popcll(m[e]);
            s.mask[(size_t)lv * MW + e] = m[e];
            if (nb[e]) s.nbmask[(size_t)lv * MW + e] = 0ull;  // newborn flag -> 1 (:968)
        }
    }
    live_out = wave_sum_i(live_out);
    if (l == 0) {
        part_live[wave_g] = live_out; s.tile_live[wave_g] = live_out > 0 ? 1 : 0;
        if ((wave_g & 63) == 0 && live_out > 0) atomicAdd(&s.fs->live_acc, 1);   // (a 1-in-64 sample of the non-empty tiles: k_predict's hint)
    }
}

// --------------------------------------------------------------------------
// One moving old particle's future status (:950-964), one integer atomic per horizon: record {px, py, vx, vy}, {w, local voxel}.
// (k_rollout's path for tiles with few moving particles; k_resample_wg's waves 1-3 run it for their tile while wave 0 resamples.)
// The accumulators are FIXED-POINT (fut_quantum, dspmap_device.h): every particle adds the same integer whichever path carries it
// -- this one, k_rollout's LDS windows, a sharded or an unsharded map -- and integer sums do not depend on the order of the adds,
// so the future status is reproducible bit for bit (the reference's own `+=` is a sequential loop, :961).
// the layer (relative to the slab) of storage voxel lv: a particle never leaves it within a frame's horizons (vz == 0)
__device__ __forceinline__ int layer_of_lv(const MapDims& d, int lv) {
    return d.tiling ? ((lv >> 6) / (d.ncx * d.ncy)) * 4 + ((lv >> 4) & 3) : lv / (d.ny * d.nx);
}
__device__ __forceinline__ void rollout_direct(const MapDims& d, const DevState& s, const float4 a, const float4 b) {
    const size_t V = (size_t)d.v_loc;
    const int zl = layer_of_lv(d, __float_as_int(b.y));
    const u64 q = fut_quantum(b.x);
    for (int t = 0; t < d.T; ++t) {
        const float pt = d.pred_t[t];
        const float fx = a.x + a.z * pt;      // :954-955
        const float fy = a.y + a.w * pt;
        if (fabsf(fx) >= d.half_x || fabsf(fy) >= d.half_y) continue;
        const int xi = (int)div_res(d, fx + d.half_x);
        const int yi = (int)div_res(d, fy + d.half_y);
        const int dl = lv_of_xyz(d, xi, yi, zl);
        if (dl < 0 || dl >= d.v_loc) continue;
        fut_add(&s.fut[(size_t)t * V + dl], q);
        s.fut_dirty[dl >> 6] = 1;
    }
}

// k_resample_wg: the same stage with FOUR waves per tile, for maps whose frame is a chain of latencies (the metric's size:
// a few hundred live tiles, one wave per SIMD -- the longest tile IS the kernel; one-word occupancy, slots <= 48).
// Only the per-voxel sums and the resampling walk are sequential in slot order (:938-1053); everything around them is not:
//   phase 1  all four waves load the tile's live rows in ONE batch (weights, velocities, x / y; rows split over the waves),
//            cull (:941) and COMPACT: every voxel's surviving particles go to consecutive entries j = 0 .. n-1 of the
//            LDS panels in slot order (j = number of the voxel's survivors in lower slots), and the MOVING old ones are
//            noted for k_rollout;
//   phase 2  one wave, one lane per voxel, walks ITS OWN n entries -- mass / mean velocity, then the systematic resampling --
//            in batches of four LDS reads: the reference's operation order, bit for bit; the loop runs as long as the tile's
//            fullest voxel, not as long as its highest occupied slot, and has no memory access in the chain;
//   phase 3  all four waves write the kept particles' new weights back and carry out the deferred copies.
// dynamic LDS: [slots][64] fp32 panel w + [slots][64] u8 slot of entry j + [64][M] u16 copy notes.
// --------------------------------------------------------------------------
#ifdef RESAMPLE_PROF
__device__ long long g_rprof[4 * 65536];
extern "C" int dspmap_debug_resample_prof(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rprof), sizeof(long long) * (size_t)n); }
#endif
#define RWB 12  // rows per wave, one occupancy word: 4 x 12 = 48 slots, the whole tile in one batch
#define RWB2 18 // ... two occupancy words (72 slots, config E's 36 particles per voxel): 4 x 18
// MW occupancy words per voxel (round 6: the two-word instantiation -- the depth-stream fill of the 264x264x80 map ran the one-wave
// k_resample<2, 8> for want of it: 77 us for 0.5 M particles, profiles/r06_*).  Bit r of a voxel's words = slot r.
template <int MW> __device__ __forceinline__ bool wbit(const u64 (&w)[MW], int r) { return ((w[MW == 1 ? 0 : (r >> 6)] >> (r & 63)) & 1ull) != 0ull; }
template <int MW> __device__ __forceinline__ int wbelow(const u64 (&w)[MW], int r) {   // set bits below slot r
    if (MW == 1) return (int)__popcll(w[0] & ((1ull << r) - 1ull));
    return r < 64 ? (int)__popcll(w[0] & ((1ull << r) - 1ull)) : (int)__popcll(w[0]) + (int)__popcll(w[MW - 1] & ((1ull << (r & 63)) - 1ull));
}
template <int MW> __device__ __forceinline__ int wcount(const u64 (&w)[MW]) { int n = 0; for (int e = 0; e < MW; ++e) n += (int)__popcll(w[e]); return n; }
template <int MW>
__global__ void __launch_bounds__(256) k_resample_wg(MapDims d, DevState s, int* __restrict__ part_live, int* __restrict__ vb_cnt,
                                                     float4* __restrict__ ro_rec, int* __restrict__ ro_cnt, int inline_ro, int* __restrict__ ro_sub) {
    constexpr int RW = MW == 1 ? RWB : RWB2;
    extern __shared__ float s_dyn[];
    __shared__ u64 s_surv[MW * 64], s_oldc[MW * 64];
    __shared__ int s_ncp[64];
    __shared__ float s_wcp[64];
    __shared__ int s_nmv;
    __shared__ float s_mvw[4];   // weight of the moving old particles each wave noted
    // (DSPMAP_P_ESTIMATOR_QUEUE) the frame's birth stage ended where this launch began: the next frame's estimator may have the rand() cursor
    // and the birth buffers
    if (blockIdx.x == 0 && threadIdx.x == 0 && s.xq && s.fpar->from_ring) xq_publish(s.xq, (int)(s.fpar->ring_pos + 1u));
    const int l = lane_id();
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int BX = (int)blockIdx.x;
    const int cells = d.slots * 64;
    float* cw = s_dyn;
    unsigned char* cs = (unsigned char*)(cw + cells);
    const int cpmax = d.M;   // a voxel makes at most M copies
    unsigned short* s_cp = (unsigned short*)(cs + cells);
    const int lv = BX * 64 + l;
    const bool inr = lv < d.v_loc;
    if (s.vis_bits && !((sload_i(reinterpret_cast<const int*>(s.vis_bits) + (BX >> 5)) >> (BX & 31)) & 1)) return;   // (tile bitmaps) empty, and nothing arrived or was born
    const int t_live = s.tile_live[BX];   // (requested together with the occupancy words: one round trip)
    u64 nb[MW], m[MW];
    bool any_m = false, any_nb = false;
#pragma unroll
    for (int e = 0; e < MW; ++e) {
        nb[e] = 0ull; m[e] = 0ull;
        if (inr) {
            nb[e] = s.nbmask[(size_t)lv * MW + e];
            m[e] = s.mask[(size_t)lv * MW + e] | nb[e];  // newborns live only in nbmask until now
        }
        any_m |= m[e] != 0ull; any_nb |= nb[e] != 0ull;
    }
    if (!t_live) return;   // empty since its last visit: result, buckets and lists are already zero
#ifdef RESAMPLE_PROF
    long long t_s0 = __builtin_readcyclecounter(), t_s1 = 0, t_s2 = 0, t_s3 = 0, t_s4 = 0;
#endif
    if (wave == 0 && inr) vb_cnt[lv] = 0;    // birth buckets of this frame are consumed: leave them empty for the next one
    if (tid == 0) s_nmv = 0;
    if (tid < 64) {
#pragma unroll
        for (int e = 0; e < MW; ++e) { s_surv[e * 64 + tid] = 0ull; s_oldc[e * 64 + tid] = 0ull; }
    }
    if (!__ballot(any_m)) {  // whole tile empty (the same answer in every wave)
        if (wave == 0) {
            if (inr) s.res4[lv] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l == 0) { part_live[BX] = 0; ro_cnt[BX] = 0; s.tile_live[BX] = 0; if (d.tiling) ro_sub[4 * BX] = -1; }
        }
        return;
    }
    const size_t tcell = (size_t)BX * cells;
    const brsrc rs_pos = __builtin_amdgcn_make_buffer_rsrc((void*)(s.pos + 3 * tcell), 0, cells * 12, 0x00020000);
    const brsrc rs_vel = __builtin_amdgcn_make_buffer_rsrc((void*)(s.vel + 2 * tcell), 0, cells * 8, 0x00020000);
    const brsrc rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(s.w + tcell), 0, cells * 4, 0x00020000);
    // ---- phase 1: this wave's rows, all in flight together
    int row[RW];
    V2 vv[RW];
    float wr[RW];
    float2 pq[RW];
    {
        // every fourth live row of the tile, starting at `wave`, counted across the occupancy words
        u64 tor[MW];
        int kk = 0;
#pragma unroll
        for (int e = 0; e < MW; ++e) {
            u64 t = wave_or_u64(m[e]), mine = 0ull;
            while (t) {
                const u64 low = t & (~t + 1ull);
                if ((kk & 3) == wave) mine |= low;
                t ^= low;
                ++kk;
            }
            tor[e] = mine;
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            row[r] = -1;
#pragma unroll
            for (int e = 0; e < MW; ++e)
                if (row[r] < 0 && tor[e]) { row[r] = e * 64 + (__ffsll((long long)tor[e]) - 1); tor[e] &= tor[e] - 1ull; }
            const int srow = (row[r] < 0 ? 0 : row[r]) * 64;
            // only the voxel's LIVE cells are fetched: a lane whose cell is empty asks for an offset beyond the tile's descriptor --
            // the buffer unit returns 0 for it without touching memory and without a branch (a predicated load would make the
            // compiler wait before it issues the next row's).  A tile of the metric's map is 8 % full: its rows were 37 MB per launch
            const bool cell = row[r] >= 0 && wbit<MW>(m, row[r]);
            const int ln = cell ? l : (1 << 24);
            wr[r] = bl_w(rs_w, ln, srow);
            vv[r] = bl_vel(rs_vel, ln, srow);
            // (x, y) of every cell too: only the moving particles need them (rollout), but asking afterwards would be a second
            // dependent round trip
            const f2w q = __builtin_bit_cast(f2w, __builtin_amdgcn_raw_buffer_load_b64(rs_pos, ln * 12, srow * 12, 0));
            pq[r] = make_float2(q.x, q.y);
        }
    }
    __syncthreads();   // (s_surv / s_oldc / s_nmv are zero)
    // which cells survive the cull (:941), per voxel
    u64 sv_mine[MW];
#pragma unroll
    for (int e = 0; e < MW; ++e) sv_mine[e] = 0ull;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        if (row[r] < 0) continue;
        const int rw = row[r];
        if (wbit<MW>(m, rw) && !(wr[r] < 1e-3f)) sv_mine[MW == 1 ? 0 : (rw >> 6)] |= 1ull << (rw & 63);
    }
#pragma unroll
    for (int e = 0; e < MW; ++e) if (sv_mine[e]) atomicOr(&s_surv[e * 64 + l], sv_mine[e]);
    __syncthreads();
#ifdef RESAMPLE_PROF2
    const long long t_sa = __builtin_readcyclecounter();
#endif
    u64 surv[MW];   // the voxel's survivors, all rows
#pragma unroll
    for (int e = 0; e < MW; ++e) surv[e] = s_surv[e * 64 + l];
    const size_t ro_base = (size_t)BX * cells;
    u64 oldc_mine[MW];
#pragma unroll
    for (int e = 0; e < MW; ++e) oldc_mine[e] = 0ull;
    float mvw = 0.f;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        if (row[r] < 0) continue;
        const int rw = row[r];
        const bool on = wbit<MW>(sv_mine, rw);
        const bool old = on && !wbit<MW>(nb, rw);
        const float vx = old ? vv[r].x : 0.f, vy = old ? vv[r].y : 0.f;
        if (on) {
            const int j = wbelow<MW>(surv, rw);   // survivors of this voxel in lower slots
            const int c = j * 64 + l;
            cw[c] = wr[r]; cs[c] = (unsigned char)rw;
            if (old) oldc_mine[MW == 1 ? 0 : (j >> 6)] |= 1ull << (j & 63);
        }
        // the rollout (:950-964) needs the MOVING old survivors: noted here, their future positions are k_rollout's job
        const bool mv = vx != 0.f || vy != 0.f;
        const int k = lds_agg_inc(&s_nmv, mv);
        if (k >= 0) {
            const size_t o = (ro_base + k) * 2;
            ro_rec[o] = make_float4(pq[r].x, pq[r].y, vx, vy);
            ro_rec[o + 1] = make_float4(wr[r], __int_as_float(lv), 0.f, 0.f);
            mvw += wr[r];
        }
    }
    mvw = wave_sum_f(mvw);
    if (l == 0) s_mvw[wave] = mvw;
#pragma unroll
    for (int e = 0; e < MW; ++e) if (oldc_mine[e]) atomicOr(&s_oldc[e * 64 + l], oldc_mine[e]);
    __syncthreads();
#ifdef RESAMPLE_PROF
    t_s1 = __builtin_readcyclecounter();
#endif
    // ---- phase 2: one lane per voxel over its own n entries, LDS only
    if (wave == 0) {
        if (l == 0) {
            ro_cnt[BX] = inline_ro ? 0 : s_nmv; ro_cnt[((d.v_loc + 63) >> 6) + BX] = __float_as_int((s_mvw[0] + s_mvw[1]) + (s_mvw[2] + s_mvw[3]));
            if (d.tiling) ro_sub[4 * BX] = -1;   // (cube storage) ONE run of records, layers mixed: k_rollout's layer workgroups pick theirs out
            if (s_nmv > RO_INLINE_MAX && (BX & 15) == 0) atomicAdd(&s.fs->mv_acc, 16);   // (the caller's hint, from every 16th tile: with many such tiles k_rollout's LDS windows pay)
        }
        // (the voxel's mass only: the walk needs it; mean velocity, static future mass and the result record are wave 1's job below --
        // the same sums in the same order, off this wave's chain)
        const int n = wcount<MW>(surv);
        int nmax = n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, WAVE));
        float wsum = 0.f;
        for (int j0 = 0; j0 < nmax; j0 += 4) {
            float w4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[q] = cw[min(j0 + q, d.slots - 1) * 64 + l];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j0 + q < n) wsum += w4[q];                // :970
        }
#ifdef RESAMPLE_PROF
        t_s2 = __builtin_readcyclecounter();
#endif
        // systematic resampling :986-1053
        int ncp = 0;
        float w_copy = 0.f;
        u64 mfin[MW];
#pragma unroll
        for (int e = 0; e < MW; ++e) mfin[e] = surv[e];
        if (n >= 5) {
            const int n_after = n > d.M ? d.M : n;                  // :992-997
            const float w_after = __fdiv_rn(wsum, (float)n_after);  // :1000
            w_copy = w_after;
            float acc_ori = 0.f, acc_new = w_after * 0.5f;          // :1005-1006
            // (copies go to the free slots of `mfin` and are not revisited: the walk covers the n survivors only, :1009)
            for (int j0 = 0; j0 < n; j0 += 4) {
                float w4[4];
                int sl4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int c = min(j0 + q, d.slots - 1) * 64 + l; w4[q] = cw[c]; sl4[q] = (int)cs[c]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (j0 + q >= n) continue;
                    acc_ori += w4[q];                               // :1011
                    if (acc_ori > acc_new) {
                        float wn = w_after;                         // keep, new weight :1014
                        acc_new += w_after;
                        bool full = false;
                        while (acc_ori > acc_new) {                 // copy heavy particles :1021
                            int fslot = -1;
                            if (!full) {
#pragma unroll
                                for (int e = 0; e < MW; ++e) {
                                    const u64 fr = ~mfin[e] & valid_bits(d, e);
                                    if (fslot < 0 && fr) { fslot = e * 64 + (__ffsll((long long)fr) - 1); mfin[e] |= fr & (~fr + 1ull); }
                                }
                            }
                            if (fslot >= 0) {
                                // the copy itself is deferred: only (source, destination) is noted here
                                if (ncp < cpmax) s_cp[l * cpmax + ncp] = (unsigned short)((sl4[q] << 8) | fslot);
                                ++ncp;
                            } else {
                                wn += w_after;                      // no free slot: fold the weight back :1037-1041
                                full = true;
                            }
                            acc_new += w_after;
                        }
                        s.w[tcell + (size_t)sl4[q] * 64 + l] = wn;   // (a store the walk does not wait for; the panels stay as wave 1 reads them)
                    } else {
                        mfin[MW == 1 ? 0 : (sl4[q] >> 6)] &= ~(1ull << (sl4[q] & 63));   // remove :1046-1049
                    }
                }
            }
        }
        if (ncp > cpmax) ncp = cpmax;  // cannot happen: a voxel makes at most M copies
        s_ncp[l] = ncp;
        s_wcp[l] = w_copy;
        if (inr) {
#pragma unroll
            for (int e = 0; e < MW; ++e) {
                s.mask[(size_t)lv * MW + e] = mfin[e];
                if (nb[e]) s.nbmask[(size_t)lv * MW + e] = 0ull;  // newborn flag -> 1 (:968)
            }
        }
        const int live_out = wave_sum_i(inr ? wcount<MW>(mfin) : 0);
        if (l == 0) {
            part_live[BX] = live_out; s.tile_live[BX] = live_out > 0 ? 1 : 0;
            if ((BX & 63) == 0 && live_out > 0) atomicAdd(&s.fs->live_acc, 1);   // (a 1-in-64 sample of the non-empty tiles: k_predict's hint)
        }
    } else {
        if (wave == 1) {
            // the voxel's result record (:970-984) and its static future mass, beside wave 0's walk (which leaves the compacted
            // panel untouched: it stores the new weights straight to their cells).  The velocities of the entries are re-read from
            // their cells (eight at a time, warm in the L2 since phase 1; nobody writes them before phase 3): keeping them in LDS
            // beside the weights made the panels 42 kB per tile, three tiles per CU; 18 kB lets the registers decide (five)
            u64 oldc[MW];
#pragma unroll
            for (int e = 0; e < MW; ++e) oldc[e] = s_oldc[e * 64 + l];
            const int n = wcount<MW>(surv);
            int nmax = n;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, WAVE));
            int n_old = 0;
            float wsum = 0.f, vxs = 0.f, vys = 0.f, stat_w = 0.f;
            for (int j0 = 0; j0 < nmax; j0 += 8) {
                float w8[8];
                V2 v8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = min(j0 + q, d.slots - 1) * 64 + l;
                    w8[q] = cw[c];
                    const int sl = j0 + q < n ? (int)cs[c] : 0;
                    const f2w v = __builtin_bit_cast(f2w, __builtin_amdgcn_raw_buffer_load_b64(rs_vel, (sl * 64 + l) * 8, 0, 0));
                    v8[q].x = v.x; v8[q].y = v.y;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (j0 + q < n) {
                        if (wbit<MW>(oldc, j0 + q)) {             // flag < 10 :944
                            ++n_old;
                            vxs += v8[q].x; vys += v8[q].y;
                            if (v8[q].x == 0.f && v8[q].y == 0.f) stat_w += w8[q];   // p + 0*t stays in this voxel for every horizon
                        }
                        wsum += w8[q];                            // :970
                    }
                }
            }
            if (inr) {
                float4 res = make_float4(wsum, 0.f, 0.f, 0.f);  // voxels_objects_number[v][0..3] :974-984
                if (n_old > 0) { res.y = __fdiv_rn(vxs, (float)n_old); res.z = __fdiv_rn(vys, (float)n_old); }
                s.res4[lv] = res;
                if (stat_w != 0.f) s.fut_stat[lv] += stat_w;  // only this lane ever writes fut_stat[lv]
            }
            if (__ballot(inr && stat_w != 0.f) && l == 0) s.fut_dirty[BX] = 1;   // (k_predict zeroes the tile's accumulators on the next clear)
        }
        if (inline_ro) {
            // the rollout of the tile's moving old particles (noted above, visible after the barrier) while wave 0 walks the voxels:
            // a map of this size is a chain of short kernels, and k_rollout's launch (~7 us) costs more than these atomics
            const int nmv = s_nmv;
            for (int it = tid - 64; it < nmv; it += 192) rollout_direct(d, s, ro_rec[(ro_base + it) * 2], ro_rec[(ro_base + it) * 2 + 1]);
        }
    }
    __syncthreads();
#ifdef RESAMPLE_PROF
    t_s3 = __builtin_readcyclecounter();
#endif
    // ---- phase 3: the copies (:1026-1031), split over the waves (the kept particles' new weights were stored by the walk)
    const int ncp = s_ncp[l];
    const float w_copy = s_wcp[l];
    int maxcp = ncp;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxcp = max(maxcp, __shfl_xor(maxcp, o, WAVE));
    for (int k = wave; k < maxcp; k += 4) {
        if (k < ncp) {
            const unsigned pr = s_cp[l * cpmax + k];
            const size_t sidx = tcell + (size_t)(pr >> 8) * 64 + l, didx = tcell + (size_t)(pr & 0xff) * 64 + l;
            const P3 cp = ld_pos(s, sidx);
            const V2 cv = ld_vel(s, sidx);
            st_pos(s, didx, cp.x, cp.y, cp.z);
            st_vel(s, didx, cv.x, cv.y);
            if (s.vz0) s.vz0[didx] = s.vz0[sidx];
            s.w[didx] = w_copy;
        }
    }
#ifdef RESAMPLE_PROF
    t_s4 = __builtin_readcyclecounter();
#ifdef RESAMPLE_PROF2
    if (tid == 0 && BX < 65536) { g_rprof[BX * 4] = t_sa - t_s0; g_rprof[BX * 4 + 1] = t_s1 - t_sa; g_rprof[BX * 4 + 2] = t_s3 - t_s2; g_rprof[BX * 4 + 3] = t_s4 - t_s3; }
#else
    if (tid == 0 && BX < 65536) { g_rprof[BX * 4] = t_s1 - t_s0; g_rprof[BX * 4 + 1] = t_s2 - t_s1; g_rprof[BX * 4 + 2] = t_s3 - t_s2; g_rprof[BX * 4 + 3] = t_s4 - t_s3; }
#endif
#endif
}

// --------------------------------------------------------------------------
// k_rollout: the future-status rollout of mapOccupancyCalculationAndResample (:950-964) for the MOVING old particles
// (static ones add the same mass to their own voxel for every horizon: fut_stat, k_resample).
// One workgroup per tile that holds moving particles.  For every horizon t the particle's future voxel (same layer:
// vz == 0) receives its weight:
//   * few moving particles (a pedestrian's newborns in an otherwise static map): one float atomic each;
//   * many (the whole tile moves): the tile's particles land in a band of neighbouring voxels, so the workgroup
//     accumulates horizon t in an LDS window over the voxel-index range [tile - RO_HALF, tile + 64 + RO_HALF) and
//     flushes the touched part of it with COALESCED atomics onto the horizon-major accumulators (a scattered float
//     atomic costs a memory transaction per lane, ~25 G/s device-wide; a row of 64 neighbouring voxels costs two).
//     Destinations outside the window take the single-atomic path.
// --------------------------------------------------------------------------
#define RO_G 8           // tiles per workgroup: their particles share the LDS windows
#define RO_GC 4          // cube storage: a group is RO_GC x RO_GC cubes of one layer of cubes
#define RO_GT (RO_GC * RO_GC)   // ... = 16 tiles
#define RO_NE 64         // runs of records a workgroup's table holds (a power of two >= RO_G, RO_GT, 4 * RO_GT)
#define RO_DENSE 384     // moving particles in a GROUP of tiles from which the LDS windows pay
#define RO_TPB 1024
#define RO_LDS_CELLS 30000   // fp32 cells of all windows together (120 kB: one workgroup per CU)
struct RolloutPlan {
    int halo[DSP_MAX_PRED];      // rows of the grid a horizon's window reaches beyond the group's voxels, either side
    int woff[DSP_MAX_PRED + 1];  // first cell of every horizon's window in the workgroup's LDS
};
// LIGHT: no LDS windows at all -- 256 threads, every contribution a global atomic.  A map with few moving particles (what the
// depth stream builds: newborns of matched clusters) holds no group that would use the windows, and the 120 kB of LDS they take
// let ONE workgroup per CU start at a time: 10 890 groups at 264x264x80 cost 25 us although next to none has anything to do.
// The handle launches this variant while last frame's count of tiles with hundreds of moving particles is low (c.ro_inline).
template <int TPB, bool LIGHT>
__global__ void __launch_bounds__(TPB) k_rollout(MapDims d, DevState s, const float4* __restrict__ ro_rec, const int* __restrict__ ro_cnt, int ntiles,
                                                    RolloutPlan pl, int* __restrict__ ro_stat, const int* __restrict__ ro_sub) {
    // One workgroup per group of RO_G consecutive tiles (512 voxels: a few rows of a layer).  EVERY particle is read once and
    // adds its weight to its future voxel at all T horizons (reading the particles once per horizon was what bound this kernel
    // with every particle moving: 10 x 0.5 GB at 132x132x60).  Horizon t has its own LDS window over the voxel-index range the
    // group's particles reach at a design speed -- halo[t] rows of the grid either side -- so that the footprints of the group's
    // tiles, which overlap almost completely, cost ONE global atomic per touched cell and horizon; faster particles fall outside
    // and take the single-atomic path.  Windows are flushed with coalesced atomics onto the horizon-major accumulators.
    // The windows are FIXED-POINT like the accumulators they are flushed to (fut_quantum: 2^-24 per unit, the same integer per
    // particle on every path): ds_add_f32 runs at a third of a lane per clock and CU on this chip, ds_add_u32 eight times faster
    // (tools/micro/lds_atomic_bench.hip), and integer sums do not depend on the order of the adds.  A 32-bit cell can receive at
    // most the group's whole moving weight W (k_resample's per-tile sums): a group with W >= FUT_WINDOW_MAX_W (256 units, never
    // seen) takes the single-atomic path for every particle -- same result.
    // ro_stat[2 * group + {0, 1}] = contributions this group sent through its windows / straight to the accumulators (diagnostics:
    // which path ran; summed on request, no atomics here)
    // CUBE storage (MapDims::tiling): a group is a block of RO_GC x RO_GC cubes of one layer of cubes -- 16 x 16 voxels, four layers deep
    // (square: the least window per particle); a particle stays in its layer (vz == 0), so the windows are planar: with the LDS windows,
    // FOUR workgroups share a group, one per layer (blockIdx & 3), and horizon t's window is the square of (16 + 2 h)^2 voxels around the
    // group in that layer.  k_resample writes a tile's records in four runs, one per layer (ro_sub: their lengths), so a layer's workgroup
    // reads ITS records only; a tile whose records came as one mixed run (k_resample_wg: ro_sub[4 t] = -1) is read whole and filtered.
    // Without windows (LIGHT) one workgroup per group takes every record.
    // The group's records as a table of RUNS: (first record, length); at most 64 (16 tiles x 4 layers, LIGHT on cubes)
    extern __shared__ unsigned s_win[];
    __shared__ int s_cnt[RO_NE + 1];
    __shared__ int s_rbase[RO_NE];
    __shared__ u64 s_mixed;
    __shared__ float s_wtot;
    __shared__ int s_stat[2];
    const bool cubes = d.tiling != 0;
    const bool split = cubes && !LIGHT;
    const int gxn = (d.ncx + RO_GC - 1) / RO_GC, gyn = (d.ncy + RO_GC - 1) / RO_GC;   // groups per row / column of a layer of cubes
    const int grp = split ? (int)blockIdx.x >> 2 : (int)blockIdx.x;
    const int sub = split ? (int)blockIdx.x & 3 : -1;                // this workgroup's layer inside the cubes
    const int gcz = cubes ? grp / (gxn * gyn) : 0, grem = cubes ? grp - gcz * gxn * gyn : 0, gy = grem / gxn, gx = grem - gy * gxn;
    const int G0 = cubes ? 0 : grp * RO_G;
    const int nt_g = cubes ? RO_GT : RO_G;                           // tiles of a group (some may lie outside the map: count 0)
    const int wx0 = gx * RO_GC * 4, wy0 = gy * RO_GC * 4;            // the group's first voxel column / row (cube storage)
    const int wzl = gcz * 4 + max(sub, 0);                           // this workgroup's layer (relative to the slab)
    const int tid = threadIdx.x;
    const int cap = 64 * d.slots;
    const int ne = cubes ? (split ? RO_GT : RO_NE) : RO_G;           // runs of this workgroup
    if (tid == 0) s_mixed = 0ull;
    __syncthreads();
    if (tid < RO_NE) {
        int cnt = 0, rb = 0;
        if (tid < ne) {
            const int k = cubes && !split ? tid >> 2 : tid, sq = cubes && !split ? tid & 3 : max(sub, 0);   // tile of the group, layer
            int bx = G0 + k;
            if (cubes) { const int cxk = gx * RO_GC + (k & (RO_GC - 1)), cyk = gy * RO_GC + k / RO_GC; bx = (cxk < d.ncx && cyk < d.ncy) ? (gcz * d.ncy + cyk) * d.ncx + cxk : -1; }
            else if (bx >= ntiles) bx = -1;
            if (bx >= 0) {
                rb = bx * cap;
                if (!cubes) cnt = ro_cnt[bx];
                else if (ro_sub[4 * bx] < 0) {   // one mixed run: the layer workgroups filter it; LIGHT takes it once (as layer 0's)
                    if (split || sq == 0) { cnt = ro_cnt[bx]; if (split && cnt > 0) atomicOr(&s_mixed, 1ull << tid); }
                } else { cnt = ro_sub[4 * bx + sq]; rb += sq * (cap >> 2); }
            }
        }
        s_cnt[tid] = cnt; s_rbase[tid] = rb;
    }
    __syncthreads();
    if (tid == 0) {
        int t = 0; float w = 0.f;
        for (int k = 0; k < RO_NE; ++k) {   // exclusive prefix of the lengths; the weights in run order (an upper bound for a layer's share)
            const int c = s_cnt[k]; s_cnt[k] = t; t += c;
            if (c > 0) w += __int_as_float(ro_cnt[ntiles + s_rbase[k] / cap]);
        }
        s_cnt[RO_NE] = t;
        s_wtot = w;
    }
    __syncthreads();
    const int total = s_cnt[RO_NE];
    const u64 mixed = s_mixed;
    if (total == 0) { if (tid < 2) ro_stat[blockIdx.x * 2 + tid] = 0; return; }
    if (tid < 2) s_stat[tid] = 0;
    int n_win = 0, n_dir = 0;
    const int T = d.T;
    const size_t V = (size_t)d.v_loc;
    // particle `it` of the group -> its record (the run whose prefix interval holds `it`: the last k with s_cnt[k] <= it; prefixes are
    // non-decreasing, empty runs share a value with their successor and are skipped); returns the run
    auto rec_of = [&](int it, float4& a, float4& b) {
        int g = 0;
#pragma unroll
        for (int st = RO_NE / 2; st > 0; st >>= 1) g += (s_cnt[g + st] <= it) ? st : 0;
        const size_t o = ((size_t)s_rbase[g] + (size_t)(it - s_cnt[g])) * 2;
        a = ro_rec[o]; b = ro_rec[o + 1];
        return g;
    };
    // (a window cell is 32 bits wide and can receive at most the group's whole moving weight: windows only while that fits)
    const bool dense = !LIGHT && total >= RO_DENSE && s_wtot < FUT_WINDOW_MAX_W;
    const int ncell = pl.woff[T];
    if (dense) for (int i = tid; i < ncell; i += TPB) s_win[i] = 0u;
    __syncthreads();
    for (int it0 = tid; it0 < total; it0 += TPB * 3) {   // three particles per step: their records are requested together
        float4 a[3], b[3];
        unsigned wq[3];
        int run[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            a[u] = make_float4(0.f, 0.f, 0.f, 0.f); b[u] = a[u]; run[u] = 0;
            if (it0 + u * TPB < total) run[u] = rec_of(it0 + u * TPB, a[u], b[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (it0 + u * TPB >= total) continue;
            if (split && ((mixed >> run[u]) & 1ull) && ((__float_as_int(b[u].y) >> 4) & 3) != sub) continue;   // (a mixed run) another layer's workgroup takes this record
            const int zl = layer_of_lv(d, __float_as_int(b[u].y));   // the particle's layer: it never changes (vz == 0)
            wq[u] = (unsigned)fut_quantum(b[u].x);   // (dense: below 2^32 because the group's sum is)
            for (int t = 0; t < T; ++t) {
                const float pt = d.pred_t[t];
                const float fx = a[u].x + a[u].z * pt;      // :954-955
                const float fy = a[u].y + a[u].w * pt;
                if (fabsf(fx) >= d.half_x || fabsf(fy) >= d.half_y) continue;
                const int xi = (int)div_res(d, fx + d.half_x);
                const int yi = (int)div_res(d, fy + d.half_y);
                const int dl = lv_of_xyz(d, xi, yi, zl);
                if (dl < 0 || dl >= d.v_loc) continue;
                int off;
                if (cubes) {   // the square of (16 + 2 h)^2 voxels around the group, row-major
                    const int h = pl.halo[t], ww = RO_GC * 4 + 2 * h, wx = xi - (wx0 - h), wy = yi - (wy0 - h);
                    off = ((unsigned)wx < (unsigned)ww && (unsigned)wy < (unsigned)ww) ? wy * ww + wx : -1;
                } else off = dl - (G0 * 64 - pl.halo[t] * d.nx);
                if (dense && off >= 0 && off < pl.woff[t + 1] - pl.woff[t]) { atomicAdd(&s_win[pl.woff[t] + off], wq[u]); ++n_win; }
                else { fut_add(&s.fut[(size_t)t * V + dl], fut_quantum(b[u].x)); s.fut_dirty[dl >> 6] = 1; ++n_dir; }
            }
        }
    }
    n_win = wave_sum_i(n_win); n_dir = wave_sum_i(n_dir);
    if (lane_id() == 0) { if (n_win) atomicAdd(&s_stat[0], n_win); if (n_dir) atomicAdd(&s_stat[1], n_dir); }
    __syncthreads();
    if (tid < 2) ro_stat[blockIdx.x * 2 + tid] = s_stat[tid];
    if (!dense) return;
    for (int t = 0; t < T; ++t) {
        const int w0 = pl.woff[t], wn = pl.woff[t + 1] - w0;
        if (cubes) {
            const int h = pl.halo[t], ww = RO_GC * 4 + 2 * h;
            for (int i = tid; i < wn; i += TPB) {
                const unsigned q = s_win[w0 + i];
                if (q) {   // (only cells of voxels inside the map ever receive anything)
                    const int wy = i / ww, wx = i - wy * ww;
                    const int dl = lv_of_xyz(d, wx0 - h + wx, wy0 - h + wy, wzl);
                    fut_add(&s.fut[(size_t)t * V + dl], (u64)q); s.fut_dirty[dl >> 6] = 1;
                }
            }
            continue;
        }
        const int g0 = G0 * 64 - pl.halo[t] * d.nx;   // local voxel index of the window's first cell (cells outside the slab stay zero)
        for (int i = tid; i < wn; i += TPB) {
            const unsigned q = s_win[w0 + i];
            if (q) { fut_add(&s.fut[(size_t)t * V + g0 + i], (u64)q); s.fut_dirty[(g0 + i) >> 6] = 1; }
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
__global__ void k_seed_uniform(MapDims d, DevState s, int per_voxel, float weight, unsigned seed, float vmax) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)d.v_loc * d.slots;
    if (t >= total) return;
    const int lv = (int)(t / d.slots), sl = (int)(t - (size_t)lv * d.slots);
    const int index = g_of_lv(d, lv);   // the reference's voxel index: the fill is the same whatever the storage order (-1: a padding voxel)
    if (sl == 0) {
        for (int e = 0; e < d.mw; ++e) {
            const int nbits = index < 0 ? 0 : max(0, min(64, per_voxel - e * 64));
            s.mask[(size_t)lv * d.mw + e] = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
            s.nbmask[(size_t)lv * d.mw + e] = 0ull;
        }
    }
    if (sl >= per_voxel || index < 0) return;
    const int zc = d.ny * d.nx;
    const int zi = index / zc, rest = index - zi * zc, yi = rest / d.nx, xi = rest - yi * d.nx;
    const unsigned h0 = hash_u32(seed ^ hash_u32((unsigned)index * 73u + (unsigned)sl));
    const unsigned h1 = hash_u32(h0 + 0x9e3779b9U), h2 = hash_u32(h1 + 0x9e3779b9U);
    const float u0 = 0.02f + 0.96f * (float)(h0 >> 8) * (1.f / 16777216.f);
    const float u1 = 0.02f + 0.96f * (float)(h1 >> 8) * (1.f / 16777216.f);
    const float u2 = 0.02f + 0.96f * (float)(h2 >> 8) * (1.f / 16777216.f);
    const size_t idx = pidx(d, lv, sl);
    st_pos(s, idx, ((float)xi + u0) * d.res - d.half_x, ((float)yi + u1) * d.res - d.half_y, ((float)zi + u2) * d.res - d.half_z);
    float vx = 0.f, vy = 0.f;
    if (vmax > 0.f) {  // benchmark variant with moving particles: velocities uniform in +-vmax
        const unsigned h3 = hash_u32(h2 + 0x9e3779b9U), h4 = hash_u32(h3 + 0x9e3779b9U);
        vx = vmax * (2.f * (float)(h3 >> 8) * (1.f / 16777216.f) - 1.f);
        vy = vmax * (2.f * (float)(h4 >> 8) * (1.f / 16777216.f) - 1.f);
    }
    st_vel(s, idx, vx, vy); s.w[idx] = weight;
    note_speed(s, vx, vy);
}

// import sparse records {flag,vx,vy,vz,px,py,pz,w} at (global voxel, slot); slot < 0 = first free
__global__ void k_import(MapDims d, DevState s, int n, const int* __restrict__ voxel, const int* __restrict__ slot,
                         const float* __restrict__ rec, int* __restrict__ n_failed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int tv = voxel[i] - d.v_base;
    bool ok = tv >= 0 && tv < d.v_true;
    const int lv = ok ? lv_of_true(d, tv) : 0;
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
    const size_t idx = pidx(d, lv, sl);
    st_vel(s, idx, r[1], r[2]);
    note_speed(s, r[1], r[2]);
    if (s.vz0) s.vz0[idx] = r[3];
    st_pos(s, idx, r[4], r[5], r[6]); s.w[idx] = r[7];
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
        const size_t idx = pidx(d, lv, sl);
        voxel[pos] = g_of_lv(d, lv);
        slot[pos] = sl;
        float* r = rec + 8 * (size_t)pos;
        r[0] = nbf ? 15.f : 1.f;
        const V2 v2 = ld_vel(s, idx);
        const P3 p3 = ld_pos(s, idx);
        r[1] = v2.x; r[2] = v2.y; r[3] = s.vz0 ? s.vz0[idx] : 0.f;
        r[4] = p3.x; r[5] = p3.y; r[6] = p3.z; r[7] = s.w[idx];
    }
}

// generateRandomFloat :1551-1553 fed from the rand() table
__device__ __forceinline__ float rand_float_t(const DevState& s, const FilterParams& fp, int c, float lo, float hi) {
    const int r = s.r_tab[c % max(fp.rtab_n, 1)];
    return lo + __fdiv_rn((float)r, __fdiv_rn((float)2147483647, (hi - lo)));
}
// addRandomParticles :594-624 from the rand() table: 6 draws per particle, newborn flag (addAParticle).
// The reference adds the particles one after another, so particle i takes the first free slot left by particles
// < i of the same voxel.  Three passes reproduce exactly that placement without a sort: (1) every particle enters
// its voxel's bucket, (2) it ranks itself among the bucket's particle indices and takes the rank-th free slot of
// the voxel's occupancy before the call (read-only in this pass), (3) the newborn bits are set.
#define SEED_BUCKET_CAP 128
struct SeedDraw { float px, py, pz, vx, vy, vz; int lv; };
__device__ __forceinline__ SeedDraw seed_draw(const MapDims& d, const DevState& s, const FilterParams& fp, int i) {
    const int c = s.fs->r_cur + 6 * i;
    SeedDraw r;
    r.px = rand_float_t(s, fp, c, -d.half_x, d.half_x);
    r.py = rand_float_t(s, fp, c + 1, -d.half_y, d.half_y);
    r.pz = rand_float_t(s, fp, c + 2, -d.half_z, d.half_z);
    r.vx = rand_float_t(s, fp, c + 3, -1.f, 1.f);
    r.vy = rand_float_t(s, fp, c + 4, -1.f, 1.f);
    r.vz = rand_float_t(s, fp, c + 5, -1.f, 1.f);
    int gv;
    r.lv = -1;
    int lv;
    if (voxel_of_lv(d, r.px, r.py, r.pz, gv, lv)) r.lv = lv;   // (-1: another rank's slab)
    return r;
}
__global__ void k_add_random_bucket(MapDims d, DevState s, FilterParams fp, int n, int* __restrict__ vb_cnt, int* __restrict__ vb_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SeedDraw r = seed_draw(d, s, fp, i);
    if (r.lv < 0) return;
    const int pos = atomicAdd(&vb_cnt[r.lv], 1);
    if (pos < SEED_BUCKET_CAP) vb_idx[(size_t)r.lv * SEED_BUCKET_CAP + pos] = i;
}
__global__ void k_add_random_place(MapDims d, DevState s, FilterParams fp, int n, float weight, const int* __restrict__ vb_cnt,
                                   const int* __restrict__ vb_idx, int* __restrict__ slot_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    slot_of[i] = -1;
    const SeedDraw r = seed_draw(d, s, fp, i);
    if (r.lv < 0) return;
    const int nb = min(vb_cnt[r.lv], SEED_BUCKET_CAP);
    int rank = 0;
    bool recorded = false;
    for (int j = 0; j < nb; ++j) {
        const int o = vb_idx[(size_t)r.lv * SEED_BUCKET_CAP + j];
        rank += o < i ? 1 : 0;
        recorded |= o == i;
    }
    if (!recorded) return;   // beyond the bucket: more than 128 earlier particles in this voxel -> it is full
    int sl = -1;
    for (int e = 0; e < d.mw && sl < 0; ++e) {
        const int nbits = min(64, d.slots - e * 64);
        const u64 valid = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
        u64 fr = ~(s.mask[(size_t)r.lv * d.mw + e] | s.nbmask[(size_t)r.lv * d.mw + e]) & valid;
        const int nf = (int)__popcll(fr);
        if (rank >= nf) { rank -= nf; continue; }
        for (int q = 0; q < rank; ++q) fr &= fr - 1ull;
        sl = e * 64 + (__ffsll((long long)fr) - 1);
    }
    if (sl < 0) return;      // voxel full :1198-1200
    const size_t idx = pidx(d, r.lv, sl);
    st_pos(s, idx, r.px, r.py, r.pz); st_vel(s, idx, r.vx, r.vy); s.w[idx] = weight;
    note_speed(s, r.vx, r.vy);
    if (s.vz0) s.vz0[idx] = r.vz;
    slot_of[i] = (r.lv << 7) | sl;
}
__global__ void k_add_random_commit(MapDims d, DevState s, int n, const int* __restrict__ slot_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || slot_of[i] < 0) return;
    const int lv = slot_of[i] >> 7, sl = slot_of[i] & 127;
    atomicOr(&s.nbmask[(size_t)lv * d.mw + (sl >> 6)], 1ull << (sl & 63));   // flag 15
}
__global__ void k_zero_ints(int* __restrict__ p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
__global__ void k_advance_rcur(DevState s, FilterParams fp, int by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) s.fs->r_cur = (int)(((long long)s.fs->r_cur + by) % max(fp.rtab_n, 1));
}

// --------------------------------------------------------------------------
// multi-GPU: particles whose new voxel lies in another Z-slab (marked in expmask by k_predict).
// k_export_slab compacts those leaving in direction `dir` into the caller's send buffer (with their source key, the
// same 2 x float4 record the inboxes hold) and frees their slots.
// --------------------------------------------------------------------------
template <int MW>
// dir = +1 / -1: that direction into rec_out / count[0].  dir = 0: both directions in one pass -- up into rec_out /
// count[0], down into rec_out2 / count[1].
__global__ void __launch_bounds__(256) k_export_slab(MapDims d, DevState s, u64* __restrict__ expmask, int dir,
                                                      float* __restrict__ rec_out, int cap, int* __restrict__ count,
                                                      float* __restrict__ rec_out2) {
    const int l = lane_id();
    const int lv = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + l;
    const bool inr = lv < d.v_loc;
    // Two passes per wave (round 5).  One aggregated atomic per slot ROW -- what this kernel did -- is a chain of up to 2 M dependent
    // same-address atomics per wave, and every wave of a boundary layer hammers the same two counters: 250 - 280 us per slab of the
    // saturated 264x264x80 map (as long as the slab's whole prediction sweep), against 6 us when nothing leaves.  Pass 1 classifies the
    // wave's leavers (up / down) and counts them; ONE atomic per wave and direction reserves their records; pass 2 writes them.
    u64 exw[MW], dnw[MW], minew[MW];
    int cu = 0, cd = 0;
#pragma unroll
    for (int e = 0; e < MW; ++e) {
        exw[e] = inr ? expmask[(size_t)lv * MW + e] : 0ull;
        dnw[e] = 0ull; minew[e] = 0ull;
        u64 bits = exw[e];
        while (bits) {   // (per lane: a voxel exports a handful of particles)
            const int sb = __ffsll((long long)bits) - 1;
            bits &= bits - 1ull;
            const P3 p3 = ld_pos(s, pidx(d, lv, e * 64 + sb));
            int gv;
            voxel_of(d, p3.x, p3.y, p3.z, gv);
            const int nlv = gv - d.v_base;   // (slabs keep index-order storage: MapDims::tiling == 0)
            const bool down = nlv < 0;
            const bool mine = dir > 0 ? nlv >= d.v_true : (dir < 0 ? down : (down || nlv >= d.v_true));
            if (mine) { minew[e] |= 1ull << sb; if (down && dir == 0) { dnw[e] |= 1ull << sb; ++cd; } else ++cu; }
        }
    }
    if (!__ballot(cu | cd)) return;
    const int tu = wave_sum_i(cu), td = wave_sum_i(cd);
    int base_u = 0, base_d = 0;
    if (l == 0) {
        if (tu) base_u = atomicAdd(count, tu);                 // (dir != 0: everything counts as "up" = count[0] / rec_out)
        if (td) base_d = atomicAdd(count + 1, td);
    }
    base_u = __builtin_amdgcn_readfirstlane(base_u); base_d = __builtin_amdgcn_readfirstlane(base_d);
    int run_u = 0, run_d = 0;   // (wave-uniform)
#pragma unroll
    for (int e = 0; e < MW; ++e) {
        u64 tor = wave_or_u64(minew[e]);
        while (tor) {
            const int sb = __ffsll((long long)tor) - 1;
            tor &= tor - 1ull;
            const bool mine = (minew[e] >> sb) & 1ull, down = (dnw[e] >> sb) & 1ull;
            const u64 bu = __ballot(mine && !down), bd = __ballot(mine && down);
            if (mine) {
                const int pos = down ? base_d + run_d + (int)__popcll(bd & lanemask_lt()) : base_u + run_u + (int)__popcll(bu & lanemask_lt());
                if (pos < cap) {
                    const size_t idx = pidx(d, lv, e * 64 + sb);
                    const P3 p3 = ld_pos(s, idx);
                    const V2 v2 = ld_vel(s, idx);
                    int gv;
                    voxel_of(d, p3.x, p3.y, p3.z, gv);
                    float* r = (down ? rec_out2 : rec_out) + 8 * (size_t)pos;
                    r[0] = __int_as_float(gv); r[1] = v2.x; r[2] = v2.y; r[3] = p3.x; r[4] = p3.y; r[5] = p3.z; r[6] = s.w[idx];
                    r[7] = __int_as_float(g_of_lv(d, lv) * d.slots + e * 64 + sb);   // source key: k_place's service order
                }
            }
            run_u += (int)__popcll(bu); run_d += (int)__popcll(bd);
        }
        if (minew[e]) {
            atomicAnd(&s.mask[(size_t)lv * MW + e], ~minew[e]);
            expmask[(size_t)lv * MW + e] = exw[e] & ~minew[e];
        }
    }
}

// Received records join the inbox of their destination tile; k_place (which the slab runs AFTER the exchange) then
// serves them together with the slab's own movers in the order of their source keys, so a sharded map fills exactly
// the slots the unsharded one does.
__global__ void __launch_bounds__(256) k_import_movers(MapDims d, int n, const float* __restrict__ rec, float4* __restrict__ in_rec,
                                                       int* __restrict__ in_cnt, int* __restrict__ dropped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool lost = false;
    if (i < n) {
        const float* r = rec + 8 * (size_t)i;
        const int tv = __float_as_int(r[0]) - d.v_base;
        if (tv >= 0 && tv < d.v_true) {
            const int nlv = lv_of_true(d, tv);
            const int tile = nlv >> 6, cap = 64 * d.slots;
            const int pos = atomicAdd(&in_cnt[tile], 1);   // beyond cap: k_place counts it as "voxel full"
            if (pos < cap) {
                const size_t o = ((size_t)tile * cap + pos) * 2;
                in_rec[o] = make_float4(__int_as_float(nlv), r[1], r[2], r[3]);   // (.x: the destination's STORAGE index, like k_predict's own records)
                in_rec[o + 1] = make_float4(r[4], r[5], r[6], r[7]);
            }
        } else lost = true;       // not a neighbouring slab's voxel (jump larger than a slab)
    }
    wave_count_add(dropped, lost);
}

// the memory skeleton of k_predict's row sweep (dspmap_debug_sweep_probe)
template <int NB>
__global__ void __launch_bounds__(256) k_sweep_probe(MapDims d, DevState s, int what, int rows) {
    const int l = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t tcell = (size_t)blockIdx.x * d.slots * 64;
    const int tcells = d.slots * 64;
    const brsrc rs_pos = __builtin_amdgcn_make_buffer_rsrc((void*)(s.pos + 3 * tcell), 0, tcells * 12, 0x00020000);
    const brsrc rs_vel = __builtin_amdgcn_make_buffer_rsrc((void*)(s.vel + 2 * tcell), 0, tcells * 8, 0x00020000);
    const brsrc rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(s.w + tcell), 0, tcells * 4, 0x00020000);
    float acc = 0.f;
    for (int r0 = wave; r0 < rows; r0 += 4 * NB) {
        P3 pp[NB]; V2 vv[NB]; float w[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int row = min(r0 + 4 * k, rows - 1);
            pp[k].x = pp[k].y = pp[k].z = 0.f; vv[k].x = vv[k].y = 0.f; w[k] = 0.f;
            if (what & 2) vv[k] = bl_vel(rs_vel, l, row * 64);
            if (what & 1) pp[k] = bl_pos(rs_pos, l, row * 64);
            if (what & 4) w[k] = bl_w(rs_w, l, row * 64);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int row = r0 + 4 * k;
            if (row >= rows) continue;
            acc += vv[k].x + vv[k].y + w[k];
            if (what & 8) bs_pos(rs_pos, l, row * 64, pp[k].x + 0.f, pp[k].y + 0.f, pp[k].z + 0.f);
            else acc += pp[k].x + pp[k].y + pp[k].z;
        }
    }
    if (acc == 1.2345e-30f) s.fs->n_valid = 1;
}
void launch_sweep_probe(const LaunchCtx& c, int what, int rows, int rows_per_batch) {
    const dim3 g(c.k.ntiles), b(256);
    if (rows_per_batch <= 1) hipLaunchKernelGGL(k_sweep_probe<1>, g, b, 0, c.stream, c.d, c.s, what, rows);
    else if (rows_per_batch == 2) hipLaunchKernelGGL(k_sweep_probe<2>, g, b, 0, c.stream, c.d, c.s, what, rows);
    else if (rows_per_batch == 3) hipLaunchKernelGGL(k_sweep_probe<3>, g, b, 0, c.stream, c.d, c.s, what, rows);
    else hipLaunchKernelGGL(k_sweep_probe<6>, g, b, 0, c.stream, c.d, c.s, what, rows);
}

// div_res (dspmap_device.h): comparison of the 3-instruction quotient with the IEEE division, bit for bit, over
//   * EVERY float of one binade [2^k, 2^(k+1)) with 2^k >= res (every quotient >= 1) -- 2^23 values.  All three operations (a * y, fma(-q0, res, a), fma(r, y, q0)) and
//     the division itself commute with a scaling of `a` by a power of two as long as nothing leaves the normal range, so this
//     proves every binade the map divides in (a >= res / 2; below that both quotients are < 1 and the voxel coordinate is 0);
//   * every `stride`-th float of the whole range [0, amax] as a direct check of exactly that argument.
__global__ void __launch_bounds__(256) k_verify_div(float res, float rcp, unsigned lo_bits, unsigned hi_bits, unsigned stride, int* __restrict__ bad) {
    int nb = 0;
    const unsigned step = gridDim.x * 256u * stride;
    for (unsigned b = lo_bits + (blockIdx.x * 256u + threadIdx.x) * stride; b <= hi_bits; b += step) {
        const float a = __uint_as_float(b);
        const float q0 = a * rcp;
        const float r = __fmaf_rn(-q0, res, a);
        const float q = __fmaf_rn(r, rcp, q0);
        const float qi = __fdiv_rn(a, res);
        // identical quotient, or at least the identical voxel coordinate where the quotient is denormal-small
        nb += (__float_as_uint(q) != __float_as_uint(qi) && !(qi < 1.f && q < 1.f && q >= 0.f)) ? 1 : 0;
        if (b > 0xffffffffu - step) break;
    }
    nb = wave_sum_i(nb);
    if (lane_id() == 0 && nb) atomicAdd(bad, nb);
}
void launch_verify_div(hipStream_t stream, float res, float rcp_res, float amax, int* bad) {
    unsigned bits;
    memcpy(&bits, &amax, sizeof(bits));
    // the exhaustive binade: [2^k, 2^(k+1)) with 2^k >= res, so that EVERY quotient of the pass is >= 1 and is compared bit for
    // bit (k_verify_div lets a pair of quotients below 1 pass: both give voxel coordinate 0) -- [1, 2) would prove nothing for a
    // resolution above 1 m
    int k = 0;
    (void)frexpf(res, &k);                       // res = f * 2^k, f in [0.5, 1)  ->  2^k > res (or == 2 res' for a power of two)
    const float lo = ldexpf(1.f, k);
    unsigned lo_bits;
    memcpy(&lo_bits, &lo, sizeof(lo_bits));
    hipLaunchKernelGGL(k_verify_div, dim3(2048), dim3(256), 0, stream, res, rcp_res, lo_bits, lo_bits + 0x7fffffu, 1u, bad);   // one whole binade
    hipLaunchKernelGGL(k_verify_div, dim3(2048), dim3(256), 0, stream, res, rcp_res, 0u, bits, 97u, bad);                       // the range, sampled
}

// PMC calibration: stream the field arrays with the sweeps' 4-byte-per-lane pattern
__global__ void __launch_bounds__(256) k_calib_read(DevState s, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    { const P3 p3 = ld_pos(s, i); const V2 v2 = ld_vel(s, i); acc += p3.x + p3.y + p3.z + v2.x + v2.y + s.w[i]; }
    if (acc == 1.2345e-30f) *sink = acc;
}
__global__ void __launch_bounds__(256) k_calib_write(DevState s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s.w[i] = s.w[i] + 0.f;
}
void launch_calib(const LaunchCtx& c, int mode, size_t n) {
    if (mode == 0) hipLaunchKernelGGL(k_calib_read, dim3(8192), dim3(256), 0, c.stream, c.s, n, (float*)c.s.fs);
    else hipLaunchKernelGGL(k_calib_write, dim3(8192), dim3(256), 0, c.stream, c.s, n);
}

// fold per-block partial counters (written without global atomics by the sweeps) into FrameScalars
__global__ void __launch_bounds__(1024) k_reduce_counters(DevState s, KernelScratch k, MapDims d, int fp_nb_num) {
    __shared__ int s_red[1024];
    const int tid = threadIdx.x;
    int acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned* pb = s.fs->bits_on ? k.tile_bits + (size_t)((k.ntiles + 63) / 64 * 2) : nullptr;   // (the last frame's k_predict visited these tiles only)
    for (int i = tid; i < k.ntiles; i += 1024) {
        if (pb && !((pb[i >> 5] >> (i & 31)) & 1u)) continue;
        acc[0] += k.part_predict[i * 4]; acc[1] += k.part_predict[i * 4 + 1];
        acc[2] += k.part_predict[i * 4 + 2]; acc[3] += k.part_predict[i * 4 + 3];
    }
    for (int i = tid; i < k.nblk_sweep * 4; i += 1024) acc[6] += k.part_resample[i];
    int accb[2] = {0, 0};
    const int nbb = (int)(((long long)birth_view(s).n * fp_nb_num + 255) / 256);  // blocks of the last k_birth_insert that covered real children
    for (int i = tid; i < nbb; i += 1024) { accb[0] += k.part_birth[i * 2]; accb[1] += k.part_birth[i * 2 + 1]; }
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
        s.fs->n_pyramid_full = out[2] + s.fs->n_place_pf + s.fs->n_pyr_removed; s.fs->n_moved = out[3];
        s.fs->n_voxel_full = s.fs->n_place_vf + s.fs->n_voxel_full_import; s.fs->n_live_out = out[6];
        int nf = 0;
        for (int b = 0; b < d.np; ++b) nf += pyr_len(d, s, b);
        s.fs->n_fov = nf;
    }
    for (int c = 0; c < 2; ++c) {
        s_red[tid] = accb[c];
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) s_red[tid] += s_red[tid + o];
            __syncthreads();
        }
        if (tid == 0) { if (c == 0) s.fs->n_born = s_red[0]; else s.fs->n_born_dropped = s_red[0]; }
        __syncthreads();
    }
}

// ==========================================================================
// launchers
// ==========================================================================
// one wave that waits a given time: the known-duration kernel dspmap_set_profiling calibrates the cost of an event bracket with
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
void launch_spin(const LaunchCtx& c, int us) {   // wall_clock64 ticks at 100 MHz
    if (us > 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, c.stream, (long long)us * 100);
}
void launch_predict_only(const LaunchCtx& c, bool with_gather, bool with_rank, int cls) {
    const int extra = (with_gather ? 1 : 0) | (with_rank ? 2 : 0) | (c.place_split ? 4 : 0);   // 4: list the tiles with a view for the split placement
    const unsigned xb = (with_gather ? (c.d.np + 3) / 4 : 0) + (with_rank ? 1 : 0) + (c.place_split ? (c.k.ntiles + 255) / 256 : 0);
    const KernelScratch* k = &c.k;
    if (c.s.vz0) {   // constructor-seeded particles take their velocity noise in the reference's sweep order
        const int nblk = (c.d.v_true + 255) / 256;
        if (c.d.mw == 1) hipLaunchKernelGGL(k_vz_count<1>, dim3(nblk), dim3(256), 0, c.stream, c.d, c.s, k->work_list, k->vz_q);
        else hipLaunchKernelGGL(k_vz_count<2>, dim3(nblk), dim3(256), 0, c.stream, c.d, c.s, k->work_list, k->vz_q);
        launch_scan_blocks(c, nblk);   // blk_cnt -> exclusive, total -> fs->occupied_count
    }
#define PRED_LAUNCH(MWV, VZ, SP) hipLaunchKernelGGL((k_predict<MWV, 4, VZ, SP>), dim3(k->ntiles + xb), dim3(256), 0, c.stream, c.d, c.s, c.fp, VZ ? 1 : 0, \
                                                k->part_predict, k->mv_rec, k->in_rec, k->in_cnt, k->expmask, k->work_list, k->vz_q, k->omask, extra, k->tile_fov, c.sweep_rev ? 1 : 0, k->view_list, cls ? k->tile_cls : nullptr, cls)
#define PRED_LAUNCH2(MWV, VZ) do { if (c.sparse) PRED_LAUNCH(MWV, VZ, true); else PRED_LAUNCH(MWV, VZ, false); } while (0)
    if (c.d.mw == 1) { if (c.s.vz0) PRED_LAUNCH2(1, true); else PRED_LAUNCH2(1, false); }
    else { if (c.s.vz0) PRED_LAUNCH2(2, true); else PRED_LAUNCH2(2, false); }
#undef PRED_LAUNCH2
#undef PRED_LAUNCH
}
void launch_claim(const LaunchCtx& c, int n_birth_grid, int part, int tile_lo, int tile_hi, int sel, int cls) {   // n_birth_grid > 0: the children of that many source points ride along
    const unsigned xb = n_birth_grid > 0 ? (unsigned)(((long long)n_birth_grid * c.fp.nb_num + 255) / 256) : 0u;
    const int nt = c.k.ntiles;
    int t0 = 0, n0 = nt, t1 = nt, n1 = 0;                       // part 0: every tile
    if (part == 1) { t0 = tile_lo; n0 = tile_hi - tile_lo; }     // part 1: the interior [tile_lo, tile_hi)
    if (part == 2) { n0 = tile_lo; t1 = tile_hi; n1 = nt - tile_hi; }   // part 2: the rest
    if (n0 + n1 <= 0 && xb == 0) return;
    const KernelScratch* k = &c.k;
    // sel == 0 runs beside the pair kernels with a small footprint: LaunchCtx::side_wg (default 3) workgroups per CU walk the tiles -- the
    // scattered stores that bound this kernel are saturated from there (measured: 7 -> 3 resident workgroups, same time),
    // and the wave slots, registers and LDS it leaves free are what the pair kernels run in
    unsigned grid = (unsigned)(n0 + n1) + xb;
    if (sel == 0) grid = std::min(grid, (unsigned)(std::max(1, c.side_wg) * c.n_cu));
    const int* vlist = (sel == 1 && c.place_split && part == 0) ? k->view_list : nullptr;   // (k_predict listed the tiles with a view)
    if (vlist) grid = std::min((unsigned)(n0 + n1), (unsigned)(PLACE_LB * c.n_cu)) + xb;    // one round of workgroups walks the list
    if (cls) grid = std::min((unsigned)(n0 + n1), (unsigned)(PLACE_LB * c.n_cu)) + xb;      // (two-branch frame) one round of workgroups walks the tiles: a workgroup
                                                                                            // that only finds out that its tile is the other branch's costs ~5 ns
    // a large sparse map: a tile receives a handful of arrivals and the launch is as long as (tiles with arrivals / resident workgroups) x
    // one tile's chain of round trips: two-wave workgroups there (264x264x80 filled by the depth stream: frame 0.435 -> 0.420 ms, two
    // alternating pairs of runs; one wave: no better; 132x132x60: +2 % with either, so only from 32 768 tiles on; the children riding
    // along need 256 threads)
    static const int sparse_threads = getenv("DSPMAP_PLACE_THREADS_SPARSE") ? atoi(getenv("DSPMAP_PLACE_THREADS_SPARSE")) : 128;
    const unsigned nthr = (c.sparse && xb == 0 && c.k.ntiles >= 32768) ? (unsigned)sparse_threads : 256u;
    if (c.d.mw == 1) hipLaunchKernelGGL(k_place<1>, dim3(grid), dim3(nthr), 0, c.stream, c.d, c.s, k->in_rec, k->in_cnt, c.s.vz0 ? 1 : 0, c.fp.tab_n, k->omask, c.fp, k->child, k->vb_cnt, k->vb_idx, (int)xb, t0, n0, t1, n1, k->tile_fov, sel, k->mv_rec, c.sweep_rev ? 0 : 1, vlist, cls ? k->tile_cls : nullptr, cls);
    else hipLaunchKernelGGL(k_place<2>, dim3(grid), dim3(nthr), 0, c.stream, c.d, c.s, k->in_rec, k->in_cnt, c.s.vz0 ? 1 : 0, c.fp.tab_n, k->omask, c.fp, k->child, k->vb_cnt, k->vb_idx, (int)xb, t0, n0, t1, n1, k->tile_fov, sel, k->mv_rec, c.sweep_rev ? 0 : 1, vlist, cls ? k->tile_cls : nullptr, cls);
}
void launch_predict(const LaunchCtx& c, bool with_gather) {
    launch_predict_only(c, with_gather, false);
    launch_claim(c);
}
// which kernels the stage runs for this map and these hints: bit 0 = the four-waves-per-tile resampler; bits 1-2 = the rollout of
// the moving particles: 0 inside the resampler (float-free integer atomics from its idle waves), 1 k_rollout LIGHT, 2 k_rollout with LDS
// windows, 3 none (no prediction horizons)
int resample_variant(const LaunchCtx& c) {
    // four waves per tile: maps below the handle's tile limit -- and, whatever their size, maps the handle takes for sparse (most tiles
    // empty: what is left is a few thousand tiles of a few hundred particles each, the metric's regime; 132x132x60 filled by the depth
    // stream, alternating inside one process: frame 0.2226 -> 0.2065 ms).  A limit of 0 keeps every map on the one-wave variant.
    const bool wg = (c.k.ntiles < c.resample_wg_tiles || (c.sparse && c.resample_wg_tiles > 0)) && ((c.d.mw == 1 && c.d.slots <= 4 * RWB) || (c.d.mw == 2 && c.d.slots <= 4 * RWB2 && c.k.ntiles < 32768));
    // (two words: 139 registers, three workgroups per CU -- on the 87 120 mostly empty tiles of a depth-stream-filled 264x264x80 map the
    // launch of four waves per EMPTY tile alone outlasts the one-wave k_resample<2, 8>: 0.441 against 0.374 ms per frame, round 6)
    int ro = c.d.T <= 0 ? 3 : (c.ro_inline ? (wg ? 0 : 1) : 2);
    return (wg ? 1 : 0) | (ro << 1);
}
void kernels_init_device() {   // per device, once (dspmap_init_device)
    (void)hipFuncSetAttribute((const void*)k_rollout<RO_TPB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RO_LDS_CELLS * 4);
}
void launch_resample(const LaunchCtx& c, int cls, bool with_rollout, int part) {   // part (DSPMAP_P_RESAMPLE_SPLIT; needs cls): 2 the early launch, 4 the late one
    const KernelScratch* k = &c.k;
    const int nw = 1;   // waves (= tiles) per workgroup (2 / 4 measured in round 4: -3 % on the realistic 264x264x80 fill, +2 % at saturation)
    const size_t lds = (size_t)nw * ((64 * c.d.M + 1) / 2) * sizeof(float);   // the copy notes; the weights are re-read (no LDS panel)
    const unsigned grid = (unsigned)((k->ntiles + nw - 1) / nw);
    // maps of the metric's size run the four-waves-per-tile variant: their frame is a chain of latencies and the longest tile is
    // the kernel; large maps keep one wave per tile (more tiles in flight per CU).  The limit is the handle's
    // (DSPMAP_P_RESAMPLE_WG_TILES), so either variant can be run on any one-word map.
    const int var = resample_variant(c);
    const int ro = var >> 1;
    if (var & 1) {
        const size_t lds4 = (size_t)(c.d.slots * 64) * sizeof(float) + (size_t)c.d.slots * 64 + (size_t)64 * c.d.M * 2;
        // ... and roll their moving particles out themselves (one integer atomic per particle and horizon from the waves that wait for
        // the sequential walk anyway): no k_rollout launch
        // -- unless many tiles hold hundreds of moving particles (c.ro_inline, the handle's choice from last frame's count):
        // then k_rollout's LDS windows are worth their launch (66x66x40 saturated, every particle moving: 0.11 vs 0.27 ms)
        if (c.d.mw == 1) hipLaunchKernelGGL(k_resample_wg<1>, dim3(k->ntiles), dim3(256), lds4, c.stream, c.d, c.s, k->part_resample, k->vb_cnt, k->ro_rec, k->ro_cnt, ro == 0 ? 1 : 0, k->ro_sub);
        else hipLaunchKernelGGL(k_resample_wg<2>, dim3(k->ntiles), dim3(256), lds4, c.stream, c.d, c.s, k->part_resample, k->vb_cnt, k->ro_rec, k->ro_cnt, ro == 0 ? 1 : 0, k->ro_sub);
    } else {
#define RS_LAUNCH(MWV, RB) hipLaunchKernelGGL((k_resample<MWV, RB>), dim3(grid), dim3(64 * nw), lds, c.stream, c.d, c.s, k->part_resample, k->vb_cnt, k->ro_rec, k->ro_cnt, (c.resample_rev ? 1 : 0) | (cls ? part : 0), cls ? k->tile_cls : nullptr, cls, k->ro_sub)
        if (c.d.mw == 1) { if (c.sparse) RS_LAUNCH(1, 8); else RS_LAUNCH(1, 4); }
        else { if (c.sparse) RS_LAUNCH(2, 8); else RS_LAUNCH(2, 4); }
#undef RS_LAUNCH
    }
    if (with_rollout) launch_rollout(c);
}
int rollout_groups(const MapDims& d, int ntiles) {   // groups of k_rollout: runs of RO_G tiles; cube storage: inside one row of cubes
    return d.tiling ? d.ncz * ((d.ncy + RO_GC - 1) / RO_GC) * ((d.ncx + RO_GC - 1) / RO_GC) : (ntiles + RO_G - 1) / RO_G;
}
void launch_rollout(const LaunchCtx& c) {
    const KernelScratch* k = &c.k;
    const int ro = resample_variant(c) >> 1;
    if (ro == 1 || ro == 2) {
        // windows: the rows a particle reaches at a design speed (1.5 m/s, a brisk pedestrian), lowered until all T windows fit the LDS
        RolloutPlan pl;
        float vdes = 1.5f;
        for (;;) {
            int tot = 0;
            for (int t = 0; t < c.d.T; ++t) {
                pl.halo[t] = (int)ceilf(vdes * fabsf(c.d.pred_t[t]) / c.d.res) + 1;
                pl.woff[t] = tot;
                // index-order storage: the group's 512 voxel indices and halo rows of the grid either side; cubes: the rectangle of voxels
                // around the group's 32 x 4 in one layer (one workgroup per layer)
                tot += c.d.tiling ? (RO_GC * 4 + 2 * pl.halo[t]) * (RO_GC * 4 + 2 * pl.halo[t]) : RO_G * 64 + 2 * pl.halo[t] * c.d.nx;
            }
            pl.woff[c.d.T] = tot;
            if (tot <= RO_LDS_CELLS || vdes < 0.02f) break;
            vdes *= 0.8f;
        }
        if (pl.woff[c.d.T] > RO_LDS_CELLS) {   // (a grid too wide even for one-row halos: every window collapses to the group itself)
            int tot = 0;
            for (int t = 0; t < c.d.T; ++t) { pl.halo[t] = 0; pl.woff[t] = tot; tot += c.d.tiling ? RO_GC * 4 * RO_GC * 4 : RO_G * 64; }
            pl.woff[c.d.T] = tot;
        }
        const unsigned ngrp = (unsigned)rollout_groups(c.d, k->ntiles);
        if (ro == 1) hipLaunchKernelGGL((k_rollout<256, true>), dim3(ngrp), dim3(256), 0, c.stream, c.d, c.s, k->ro_rec, k->ro_cnt,
                                        k->ntiles, pl, k->ro_stat, k->ro_sub);
        else hipLaunchKernelGGL((k_rollout<RO_TPB, false>), dim3(ngrp * (c.d.tiling ? 4u : 1u)), dim3(RO_TPB), (size_t)pl.woff[c.d.T] * 4, c.stream, c.d, c.s, k->ro_rec, k->ro_cnt,
                                k->ntiles, pl, k->ro_stat, k->ro_sub);
    }
}
__global__ void k_set_live_sample(DevState s, int v) { s.fs->live_acc = v; }
static void mark_all_live(const LaunchCtx& c) {   // particles were written outside a frame: every tile may hold some
    (void)hipMemsetAsync(c.s.tile_live, 1, sizeof(int) * (size_t)c.k.ntiles, c.stream);
    (void)hipMemsetAsync(c.s.tile_moving, 1, sizeof(int) * (size_t)c.k.ntiles, c.stream);   // (nothing known about the velocities)
    // ... and the next frame's estimate of the non-empty tiles (FrameScalars::live_hint) says so too
    hipLaunchKernelGGL(k_set_live_sample, dim3(1), dim3(1), 0, c.stream, c.s, (c.k.ntiles + 63) / 64);
}
void launch_seed_uniform(const LaunchCtx& c, int per_voxel, float weight, unsigned seed, float vmax) {
    mark_all_live(c);
    const size_t total = (size_t)c.d.v_loc * c.d.slots;
    hipLaunchKernelGGL(k_seed_uniform, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, c.d, c.s, per_voxel, weight, seed, vmax);
}
void launch_import(const LaunchCtx& c, int n, const int* voxel_dev, const int* slot_dev, const float* rec8_dev, int* n_failed_dev) {
    if (n <= 0) return;
    mark_all_live(c);
    hipLaunchKernelGGL(k_import, dim3((n + 255) / 256), dim3(256), 0, c.stream, c.d, c.s, n, voxel_dev, slot_dev, rec8_dev, n_failed_dev);
}
void launch_export(const LaunchCtx& c, int* voxel_out, int* slot_out, float* rec8_out, int* count_dev, int cap) {
    const size_t total = (size_t)c.d.v_loc * c.d.slots;
    hipLaunchKernelGGL(k_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.stream, c.d, c.s, voxel_out, slot_out, rec8_out, count_dev, cap);
}
void launch_add_random(const LaunchCtx& c, int n, float weight, int* slot_of_tmp) {
    if (n <= 0) return;
    mark_all_live(c);
    const dim3 g((n + 255) / 256), b(256);
    hipLaunchKernelGGL(k_add_random_bucket, g, b, 0, c.stream, c.d, c.s, c.fp, n, c.k.vb_cnt, c.k.vb_idx);
    hipLaunchKernelGGL(k_add_random_place, g, b, 0, c.stream, c.d, c.s, c.fp, n, weight, c.k.vb_cnt, c.k.vb_idx, slot_of_tmp);
    hipLaunchKernelGGL(k_add_random_commit, g, b, 0, c.stream, c.d, c.s, n, slot_of_tmp);
    hipLaunchKernelGGL(k_zero_ints, dim3((c.d.v_loc + 255) / 256), b, 0, c.stream, c.k.vb_cnt, c.d.v_loc);   // buckets empty again
    hipLaunchKernelGGL(k_advance_rcur, dim3(1), dim3(64), 0, c.stream, c.s, c.fp, 6 * n);
}
void launch_export_slab(const LaunchCtx& c, int dir, float* rec_out, int cap, int* count_dev, float* rec_out_down) {
    const KernelScratch* k = &c.k;
    if (c.d.mw == 1) hipLaunchKernelGGL(k_export_slab<1>, dim3(k->nblk_sweep), dim3(256), 0, c.stream, c.d, c.s, k->expmask, dir, rec_out, cap, count_dev, rec_out_down);
    else hipLaunchKernelGGL(k_export_slab<2>, dim3(k->nblk_sweep), dim3(256), 0, c.stream, c.d, c.s, k->expmask, dir, rec_out, cap, count_dev, rec_out_down);
}
void launch_import_movers(const LaunchCtx& c, int n, const float* rec) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_import_movers, dim3((n + 255) / 256), dim3(256), 0, c.stream, c.d, n, rec, c.k.in_rec, c.k.in_cnt, &c.s.fs->n_voxel_full_import);
}
void launch_reduce_counters(const LaunchCtx& c) {
    hipLaunchKernelGGL(k_reduce_counters, dim3(1), dim3(1024), 0, c.stream, c.s, c.k, c.d, c.fp.nb_num);
}
