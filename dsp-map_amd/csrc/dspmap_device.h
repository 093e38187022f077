// dspmap_device.h -- device-side primitives (gfx950, wave64).
// Geometry / index math restates include/dsp_dynamic.h:1062-1125,1303-1367 of
// the reference with the SAME operation order (this TU is compiled with
// -ffp-contract=off so that a*b+c is two roundings, like the strict oracle).
#pragma once
#include <hip/hip_runtime.h>
#include "dspmap_types.h"

#define WAVE 64

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }
// one int at a wave-uniform address through the scalar unit, waited for (the compiler takes the vector path for memory that
// kernels write; the scalar cache is invalidated at every launch, so values of EARLIER kernels are seen)
__device__ __forceinline__ int sload_i(const int* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
}
__device__ __forceinline__ void sload_i3(const int* p0, const int* p1, const int* p2, int& v0, int& v1, int& v2) {   // three at once: one round trip
    asm volatile("s_load_dword %0, %3, 0x0\n\ts_load_dword %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v0), "=&s"(v1), "=&s"(v2) : "s"(p0), "s"(p1), "s"(p2));
}
__device__ __forceinline__ void sload_i5(const int* p0, const int* p1, const int* p2, const int* p3, const int* p4, int& v0, int& v1, int& v2, int& v3, int& v4) {
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %6, 0x0\n\ts_load_dword %2, %7, 0x0\n\ts_load_dword %3, %8, 0x0\n\ts_load_dword %4, %9, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3), "=&s"(v4) : "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4));
}
__device__ __forceinline__ void sload_i2(const int* p0, const int* p1, int& v0, int& v1) {   // two at once: one round trip
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v0), "=&s"(v1) : "s"(p0), "s"(p1));
}
__device__ __forceinline__ void sload_i4(const int* p0, const int* p1, const int* p2, const int* p3, int& v0, int& v1, int& v2, int& v3) {   // four at once
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3) : "s"(p0), "s"(p1), "s"(p2), "s"(p3));
}
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// ---- wavefront scan / reduction on the DPP network (no LDS traffic).
// gfx9 pattern: 4 row_shr steps scan each row of 16 lanes, row_bcast:15 / :31
// carry the row totals across (same sequence LLVM's atomic optimizer emits).
// Requires all 64 lanes active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_i(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ float wave_incl_scan(float v) {
    v = dpp_add_f<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add_f<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add_f<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add_f<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add_f<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_add_f<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ int wave_incl_scan_i(int v) {
    v = dpp_add_i<0x111, 0xf>(v);
    v = dpp_add_i<0x112, 0xf>(v);
    v = dpp_add_i<0x114, 0xf>(v);
    v = dpp_add_i<0x118, 0xf>(v);
    v = dpp_add_i<0x142, 0xa>(v);
    v = dpp_add_i<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_incl_scan(v)), 63));
}
__device__ __forceinline__ float wave_sum_f(float v) {   // butterfly: the same order on every run, the total in every lane
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_i(v), 63); }

// Wave-aggregated "append": every active lane gets a distinct position in the
// list counted by cnt[key]; one global atomic per distinct key per wave.
// Must be called from wave-uniform control flow (inactive lanes pass active=false).
__device__ __forceinline__ int wave_agg_inc(int* cnt, int key, bool active) {
    int pos = -1;
    u64 todo = __ballot(active);
    const int l = lane_id();
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader, WAVE);
        const bool mine = active && key == k;
        const u64 grp = __ballot(mine);
        int base = 0;
        if (l == leader) base = atomicAdd(&cnt[k], (int)__popcll(grp));
        base = __shfl(base, leader, WAVE);
        if (mine) pos = base + (int)__popcll(grp & ((1ull << l) - 1ull));
        todo &= ~grp;
    }
    return pos;
}
// single-counter variant
__device__ __forceinline__ int wave_agg_inc1(int* cnt, bool active) {
    const u64 grp = __ballot(active);
    if (!grp) return -1;
    const int l = lane_id();
    const int leader = __ffsll((long long)grp) - 1;
    int base = 0;
    if (l == leader) base = atomicAdd(cnt, (int)__popcll(grp));
    base = __shfl(base, leader, WAVE);
    return active ? base + (int)__popcll(grp & ((1ull << l) - 1ull)) : -1;
}
__device__ __forceinline__ void wave_count_add(int* cnt, bool active) {
    const u64 grp = __ballot(active);
    if (grp && lane_id() == __ffsll((long long)grp) - 1) atomicAdd(cnt, (int)__popcll(grp));
}

// OR of a 64-bit value over the wave (DPP network, both halves)
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u64 wave_or_u64(u64 v) {
    return ((u64)wave_or_u32((unsigned)(v >> 32)) << 32) | (u64)wave_or_u32((unsigned)v);
}
// ---- quaternion rotation: rotateVectorByQuaternion dsp_dynamic.h:1303-1322
// (att * (0,v) * att.inverse(); Hamilton product in Eigen's generic operand order)
__device__ __forceinline__ void quat_mul(const float a[4], const float b[4], float r[4]) {
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    r[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}
__device__ __forceinline__ void rotate_by_quat(float vx, float vy, float vz, const float q[4], float out[3]) {
    float vq[4] = {0.f, vx, vy, vz};
    float n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + q[0] * q[0];
    float inv[4] = {__fdiv_rn(q[0], n2), __fdiv_rn(-q[1], n2), __fdiv_rn(-q[2], n2), __fdiv_rn(-q[3], n2)};
    float t[4], r[4];
    quat_mul(q, vq, t);
    quat_mul(t, inv, r);
    out[0] = r[1]; out[1] = r[2]; out[2] = r[3];
}

// ---- vectorMultiply :1324-1326
__device__ __forceinline__ float dot3(float x, float y, float z, const float* n) {
    return x * n[0] + y * n[1] + z * n[2];
}

// ---- ifInPyramidsArea :1329-1339 + findPointPyramid{Horizontal,Vertical}Index :1341-1367.
// The reference scans the 28 / 16 boundary planes linearly for the first sign
// change; the dot products are monotone along the plane index for any point
// inside the FOV wedge, so a binary search with the SAME predicate (same
// rotated normals, same dot expression) returns the same cell with 9 instead
// of 44 dot products.  Returns h*np_v+v, or -1 outside the FOV.
__device__ __forceinline__ int pyramid_of(const MapDims& d, const float* ph, const float* pv, float x, float y, float z) {
    if (!(dot3(x, y, z, ph) >= 0.f && dot3(x, y, z, ph + 3 * d.np_h) <= 0.f &&
          dot3(x, y, z, pv) <= 0.f && dot3(x, y, z, pv + 3 * d.np_v) >= 0.f))
        return -1;
    int lo = 0, hi = d.np_h - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (dot3(x, y, z, ph + 3 * (mid + 1)) <= 0.f) hi = mid; else lo = mid + 1;
    }
    const int h = lo;
    lo = 0; hi = d.np_v - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (dot3(x, y, z, pv + 3 * (mid + 1)) >= 0.f) hi = mid; else lo = mid + 1;
    }
    return h * d.np_v + lo;
}

// ---- getParticleVoxelsIndex :1076-1088 + ifParticleIsOut :1118-1125.
// (int)((p + half) / res) needs the CORRECTLY ROUNDED fp32 quotient (Appendix A-10: a plain reciprocal multiply would move
// particles that sit on voxel faces).  The IEEE division costs ~11 VALU instructions; with y = RN(1 / res) the sequence
//   q0 = a * y;  r = fma(-q0, res, a);  q = fma(r, y, q0)
// (Markstein) returns the same correctly rounded quotient in 3.  MapDims::div_ok is set only after a kernel has compared
// the two, bit for bit, for EVERY float a in [0, 2 * half] the map can produce (k_verify_div at device initialisation);
// otherwise the IEEE division stays.  Returns the GLOBAL index.
__device__ __forceinline__ float div_res(const MapDims& d, float a) {
    if (d.div_ok) {
        const float q0 = a * d.rcp_res;
        const float r = __fmaf_rn(-q0, d.res, a);
        return __fmaf_rn(r, d.rcp_res, q0);
    }
    return __fdiv_rn(a, d.res);
}
__device__ __forceinline__ bool voxel_of(const MapDims& d, float px, float py, float pz, int& gidx) {
    if (fabsf(px) >= d.half_x || fabsf(py) >= d.half_y || fabsf(pz) >= d.half_z) return false;   // (no NaN ever gets here: >= is false, as in :1118-1125)
    const int x = (int)div_res(d, px + d.half_x);
    const int y = (int)div_res(d, py + d.half_y);
    const int z = (int)div_res(d, pz + d.half_z);
    // z * ny * nx + y * nx + x, all factors below 2^24 (z * ny * nx < v_glob < 2^31; 24-bit multiplies are full rate)
    gidx = (int)(__umul24((unsigned)z, (unsigned)(d.ny * d.nx)) + __umul24((unsigned)y, (unsigned)d.nx) + (unsigned)x);
    return (unsigned)gidx < (unsigned)d.v_glob;
}

// ---- storage order of the voxels (MapDims::tiling): the device arrays are indexed with lv; the reference's voxel index (:1081) is kept for
// everything that ORDERS (sweep keys) or leaves the library (results, state records).
__device__ __forceinline__ int lv_of_xyz(const MapDims& d, int x, int y, int zl) {   // zl = z - z_lo, inside the slab
    if (!d.tiling) return (int)(__umul24((unsigned)zl, (unsigned)(d.ny * d.nx)) + __umul24((unsigned)y, (unsigned)d.nx) + (unsigned)x);
    return ((((zl >> 2) * d.ncy + (y >> 2)) * d.ncx + (x >> 2)) << 6) | ((zl & 3) << 4) | ((y & 3) << 2) | (x & 3);
}
// the slab's voxel t in index order (t = global index - v_base) -> storage
__device__ __forceinline__ int lv_of_true(const MapDims& d, int t) {
    if (!d.tiling) return t;
    const int zc = d.ny * d.nx;
    const int zl = t / zc, rest = t - zl * zc, y = rest / d.nx, x = rest - y * d.nx;
    return lv_of_xyz(d, x, y, zl);
}
// the reference's GLOBAL voxel index -> storage; -1 outside this rank's slab
__device__ __forceinline__ int lv_of_g(const MapDims& d, int gv) {
    const int t = gv - d.v_base;
    return (t >= 0 && t < d.v_true) ? lv_of_true(d, t) : -1;
}
// storage -> the reference's GLOBAL voxel index; -1 for a padding voxel of a cube that sticks out of the map
__device__ __forceinline__ int g_of_lv(const MapDims& d, int lv) {
    if (!d.tiling) return lv + d.v_base;
    const int c = lv >> 6;
    const int cx = c % d.ncx, r = c / d.ncx, cy = r % d.ncy, cz = r / d.ncy;
    const int x = cx * 4 + (lv & 3), y = cy * 4 + ((lv >> 2) & 3), zl = cz * 4 + ((lv >> 4) & 3);
    if (x >= d.nx || y >= d.ny || zl >= d.z_hi - d.z_lo) return -1;
    return ((d.z_lo + zl) * d.ny + y) * d.nx + x;
}
// the same for the sweeps, one tile at a time: global index of (tile, lane) = tile_gbase(tile) + lane_goff(lane); the base is wave-uniform
// (integer divisions on the scalar unit, once per tile), the lane's offset depends on the lane only.  (Padding lanes get an index of no
// meaning: no particle ever lives there.)
__device__ __forceinline__ int tile_gbase(const MapDims& d, int BX) {
    if (!d.tiling) return d.v_base + BX * 64;
    const int cx = BX % d.ncx, r = BX / d.ncx, cy = r % d.ncy, cz = r / d.ncy;
    return ((d.z_lo + cz * 4) * d.ny + cy * 4) * d.nx + cx * 4;
}
__device__ __forceinline__ int lane_goff(const MapDims& d, int lane) {
    return d.tiling ? (((lane >> 4) & 3) * d.ny + ((lane >> 2) & 3)) * d.nx + (lane & 3) : lane;
}
// the box of a tile's voxels in voxel units: [x0, x1] x [y0, y1] x [z0, z1] (global z), cube tiling only
__device__ __forceinline__ void cube_box(const MapDims& d, int BX, int& x0, int& y0, int& z0) {
    const int cx = BX % d.ncx, r = BX / d.ncx, cy = r % d.ncy, cz = r / d.ncy;
    x0 = cx * 4; y0 = cy * 4; z0 = d.z_lo + cz * 4;
}
// getParticleVoxelsIndex (:1076-1088) with the storage index as well: lv = -1 when the voxel lies in another rank's slab
__device__ __forceinline__ bool voxel_of_lv(const MapDims& d, float px, float py, float pz, int& gidx, int& lv) {
    if (fabsf(px) >= d.half_x || fabsf(py) >= d.half_y || fabsf(pz) >= d.half_z) return false;
    const int x = (int)div_res(d, px + d.half_x);
    const int y = (int)div_res(d, py + d.half_y);
    const int z = (int)div_res(d, pz + d.half_z);
    gidx = (int)(__umul24((unsigned)z, (unsigned)(d.ny * d.nx)) + __umul24((unsigned)y, (unsigned)d.nx) + (unsigned)x);
    if (!((unsigned)gidx < (unsigned)d.v_glob)) return false;
    const int zl = z - d.z_lo;
    lv = (unsigned)zl < (unsigned)(d.z_hi - d.z_lo) ? lv_of_xyz(d, x, y, zl) : -1;
    return true;
}

// ---- queryNormalPDF :1294-1301 reproduced arithmetically.  The reference's
// LUT (calculateNormalPDFBuffer :1288-1292) holds c*exp(-t^2/2) at
// t = (i-10000)*0.001 with c = 1/sqrt(pi); the index truncates
// z*1000+10000 after clamping z to +-9.9.  We compute the same quantised t per
// axis and ONE exp for the product of the three axis factors:
//   g(x)g(y)g(z) = c^3 * exp(-(tx^2+ty^2+tz^2)/2).
// The division (x-mu)/sigma by the frame-constant sigma is done as a reciprocal
// multiply plus one FMA residual correction (q = d*y; r = fma(-q,sigma,d);
// q' = fma(r,y,q)), which returns the correctly rounded quotient (Markstein),
// i.e. the same quantisation bin as the reference, for 3 VALU ops instead of ~10.
// axis_u returns the LUT index relative to the table centre as a float, u = i - 10000 with
// i = (int)(z*1000 + 10000) (:1296-1299).  z*1000 + 10000 is positive after the clamp, so the truncation is a
// floor (one v_floor instead of two conversions) and the clamp is one v_med3.  The axis factor is
// c * exp(-t^2/2) with t = u * 0.001; pair_gk sums the three u^2 (integers up to 9.8e7, fp32-exact to 6e-8
// relative) and folds 0.001^2 / 2 and log2(e) into the single exp2: ~30 VALU operations per pair.
// pair_gk2 below does two pairs per lane with packed fp32 (v_pk_add / v_pk_mul / v_pk_fma).  On gfx950 that is NOT a
// higher FLOP rate (a wave64 fp32 instruction issues in ~2.8 cycles at full occupancy, the packed one in ~5.3,
// tools/micro/valu_bench.hip) but fewer instructions per pair, which helps when few waves share a SIMD: 3-4 % on the
// pair kernels at the metric's workload, nothing at saturation.
__device__ __forceinline__ float axis_u(float a, float mu, float sigma, float inv_sigma) {
    const float dlt = a - mu;
    const float q0 = dlt * inv_sigma;
    const float rr = __fmaf_rn(-q0, sigma, dlt);
    float z = __fmaf_rn(rr, inv_sigma, q0);
    z = __builtin_amdgcn_fmed3f(z, -9.9f, 9.9f);
    return floorf(z * 1000.f + 10000.f) - 10000.f;   // same two roundings as the reference's (int)(z*1000+10000)
}
__device__ __forceinline__ float pair_gk(float px, float py, float pz, float ox, float oy, float oz,
                                         float sigma, float inv_sigma, float c3) {
    const float ux = axis_u(px, ox, sigma, inv_sigma);
    const float uy = axis_u(py, oy, sigma, inv_sigma);
    const float uz = axis_u(pz, oz, sigma, inv_sigma);
    const float s = ux * ux + uy * uy + uz * uz;
    // exp(-0.5 * 1e-6 * s) = exp2(s * (-0.5e-6 * log2(e)))
    return c3 * __builtin_amdgcn_exp2f(s * -7.213475204444817e-07f);
}

// Exactly the operations of axis_u / pair_gk, component by component: bit-identical results; only the clamp, the floor
// and the exp2 have no packed form.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v axis_u2(f2v a, f2v mu, float sigma, float inv_sigma) {
    const f2v dlt = a - mu;
    const f2v q0 = dlt * inv_sigma;
    const f2v rr = __builtin_elementwise_fma(-q0, (f2v)(sigma), dlt);
    f2v z = __builtin_elementwise_fma(rr, (f2v)(inv_sigma), q0);
    z.x = __builtin_amdgcn_fmed3f(z.x, -9.9f, 9.9f);
    z.y = __builtin_amdgcn_fmed3f(z.y, -9.9f, 9.9f);
    f2v t = z * 1000.f + 10000.f;
    t.x = floorf(t.x); t.y = floorf(t.y);
    return t - 10000.f;
}
__device__ __forceinline__ f2v pair_gk2(f2v px, f2v py, f2v pz, f2v ox, f2v oy, f2v oz, float sigma, float inv_sigma, float c3) {
    const f2v ux = axis_u2(px, ox, sigma, inv_sigma);
    const f2v uy = axis_u2(py, oy, sigma, inv_sigma);
    const f2v uz = axis_u2(pz, oz, sigma, inv_sigma);
    const f2v s = (ux * ux + uy * uy + uz * uz) * -7.213475204444817e-07f;
    f2v e;
    e.x = __builtin_amdgcn_exp2f(s.x); e.y = __builtin_amdgcn_exp2f(s.y);
    return e * c3;
}

// Ck accumulators are 64-bit fixed point in units of 2^-34 (5.8e-11; range +-5e8).  Ck >= kappa (1e-2 by default), whose
// fp32 ulp is 9e-10, so the fixed-point grid is finer than the float the sum is read back into, and integer atomics make
// the sum independent of arrival order (float atomics differ by an ulp from run to run, which the resampler's
// equal-weight ties amplify into different survivors).
#define CK_FIX_SCALE 17179869184.0
// round to the nearest multiple of 2^-34 (magic-number rounding, valid below 2^18): sums of snapped terms are exact in
// double while they stay below 2^19, far above any Ck the filter can produce
__device__ __forceinline__ double ck_snap(float a) { return __dsub_rn(__dadd_rn((double)a, 393216.0), 393216.0); }
__device__ __forceinline__ float ck_from_fix(long long v) { return (float)((double)v * (1.0 / CK_FIX_SCALE)); }

// Future-status accumulators (voxels_objects_number[v][4..], `+=` at :961): 64-bit fixed point in units of 2^-24 (6e-8 -- the
// rounding of an fp32 accumulator that holds ~1; weights entering the rollout are >= 1e-3, :941).  Every moving particle adds
// the SAME integer whichever path carries it (resampler's idle waves, k_rollout's LDS windows or its single atomics, any slab of a
// sharded map), and integer sums are associative: the future status is reproducible bit for bit and independent of the variant
// the handle picked.  A 32-bit LDS window cell holds up to 2^32 * 2^-24 = 256 units of weight.
#define FUT_FIX_SCALE 16777216.0f
#define FUT_FIX_INV 5.9604644775390625e-08
#define FUT_WINDOW_MAX_W 250.0f
__device__ __forceinline__ u64 fut_quantum(float w) { return (u64)__float2ull_rn(w * FUT_FIX_SCALE); }   // (NaN / negative -> 0)
__device__ __forceinline__ void fut_add(u64* cell, u64 q) { __hip_atomic_fetch_add(cell, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float fut_value(u64 q) { return (float)((double)q * FUT_FIX_INV); }

// entries of pyramid b's range-sorted particle list (what the pair kernels read): the list as registered, cut to the reference's
// capacity -- on a sharded map in a frame with a global cut, what THIS rank keeps of it
__device__ __forceinline__ int pyr_len(const MapDims& d, const DevState& s, int b) {
    return s.pyr_kept ? s.pyr_kept[b] : min(s.pyr_cnt[b], d.capp);
}

#define GU 16
// obs_gather_wave: one wave per pyramid (k_obs_gather; in a whole frame the extra workgroups of k_predict).  Appends matching points in INPUT order
// (stable, ballot + prefix popcount) to the pyramid's bin, keeps the first 99
// (count saturates, :279-284), tracks the max range over ALL matches (:275-277).
__device__ __forceinline__ void obs_gather_wave(const MapDims& d, const DevState& s, const int b) {
    const int n_pts = s.fpar->n_pts;
    const int l = lane_id();
    int count = 0;
    float maxlen = -1.f;
    for (int base = 0; base < n_pts; base += GU * WAVE) {
        int pid[GU];
#pragma unroll
        for (int k = 0; k < GU; ++k) {  // GU independent loads in flight: the scan is a chain of L2 round trips
            const int i = base + k * WAVE + l;
            pid[k] = i < n_pts ? s.pt_pyr[i] : -1;
        }
#pragma unroll
        for (int k = 0; k < GU; ++k) {
            const int i = base + k * WAVE + l;
            const bool match = pid[k] == b;
            const u64 m = __ballot(match);
            if (match) {
                const int pos = count + (int)__popcll(m & lanemask_lt());
                const float4 p = s.pt_rot[i];
                if (pos < DSP_OBS_CAP - 1) s.obs[b * DSP_OBS_CAP + pos] = p;
                maxlen = fmaxf(maxlen, p.w);
            }
            count += (int)__popcll(m);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxlen = fmaxf(maxlen, __shfl_xor(maxlen, o, WAVE));
    if (l == 0) {
        const int c = min(count, DSP_OBS_CAP - 1);
        s.obs_cnt[b] = c;
        s.obs_maxlen[b] = maxlen;
        if (c) atomicAdd(&s.fs->n_obs, c);
        if (count) atomicAdd(&s.fs->n_valid, count);   // valid_points :286 counts the overflowed points too
    }
}

// particle storage index: tiles of 64 voxels, slot-major inside a tile (see dspmap_sweep.hip)
__device__ __forceinline__ size_t pidx(const MapDims& d, int lv, int slot) {
    return ((size_t)(lv >> 6) * d.slots + slot) * 64 + (lv & 63);
}

// particle field records (see DevState): 12-byte positions, 8-byte velocities; one vector access each
struct P3 { float x, y, z; };
struct V2 { float x, y; };
__device__ __forceinline__ P3 ld_pos(const DevState& s, size_t idx) { return reinterpret_cast<const P3*>(s.pos)[idx]; }
__device__ __forceinline__ V2 ld_vel(const DevState& s, size_t idx) { return reinterpret_cast<const V2*>(s.vel)[idx]; }
__device__ __forceinline__ void st_pos(const DevState& s, size_t idx, float x, float y, float z) {
    P3 v; v.x = x; v.y = y; v.z = z;
    reinterpret_cast<P3*>(s.pos)[idx] = v;
}
__device__ __forceinline__ void st_vel(const DevState& s, size_t idx, float x, float y) {
    V2 v; v.x = x; v.y = y;
    reinterpret_cast<V2*>(s.vel)[idx] = v;
}

// Whoever gives a particle a velocity the map has not seen before notes its size (FrameScalars::vmax_bits; non-negative floats order like
// their bit patterns).  `cur` = the word as the caller read it earlier (any stale value is fine: the maximum only grows); the atomic is
// issued only by a lane that exceeds what memory holds NOW, i.e. a handful of times in a map's life.
__device__ __forceinline__ void note_speed(const DevState& s, float vx, float vy) {
    const int b = __float_as_int(fmaxf(fabsf(vx), fabsf(vy)));   // (a NaN component is ignored by fmaxf; inf makes every tile a halo tile: still correct)
    if (b > 0 && b > __hip_atomic_load(&s.fs->vmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&s.fs->vmax_bits, b);
}

// The same records through a buffer descriptor of one tile's cells: the lane's byte offset is a single register whatever
// the row, the row enters as a wave-uniform scalar offset (cells before the row) -- no 64-bit address per load, so a whole
// batch of rows can be in flight without an address register pair each.
typedef __amdgpu_buffer_rsrc_t brsrc;
typedef float f3v __attribute__((ext_vector_type(3)));
typedef float f2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ P3 bl_pos(brsrc r, int lane, int srow_cells) {
    const f3v v = __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(r, lane * 12, srow_cells * 12, 0));
    P3 p; p.x = v.x; p.y = v.y; p.z = v.z;
    return p;
}
__device__ __forceinline__ V2 bl_vel(brsrc r, int lane, int srow_cells) {
    const f2w v = __builtin_bit_cast(f2w, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, srow_cells * 8, 0));
    V2 p; p.x = v.x; p.y = v.y;
    return p;
}
__device__ __forceinline__ float bl_w(brsrc r, int lane, int srow_cells) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, srow_cells * 4, 0));
}
__device__ __forceinline__ void bs_pos(brsrc r, int lane, int srow_cells, float x, float y, float z) {
    f3v v; v.x = x; v.y = y; v.z = z;
    typedef unsigned u3v __attribute__((ext_vector_type(3)));
    __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u3v, v), r, lane * 12, srow_cells * 12, 0);
}

// claim the lowest free slot of a voxel: first-free-slot rule of addAParticle /
// moveParticle (:1184-1185,1214-1215) as one atomic OR per attempt.
// Returns the slot or -1 if the voxel is full.
__device__ __forceinline__ int claim_slot(u64* mask, int lv, const MapDims& d) {
    for (int wi = 0; wi < d.mw; ++wi) {
        u64* wp = mask + (size_t)lv * d.mw + wi;
        const int nbits = min(64, d.slots - wi * 64);
        const u64 valid = nbits == 64 ? ~0ull : ((1ull << nbits) - 1ull);
        u64 cur = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (true) {
            const u64 free_bits = ~cur & valid;
            if (!free_bits) break;
            const u64 bit = free_bits & (~free_bits + 1ull);
            const u64 prev = atomicOr(wp, bit);
            if (!(prev & bit)) return wi * 64 + (__ffsll((long long)bit) - 1);
            cur = prev | bit;
        }
    }
    return -1;
}

// ---- cross-queue hand-over words (DevState::xq; DSPMAP_P_ESTIMATOR_QUEUE).  The estimator's kernels run on a hardware queue of their own and
// meet the captured frame through sequence numbers in HBM: a publisher stores "ring position + 1" with an agent-scope atomic (after a
// release fence where data it wrote inside the SAME kernel has to be visible); a waiter -- ONE lane of a workgroup, before the workgroup
// reads anything the other queue wrote -- polls until the word has reached its frame's number.  In the usual case the word is there at
// the first look and the kernel boundary in front of the waiter has already made the data visible; after a real wait an acquire fence
// drops what the caches may hold.  A wait is bounded (200 ms of the 100 MHz wall clock): the pairing of the two queues is the host's job
// and a missing partner must not hang the device -- the give-up is noted in host-mapped memory and the next call fails on it.
__device__ __forceinline__ void xq_publish(int* word, int seq) {
    __hip_atomic_store(word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xq_wait(int* word, int want, int* gave_up) {
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want < 0) {
            if (wall_clock64() - t0 > 20000000ll) { *gave_up = want; break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    // ALWAYS an acquire (one lane: it invalidates this CU's L1, nothing is written back -- the cost that mattered was the release's): a
    // workgroup that finds the word set at its first look may still hold lines of the other queue's data from before they were written
    // (the word can arrive between this kernel's launch-time invalidate and the look), ADVICE r5
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// the same for a workgroup that only LOOKS (no wait): call after a look that found the word ready, before reading the other queue's data
__device__ __forceinline__ void xq_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
