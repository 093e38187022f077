// dspmap_api.hip -- host runtime + C ABI (include/dspmap.h) of libdspmap_hip.so.
// Host logic restates DSPMap::update's gating / delta-pose preamble
// (reference include/dsp_dynamic.h:187-218) and owns the device state; all
// per-particle / per-voxel work is in dspmap_kernels.hip.  There is no CPU
// compute path here: without a usable HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <random>
#include <string>
#include <thread>
#include <map>
#include <mutex>
#include <vector>

#include "dspmap_internal.h"

void dspmap_prof_mark(dspmap* m, int i) {
    if (m->prof) (void)hipEventRecord(m->pev[i], m->stream);
}
void dspmap_prof_collect(dspmap* m) {
    if (!m->prof || !m->prof_pending) return;
    (void)hipEventSynchronize(m->pev[DSPMAP_N_STAGES]);
    for (int i = 0; i < DSPMAP_N_STAGES; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->pev[i], m->pev[i + 1]) == hipSuccess) m->stage_ms[i] += ms;
    }
    m->prof_frames++;
    m->prof_pending = false;
}

int dspmap_fail(dspmap* m, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (m) m->err = buf;
    return code;
}


void dspmap_resample(dspmap* m, const LaunchCtx& c) {   // the stage's launches + which variant they were (dspmap_debug_rollout_paths)
    m->last_resample_variant = resample_variant(c);
    launch_resample(c);
}

LaunchCtx dspmap_ctx_of(dspmap* m) {
    LaunchCtx c;
    c.d = m->d; c.fp = m->fp; c.s = m->s; c.k = m->k; c.stream = m->stream;
    c.pt_cap = m->pt_cap; c.birth_cap = m->birth_cap; c.n_cu = m->n_cu;
    c.k.nbsnap = m->nb_dirty ? m->nbsnap_buf : nullptr;
    c.ve = m->ve;
    // which k_predict: the SPARSE variant while most tiles are empty (both give the same result; the estimate is a few frames old)
    if (m->sparse_force >= 0) m->sparse_mode = m->sparse_force != 0;
    else if (m->hint_host && m->k.ntiles >= 4096) {
        const long long est = 64ll * *m->hint_host;
        if (!m->sparse_mode && est * 4 < m->k.ntiles) m->sparse_mode = true;
        else if (m->sparse_mode && est * 2 > m->k.ntiles) m->sparse_mode = false;
    }
    c.sparse = m->sparse_mode;
    if (m->ro_force >= 0) m->ro_kernel = m->ro_force == 0;
    else if (m->hint_host) {
        const int heavy = m->hint_host[1];
        if (!m->ro_kernel && heavy * 32 > m->k.ntiles) m->ro_kernel = true;
        else if (m->ro_kernel && heavy * 128 < m->k.ntiles) m->ro_kernel = false;
    }
    c.ro_inline = !m->ro_kernel;
    c.resample_wg_tiles = m->resample_wg_tiles;
    c.side_wg = m->side_wg;
    c.sweep_rev = (m->sweep_alt < 0 ? m->k.ntiles >= 4096 : m->sweep_alt == 1) && (m->frame_parity & 1u);
    c.resample_rev = m->sweep_alt == 2 ? true : c.sweep_rev;   // 2: k_predict up, k_place down, k_resample down -- and the next k_predict starts where it ended
    return c;
}

extern "C" void dspmap_default_config(dspmap_config* c) {  // dsp_dynamic.h:38-50
    memset(c, 0, sizeof(*c));
    c->nx = 66; c->ny = 66; c->nz = 40;
    c->voxel_resolution = 0.15f;
    c->angle_resolution = 3;
    c->max_particle_num_voxel = 9;
    c->half_fov_h = 42; c->half_fov_v = 24;
    c->prediction_times = 6;
    const float t[6] = {0.05f, 0.2f, 0.5f, 1.f, 1.5f, 2.f};
    memcpy(c->prediction_future_time, t, sizeof(t));
    c->device = -1;
}

static void derive_dims(dspmap* m) {
    const dspmap_config& c = m->cfg;
    MapDims& d = m->d;
    memset(&d, 0, sizeof(d));
    d.nx = c.nx; d.ny = c.ny; d.nz = c.nz;
    d.z_lo = c.z_lo; d.z_hi = c.z_hi;
    if (d.z_lo == 0 && d.z_hi == 0) d.z_hi = c.nz;
    d.v_true = c.nx * c.ny * (d.z_hi - d.z_lo);
    d.v_base = d.z_lo * c.nx * c.ny;
    // storage order (MapDims::tiling): cubes of 4 x 4 x 4 voxels on unsharded maps large enough for the two-branch frame (the maps that split
    // their placement), runs of 64 voxel indices otherwise -- DSPMAP_P_TILING / DSPMAP_TILING force either (before the device is initialised)
    const bool whole = d.z_lo == 0 && d.z_hi == c.nz;
    const long long needle_tiles = ((long long)d.v_true + 63) / 64;
    d.tiling = m->tiling_req >= 0 ? (m->tiling_req != 0 ? 1 : 0) : ((whole && needle_tiles >= m->place_split_tiles && m->place_split_tiles > 1) ? 1 : 0);
    d.ncx = (c.nx + 3) / 4; d.ncy = (c.ny + 3) / 4; d.ncz = (d.z_hi - d.z_lo + 3) / 4;
    if (d.tiling && (double)d.ncx * d.ncy * d.ncz * 64.0 * ((c.safe_particle_factor > 0 ? c.safe_particle_factor : 2) * c.max_particle_num_voxel) >= 2147483648.0) d.tiling = 0;   // (cell indices are 31-bit, padding included)
    d.v_loc = d.tiling ? d.ncx * d.ncy * d.ncz * 64 : d.v_true;
    d.v_glob = c.nx * c.ny * c.nz;                                     // :62
    d.M = c.max_particle_num_voxel;
    d.slots = (c.safe_particle_factor > 0 ? c.safe_particle_factor : 2) * d.M;   // :65 (x2); dsp_static.h:63 uses x5
    d.nn = c.pyramid_neighbor_n > 0 ? c.pyramid_neighbor_n : 1;
    d.nbins = (2 * d.nn + 1) * (2 * d.nn + 1);
    d.static_model = c.static_model ? 1 : 0;
    d.tile_skip = 1;
    d.mw = (d.slots + 63) / 64;
    const int A = c.angle_resolution;
    d.np_h = c.half_fov_h * 2 / A;                                     // :58
    d.np_v = c.half_fov_v * 2 / A;                                     // :59
    d.np = d.np_h * d.np_v;                                            // :60
    const int pyramid_num = 360 * 180 / A / A;                         // :63
    const int safe_particle_num = (int)((double)d.v_glob * d.M + 1e5); // :64
    d.capp = safe_particle_num / pyramid_num * 2;                      // :66
    d.capa = 2 * d.capp + 64;
    d.T = c.prediction_times;
    d.res = c.voxel_resolution;
    d.half_x = (d.res * (float)c.nx) * 0.5f;                           // :528-530
    d.half_y = (d.res * (float)c.ny) * 0.5f;
    d.half_z = (d.res * (float)c.nz) * 0.5f;
    for (int i = 0; i < d.T; ++i) d.pred_t[i] = c.prediction_future_time[i];
    d.rng_inv_bw = (float)PS_NBK / sqrtf(d.half_x * d.half_x + d.half_y * d.half_y + d.half_z * d.half_z);
}

int dspmap_need_index_order(dspmap* m) {
    if (!m->d.tiling) return DSPMAP_OK;
    if (m->device_ready) return dspmap_fail(m, DSPMAP_E_STATE, "the sharded frame needs index-order storage: set DSPMAP_P_TILING = 0 before the handle's first use");
    m->tiling_req = 0;
    derive_dims(m);
    return DSPMAP_OK;
}

static void refresh_fp(dspmap* m) {
    FilterParams& f = m->fp;
    f.inv_sigma_ob = 1.f / f.sigma_ob;
    const float pi_2 = 1.57079632679489661923f;
    f.pdf_c = 1.f / sqrtf(2.f * pi_2);  // standardNormalPDF :1284 at 0
    f.pdf_c3 = f.pdf_c * f.pdf_c * f.pdf_c;
    f.cull_r = m->cull_sigmas * f.sigma_ob;
}

extern "C" dspmap_t* dspmap_create(const dspmap_config* cfg) {
    if (!cfg) return nullptr;
    if (cfg->nx <= 0 || cfg->ny <= 0 || cfg->nz <= 0 || cfg->voxel_resolution <= 0.f) return nullptr;
    if ((long long)cfg->nx * cfg->ny >= (1ll << 24) || cfg->nz >= (1 << 24)) return nullptr;   // voxel_of multiplies 24-bit factors
    if (cfg->angle_resolution <= 0 || cfg->max_particle_num_voxel <= 0 || cfg->max_particle_num_voxel > 64) return nullptr;
    if (cfg->prediction_times < 0 || cfg->prediction_times > DSPMAP_MAX_PRED_TIMES) return nullptr;
    if (cfg->z_lo < 0 || cfg->z_hi > cfg->nz || cfg->z_lo > cfg->z_hi) return nullptr;
    if (cfg->pyramid_neighbor_n < 0 || cfg->pyramid_neighbor_n > 2 || cfg->safe_particle_factor < 0) return nullptr;
    if ((cfg->safe_particle_factor > 0 ? cfg->safe_particle_factor : 2) * cfg->max_particle_num_voxel > 128) return nullptr;  // two occupancy words
    if ((double)cfg->nx * cfg->ny * cfg->nz * (cfg->safe_particle_factor > 0 ? cfg->safe_particle_factor : 2) * cfg->max_particle_num_voxel >= 2147483648.0)
        return nullptr;  // cell indices and sweep keys are 31-bit
    dspmap* m = new dspmap();
    m->cfg = *cfg;
    derive_dims(m);
    if (m->d.np_h + 1 > DSP_MAX_PLANES_H || m->d.np_v + 1 > DSP_MAX_PLANES_V || m->d.np <= 0) { delete m; return nullptr; }
    memset(&m->s, 0, sizeof(m->s));
    memset(&m->k, 0, sizeof(m->k));
    FilterParams& f = m->fp;
    memset(&f, 0, sizeof(f));
    f.sigma_ob = 0.2f; f.kappa = 0.01f; f.p_det = 0.95f;  // :156-158
    f.nb_weight = 0.04f; f.nb_num = 20;                   // :162-163
    f.occl_margin = 0.3f;                                 // :70
    f.tab_n = 1; f.rtab_n = 1;
    refresh_fp(m);
    m->device = cfg->device;
    m->vel.configure(cfg->half_fov_h, cfg->half_fov_v, cfg->angle_resolution);
    if (const char* e = getenv("DSPMAP_PLACE_SPLIT_TILES")) { const long v = atol(e); if (v > 0) m->place_split_tiles = (int)std::min(v, 2000000000l); }
    if (const char* e = getenv("DSPMAP_SWEEP_ALTERNATE")) m->sweep_alt = atoi(e) < 0 ? -1 : (atoi(e) >= 2 ? 2 : (atoi(e) != 0 ? 1 : 0));
    if (const char* e = getenv("DSPMAP_USE_GRAPH")) { m->use_graph = atoi(e) == 1; m->direct_ring = atoi(e) == 2; }
    if (const char* e = getenv("DSPMAP_ESTIMATOR_QUEUE")) m->est_queue = atoi(e) != 0;
    if (const char* e = getenv("DSPMAP_XQ_TEST_DELAY_US")) m->xq_test_delay_us = std::max(0, std::min(atoi(e), 100000));
    m->xq_test_break = getenv("DSPMAP_XQ_TEST_BREAK") != nullptr;   // (test hooks are read HERE, once: never in a frame)
    if (const char* e = getenv("DSPMAP_XQ_FORCE")) m->xq_force = !strcmp(e, "shared") ? 1 : (!strcmp(e, "apart") ? 2 : 0);
    if (const char* e = getenv("DSPMAP_TILING")) m->tiling_req = atoi(e) < 0 ? -1 : (atoi(e) != 0 ? 1 : 0);
    if (const char* e = getenv("DSPMAP_FRAME_BRANCHES")) m->frame_branches = atoi(e) < 0 ? -1 : (atoi(e) != 0 ? 1 : 0);
    if (const char* e = getenv("DSPMAP_RESAMPLE_SPLIT")) m->resample_split = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("DSPMAP_TILE_BITMAPS")) m->tile_bitmaps = atoi(e) != 0;
    if (const char* e = getenv("DSPMAP_SIDE_PLACEMENT")) { const int iv = atoi(e); if (iv > 0 && (iv >> 4) <= 2 && (iv & 15)) { m->side_fork = iv >> 4; m->side_wg = iv & 15; } }
    if (const char* e = getenv("DSPMAP_RESAMPLE_WG_TILES")) { const long v = atol(e); if (v >= 0) m->resample_wg_tiles = (int)std::min(v, 2000000000l); }
    derive_dims(m);   // (the storage order follows DSPMAP_TILING / DSPMAP_PLACE_SPLIT_TILES)
    return m;
}

static void free_dev(dspmap* m) {
    if (!m->device_ready) return;
    const bool dbg = getenv("DSPMAP_DEBUG_DESTROY") != nullptr;
    auto chk = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && dbg) fprintf(stderr, "[dspmap destroy] %s: %s\n", what, hipGetErrorString(e));
    };
    if (m->device >= 0) chk(hipSetDevice(m->device), "hipSetDevice");
    if (m->stream) chk(hipStreamSynchronize(m->stream), "hipStreamSynchronize");   // nothing of this handle may be in flight
    if (m->stream2) chk(hipStreamSynchronize(m->stream2), "hipStreamSynchronize(2)");
    if (m->stream4) chk(hipStreamSynchronize(m->stream4), "hipStreamSynchronize(4)");
    if (m->stream3) chk(hipStreamSynchronize(m->stream3), "hipStreamSynchronize(3)");
    for (hipGraphExec_t& g : m->graph_exec) if (g) { chk(hipGraphExecDestroy(g), "hipGraphExecDestroy"); g = nullptr; }
    if (m->graph) chk(hipGraphDestroy(m->graph), "hipGraphDestroy");
    DevState& s = m->s;
    dspmap_dist_free(m);
    if (m->mgpu_bound && !m->mgpu_self_bound) { s.obs_ck = nullptr; s.nstatic = nullptr; }  // caller-owned
    if (m->mgpu_count) chk(hipFree(m->mgpu_count), "hipFree");
    void* ptrs[] = {s.fpar, s.obs_ckf, s.part_inv, s.fut_stat, s.mask, s.nbmask, s.pos, s.vel, s.w, s.vz0, s.res4, s.fut, s.fut_out, s.obs, s.obs_ck,
                    s.obs_cnt, s.obs_maxlen, s.planes_h, s.planes_v, s.planes_h0, s.planes_v0, s.pt_rot, s.pt_pyr,
                    s.birth, s.plan, s.plan_pbase, s.plan_inside, s.nstatic, s.fov_rec, s.fov_slot, s.fov_key, s.fov_spos, s.fov_rec_s, s.fov_slot_s, s.pyr_cnt, s.in_n, s.pmask, s.ta, s.dflag, s.dirty,
                    s.blk_cnt, s.occ_xyz, s.p_tab, s.v_tab, s.r_tab, s.fs, m->k.mv_rec, m->k.ro_cnt, m->k.in_rec, m->k.in_cnt, m->k.tile_bits, m->k.omask, m->k.ck_items, m->k.wu_items, m->k.n_items, m->k.nb_tab, m->k.expmask,
                    s.tile_moving, m->k.ro_stat, m->k.ro_sub, m->k.part_predict, m->k.tile_fov, m->k.view_list, m->k.tile_cls, s.tile_live, s.fut_dirty, m->k.part_resample, m->k.vb_cnt, m->k.vb_idx, m->k.work_list, m->k.child, m->k.part_birth, m->k.vz_q, m->pts_dev, s.birth_ovf, s.birth_cvr};
    for (void* p : ptrs) if (p) chk(hipFree(p), "hipFree");
    if (m->res_true) chk(hipFree(m->res_true), "hipFree");
    if (m->nbsnap_buf) chk(hipFree(m->nbsnap_buf), "hipFree");
    {
        void* vp[] = {m->ve.ng_view, m->ve.edges, m->ve.ecnt, m->ve.w, m->ve.root, m->ve.rank, m->ve.by_rank, m->ve.dyn_list, m->ve.cl, m->ve.last, m->ve.n,
                      m->ve.v_rot, m->ve.v_pyr, m->ve.v_fpar};
        for (void* q : vp) if (q) chk(hipFree(q), "hipFree");
    }
    if (m->pp_box) chk(hipFree(m->pp_box), "hipFree");
    if (m->pp_acc) chk(hipFree(m->pp_acc), "hipFree");
    if (m->pp_blk) chk(hipFree(m->pp_blk), "hipFree");
    for (int k = 0; k < DSPMAP_PTS_RING; ++k) {
        if (m->pts_ring[k]) chk(hipHostFree(m->pts_ring[k]), "hipHostFree");
        if (m->pts_ring_ev[k]) chk(hipEventDestroy(m->pts_ring_ev[k]), "hipEventDestroy");
        m->pts_ring[k] = nullptr; m->pts_ring_ev[k] = nullptr; m->pts_ring_busy[k] = false; m->pts_ring_cap[k] = 0;
    }
    m->pts_pin = nullptr; m->pts_pin_cap = 0;
    if (m->cring_host) { chk(hipHostFree(m->cring_host), "hipHostFree"); m->cring_host = nullptr; m->cring_dev = nullptr; m->cring_cap = 0; }
    if (m->birth_pin) chk(hipHostFree(m->birth_pin), "hipHostFree");
    if (m->birth_ev) { chk(hipEventDestroy(m->birth_ev), "hipEventDestroy"); m->birth_ev = nullptr; m->birth_ev_set = false; }
    if (m->ev_fork) chk(hipEventDestroy(m->ev_fork), "hipEventDestroy");
    if (m->ev_join) chk(hipEventDestroy(m->ev_join), "hipEventDestroy");
    if (m->ev_fork2) chk(hipEventDestroy(m->ev_fork2), "hipEventDestroy");
    for (hipEvent_t& e : m->ring_ev) if (e) { chk(hipEventDestroy(e), "hipEventDestroy"); e = nullptr; }
    if (m->ring_host) { chk(hipHostFree(m->ring_host), "hipHostFree"); m->ring_host = nullptr; }
    if (m->hint_host) { chk(hipHostFree((void*)m->hint_host), "hipHostFree"); m->hint_host = nullptr; }
    if (m->s.ring_seq) { chk(hipFree(m->s.ring_seq), "hipFree"); m->s.ring_seq = nullptr; }
    if (m->xq_dev) { chk(hipFree(m->xq_dev), "hipFree"); m->xq_dev = nullptr; }
    for (hipEvent_t e : m->pev) if (e) chk(hipEventDestroy(e), "hipEventDestroy(prof)");
    if (m->stream2) chk(hipStreamDestroy(m->stream2), "hipStreamDestroy(2)");
    if (m->stream4) chk(hipStreamDestroy(m->stream4), "hipStreamDestroy(4)");
    for (hipEvent_t e : m->ev_br) if (e) chk(hipEventDestroy(e), "hipEventDestroy");
    if (m->stream3) chk(hipStreamDestroy(m->stream3), "hipStreamDestroy(3)");
    if (m->ev0) chk(hipEventDestroy(m->ev0), "hipEventDestroy");
    if (m->ev1) chk(hipEventDestroy(m->ev1), "hipEventDestroy");
    if (m->own_stream && m->stream) chk(hipStreamDestroy(m->stream), "hipStreamDestroy");
    (void)hipGetLastError();   // a failure while tearing this handle down must not surface in another handle's next call
    m->device_ready = false;
}

extern "C" void dspmap_destroy(dspmap_t* m) {
    if (!m) return;
    free_dev(m);
    delete m;
}

extern "C" const char* dspmap_last_error(const dspmap_t* m) { return m ? m->err.c_str() : "null handle"; }

// the slab's voxel t in the reference's index order -> storage (lv_of_true of dspmap_device.h on the host): checkpoints hold the
// result grid and the accumulators in the reference's order, whatever order the map that wrote them stored them in
static inline size_t host_lv_of_true(const MapDims& d, size_t t) {
    if (!d.tiling) return t;
    const size_t zc = (size_t)d.ny * d.nx;
    const size_t zl = t / zc, rest = t - zl * zc, y = rest / d.nx, x = rest - y * d.nx;
    return ((((zl >> 2) * d.ncy + (y >> 2)) * d.ncx + (x >> 2)) << 6) | ((zl & 3) << 4) | ((y & 3) << 2) | (x & 3);
}

template <typename T>
static hipError_t dalloc(T** p, size_t n) {
    return hipMalloc((void**)p, sizeof(T) * (n ? n : 1));
}

// generateGaussianRandomsVectorZeroCenter :1150-1160 (same engine/distribution as the reference)
static void gen_gauss_tables(dspmap* m, unsigned seed) {
    const int n = m->cfg.gaussian_table_size > 0 ? m->cfg.gaussian_table_size : 10000000;  // :72
    m->h_ptab.resize(n); m->h_vtab.resize(n);
    std::default_random_engine random(seed);
    std::normal_distribution<double> n1(0, m->p_stddev);
    std::normal_distribution<double> n2(0, m->v_stddev);
    for (int i = 0; i < n; i++) { m->h_ptab[i] = (float)n1(random); m->h_vtab[i] = (float)n2(random); }
}
static void gen_rand_table(dspmap* m, unsigned seed) {
    // the reference draws uniforms with libc rand() after srand(time(0)) (:586,1551-1553);
    // a private random_r stream of the same generator is tabulated instead
    const int n = 1 << 22;
    m->h_rtab.resize(n);
    struct random_data rd;
    memset(&rd, 0, sizeof(rd));
    char state[128];
    initstate_r(seed, state, sizeof(state), &rd);
    for (int i = 0; i < n; i++) { int32_t r; random_r(&rd, &r); m->h_rtab[i] = r; }
}

static int upload_tables(dspmap* m) {
    DevState& s = m->s;
    m->graph_epoch++;
    if (s.p_tab) { (void)hipFree(s.p_tab); s.p_tab = nullptr; }
    if (s.v_tab) { (void)hipFree(s.v_tab); s.v_tab = nullptr; }
    const size_t n = m->h_ptab.size();
    HIPCHK(m, dalloc(&s.p_tab, n));
    HIPCHK(m, dalloc(&s.v_tab, n));
    HIPCHK(m, hipMemcpy(s.p_tab, m->h_ptab.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    HIPCHK(m, hipMemcpy(s.v_tab, m->h_vtab.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    m->fp.tab_n = (int)n;
    float pm = 0.f;   // how far from its observation a newborn can land (:871-873): FrameParams::birth_reach, k_tile_class
    for (size_t i = 0; i < n; ++i) { const float a = fabsf(m->h_ptab[i]); if (a > pm || a != a) pm = a != a ? INFINITY : a; }
    m->ptab_max = pm;
    return DSPMAP_OK;
}
static int upload_rtab(dspmap* m) {
    DevState& s = m->s;
    m->graph_epoch++;
    if (s.r_tab) { (void)hipFree(s.r_tab); s.r_tab = nullptr; }
    const size_t n = m->h_rtab.size();
    HIPCHK(m, dalloc(&s.r_tab, n));
    HIPCHK(m, hipMemcpy(s.r_tab, m->h_rtab.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    m->fp.rtab_n = (int)n;
    return DSPMAP_OK;
}

int dspmap_ensure_point_cap(dspmap* m, int n) {
    if (n <= m->pt_cap) return DSPMAP_OK;
    m->graph_epoch++;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    DevState& s = m->s;
    const int cap = n + n / 2 + 1024;
    if (m->mgpu_bound && !m->mgpu_self_bound) return dspmap_fail(m, DSPMAP_E_ARG, "%d points exceed the capacity bound with dspmap_mgpu_bind", n);
    BirthSrc* old_birth = s.birth;   // holds the cloud of the last non-empty view (re-used by frames with an empty one): carried over
    void* olds[] = {s.pt_rot, s.pt_pyr, s.plan, s.plan_pbase, s.plan_inside, s.nstatic, m->pts_dev, m->k.child, m->k.part_birth, s.birth_ovf, s.birth_cvr};
    for (void* p : olds) if (p) (void)hipFree(p);
    HIPCHK(m, dalloc(&s.pt_rot, (size_t)cap));
    HIPCHK(m, dalloc(&s.pt_pyr, (size_t)cap));
    HIPCHK(m, dalloc(&s.birth, (size_t)cap));
    if (old_birth) {
        if (m->pt_cap > 0) HIPCHK(m, hipMemcpy(s.birth, old_birth, sizeof(BirthSrc) * (size_t)m->pt_cap, hipMemcpyDeviceToDevice));
        (void)hipFree(old_birth);
    }
    HIPCHK(m, dalloc(&s.plan, (size_t)cap));
    HIPCHK(m, dalloc(&s.plan_pbase, (size_t)cap));
    HIPCHK(m, dalloc(&s.plan_inside, (size_t)cap));
    HIPCHK(m, dalloc(&s.nstatic, (size_t)cap));
    HIPCHK(m, dalloc(&s.birth_cvr, (size_t)cap + (size_t)cap / 16 + 2));
    HIPCHK(m, dalloc(&m->pts_dev, (size_t)cap * 3));
    HIPCHK(m, dalloc(&m->k.child, (size_t)cap * 32));
    HIPCHK(m, dalloc(&s.birth_ovf, (size_t)cap * 32));
    HIPCHK(m, dalloc(&m->k.part_birth, ((size_t)cap * 32 + 255) / 256 * 2));
    HIPCHK(m, hipMemset(m->k.part_birth, 0, sizeof(int) * (((size_t)cap * 32 + 255) / 256 * 2)));
    m->pt_cap = cap; m->birth_cap = cap;
    if (m->cring_host && m->cring_cap < std::min(cap, m->ve.cap)) {   // (the stream is idle: synchronised above) the cloud ring follows the capacity, up to what the device estimator takes
        (void)hipHostFree(m->cring_host);
        m->cring_host = nullptr; m->cring_dev = nullptr; m->cring_cap = 0;
    }
    return DSPMAP_OK;
}

extern "C" int dspmap_init_device(dspmap_t* m) {
    if (!m) return DSPMAP_E_ARG;
    if (m->device_ready) return DSPMAP_OK;
    int ndev = 0;
    const hipError_t e_cnt = hipGetDeviceCount(&ndev);
    if (e_cnt != hipSuccess || ndev <= 0)
        return dspmap_fail(m, DSPMAP_E_DEVICE, "no HIP device available (hipGetDeviceCount: %s, %d devices; libdspmap_hip has no CPU fallback)",
                           hipGetErrorString(e_cnt), ndev);
    if (m->device >= 0) HIPCHK(m, hipSetDevice(m->device));
    else HIPCHK(m, hipGetDevice(&m->device));
    if (!m->stream) { HIPCHK(m, hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)); m->own_stream = true; }
    HIPCHK(m, hipStreamCreateWithFlags(&m->stream2, hipStreamNonBlocking));
    for (hipEvent_t& e : m->ev_br) HIPCHK(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(m, hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    HIPCHK(m, hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    HIPCHK(m, hipEventCreateWithFlags(&m->ev_fork2, hipEventDisableTiming));
    for (hipEvent_t& e : m->ring_ev) HIPCHK(m, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(m, hipHostMalloc((void**)&m->ring_host, sizeof(FrameParams) * DSPMAP_RING, hipHostMallocMapped));
    memset(m->ring_host, 0, sizeof(FrameParams) * DSPMAP_RING);
    { void* dp = nullptr; HIPCHK(m, hipHostGetDevicePointer(&dp, m->ring_host, 0)); m->ring_dev = (const FrameParams*)dp; }
    HIPCHK(m, hipHostMalloc((void**)&m->hint_host, 4 * sizeof(int), hipHostMallocMapped));
    m->hint_host[0] = 1 << 24;   // (nothing known yet: not sparse)
    m->hint_host[1] = 0;
    m->hint_host[2] = 0;         // ring position behind the last frame whose first kernel is done with its parameter / cloud slot
    m->hint_host[3] = 0;
    { void* dp = nullptr; HIPCHK(m, hipHostGetDevicePointer(&dp, (void*)m->hint_host, 0)); m->s.hint_out = (int*)dp; }
    HIPCHK(m, hipMalloc((void**)&m->s.ring_seq, sizeof(int)));
    HIPCHK(m, hipMemset(m->s.ring_seq, 0, sizeof(int)));
    HIPCHK(m, hipMalloc((void**)&m->xq_dev, (XQ_LIST + DSPMAP_XQ_LIST) * sizeof(int)));
    HIPCHK(m, hipMemset(m->xq_dev, 0, (XQ_LIST + DSPMAP_XQ_LIST) * sizeof(int)));
    if (m->xq_test_delay_us > 0) { const int one = 1; HIPCHK(m, hipMemcpy(m->xq_dev + 10, &one, sizeof(int), hipMemcpyHostToDevice)); }   // (test hook, see queue_estimator)
    m->ring_head = 0;
    { int dev = 0, cu = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) m->n_cu = cu; }
    HIPCHK(m, hipEventCreate(&m->ev0));
    HIPCHK(m, hipEventCreate(&m->ev1));
    const MapDims& d = m->d;
    DevState& s = m->s;
    const size_t ntiles = ((size_t)d.v_loc + 63) / 64;
    const size_t S = ntiles * 64 * d.slots, W = (size_t)d.v_loc * d.mw;
    HIPCHK(m, dalloc(&s.mask, W)); HIPCHK(m, dalloc(&s.nbmask, W));
    HIPCHK(m, dalloc(&s.pos, 3 * S)); HIPCHK(m, dalloc(&s.vel, 2 * S)); HIPCHK(m, dalloc(&s.w, S));
    HIPCHK(m, dalloc(&s.res4, (size_t)d.v_loc));
    HIPCHK(m, dalloc(&s.fut, (size_t)d.v_loc * (d.T ? d.T : 1)));
    HIPCHK(m, dalloc(&s.fut_out, (size_t)d.v_loc * (d.T ? d.T : 1)));
    HIPCHK(m, dalloc(&s.fut_stat, (size_t)d.v_loc));
    HIPCHK(m, hipMemset(s.fut_stat, 0, sizeof(float) * (size_t)d.v_loc));
    HIPCHK(m, hipMemset(s.pos, 0, sizeof(float) * 3 * S)); HIPCHK(m, hipMemset(s.vel, 0, sizeof(float) * 2 * S));
    HIPCHK(m, hipMemset(s.w, 0, sizeof(float) * S));
    HIPCHK(m, dalloc(&s.obs, (size_t)d.np * DSP_OBS_CAP));
    HIPCHK(m, dalloc(&s.obs_ck, (size_t)d.np * DSP_OBS_CAP));
    HIPCHK(m, dalloc(&s.obs_ckf, (size_t)d.np * DSP_OBS_CAP));
    HIPCHK(m, dalloc(&s.part_inv, (size_t)d.np));
    HIPCHK(m, hipMemset(s.obs_ckf, 0, sizeof(float) * d.np * DSP_OBS_CAP));
    HIPCHK(m, hipMemset(s.part_inv, 0, sizeof(float) * d.np));
    HIPCHK(m, dalloc(&s.obs_cnt, (size_t)d.np));
    HIPCHK(m, dalloc(&s.obs_maxlen, (size_t)d.np));
    HIPCHK(m, dalloc(&s.planes_h, (size_t)(d.np_h + 1) * 3)); HIPCHK(m, dalloc(&s.planes_v, (size_t)(d.np_v + 1) * 3));
    HIPCHK(m, dalloc(&s.planes_h0, (size_t)(d.np_h + 1) * 3)); HIPCHK(m, dalloc(&s.planes_v0, (size_t)(d.np_v + 1) * 3));
    HIPCHK(m, dalloc(&s.fov_rec, (size_t)d.np * d.capa));
    HIPCHK(m, dalloc(&s.fov_slot, (size_t)d.np * d.capa));
    HIPCHK(m, dalloc(&s.fov_key, (size_t)d.np * d.capa));
    HIPCHK(m, dalloc(&s.fov_spos, (size_t)d.np * d.capa));
    HIPCHK(m, dalloc(&s.fov_rec_s, (size_t)d.np * d.capp));
    HIPCHK(m, dalloc(&s.fov_slot_s, (size_t)d.np * d.capp));
    HIPCHK(m, dalloc(&s.pyr_cnt, (size_t)d.np));
    HIPCHK(m, dalloc(&s.fs, (size_t)1));
    HIPCHK(m, dalloc(&s.fpar, (size_t)1));
    HIPCHK(m, hipMemset(s.fpar, 0, sizeof(FrameParams)));
    KernelScratch& k = m->k;
    k.ntiles = (int)ntiles;
    k.nblk_sweep = (int)((ntiles + 3) / 4);  // k_resample: 4 tiles (waves) per 256-thread block
    HIPCHK(m, dalloc(&k.mv_rec, ntiles * 64 * d.slots * 2));
    HIPCHK(m, dalloc(&k.in_rec, ntiles * 64 * d.slots * 2));
    HIPCHK(m, dalloc(&k.in_cnt, ntiles));
    {   // the tile bitmaps (DevState::vis_bits): three tables of one bit per tile, whole 64-bit words
        const size_t nw = (ntiles + 63) / 64 * 2;
        HIPCHK(m, dalloc(&k.tile_bits, 3 * nw));
        HIPCHK(m, hipMemset(k.tile_bits, 0, sizeof(unsigned) * 3 * nw));
    }
    HIPCHK(m, dalloc(&s.in_n, 2 * ntiles));
    HIPCHK(m, hipMemset(s.in_n, 0xff, sizeof(int) * 2 * ntiles));   // (no prediction's stamp)
    HIPCHK(m, dalloc(&s.pmask, W)); HIPCHK(m, dalloc(&s.ta, W)); HIPCHK(m, dalloc(&s.dflag, (size_t)d.v_loc)); HIPCHK(m, dalloc(&s.dirty, (size_t)DSP_DIRTY_CAP));
    HIPCHK(m, hipMemset(s.pmask, 0, sizeof(u64) * W)); HIPCHK(m, hipMemset(s.ta, 0, sizeof(u64) * W)); HIPCHK(m, hipMemset(s.dflag, 0, sizeof(int) * (size_t)d.v_loc));
    k.ro_rec = k.mv_rec;   // k_predict's staging area is dead once k_predict has ended: k_resample -> k_rollout reuse it
    HIPCHK(m, dalloc(&k.ro_cnt, 2 * ntiles));   // [ntiles] counts, then [ntiles] the float bits of the tiles' moving weight
    HIPCHK(m, hipMemset(k.ro_cnt, 0, sizeof(int) * 2 * ntiles));
    HIPCHK(m, dalloc(&k.ro_sub, 4 * ntiles));
    HIPCHK(m, hipMemset(k.ro_sub, 0, sizeof(int) * 4 * ntiles));
    const size_t n_ro_wg = (size_t)rollout_groups(d, (int)ntiles) * (d.tiling ? 4 : 1);   // workgroups of k_rollout
    HIPCHK(m, dalloc(&k.ro_stat, 2 * n_ro_wg));
    HIPCHK(m, hipMemset(k.ro_stat, 0, sizeof(int) * 2 * n_ro_wg));
    kernels_init_device();
    HIPCHK(m, dalloc(&k.omask, W));
    HIPCHK(m, hipMemset(k.omask, 0, sizeof(u64) * W));
    HIPCHK(m, dalloc(&k.ck_items, (size_t)d.np * ((d.capp + 63) / 64 + 1)));
    HIPCHK(m, dalloc(&k.wu_items, (size_t)d.np * ((d.capp + 31) / 32 + 1)));
    HIPCHK(m, dalloc(&k.n_items, (size_t)4));
    HIPCHK(m, dalloc(&k.nb_tab, (size_t)d.np * NB_TAB_STRIDE));
    HIPCHK(m, hipMemset(k.in_cnt, 0, sizeof(int) * ntiles));
    const bool slab = !(d.z_lo == 0 && d.z_hi == d.nz);
    if (slab) HIPCHK(m, dalloc(&k.expmask, W));
    HIPCHK(m, dalloc(&k.part_predict, (size_t)k.ntiles * 4));
    HIPCHK(m, dalloc(&s.tile_live, (size_t)k.ntiles));
    HIPCHK(m, hipMemset(s.tile_live, 1, sizeof(int) * (size_t)k.ntiles));
    HIPCHK(m, dalloc(&s.tile_moving, (size_t)k.ntiles));
    HIPCHK(m, hipMemset(s.tile_moving, 1, sizeof(int) * (size_t)k.ntiles));
    HIPCHK(m, dalloc(&s.fut_dirty, (size_t)k.ntiles));
    HIPCHK(m, hipMemset(s.fut_dirty, 0, sizeof(int) * (size_t)k.ntiles));   // (the accumulators start zeroed)
    HIPCHK(m, dalloc(&k.view_list, (size_t)k.ntiles));
    HIPCHK(m, dalloc(&k.tile_cls, (size_t)k.ntiles));
    HIPCHK(m, hipMemset(k.tile_cls, 0, sizeof(int) * (size_t)k.ntiles));
    HIPCHK(m, dalloc(&k.tile_fov, (size_t)k.ntiles));
    HIPCHK(m, hipMemset(k.tile_fov, 0xff, sizeof(int) * (size_t)k.ntiles));   // no frame's tag
    HIPCHK(m, dalloc(&k.part_resample, (size_t)k.nblk_sweep * 4));
    HIPCHK(m, dalloc(&k.work_list, (size_t)d.v_loc));
    HIPCHK(m, dalloc(&k.vb_cnt, (size_t)d.v_loc));
    HIPCHK(m, dalloc(&k.vb_idx, (size_t)d.v_loc * 128));
    HIPCHK(m, dalloc(&s.blk_cnt, (size_t)(d.v_loc + 255) / 256 + 1));
    HIPCHK(m, hipMemset(s.mask, 0, sizeof(u64) * W)); HIPCHK(m, hipMemset(s.nbmask, 0, sizeof(u64) * W));
    if (k.expmask) HIPCHK(m, hipMemset(k.expmask, 0, sizeof(u64) * W));
    HIPCHK(m, hipMemset(s.res4, 0, sizeof(float4) * (size_t)d.v_loc));
    HIPCHK(m, hipMemset(s.fut, 0, sizeof(u64) * (size_t)d.v_loc * (d.T ? d.T : 1)));
    HIPCHK(m, hipMemset(s.fs, 0, sizeof(FrameScalars)));
    HIPCHK(m, hipMemset(s.obs_cnt, 0, sizeof(int) * d.np));
    HIPCHK(m, hipMemset(s.obs_ck, 0, sizeof(long long) * d.np * DSP_OBS_CAP));
    HIPCHK(m, hipMemset(s.pyr_cnt, 0, sizeof(int) * d.np));
    HIPCHK(m, hipMemset(k.part_predict, 0, sizeof(int) * (size_t)k.ntiles * 4));
    HIPCHK(m, hipMemset(k.part_resample, 0, sizeof(int) * (size_t)k.nblk_sweep * 4));
    HIPCHK(m, hipMemset(k.vb_cnt, 0, sizeof(int) * (size_t)d.v_loc));   // invariant: empty outside a birth stage
    {   // boundary-plane normals, sensor frame (:563-578; float sin/cos like the C++ overloads)
        std::vector<float> h((size_t)(d.np_h + 1) * 3), v((size_t)(d.np_v + 1) * 3);
        const float pi_f = 3.14159265358979323846f;
        const int A = m->cfg.angle_resolution;
        const float ang = (float)A / 180.f * pi_f;  // :543
        const int he = m->cfg.half_fov_h / A, ve = m->cfg.half_fov_v / A;
        for (int i = -he; i <= he; i++) { h[(i + he) * 3] = -sinf((float)i * ang); h[(i + he) * 3 + 1] = cosf((float)i * ang); h[(i + he) * 3 + 2] = 0.f; }
        for (int i = -ve; i <= ve; i++) { v[(i + ve) * 3] = sinf((float)i * ang); v[(i + ve) * 3 + 1] = 0.f; v[(i + ve) * 3 + 2] = cosf((float)i * ang); }
        HIPCHK(m, hipMemcpy(s.planes_h0, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
        HIPCHK(m, hipMemcpy(s.planes_v0, v.data(), sizeof(float) * v.size(), hipMemcpyHostToDevice));
        HIPCHK(m, hipMemcpy(s.planes_h, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
        HIPCHK(m, hipMemcpy(s.planes_v, v.data(), sizeof(float) * v.size(), hipMemcpyHostToDevice));
    }
    {   // device velocity estimator (dspmap_velest.hip)
        VelEst& ve = m->ve;
        ve.cap = velocity_estimator_capacity();
        const size_t nc = (size_t)ve.cap / 5 + 8;
        HIPCHK(m, dalloc(&ve.w, (size_t)ve.cap)); HIPCHK(m, dalloc(&ve.root, (size_t)ve.cap));
        HIPCHK(m, dalloc(&ve.ng_view, (size_t)ve.cap));
        HIPCHK(m, dalloc(&ve.edges, (size_t)ve.cap * velocity_estimator_slices())); HIPCHK(m, dalloc(&ve.ecnt, (size_t)velocity_estimator_slices()));
        HIPCHK(m, dalloc(&ve.rank, nc)); HIPCHK(m, dalloc(&ve.by_rank, nc));
        HIPCHK(m, dalloc(&ve.dyn_list, nc)); HIPCHK(m, dalloc(&ve.cl, nc)); HIPCHK(m, dalloc(&ve.last, nc * 5));
        HIPCHK(m, dalloc(&ve.n, (size_t)4));
        HIPCHK(m, hipMemset(ve.n, 0, sizeof(int) * 4));
        HIPCHK(m, dalloc(&ve.v_rot, (size_t)ve.cap)); HIPCHK(m, dalloc(&ve.v_pyr, (size_t)ve.cap)); HIPCHK(m, dalloc(&ve.v_fpar, (size_t)1));
    }
    {   // may a / res be computed as reciprocal + two FMAs?  Compared with the IEEE quotient on the device (k_verify_div), once
        // per resolution and process
        static std::mutex mu;
        static std::map<std::pair<unsigned, unsigned>, int> known;
        m->d.rcp_res = 1.0f / m->d.res;
        m->d.div_ok = 0;
        const float amax = 2.f * std::max(m->d.half_x, std::max(m->d.half_y, m->d.half_z));
        unsigned rb, ab;
        memcpy(&rb, &m->d.res, 4); memcpy(&ab, &amax, 4);
        std::lock_guard<std::mutex> lk(mu);
        auto it = known.find({rb, ab});
        if (it == known.end()) {
            int* bad = nullptr;
            HIPCHK(m, hipMalloc((void**)&bad, sizeof(int)));
            HIPCHK(m, hipMemsetAsync(bad, 0, sizeof(int), m->stream));
            launch_verify_div(m->stream, m->d.res, m->d.rcp_res, amax, bad);
            int nbad = 1;
            HIPCHK(m, hipMemcpyAsync(&nbad, bad, sizeof(int), hipMemcpyDeviceToHost, m->stream));
            HIPCHK(m, hipStreamSynchronize(m->stream));
            (void)hipFree(bad);
            it = known.emplace(std::make_pair(rb, ab), nbad == 0 ? 1 : 0).first;
        }
        m->d.div_ok = m->div_forced_off ? 0 : it->second;
    }
    m->device_ready = true;  // from here on free_dev() releases everything
    unsigned seed = m->cfg.seed ? m->cfg.seed : (unsigned)time(nullptr);  // :586,1151
    if (!m->tables_injected) gen_gauss_tables(m, seed);
    if (!m->rtab_injected) gen_rand_table(m, seed);
    int rc = upload_tables(m);
    if (rc != DSPMAP_OK) return rc;
    rc = upload_rtab(m);
    if (rc != DSPMAP_OK) return rc;
    {   // cursors
        FrameScalars fs;
        memset(&fs, 0, sizeof(fs));
        fs.p_cur = m->pend_cursor[0]; fs.v_cur = m->pend_cursor[1]; fs.r_cur = m->pend_cursor[2];
        HIPCHK(m, hipMemcpy(s.fs, &fs, sizeof(fs), hipMemcpyHostToDevice));
    }
    rc = dspmap_ensure_point_cap(m, 8192);
    if (rc != DSPMAP_OK) return rc;
    HIPCHK(m, hipDeviceSynchronize());
    return DSPMAP_OK;
}



extern "C" int dspmap_sync(dspmap_t* m) {
    if (!m) return DSPMAP_E_ARG;
    if (!m->device_ready) return DSPMAP_OK;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}

extern "C" int dspmap_set_stream(dspmap_t* m, void* hip_stream) {
    if (!m) return DSPMAP_E_ARG;
    if (m->device_ready) HIPCHK(m, hipStreamSynchronize(m->stream));
    if (m->own_stream && m->stream) { (void)hipStreamDestroy(m->stream); m->own_stream = false; }
    m->stream = (hipStream_t)hip_stream;
    m->graph_epoch++;
    return DSPMAP_OK;
}

// ------------------------------------------------------------------ setters
extern "C" int dspmap_set_param(dspmap_t* m, int key, double v) {
    if (!m) return DSPMAP_E_ARG;
    m->graph_epoch++;
    switch (key) {
        case DSPMAP_P_POSITION_STDDEV: m->p_stddev = (float)v; break;
        case DSPMAP_P_VELOCITY_STDDEV: m->v_stddev = (float)v; break;
        case DSPMAP_P_OBSERVATION_STDDEV: m->fp.sigma_ob = (float)v; refresh_fp(m); break;
        case DSPMAP_P_NEWBORN_WEIGHT: m->fp.nb_weight = (float)v; break;
        case DSPMAP_P_NEWBORN_NUMBER:
            if (v < 1 || v > 32) return dspmap_fail(m, DSPMAP_E_ARG, "newborn number must be in [1,32]");
            m->fp.nb_num = (int)v; break;
        case DSPMAP_P_VOXEL_FILTER_RES: m->voxel_filter_res = (float)v; break;
        case DSPMAP_P_KAPPA: m->fp.kappa = (float)v; break;
        case DSPMAP_P_DETECTION: m->fp.p_det = (float)v; break;
        case DSPMAP_P_VELOCITY_ESTIMATOR:
            if (v != 0 && v != 1 && v != 2) return dspmap_fail(m, DSPMAP_E_ARG, "velocity estimator: 0 off, 1 host stage, 2 device");
            m->use_vel_est = (int)v; break;
        case DSPMAP_P_USE_GRAPH: m->use_graph = v == 1; m->direct_ring = v == 2; break;
        case DSPMAP_P_HOST_CLOUD_DIRECT: m->host_direct = v != 0; break;
        case DSPMAP_P_ESTIMATOR_QUEUE:   // (part of the captured frame's key)
            m->est_queue = v != 0;
            m->xq_failed = false;        // (setting the switch, either way, forgets an earlier give-up)
            if (m->hint_host) m->hint_host[3] = 0;
            break;
        case DSPMAP_P_FRAME_BRANCHES: m->frame_branches = v < 0 ? -1 : (v != 0 ? 1 : 0); m->graph_epoch++; break;
        case DSPMAP_P_RESAMPLE_SPLIT: m->resample_split = v != 0 ? 1 : 0; m->graph_epoch++; break;
        case DSPMAP_P_TILE_BITMAPS: m->tile_bitmaps = v != 0; m->graph_epoch++; break;
        case DSPMAP_P_SIDE_PLACEMENT: {
            const int iv = v < 0 ? 16 + 3 : (int)v;
            m->side_fork = (iv >> 4) > 2 ? 0 : (iv >> 4);
            m->side_wg = (iv & 15) ? (iv & 15) : 3;
            m->graph_epoch++;
            break;
        }
        case DSPMAP_P_TILING:
            if (m->device_ready) return dspmap_fail(m, DSPMAP_E_STATE, "DSPMAP_P_TILING must be set before the device state is allocated");
            m->tiling_req = v < 0 ? -1 : (v != 0 ? 1 : 0);
            derive_dims(m);
            break;
        case DSPMAP_P_SPARSE_SWEEP: m->sparse_force = v < 0 ? -1 : (v != 0 ? 1 : 0); break;
        case DSPMAP_P_ROLLOUT_INLINE: m->ro_force = v < 0 ? -1 : (v != 0 ? 1 : 0); break;
        case DSPMAP_P_FAST_DIVISION: if (v == 0) { m->d.div_ok = 0; m->div_forced_off = true; m->graph_epoch++; } break;
        case DSPMAP_P_PLACE_SPLIT_TILES:
            m->place_split_tiles = v < 1 ? 1 : (v > 2e9 ? 2000000000 : (int)v); m->graph_epoch++;
            if (!m->device_ready) derive_dims(m);   // (the storage order the handle picks by itself follows this limit)
            break;
        case DSPMAP_P_RESAMPLE_WG_TILES: m->resample_wg_tiles = v < 0 ? 0 : (v > 2e9 ? 2000000000 : (int)v); m->graph_epoch++; break;
        case DSPMAP_P_SWEEP_ALTERNATE: m->sweep_alt = v < 0 ? -1 : (v >= 2 ? 2 : (v != 0 ? 1 : 0)); m->graph_epoch++; break;
        case DSPMAP_P_STATIC_TILE_SKIP: m->d.tile_skip = v != 0 ? 1 : 0; m->graph_epoch++; break;
        case DSPMAP_P_OCCLUSION_MARGIN: m->fp.occl_margin = (float)v; break;
        case DSPMAP_P_PAIR_CULL_SIGMAS: if (!(v > 0)) return dspmap_fail(m, DSPMAP_E_ARG, "pair cull radius must be positive"); m->cull_sigmas = (float)v; refresh_fp(m); break;
        case DSPMAP_P_REGENERATE_TABLES:
            // setPredictionVariance regenerates both tables with a fresh seed (:355-360)
            if (v != 0 && !m->tables_injected) {
                gen_gauss_tables(m, m->cfg.seed ? m->cfg.seed + 1 : (unsigned)time(nullptr));
                if (m->device_ready) { HIPCHK(m, hipStreamSynchronize(m->stream)); return upload_tables(m); }
            }
            break;
        default: return dspmap_fail(m, DSPMAP_E_ARG, "unknown parameter key %d", key);
    }
    return DSPMAP_OK;
}
extern "C" double dspmap_get_param(const dspmap_t* m, int key) {
    if (!m) return 0;
    switch (key) {
        case DSPMAP_P_POSITION_STDDEV: return m->p_stddev;
        case DSPMAP_P_VELOCITY_STDDEV: return m->v_stddev;
        case DSPMAP_P_OBSERVATION_STDDEV: return m->fp.sigma_ob;
        case DSPMAP_P_NEWBORN_WEIGHT: return m->fp.nb_weight;
        case DSPMAP_P_NEWBORN_NUMBER: return m->fp.nb_num;
        case DSPMAP_P_VOXEL_FILTER_RES: return m->voxel_filter_res;
        case DSPMAP_P_KAPPA: return m->fp.kappa;
        case DSPMAP_P_DETECTION: return m->fp.p_det;
        case DSPMAP_P_VELOCITY_ESTIMATOR: return m->use_vel_est;
        case DSPMAP_P_OCCLUSION_MARGIN: return m->fp.occl_margin;
        case DSPMAP_P_PAIR_CULL_SIGMAS: return m->cull_sigmas;
        case DSPMAP_P_UPDATE_TIME: return m->update_time;
        case DSPMAP_P_UPDATE_COUNTER: return m->update_counter;
        case DSPMAP_P_PLACE_SPLIT_TILES: return m->place_split_tiles;
        case DSPMAP_P_RESAMPLE_WG_TILES: return m->resample_wg_tiles;
        case DSPMAP_P_SWEEP_ALTERNATE: return m->sweep_alt;
        case DSPMAP_P_STATIC_TILE_SKIP: return m->d.tile_skip;
        case DSPMAP_P_SPARSE_SWEEP: return m->sparse_mode ? 1 : 0;
        case DSPMAP_P_ROLLOUT_INLINE: return m->ro_kernel ? 0 : 1;
        case DSPMAP_P_FAST_DIVISION: return m->d.div_ok;
        case DSPMAP_P_HOST_CLOUD_DIRECT: return m->host_direct ? 1 : 0;
        case DSPMAP_P_ESTIMATOR_QUEUE: return m->est_queue ? 1 : 0;
        case DSPMAP_P_FRAME_BRANCHES: return m->frame_branches;
        case DSPMAP_P_SIDE_PLACEMENT: return m->side_fork * 16 + m->side_wg;
        case DSPMAP_P_RESAMPLE_SPLIT: return m->resample_split;
        case DSPMAP_P_TILE_BITMAPS: return m->tile_bitmaps ? 1 : 0;
        case DSPMAP_P_TILING: return m->d.tiling;
        case DSPMAP_P_USE_GRAPH: return m->use_graph ? 1 : (m->direct_ring ? 2 : 0);
        default: return 0;
    }
}

extern "C" int dspmap_set_gaussian_tables(dspmap_t* m, const float* p, const float* v, int n) {
    if (!m || !p || !v || n <= 0) return DSPMAP_E_ARG;
    m->h_ptab.assign(p, p + n);
    m->h_vtab.assign(v, v + n);
    m->tables_injected = true;
    if (m->device_ready) {
        HIPCHK(m, hipStreamSynchronize(m->stream));
        int rc = upload_tables(m);
        if (rc != DSPMAP_OK) return rc;
        return dspmap_set_cursors(m, 0, 0, -1);
    }
    m->pend_cursor[0] = m->pend_cursor[1] = 0;
    return DSPMAP_OK;
}
extern "C" int dspmap_set_rand_table(dspmap_t* m, const int* r, int n) {
    if (!m || !r || n <= 0) return DSPMAP_E_ARG;
    m->h_rtab.assign(r, r + n);
    m->rtab_injected = true;
    if (m->device_ready) {
        HIPCHK(m, hipStreamSynchronize(m->stream));
        int rc = upload_rtab(m);
        if (rc != DSPMAP_OK) return rc;
        return dspmap_set_cursors(m, -1, -1, 0);
    }
    m->pend_cursor[2] = 0;
    return DSPMAP_OK;
}
extern "C" int dspmap_set_cursors(dspmap_t* m, int pc, int vc, int rc) {  // negative = leave unchanged
    if (!m) return DSPMAP_E_ARG;
    if (!m->device_ready) {
        if (pc >= 0) m->pend_cursor[0] = pc;
        if (vc >= 0) m->pend_cursor[1] = vc;
        if (rc >= 0) m->pend_cursor[2] = rc;
        return DSPMAP_OK;
    }
    HIPCHK(m, hipStreamSynchronize(m->stream));
    FrameScalars fs;
    HIPCHK(m, hipMemcpy(&fs, m->s.fs, sizeof(fs), hipMemcpyDeviceToHost));
    if (pc >= 0) fs.p_cur = pc;
    if (vc >= 0) fs.v_cur = vc;
    if (rc >= 0) fs.r_cur = rc;
    HIPCHK(m, hipMemcpy(m->s.fs, &fs, sizeof(fs), hipMemcpyHostToDevice));
    return DSPMAP_OK;
}
extern "C" int dspmap_get_cursors(dspmap_t* m, int* pc, int* vc, int* rc) {
    READY(m);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    FrameScalars fs;
    HIPCHK(m, hipMemcpy(&fs, m->s.fs, sizeof(fs), hipMemcpyDeviceToHost));
    if (pc) *pc = fs.p_cur;
    if (vc) *vc = fs.v_cur;
    if (rc) *rc = fs.r_cur;
    return DSPMAP_OK;
}

// --------------------------------------------------------------- the frame
void dspmap_freeze_birth_statics(dspmap* m) {
    if (m->nb_frozen) return;
    m->graph_epoch++;  // function statics initialised at first call (:808-811)
    m->fp.min_static_nb = (int)((float)m->fp.nb_num * 0.15f);
    m->fp.model_nb = (int)((float)m->fp.nb_num * 0.8f);
    m->nb_frozen = true;
}

void dspmap_flush_future_clear(dspmap* m) {
    if (!m->fut_clear_pending) return;
    LaunchCtx c = dspmap_ctx_of(m);
    launch_clear_future(c);
    m->fut_clear_pending = false;
}
int dspmap_push_frame_params(dspmap* m) {
    // a pending clear of the future accumulators rides on the frame: its k_predict does it (no extra launch)
    m->hp.clear_fut = m->fut_clear_pending ? 1 : 0;
    m->fut_clear_pending = false;
    m->hp.from_ring = 0;
    // pageable source: the runtime stages the bytes before returning, so m->hp can be reused at once
    HIPCHK(m, hipMemcpyAsync(m->s.fpar, &m->hp, sizeof(FrameParams), hipMemcpyHostToDevice, m->stream));
    return DSPMAP_OK;
}
// A new cloud is about to be binned: bump the frame epoch (FrameScalars::view_epoch refers to it) and return the
// bound for the grids of the birth launches of a synthesised cloud.
int dspmap_begin_cloud(dspmap* m, int n_points, bool static_birth) {
    m->hp.epoch++;
    if (!static_birth) return n_points;
    if (n_points > m->static_hi) m->static_hi = n_points;
    return m->static_hi;
}
static void fill_pose(dspmap* m, const float dp[3], float dt) {
    for (int i = 0; i < 4; i++) m->hp.quat[i] = m->quat[i];
    for (int i = 0; i < 3; i++) { m->hp.cur_pos[i] = m->cur_pos[i]; m->hp.od[i] = -dp[i]; }  // particles move opposite to the sensor (:300)
    m->hp.dt = dt;
    m->hp.res_filter = m->voxel_filter_res;
    m->hp.birth_reach = m->ptab_max;
}

// A cross-queue wait of an earlier frame gave up (DSPMAP_P_ESTIMATOR_QUEUE; bounded at 200 ms: a contended GPU, a debugger): that frame ran
// WITHOUT its birth stage (k_birth_insert returns when it finds the give-up word: no partial birth cloud is consumed).  Checked by every frame
// entry point BEFORE anything of the handle is touched: the call that finds the note fails once, with the state as the failed frame left
// it, and the handle goes on with the estimator as a forked branch of the graph (xq_failed) -- until the state is cleared or restored or
// the switch is set again.
int dspmap_check_estimator_queue(dspmap* m) {
    if (!m->hint_host || m->hint_host[3] == 0) return DSPMAP_OK;
    const int at = m->hint_host[3] - 1;
    (void)hipStreamSynchronize(m->stream);
    if (m->stream3) (void)hipStreamSynchronize(m->stream3);
    m->hint_host[3] = 0;
    m->xq_failed = true;
    m->graph_epoch++;
    return dspmap_fail(m, DSPMAP_E_DEVICE, "estimator queue: a cross-queue wait gave up at ring position %d; that frame ran without its birth stage, the handle continues with the estimator inside the captured frame (DSPMAP_P_ESTIMATOR_QUEUE)", at);
}

// C0 gate + deltas, update() :187-218.  returns 1 (accepted) / 0 (rejected)
int dspmap_gate_and_delta(dspmap* m, const float pos[3], double stamp, const float q[4], float dp[3], float* dt) {
    if (!m->have_last) {
        m->last_p[0] = pos[0]; m->last_p[1] = pos[1]; m->last_p[2] = pos[2];
        m->last_stamp = stamp;
        m->have_last = true;
    }
    if (fabsf(q[0]) > 1.001f || fabsf(q[1]) > 1.001f || fabsf(q[2]) > 1.001f || fabsf(q[3]) > 1.001f) {
        printf("Invalid quaternion.\n");  // :194
        return 0;
    }
    dp[0] = pos[0] - m->last_p[0]; dp[1] = pos[1] - m->last_p[1]; dp[2] = pos[2] - m->last_p[2];
    *dt = (float)(stamp - m->last_stamp);
    if (fabsf(dp[0]) > 10.f || fabsf(dp[1]) > 10.f || fabsf(dp[2]) > 10.f || *dt < 0.f || *dt > 10.f) {
        printf("!!! delt_t = %f\n", *dt);  // :204-206
        return 0;
    }
    for (int i = 0; i < 3; i++) m->cur_pos[i] = m->last_p[i] = pos[i];
    m->last_stamp = stamp;
    m->dt_last = *dt;
    m->update_time += *dt; m->update_counter += 1;   // mapPrediction :634-635
    for (int i = 0; i < 4; i++) m->quat[i] = q[i];
    return 1;
}

// does this frame place the arrivals of the tiles with a view first and the others beside the pair kernels / register the movers in k_predict?
static bool frame_splits_placement(const dspmap* m, const LaunchCtx& c, bool fork) {
    return !fork && !m->prof && (!c.sparse || m->place_split_tiles <= 1) && c.k.ntiles >= m->place_split_tiles;   // (1 = always, as documented)
}

// does this frame run as two branches (DSPMAP_P_FRAME_BRANCHES; KernelScratch::tile_cls)?  The same maps that split their placement -- dense and
// large: the branches cost a classification launch and two passes of workgroups over the tiles --, unless forced; never a frame whose
// prediction changes velocities (vz0: constructor-seeded particles draw their noise there, after the classes were sized), a profiled
// frame (one stream), a map that runs the four-waves-per-tile resampler (no class filter there: small maps), or index-order storage
// (MapDims::tiling == 0: a run of 64 voxel indices that points away from the sensor is cut by the field of view almost wherever it
// lies -- 58 % of the 132x132x60 map's runs have a view: there is nothing to leave to a second branch)
static bool frame_runs_two_branches(const dspmap* m, const LaunchCtx& c, bool fork) {
    if (fork || m->prof || m->frame_branches == 0 || c.s.vz0 || !c.k.tile_cls || !c.d.tiling) return false;   // (cube storage: k_tile_class)
    if (resample_variant(c) & 1) return false;
    return m->frame_branches == 1 || frame_splits_placement(m, c, fork);
}

// does this frame resample the tiles no newborn can reach BESIDE the weight update and the births (DSPMAP_P_RESAMPLE_SPLIT)?  A frame that
// splits its placement (the side stream exists and ends with the placement of exactly such tiles), cube storage (k_tile_class), the
// one-wave-per-tile resampler (class filter), no velocity noise pending, and a birth cloud made on the device from THIS frame's view
// (every point in view a static source, or the device estimator's: a caller-supplied cloud may hold points anywhere)
static bool frame_splits_resampling(const dspmap* m, const LaunchCtx& c, bool split, bool device_cloud) {
    if (!split || m->resample_split == 0 || !device_cloud || c.s.vz0 || !c.k.tile_cls || !c.d.tiling) return false;
    return !(resample_variant(c) & 1);
}

// enqueue one whole device-resident frame (setup .. resample); every per-frame value is read from s.fpar.
// When `fork` is set (graph capture) the observation binning runs on a second stream concurrently with
// prediction + re-binning: the two only share the rotated planes written by k_reset and meet again at
// the Ck kernel.
static void enqueue_frame(dspmap* m, LaunchCtx& c, int pts_grid, int birth_grid, bool fork, bool all_static, bool est = false) {
    // (per-stage timing keeps the frame on one stream; so does a sparse map -- most tiles empty: two passes over all the tiles cost
    // more than the overlap gives: 264x264x80 filled by the depth stream 0.445 -> 0.434 ms, 132x132x60 0.232 -> 0.228; saturated
    // maps keep the split: 0.659 against 0.667 ms and 4.61 against 4.80 ms, interleaved runs on one box)
    if (frame_runs_two_branches(m, c, fork)) {
        // TWO BRANCHES (round 6).  The reference's frame is four sweeps over every voxel (:300-322); here most of a large map is only moved
        // and resampled -- bandwidth-bound sweeps -- while pyramid lists, Ck, weights and births (a chain of latency- and VALU-bound
        // kernels) concern the part the sensor sees.  k_tile_class cuts the tiles into that part, grown by the reach of a newborn (Q) and
        // again by the frame's largest displacement (P); then
        //   main:  predict(P) -> place(Q) -> lists -> Ck -> weights -> births -> resample(Q) -+-> rollout
        //   side:  predict(not P) -> [predict(P) done] -> place(not Q) -> resample(not Q) ----+
        // run beside each other.  Same kernels, same per-tile work, same result slot for slot (tests/test_gpu_round6.py).
        c.place_split = false;
        c.branches = true;
        m->rsplit_enq = false;
        if (!m->stream4) {
            // the bulk branch's stream, created at the first frame that needs it, with the LOWEST priority the device offers: its sweeps would
            // otherwise keep every CU's wave slots and LDS filled and the in-view chain's kernels -- the frame's critical path -- would wait
            // for them (list preparation 13 -> 73 us, weights 40 -> 71 us beside them, profiles/r06_b_C_sat_timeline.md)
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            static const bool flat = getenv("DSPMAP_BULK_PRIORITY_FLAT") != nullptr;
            if (flat || hipStreamCreateWithPriority(&m->stream4, hipStreamNonBlocking, lo) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamCreateWithFlags(&m->stream4, hipStreamNonBlocking); }
        }
        launch_setup_and_bin(c, pts_grid, false, m->frame_ring ? m->ring_dev : nullptr, DSPMAP_RING - 1);
        launch_tile_class(c);
        (void)hipEventRecord(m->ev_br[0], m->stream);
        (void)hipStreamWaitEvent(m->stream4, m->ev_br[0], 0);
        LaunchCtx cb = c;
        cb.stream = m->stream4;
        const bool with_est = est && birth_grid > 0;
        const bool early_birth = !with_est && birth_grid > 0;
        launch_predict_only(c, true, early_birth, TILE_P);
        (void)hipEventRecord(m->ev_br[1], m->stream);                 // predict(P) has ended: every arrival of a Q tile is in its inbox
        launch_predict_only(cb, false, false, -TILE_P);
        (void)hipEventRecord(m->ev_br[2], m->stream4);                // predict(not P) has ended: every tile's pending clear is done
        (void)hipStreamWaitEvent(m->stream4, m->ev_br[1], 0);
        launch_claim(cb, 0, 0, 0, 0, -1, -TILE_Q);
        (void)hipEventRecord(m->ev_br[4], m->stream4);                // place(not Q) has ended
        launch_resample(cb, -TILE_Q, false, 2);                       // (the early launch: leaves a frame with an empty view alone, see k_resample)
        (void)hipEventRecord(m->ev_br[3], m->stream4);
        if (with_est) {   // the estimator's branch (the reference's helper thread, :297,311): a third one, from the binning to the birth stage
            (void)hipStreamWaitEvent(m->stream2, m->ev_br[0], 0);
            LaunchCtx c2 = c;
            c2.stream = m->stream2;
            launch_velocity_estimator(c2, true);
            launch_birth_early(c2, birth_grid, false);
            (void)hipEventRecord(m->ev_join, m->stream2);
        }
        launch_claim(c, early_birth ? birth_grid : 0, 0, 0, 0, -1, TILE_Q);
        launch_pyr_prepare(c);
        launch_ck_partial(c, true);
        launch_weight_update(c);
        if (with_est) {
            (void)hipStreamWaitEvent(m->stream, m->ev_join, 0);
            launch_birth_late(c, birth_grid, false, false);
        } else {
            if (birth_grid <= 0) launch_ck_finalize(c);
            if (early_birth) launch_birth_late(c, birth_grid, all_static);
            else launch_birth(c, birth_grid, true, all_static);
        }
        (void)hipStreamWaitEvent(m->stream, m->ev_br[2], 0);          // (the rollout of the Q tiles adds to accumulators anywhere: after every clear)
        // a frame with an empty view re-uses the birth cloud of the last non-empty one (:1379-1381): its newborns land where THAT frame's
        // field of view was, Q says nothing about them -- this launch then takes every tile, so every tile's placement must have ended
        (void)hipStreamWaitEvent(m->stream, m->ev_br[4], 0);
        launch_resample(c, TILE_Q, false, 4);
        (void)hipStreamWaitEvent(m->stream, m->ev_br[3], 0);
        launch_rollout(c);
        m->last_resample_variant = resample_variant(c);
        m->branch_pending = true;
        return;
    }
    // (DSPMAP_P_TILE_BITMAPS) a sparse unsharded map: the sweeps of this frame find their empty tiles in the bitmaps k_obs_points rebuilds
    c.tile_bits = !fork && c.sparse && m->tile_bitmaps && c.k.tile_bits && m->d.v_true == m->d.v_glob && !m->mgpu_bound;
    if (c.tile_bits) {
        const size_t nw = ((size_t)c.k.ntiles + 63) / 64 * 2;
        c.s.vis_bits = c.k.tile_bits; c.s.pred_bits = c.k.tile_bits + nw; c.s.arr_bits = c.k.tile_bits + 2 * nw;
    }
    const bool split0 = frame_splits_placement(m, c, fork);
    const bool split = split0;
    c.place_split = split;
    // (DSPMAP_P_RESAMPLE_SPLIT) Q = the tiles a newborn of this frame can reach (k_tile_class: the field of view grown by the position
    // table's largest value; every tile with a view is one).  Weights, births and the list re-slotting only touch Q tiles, and the side
    // stream's placement serves exactly the tiles without a view: once it is done the tiles outside Q are final for this frame, and the
    // side stream resamples them while the main chain is still in its weight update and birth stage; the main chain resamples Q behind
    // the births, the rollout follows both.  Same per-tile work, same result slot for slot.
    const bool rsplit = frame_splits_resampling(m, c, split, all_static || est);
    m->rsplit_enq = rsplit;
    auto side_place = [&]() {   // (queued behind the main chain's next kernel: the branch whose node comes first after the fork stays on the parent's hardware queue)
        (void)hipStreamWaitEvent(m->stream2, m->ev_fork2, 0);
        LaunchCtx c2 = c;
        c2.stream = m->stream2;
        if (rsplit) launch_tile_class(c2);
        launch_claim(c2, 0, 0, 0, 0, 0);
        (void)hipEventRecord(m->ev_join, m->stream2);
        if (rsplit) {
            launch_resample(c2, -TILE_Q, false, 2);
            (void)hipEventRecord(m->ev_br[3], m->stream2);
        }
    };
    auto final_resample = [&]() {
        if (!rsplit) { dspmap_resample(m, c); return; }
        m->last_resample_variant = resample_variant(c);
        launch_resample(c, TILE_Q, false, 4);
        (void)hipStreamWaitEvent(m->stream, m->ev_br[3], 0);
        launch_rollout(c);
    };
    // where the side placement leaves the main chain (DSPMAP_P_SIDE_PLACEMENT): 0 behind the list preparation, 1 behind the placement of
    // the tiles with a view, 2 behind the prediction
    const int side_fork = m->side_fork;
    dspmap_prof_mark(m, 0);
    if (!fork) {
        launch_setup_and_bin(c, pts_grid, false, m->frame_ring ? m->ring_dev : nullptr, DSPMAP_RING - 1);   // the gather rides on k_predict's launch
    } else {
        launch_frame_setup(c, true);
        (void)hipEventRecord(m->ev_fork, m->stream);
        (void)hipStreamWaitEvent(m->stream2, m->ev_fork, 0);
        LaunchCtx c2 = c;
        c2.stream = m->stream2;
        launch_obs_bin(c2, pts_grid);
        (void)hipEventRecord(m->ev_join, m->stream2);
    }
    dspmap_prof_mark(m, 1);
    if (est && birth_grid > 0) {
        // The velocity estimator runs BESIDE prediction and weight update, like the reference's helper thread (:297,311):
        // a second branch of the frame (side stream; a forked branch of the captured graph) takes the binned view through
        // k_ve_components -> k_ve_clusters (+ the birth rank) -> the newborn children, and joins before the birth stage.
        // (The main chain's next kernel is queued BEFORE the side branch's: the branch whose node comes first after the
        // fork stays on the parent's hardware queue, the other one pays the cross-queue hand-over.)
        // DSPMAP_P_ESTIMATOR_QUEUE (c.s.xq set; never with a split placement / early registration): no branch at all in this graph -- the caller
        // has queued the estimator's kernels on the other stream itself, k_predict and the split kernel below meet them through DevState::xq
        const bool xq = c.s.xq != nullptr;
        if (!xq) (void)hipEventRecord(m->ev_fork, m->stream);
        launch_predict_only(c, true, false);
        LaunchCtx c2 = c;
        c2.stream = m->stream2;
        if (!xq) {
            (void)hipStreamWaitEvent(m->stream2, m->ev_fork, 0);
            launch_velocity_estimator(c2, true);
            // the children (the rank ran inside k_ve_clusters): on the side branch when it goes on with the placement of the tiles without
            // a view (large maps); otherwise the branch -- the longer one at the metric's size -- ends here and the waves of the split
            // generate them (launch_birth_late)
            if (split) launch_birth_early(c2, birth_grid, false);
            (void)hipEventRecord(m->ev_join, m->stream2);
        }
        dspmap_prof_mark(m, 2);
        launch_claim(c, 0, 0, 0, 0, split ? 1 : -1);
        // (the side stream carries the estimator first: "behind the prediction" -- fork 2 -- is "behind the placement of the tiles with a view" here)
        if (split && side_fork >= 1) (void)hipEventRecord(m->ev_fork2, m->stream);
        if (split) launch_pyr_prepare(c);
        if (split && side_fork >= 1) side_place();   // (behind the estimator's kernels on that stream)
        dspmap_prof_mark(m, 3);
        if (split && side_fork < 1) (void)hipEventRecord(m->ev_fork2, m->stream);
        launch_ck_partial(c, split);
        if (split && side_fork < 1) side_place();   // the side stream places the arrivals of the tiles outside the field of view (behind the estimator's kernels)
        dspmap_prof_mark(m, 4);
        launch_weight_update(c);
        dspmap_prof_mark(m, 5);
        if (!xq) (void)hipStreamWaitEvent(m->stream, m->ev_join, 0);
        dspmap_prof_mark(m, 6);
        launch_birth_late(c, birth_grid, false, !split);
        dspmap_prof_mark(m, 7);
        final_resample();
        dspmap_prof_mark(m, 8);
        if (m->prof) m->prof_pending = true;
        return;
    }
    // births: the rank and the children need nothing but the frame's birth cloud -- they ride on the launches of
    // k_predict and k_place and leave the frame's critical path; split, cursors and insert follow the weight update
    const bool early_birth = !fork && birth_grid > 0;
    launch_predict_only(c, !fork, early_birth);
    dspmap_prof_mark(m, 2);
    if (split && side_fork == 2) (void)hipEventRecord(m->ev_fork2, m->stream);
    launch_claim(c, early_birth ? birth_grid : 0, 0, 0, 0, split ? 1 : -1);
    if (split && side_fork == 2) side_place();
    if (split) {
        // Only the arrivals of tiles that can see the field of view are registered in pyramids, so only their placement
        // has to precede the weight update: the others get their slots on the side stream WHILE the pair kernels run
        // (VALU-bound; the list preparation before them is itself a scatter and would only share the memory system).
        if (side_fork == 1) (void)hipEventRecord(m->ev_fork2, m->stream);
        launch_pyr_prepare(c);
        if (side_fork == 1) side_place();
        if (side_fork != 1 && side_fork != 2) (void)hipEventRecord(m->ev_fork2, m->stream);
    }
    if (fork) (void)hipStreamWaitEvent(m->stream, m->ev_join, 0);
    dspmap_prof_mark(m, 3);
    launch_ck_partial(c, split);
    if (split && side_fork != 1 && side_fork != 2) side_place();
    dspmap_prof_mark(m, 4);
    launch_weight_update(c);
    if (split) (void)hipStreamWaitEvent(m->stream, m->ev_join, 0);
    dspmap_prof_mark(m, 5);
    if (birth_grid <= 0) launch_ck_finalize(c);   // otherwise k_birth_rank reduces the 1/Ck sums (one launch less)
    dspmap_prof_mark(m, 6);
    if (early_birth) launch_birth_late(c, birth_grid, all_static);
    else launch_birth(c, birth_grid, true, all_static);
    dspmap_prof_mark(m, 7);
    final_resample();
    dspmap_prof_mark(m, 8);
    if (m->prof) m->prof_pending = true;
}

int dspmap_upload_birth(dspmap* m, const dspmap_vpoint* pts, int n);
static int upload_birth(dspmap* m, const dspmap_vpoint* pts, int n) { return dspmap_upload_birth(m, pts, n); }

// The reference keeps ONE clusters_feature_vector_dynamic_last (a function static, :1401,1542).  Here the device estimator
// (dspmap_velest.hip) and the host stage (velocity_estimator.cpp: clouds beyond the device estimator's capacity, or
// DSPMAP_P_VELOCITY_ESTIMATOR = 1) each hold a copy: whenever a frame is about to run on the one that does not hold the
// newer copy, the state is handed over first.
int dspmap_ve_state_to_host(dspmap* m);
static int ve_state_to_host(dspmap* m) { return dspmap_ve_state_to_host(m); }
int dspmap_ve_state_to_host(dspmap* m) {
    if (m->ve_last_at != 2) return DSPMAP_OK;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    int n3[4] = {0, 0, 0, 0};
    HIPCHK(m, hipMemcpy(n3, m->ve.n, sizeof(n3), hipMemcpyDeviceToHost));
    const int n = std::max(0, std::min(n3[2], m->ve.cap / 5 + 8));
    std::vector<float> buf((size_t)n * 5 + 1);
    if (n > 0) HIPCHK(m, hipMemcpy(buf.data(), m->ve.last, sizeof(float) * 5 * (size_t)n, hipMemcpyDeviceToHost));
    m->vel.import_last(buf.data(), n);
    m->ve_last_at = 1;
    return DSPMAP_OK;
}
int dspmap_ve_state_to_device(dspmap* m);
static int ve_state_to_device(dspmap* m) { return dspmap_ve_state_to_device(m); }
int dspmap_ve_state_to_device(dspmap* m) {
    if (m->ve_last_at != 1) return DSPMAP_OK;
    const int cap = m->ve.cap / 5 + 8;
    std::vector<float> buf((size_t)cap * 5);
    const int n = m->vel.export_last(buf.data(), cap);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (n > 0) HIPCHK(m, hipMemcpy(m->ve.last, buf.data(), sizeof(float) * 5 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(m, hipMemcpy(m->ve.n + 2, &n, sizeof(int), hipMemcpyHostToDevice));
    m->ve_last_at = 2;
    return DSPMAP_OK;
}

// One frame with the host stages in the loop (velocity estimator and/or a caller-supplied birth cloud): prediction and
// the weight update are queued first, the host estimator runs while they execute (the reference forks
// velocityEstimationThread before prediction and joins it before the birth stage, :297,311), then the tagged cloud is
// uploaded and birth + resampling follow.  `pts_dev` is the frame's cloud on the device, m->pts_pin its host copy (valid
// once `pts_ready`, if given, has completed).
static int frame_with_host_stages(dspmap* m, int np, const float* pts_dev, const float q[4], const float dp[3], float dt,
                                  hipEvent_t pts_ready) {
    dspmap_freeze_birth_statics(m);
    m->frame_parity ^= 1u;
    LaunchCtx c = dspmap_ctx_of(m);
    if (m->vz_frames <= 0) c.s.vz0 = nullptr;
    // dsp_static.h has no velocity estimation: every in-FOV point is a zero-velocity birth source
    const bool have_cloud = !m->cfg.static_model && (m->use_vel_est != 0 || m->h_birth_valid);
    fill_pose(m, dp, dt);
    m->hp.n_pts = np; m->hp.n_birth = np; m->hp.static_birth = have_cloud ? 0 : 1;
    m->hp.pts = pts_dev; m->hp.birth = m->s.birth;
    const int nb_static_grid = dspmap_begin_cloud(m, np, !have_cloud);
    int rc = dspmap_push_frame_params(m);
    if (rc != DSPMAP_OK) return rc;
    HIPCHK(m, hipEventRecord(m->ev0, m->stream));
    launch_setup_and_bin(c, np, false);
    launch_predict(c, true);
    launch_ck_partial(c);
    launch_weight_update(c);
    int nb = nb_static_grid;
    if (m->use_vel_est != 0 && !m->cfg.static_model) {
        if (pts_ready) HIPCHK(m, hipEventSynchronize(pts_ready));
        std::vector<float> view;
        view.reserve((size_t)np * 3);
        m->vel.rotate_and_filter(m->pts_pin, np, q, view);
        { const int rcs = ve_state_to_host(m); if (rcs != DSPMAP_OK) return rcs; }
        m->vel.run(view, m->cur_pos, dt, m->voxel_filter_res, m->h_birth);
        m->ve_last_at = 1;
        m->h_birth_valid = true;
    }
    if (have_cloud) {
        nb = (int)m->h_birth.size();
        rc = upload_birth(m, m->h_birth.data(), nb);
        if (rc != DSPMAP_OK) return rc;
        c.s = m->s;  // pointers may have been re-allocated
        if (m->vz_frames <= 0) c.s.vz0 = nullptr;
        m->hp.n_birth = nb; m->hp.birth = m->s.birth;
        rc = dspmap_push_frame_params(m);
        if (rc != DSPMAP_OK) return rc;
    }
    if (nb > 0) launch_birth(c, nb, true, !have_cloud);  // :314-316
    else launch_ck_finalize(c);
    dspmap_resample(m, c);
    if (m->vz_frames > 0) --m->vz_frames;
    if (m->nb_dirty) { m->nb_dirty = false; m->graph_epoch++; }
    HIPCHK(m, hipEventRecord(m->ev1, m->stream));
    m->ev_valid = true;
    m->last_n_points = np;
    m->last_n_birth = nb;
    m->last_birth_static = !have_cloud;
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}


// The frame's parameter block through the pinned ring instead of a pageable H2D copy (which makes the host wait for the stream to
// drain: every kernel of the frame is then launched into an empty queue).  For frames that are NOT replayed as a graph (the sharded
// frame of dspmap_mgpu_begin): fills the ring slot and returns the ring to pass to launch_setup_and_bin; dspmap_ring_pushed() after
// the frame's first launches were queued.  Falls back to the copy when the ring does not exist.
const FrameParams* dspmap_ring_push(dspmap* m) {
    if (!m->ring_host) return nullptr;
    const unsigned q = (m->ring_head / (DSPMAP_RING / 4)) % 4;
    if (m->ring_head % (DSPMAP_RING / 4) == 0 && m->ring_ev_set[q]) (void)hipEventSynchronize(m->ring_ev[q]);
    m->hp.clear_fut = m->fut_clear_pending ? 1 : 0;
    m->fut_clear_pending = false;
    m->hp.from_ring = 1;
    m->hp.ring_pos = m->ring_head;
    m->ring_host[m->ring_head % DSPMAP_RING] = m->hp;
    return m->ring_dev;
}
void dspmap_ring_pushed(dspmap* m) {
    if (m->ring_head % (DSPMAP_RING / 4) == DSPMAP_RING / 4 - 1) {
        const unsigned q = (m->ring_head / (DSPMAP_RING / 4)) % 4;
        if (hipEventRecord(m->ring_ev[q], m->stream) == hipSuccess) m->ring_ev_set[q] = true;
    }
    ++m->ring_head;
}

// The estimator's own stream (DSPMAP_P_ESTIMATOR_QUEUE), created at the first frame that uses it.  The runtime maps more streams than it has
// hardware queues (GPU_MAX_HW_QUEUES = 4) onto shared ones; if this stream and the handle's main stream land on ONE hardware queue everything
// stays correct (every cross-stream wait is for earlier work) but the estimator runs after the frame instead of beside it (66x66x40: 0.148 ->
// 0.214 ms, seen in bench.py with five streams alive).  So the pairing is TESTED: the main stream is kept busy for 2 ms, a one-microsecond kernel
// goes to the candidate -- if it ends while the main stream is still busy the two do not share a queue; otherwise the candidate is kept aside
// (so that the next one lands elsewhere) and another is tried.  (A high stream priority -- its own pool of hardware queues -- was the first fix:
// with such a stream alive, graph replays WITH a forked branch ran 0.15 ms longer, 132x132x60 saturated + device estimator 0.58 -> 0.73 ms, and
// forking into a prioritised stream during capture crashed the runtime.)
// When NO candidate is apart from the main stream's hardware queue (seen in processes that had created hundreds of streams), the handle does
// not take a shared one (0.214 instead of 0.151 ms per frame at the metric's size): m->xq_shared is set and its frames keep the estimator as
// a forked branch of the captured graph (device_frame).  Test hooks (DSPMAP_XQ_FORCE, read at dspmap_create): "shared" = every candidate counts
// as sharing the queue (the fallback is what runs), "apart" = a handle that finds no candidate apart fails loudly instead of falling back.
static int ensure_estimator_stream(dspmap* m, const LaunchCtx& c) {
    if (m->xq_shared && m->stream3_for == m->stream) return DSPMAP_OK;   // (tested before, for this main stream: nothing apart)
    if (m->stream3 && m->stream3_for == m->stream) return DSPMAP_OK;
    m->xq_shared = false;
    if (m->stream3) { HIPCHK(m, hipStreamSynchronize(m->stream3)); (void)hipStreamDestroy(m->stream3); m->stream3 = nullptr; }
    HIPCHK(m, hipStreamSynchronize(m->stream));
    hipStream_t good = nullptr;
    hipStream_t aside[16]; int n_aside = 0;
    for (int attempt = 0; attempt < (m->xq_force == 2 ? 16 : 6) && !good; ++attempt) {
        hipStream_t cand = nullptr;
        HIPCHK(m, hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
        LaunchCtx cm = c; cm.stream = m->stream;
        LaunchCtx cs = c; cs.stream = cand;
        launch_spin(cm, 2000);
        launch_spin(cs, 1);
        HIPCHK(m, hipStreamSynchronize(cand));
        const bool apart = hipStreamQuery(m->stream) == hipErrorNotReady && m->xq_force != 1;
        (void)hipGetLastError();
        HIPCHK(m, hipStreamSynchronize(m->stream));
        if (apart) good = cand; else aside[n_aside++] = cand;
    }
    for (int i = 0; i < n_aside; ++i) (void)hipStreamDestroy(aside[i]);
    m->stream3 = good; m->stream3_for = m->stream;
    if (!good) {   // every candidate shared the main stream's hardware queue: the estimator stays a forked branch of the graph
        m->xq_shared = true;
        if (m->xq_force == 2) return dspmap_fail(m, DSPMAP_E_DEVICE, "estimator queue: no stream apart from the main stream's hardware queue (DSPMAP_XQ_FORCE=apart)");
    }
    return DSPMAP_OK;
}

// One device-resident frame after the gate (dspmap_update_device; dspmap_update with the device estimator).
static int device_frame(dspmap* m, int n_points, const float* points_dev, int n_birth, const dspmap_vpoint* birth_dev,
                        const float dp[3], float dt, const float q[4]) {
    int rc = dspmap_ensure_point_cap(m, n_points > n_birth ? n_points : n_birth);
    if (rc != DSPMAP_OK) return rc;
    dspmap_freeze_birth_statics(m);
    const bool want_est = !birth_dev && m->use_vel_est != 0 && !m->cfg.static_model;
    const bool est_dev = want_est && m->use_vel_est == 2 && n_points <= m->ve.cap;
    if (want_est && !est_dev && n_points > 0) {
        // device-resident cloud + the HOST velocity estimator (DSPMAP_P_VELOCITY_ESTIMATOR = 1, or a cloud larger than
        // the device estimator orders in one workgroup): the (<= 60 kB) cloud is copied to the host, clustered and
        // matched there WHILE the device predicts and re-weights (the reference's fork/join, :297,311), and the tagged
        // birth cloud is uploaded for the birth stage.
        rc = dspmap_pts_slot_acquire(m, n_points);
        if (rc != DSPMAP_OK) return rc;
        HIPCHK(m, hipMemcpyAsync(m->pts_pin, points_dev, sizeof(float) * 3 * (size_t)n_points, hipMemcpyDeviceToHost, m->stream));
        rc = dspmap_pts_slot_release(m);
        if (rc != DSPMAP_OK) return rc;
        HIPCHK(m, hipEventRecord(m->ev_fork, m->stream));
        dspmap_prof_collect(m);
        rc = ve_state_to_host(m);   // (the device estimator may hold the previous frame's clusters)
        if (rc != DSPMAP_OK) return rc;
        return frame_with_host_stages(m, n_points, points_dev, q, dp, dt, m->ev_fork);
    }
    if (est_dev) { if (m->ve_last_at != 2) m->xq_break = true; rc = ve_state_to_device(m); if (rc != DSPMAP_OK) return rc; m->ve_last_at = 2; }
    m->frame_parity ^= 1u;
    LaunchCtx c = dspmap_ctx_of(m);
    const bool has_vz = m->vz_frames > 0;
    if (!has_vz) c.s.vz0 = nullptr;
    // birth cloud: the caller's (0), synthesised from the view (1: every point in view a static source), or the device
    // velocity estimator's (2)
    const int mode = birth_dev ? 0 : (est_dev ? 2 : 1);
    const int nb = mode == 0 ? n_birth : n_points;
    fill_pose(m, dp, dt);
    m->hp.n_pts = n_points; m->hp.n_birth = nb; m->hp.static_birth = mode;
    m->hp.pts = points_dev;
    m->hp.birth = mode == 0 ? (BirthSrc*)birth_dev : m->s.birth;
    const int nb_grid = mode != 0 ? dspmap_begin_cloud(m, n_points, true) : (dspmap_begin_cloud(m, n_points, false), nb);
    // A replayed frame reads its parameter block from a pinned ring (its first kernel fetches the slot over the bus):
    // no copy node between two graph launches.  Slot k of the ring is reused DSPMAP_RING frames later; an event per
    // quarter of the ring makes sure the frames that read it have ended (the host never runs that far ahead in practice).
    m->frame_ring = (m->use_graph || m->direct_ring) && !m->prof && m->ring_host != nullptr;
    // the estimator on a queue of its own (DSPMAP_P_ESTIMATOR_QUEUE): replayed frames with the device estimator whose graph would otherwise fork
    // for it alone -- a split placement / early registration keeps its side branch, and the estimator on it
    bool xq = m->est_queue && !m->xq_failed && m->xq_dev && mode == 2 && m->frame_ring && (m->birth_cap + 15) / 16 + 1 <= DSPMAP_XQ_LIST && !frame_splits_placement(m, c, false) && !frame_runs_two_branches(m, c, false);
    if (xq) {   // ... and only on a stream that does not share the main stream's hardware queue (tested once per main stream)
        const int rs = ensure_estimator_stream(m, c);
        if (rs != DSPMAP_OK) return rs;
        if (m->xq_shared) xq = false;
    }
    if (xq) c.s.xq = m->xq_dev;
    if (mode == 2) m->est_path = xq ? 1 : ((m->est_queue && m->xq_shared) ? 2 : 3);
    if (m->frame_ring) {
        const unsigned q = (m->ring_head / (DSPMAP_RING / 4)) % 4;
        if (m->ring_head % (DSPMAP_RING / 4) == 0 && m->ring_ev_set[q]) HIPCHK(m, hipEventSynchronize(m->ring_ev[q]));
        m->hp.clear_fut = m->fut_clear_pending ? 1 : 0;
        m->fut_clear_pending = false;
        m->hp.from_ring = 1;
        m->hp.ring_pos = m->ring_head;
        if (m->host_cloud && n_points > 0) {
            // the boundary's own call (update(float* host, ...), reference :181): the cloud goes into this frame's slot of the
            // mapped cloud ring and k_obs_points fetches it over the bus with the parameter block -- the graph launch below is
            // the only thing queued for the frame
            if (!m->cring_host) {
                m->cring_cap = std::max(1, std::min(m->pt_cap, m->ve.cap));   // (this path only carries clouds the device estimator takes: <= ve.cap points; 64 slots x 12 B each)
                HIPCHK(m, hipHostMalloc((void**)&m->cring_host, sizeof(float) * 3 * (size_t)m->cring_cap * DSPMAP_CLOUD_RING, hipHostMallocMapped));
                void* dp2 = nullptr;
                HIPCHK(m, hipHostGetDevicePointer(&dp2, m->cring_host, 0));
                m->cring_dev = (const float*)dp2;
            }
            if (m->ring_head >= DSPMAP_CLOUD_RING) {   // the frame that read this slot last must be past its first kernel
                const unsigned need = m->ring_head - DSPMAP_CLOUD_RING + 2u;   // (+ 1: the estimator on its own queue reads the slot as well, and is only known to be done with it when the FOLLOWING frame's prediction starts)
                const volatile int* seen = m->hint_host + 2;
                for (long spin = 0; (int)((unsigned)*seen - need) < 0; ++spin) {
                    if (spin > 2000) { HIPCHK(m, hipStreamSynchronize(m->stream)); break; }   // (a queue more than 64 frames deep: wait for it)
                    std::this_thread::yield();
                }
            }
            float* dst = m->cring_host + (size_t)(m->ring_head % DSPMAP_CLOUD_RING) * 3 * (size_t)m->cring_cap;
            const float* src = m->host_cloud;
            const int st = m->host_cloud_stride;
            if (st == 3) memcpy(dst, src, sizeof(float) * 3 * (size_t)n_points);
            else for (int i = 0; i < n_points; i++) {  // xyz are the first three floats of each point (:247,289)
                dst[3 * i] = src[(size_t)i * st]; dst[3 * i + 1] = src[(size_t)i * st + 1]; dst[3 * i + 2] = src[(size_t)i * st + 2];
            }
            m->hp.pts = m->cring_dev + (size_t)(m->ring_head % DSPMAP_CLOUD_RING) * 3 * (size_t)m->cring_cap;
        }
        m->ring_host[m->ring_head % DSPMAP_RING] = m->hp;
    } else {
        rc = dspmap_push_frame_params(m);
        if (rc != DSPMAP_OK) return rc;
    }
    dspmap_prof_collect(m);
    // update_ms (dspmap_get_counters): an event record between two graph launches costs ~5 us of device time each (measured:
    // 0.162 -> 0.150 ms per frame at the metric's workload without them), so a replayed frame carries the pair only every
    // 32nd time; direct launches (profiling, DSPMAP_P_USE_GRAPH = 0) are timed every frame
    const bool timed = !((m->use_graph || m->direct_ring) && !m->prof) || (m->frame_no++ % 32u) == 0;
    if (timed) HIPCHK(m, hipEventRecord(m->ev0, m->stream));
    auto queue_estimator = [&]() -> int {
        if (!xq) return DSPMAP_OK;
            // this frame's estimator on its own queue, queued BEFORE the frame itself.  It waits for nothing of THIS frame (k_ve_view makes its
            // own picture of the view from the ring slot) -- only for the previous frame's birth stage, which hands over the rand() cursor and
            // the birth buffers: through the word that frame's resampling kernel publishes when that frame was the handle's previous call
            // (nothing else can have touched the estimator's state in between), through an event on the handle's stream otherwise (after
            // another entry point -- a pre-processed cloud, an import, new cursors, a frame of another kind --, or on a stream the caller
            // owns and may have queued the cloud's producer on).  The frame's first birth kernel waits for k_ve_clusters' word.  Every wait
            // is for work queued EARLIER, whatever hardware queues the two streams share: nothing to deadlock on.
            LaunchCtx c2 = c;
            c2.stream = m->stream3;
            if (m->xq_test_break) m->xq_break = true;
            const bool chained = m->own_stream && !m->xq_break && m->xq_chain_api + 1 == m->api_seq;
            m->xq_break = false;
            if (!chained) {
                HIPCHK(m, hipEventRecord(m->ev_fork, m->stream));
                HIPCHK(m, hipStreamWaitEvent(m->stream3, m->ev_fork, 0));
            }
            const int seq = (int)(m->hp.ring_pos + 1u);
            // (test hook: every third frame's estimator is held back by the clock; in the same handles the frame's first birth kernel takes
            // every third frame's cloud for unfinished at its first look whatever the clock says -- DevState::xq[10] -- so that the
            // one-waiting-workgroup / deferred-shares path runs in a known set of frames whichever hardware queues the streams share)
            if (m->xq_test_delay_us > 0 && m->xq_frames % 3 == 1) launch_spin(c2, m->xq_test_delay_us);
            launch_velocity_estimator_xq(c2, true, m->ring_dev + (m->ring_head % DSPMAP_RING), m->xq_dev, m->s.hint_out + 3, chained ? m->xq_last_seq : 0, seq);
            m->xq_last_seq = seq;
            m->xq_chain_api = m->api_seq;
            ++m->xq_frames;
            return DSPMAP_OK;
    };
    // A two-branch frame is queued as PLAIN launches on the handle's two streams, parameter block through the same pinned ring: replayed as
    // one captured graph its branches did not overlap (the runtime put the bulk branch's prediction behind the in-view chain's birth
    // kernels on one of its internal streams, profiles/r06_*_timeline*), and at this size (0.4 - 5 ms per frame) the ~0.1 ms of host time its
    // fifteen launches take is hidden behind the device
    const bool two = frame_runs_two_branches(m, c, false);
    if (m->use_graph && !m->prof && !two) {
        // the kernel arguments of a frame are constant (per-frame values live in s.fpar): capture once, replay
        const unsigned long long key = ((unsigned long long)m->graph_epoch << 8) | (has_vz ? 1u : 0u) | ((unsigned)mode << 1) | (c.sparse ? 8u : 0u) | (c.ro_inline ? 16u : 0u) | (xq ? 32u : 0u);
        const int gi = c.sweep_rev ? 1 : 0;   // (one executable graph per sweep direction: the direction is a kernel argument)
        if (!m->graph_exec[gi] || m->graph_key[gi] != key) {
            if (m->graph_exec[gi]) {   // (replays of the old executable graph may still be queued: let them finish before it goes)
                HIPCHK(m, hipStreamSynchronize(m->stream));
                (void)hipGraphExecDestroy(m->graph_exec[gi]); m->graph_exec[gi] = nullptr;
            }
            if (m->graph) { (void)hipGraphDestroy(m->graph); m->graph = nullptr; }
            HIPCHK(m, hipStreamBeginCapture(m->stream, hipStreamCaptureModeRelaxed));
            m->graph_rsplit[gi] = false;
            enqueue_frame(m, c, m->pt_cap, m->birth_cap, false, mode == 1, mode == 2);  // (fork=true measured slower: HIP replays multi-branch graphs with a much higher launch cost) grids sized for the capacity; kernels bound-check against fpar
            HIPCHK(m, hipStreamEndCapture(m->stream, &m->graph));
            m->graph_rsplit[gi] = m->rsplit_enq;
            if (const char* dot = getenv("DSPMAP_GRAPH_DOT")) (void)hipGraphDebugDotPrint(m->graph, dot, 0);   // diagnostics: the frame's nodes and edges
            HIPCHK(m, hipGraphInstantiate(&m->graph_exec[gi], m->graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(m->graph);   // the executable graph keeps its own copy of the topology
            m->graph = nullptr;
            m->graph_key[gi] = key;
        }
        m->last_resample_variant = resample_variant(c);   // (baked into the graph: c.ro_inline is part of its key)
        rc = queue_estimator();
        if (rc != DSPMAP_OK) return rc;
        HIPCHK(m, hipGraphLaunch(m->graph_exec[gi], m->stream));
        if (m->graph_rsplit[gi]) ++m->rsplit_frames;
        if (m->frame_ring) {
            if (m->ring_head % (DSPMAP_RING / 4) == DSPMAP_RING / 4 - 1) {
                const unsigned q = (m->ring_head / (DSPMAP_RING / 4)) % 4;
                HIPCHK(m, hipEventRecord(m->ring_ev[q], m->stream));
                m->ring_ev_set[q] = true;
            }
            ++m->ring_head;
        }
    } else {
        if (m->frame_ring) { rc = queue_estimator(); if (rc != DSPMAP_OK) return rc; }
        m->branch_pending = false;
        enqueue_frame(m, c, n_points, nb_grid, false, mode == 1, mode == 2);
        if (m->branch_pending) ++m->branch_frames;
        if (m->rsplit_enq) ++m->rsplit_frames;
        if (m->frame_ring) {
            if (m->ring_head % (DSPMAP_RING / 4) == DSPMAP_RING / 4 - 1) {
                const unsigned q = (m->ring_head / (DSPMAP_RING / 4)) % 4;
                HIPCHK(m, hipEventRecord(m->ring_ev[q], m->stream));
                m->ring_ev_set[q] = true;
            }
            ++m->ring_head;
        }
    }
    if (m->vz_frames > 0) --m->vz_frames;
    if (m->nb_dirty) { m->nb_dirty = false; m->graph_epoch++; }
    if (timed) { HIPCHK(m, hipEventRecord(m->ev1, m->stream)); m->ev_valid = true; }
    m->last_n_points = n_points;
    m->last_n_birth = nb_grid;
    m->last_birth_static = mode != 0;   // the cloud lives on the device (dspmap_get_birth_cloud materialises it)
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}

extern "C" int dspmap_update_device(dspmap_t* m, int n_points, const float* points_dev, int n_birth,
                                    const dspmap_vpoint* birth_dev, const float pos[3], double stamp,
                                    const float q[4]) {
    READY(m);
    if (n_points < 0 || (n_points > 0 && !points_dev) || !pos || !q) return dspmap_fail(m, DSPMAP_E_ARG, "bad arguments");
    { const int rq = dspmap_check_estimator_queue(m); if (rq != DSPMAP_OK) return rq; }
    float dp[3], dt;
    if (!dspmap_gate_and_delta(m, pos, stamp, q, dp, &dt)) return DSPMAP_REJECTED;
    return device_frame(m, n_points, points_dev, n_birth, birth_dev, dp, dt, q);
}

// The pinned staging buffers rotate: the copy queued from (or into) a slot may still be waiting behind a whole frame when
// the caller comes back with its next cloud, so a slot is refilled only after the event behind its last copy has completed.
int dspmap_pts_slot_acquire(dspmap* m, int n) {
    const unsigned k = m->pts_ring_pos++ % DSPMAP_PTS_RING;
    if (m->pts_ring_busy[k]) { HIPCHK(m, hipEventSynchronize(m->pts_ring_ev[k])); m->pts_ring_busy[k] = false; }
    if (!m->pts_ring_ev[k]) HIPCHK(m, hipEventCreateWithFlags(&m->pts_ring_ev[k], hipEventDisableTiming));
    if (n > m->pts_ring_cap[k] || !m->pts_ring[k]) {
        if (m->pts_ring[k]) (void)hipHostFree(m->pts_ring[k]);
        m->pts_ring[k] = nullptr;
        m->pts_ring_cap[k] = n + n / 2 + 1024;
        HIPCHK(m, hipHostMalloc((void**)&m->pts_ring[k], sizeof(float) * 3 * (size_t)m->pts_ring_cap[k]));
    }
    m->pts_pin = m->pts_ring[k];
    m->pts_pin_cap = m->pts_ring_cap[k];
    return DSPMAP_OK;
}
int dspmap_pts_slot_release(dspmap* m) {
    const unsigned k = (m->pts_ring_pos - 1u) % DSPMAP_PTS_RING;
    HIPCHK(m, hipEventRecord(m->pts_ring_ev[k], m->stream));
    m->pts_ring_busy[k] = true;
    return DSPMAP_OK;
}
int dspmap_stage_points(dspmap* m, int n, int stride, const float* pts) {
    int rc = dspmap_ensure_point_cap(m, n);
    if (rc != DSPMAP_OK) return rc;
    rc = dspmap_pts_slot_acquire(m, n);
    if (rc != DSPMAP_OK) return rc;
    for (int i = 0; i < n; i++) {  // xyz are the first three floats of each point (:247,289)
        m->pts_pin[3 * i] = pts[(size_t)i * stride];
        m->pts_pin[3 * i + 1] = pts[(size_t)i * stride + 1];
        m->pts_pin[3 * i + 2] = pts[(size_t)i * stride + 2];
    }
    if (n > 0) {
        HIPCHK(m, hipMemcpyAsync(m->pts_dev, m->pts_pin, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, m->stream));
        rc = dspmap_pts_slot_release(m);
        if (rc != DSPMAP_OK) return rc;
    }
    return DSPMAP_OK;
}

int dspmap_upload_birth(dspmap* m, const dspmap_vpoint* pts, int n) {
    int rc = dspmap_ensure_point_cap(m, n);
    if (rc != DSPMAP_OK) return rc;
    if (n > m->birth_pin_cap) {
        if (m->birth_ev_set) { HIPCHK(m, hipEventSynchronize(m->birth_ev)); m->birth_ev_set = false; }
        if (m->birth_pin) (void)hipHostFree(m->birth_pin);
        m->birth_pin_cap = n + n / 2 + 1024;
        HIPCHK(m, hipHostMalloc((void**)&m->birth_pin, sizeof(BirthSrc) * (size_t)m->birth_pin_cap));
    }
    static_assert(sizeof(BirthSrc) == sizeof(dspmap_vpoint), "layout");
    if (n > 0) {
        // the previous frame's copy out of this buffer may still be queued behind that frame's kernels
        if (m->birth_ev_set) { HIPCHK(m, hipEventSynchronize(m->birth_ev)); m->birth_ev_set = false; }
        if (!m->birth_ev) HIPCHK(m, hipEventCreateWithFlags(&m->birth_ev, hipEventDisableTiming));
        memcpy(m->birth_pin, pts, sizeof(BirthSrc) * (size_t)n);
        HIPCHK(m, hipMemcpyAsync(m->s.birth, m->birth_pin, sizeof(BirthSrc) * (size_t)n, hipMemcpyHostToDevice, m->stream));
        HIPCHK(m, hipEventRecord(m->birth_ev, m->stream));
        m->birth_ev_set = true;
    }
    return DSPMAP_OK;
}

extern "C" int dspmap_update(dspmap_t* m, int n, int stride, const float* pts, float sx, float sy, float sz,
                             double stamp, float qw, float qx, float qy, float qz) {
    READY(m);
    if (n > 0 && (!pts || stride < 3)) return dspmap_fail(m, DSPMAP_E_ARG, "bad point cloud arguments");
    { const int rq = dspmap_check_estimator_queue(m); if (rq != DSPMAP_OK) return rq; }
    const float pos[3] = {sx, sy, sz};
    const float q[4] = {qw, qx, qy, qz};
    float dp[3], dt;
    if (!dspmap_gate_and_delta(m, pos, stamp, q, dp, &dt)) return DSPMAP_REJECTED;
    const int np = n > 0 ? n : 0;
    const bool dev_frame = m->use_vel_est == 2 && !m->cfg.static_model && !m->h_birth_valid && np <= m->ve.cap;
    if (dev_frame && (m->use_graph || m->direct_ring) && !m->prof && m->ring_host && m->host_direct) {
        // velocity estimator on the device + captured frame: the cloud rides in the pinned cloud ring (device_frame), the frame is one
        // graph launch -- no copy node, no event in front of it (round 4: 5 837 against 6 913 frames/s with the cloud resident in HBM)
        int rc0 = dspmap_ensure_point_cap(m, np);
        if (rc0 != DSPMAP_OK) return rc0;
        m->host_cloud = pts; m->host_cloud_stride = stride;
        rc0 = device_frame(m, np, m->pts_dev, 0, nullptr, dp, dt, q);   // (pts_dev: a valid address; replaced by the ring slot when np > 0)
        m->host_cloud = nullptr;
        return rc0;
    }
    int rc = dspmap_stage_points(m, np, stride, pts);
    if (rc != DSPMAP_OK) return rc;
    m->xq_break = true;   // (the cloud reaches pts_dev through a copy on the handle's stream)
    if (dev_frame)
        return device_frame(m, np, m->pts_dev, 0, nullptr, dp, dt, q);   // velocity estimator on the device: no host stage in the frame
    return frame_with_host_stages(m, np, m->pts_dev, q, dp, dt, nullptr);
}

extern "C" int dspmap_set_birth_cloud(dspmap_t* m, const dspmap_vpoint* pts, int n) {
    if (!m || n < 0 || (n > 0 && !pts)) return DSPMAP_E_ARG;
    m->h_birth.assign(pts, pts + n);
    m->h_birth_valid = true;
    return DSPMAP_OK;
}
extern "C" int dspmap_get_birth_cloud(dspmap_t* m, dspmap_vpoint* out, int cap, int* n_out) {
    READY(m);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (m->last_birth_static) {
        // the synthesised cloud is rebuilt from the frame's view (or is the kept cloud of the last non-empty view)
        BirthSrc* dtmp = nullptr;
        int* dn = nullptr;
        const int capb = m->pt_cap > 0 ? m->pt_cap : 1;
        HIPCHK(m, dalloc(&dtmp, (size_t)capb));
        HIPCHK(m, dalloc(&dn, (size_t)1));
        LaunchCtx c = dspmap_ctx_of(m);
        launch_birth_materialize(c, dtmp, capb, dn);
        int nsrc = 0;
        HIPCHK(m, hipMemcpyAsync(&nsrc, dn, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipStreamSynchronize(m->stream));
        if (nsrc > capb) nsrc = capb;
        std::vector<BirthSrc> tmp((size_t)nsrc);
        if (nsrc) HIPCHK(m, hipMemcpy(tmp.data(), dtmp, sizeof(BirthSrc) * tmp.size(), hipMemcpyDeviceToHost));
        (void)hipFree(dtmp); (void)hipFree(dn);
        int k = 0;
        for (auto& b : tmp)
            if (b.intensity > -1.5f) { if (out && k < cap) memcpy(&out[k], &b, sizeof(b)); ++k; }
        if (n_out) *n_out = k;
    } else {
        const int n = (int)m->h_birth.size();
        for (int i = 0; i < n && i < cap && out; i++) out[i] = m->h_birth[i];
        if (n_out) *n_out = n;
    }
    return DSPMAP_OK;
}

// ----------------------------------------------------------------- readout
static int readout(dspmap* m, float thr, float* xyz, int cap, int* n_out, float* fut_out, bool want_occ) {
    READY(m);
    LaunchCtx c = dspmap_ctx_of(m);
    const MapDims& d = m->d;
    int n = 0;
    if (want_occ) {
        if (!m->s.occ_xyz) { HIPCHK(m, dalloc(&m->s.occ_xyz, (size_t)d.v_loc * 3)); c.s = m->s; }
        launch_occupied_compact(c, thr);
        HIPCHK(m, hipMemcpyAsync(&n, &m->s.fs->occupied_count, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipStreamSynchronize(m->stream));
        const int ncopy = n < cap ? n : cap;
        if (xyz && ncopy > 0) HIPCHK(m, hipMemcpyAsync(xyz, m->s.occ_xyz, sizeof(float) * 3 * (size_t)ncopy, hipMemcpyDeviceToHost, m->stream));
    }
    if (fut_out && d.T > 0) {
        dspmap_flush_future_clear(m);
        launch_future_combine(c);
        HIPCHK(m, hipMemcpyAsync(fut_out, m->s.fut_out, sizeof(float) * (size_t)d.v_true * d.T, hipMemcpyDeviceToHost, m->stream));
    }
    m->fut_clear_pending = true;  // :397-400, :420-424
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (n_out) *n_out = n;
    return DSPMAP_OK;
}
extern "C" int dspmap_get_occupancy(dspmap_t* m, float thr, float* xyz, int cap, int* n_out) {
    return readout(m, thr, xyz, cap, n_out, nullptr, true);
}
extern "C" int dspmap_get_occupancy_with_future(dspmap_t* m, float thr, float* xyz, int cap, int* n_out, float* fut) {
    return readout(m, thr, xyz, cap, n_out, fut, true);
}
extern "C" int dspmap_get_future(dspmap_t* m, float* fut) { return readout(m, 0.f, nullptr, 0, nullptr, fut, false); }
extern "C" int dspmap_clear_future(dspmap_t* m) {
    READY(m);
    BENIGN(m);
    m->fut_clear_pending = true;   // :431-438, carried out by the next frame's k_predict or before the next read
    return DSPMAP_OK;
}
extern "C" int dspmap_get_results(dspmap_t* m, float* out) {
    READY(m);
    BENIGN(m);
    const float4* src = m->s.res4;
    if (m->d.tiling) {   // cube storage: the caller's array is in the reference's voxel order
        if (!m->res_true) HIPCHK(m, dalloc(&m->res_true, (size_t)m->d.v_true));
        launch_results_true(dspmap_ctx_of(m), m->res_true);
        src = m->res_true;
    }
    HIPCHK(m, hipMemcpyAsync(out, src, sizeof(float4) * (size_t)m->d.v_true, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    return DSPMAP_OK;
}
extern "C" const float* dspmap_results_device(dspmap_t* m) {
    if (!m || !m->device_ready) return nullptr;
    if (!m->d.tiling) return (const float*)m->s.res4;
    if (!m->res_true && dalloc(&m->res_true, (size_t)m->d.v_true) != hipSuccess) return nullptr;
    launch_results_true(dspmap_ctx_of(m), m->res_true);   // (stream-ordered like the future view below)
    return (const float*)m->res_true;
}
extern "C" const float* dspmap_future_device(dspmap_t* m) {
    if (!m || !m->device_ready) return nullptr;
    dspmap_flush_future_clear(m);
    LaunchCtx c = dspmap_ctx_of(m);
    launch_future_combine(c);  // [V][T] view of the horizon-major accumulators + the static-particle mass
    return m->s.fut_out;
}

extern "C" void dspmap_voxel_center(const dspmap_t* m, int index, float* px, float* py, float* pz) {  // :1556-1572
    const MapDims& d = m->d;
    const int zc = d.ny * d.nx;
    const int zi = index / zc, rest = index - zi * zc, yi = rest / d.nx, xi = rest - yi * d.nx;
    const float cx = -d.half_x + d.res * 0.5f, cy = -d.half_y + d.res * 0.5f, cz = -d.half_z + d.res * 0.5f;
    *px = (float)xi * d.res + cx; *py = (float)yi * d.res + cy; *pz = (float)zi * d.res + cz;
}
extern "C" int dspmap_point_voxel_index(const dspmap_t* m, float px, float py, float pz, int* index) {  // :1574-1584
    const MapDims& d = m->d;
    if (px >= d.half_x || px <= -d.half_x || py >= d.half_y || py <= -d.half_y || pz >= d.half_z || pz <= -d.half_z) return 0;
    const int x = (int)((px + d.half_x) / d.res), y = (int)((py + d.half_y) / d.res), z = (int)((pz + d.half_z) / d.res);
    *index = z * d.ny * d.nx + y * d.nx + x;
    if (*index < 0 || *index >= d.v_glob) return 0;
    return 1;
}

extern "C" int dspmap_voxel_num(const dspmap_t* m) { return m ? m->d.v_glob : 0; }
extern "C" int dspmap_local_voxel_num(const dspmap_t* m) { return m ? m->d.v_true : 0; }
extern "C" int dspmap_local_voxel_base(const dspmap_t* m) { return m ? m->d.v_base : 0; }
extern "C" int dspmap_slots_per_voxel(const dspmap_t* m) { return m ? m->d.slots : 0; }
extern "C" int dspmap_pyramid_num(const dspmap_t* m) { return m ? m->d.np : 0; }
extern "C" int dspmap_pyramid_capacity(const dspmap_t* m) { return m ? m->d.capp : 0; }

extern "C" int dspmap_get_counters(dspmap_t* m, dspmap_counters* out) {
    READY(m);
    if (!out) return DSPMAP_E_ARG;
    LaunchCtx c = dspmap_ctx_of(m);
    launch_reduce_counters(c);
    FrameScalars fs;
    HIPCHK(m, hipMemcpyAsync(&fs, m->s.fs, sizeof(fs), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    memset(out, 0, sizeof(*out));
    out->n_points_in = m->last_n_points;
    out->n_valid = fs.n_valid; out->n_obs = fs.n_obs; out->n_live_in = fs.n_live_in; out->n_moved = fs.n_moved;
    out->n_out_of_map = fs.n_out_of_map; out->n_voxel_full = fs.n_voxel_full; out->n_pyramid_full = fs.n_pyramid_full;
    out->n_fov = fs.n_fov; out->n_born = fs.n_born; out->n_born_dropped = fs.n_born_dropped;
    out->n_live_out = fs.n_live_out; out->n_exported_up = m->last_exp[1]; out->n_exported_down = m->last_exp[0];
    out->n_reslotted = fs.n_dirty; out->n_overflow_inexact = fs.n_overflow_inexact;
    out->newborn_weight = fs.newborn_w;
    if (m->ev_valid) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->ev0, m->ev1) == hipSuccess) out->update_ms = ms;
    }
    return DSPMAP_OK;
}

// ------------------------------------------------------------ state access
static int ensure_vz(dspmap* m) {
    m->graph_epoch++;
    if (!m->s.vz0) {
        const size_t S = (((size_t)m->d.v_loc + 63) / 64) * 64 * m->d.slots;
        HIPCHK(m, dalloc(&m->s.vz0, S));
        HIPCHK(m, hipMemset(m->s.vz0, 0, sizeof(float) * S));
        if (!m->k.vz_q) HIPCHK(m, dalloc(&m->k.vz_q, (size_t)m->d.v_loc * m->d.mw));
    }
    m->vz_frames = 2;  // seeded (flag 15) particles are first predicted in the second frame
    return DSPMAP_OK;
}

int dspmap_mark_nb_dirty(dspmap* m) {
    if (!m->nbsnap_buf) HIPCHK(m, dalloc(&m->nbsnap_buf, (size_t)m->d.v_loc * m->d.mw));
    if (!m->nb_dirty) m->graph_epoch++;
    m->nb_dirty = true;
    return DSPMAP_OK;
}

extern "C" int dspmap_clear_state(dspmap_t* m) {
    READY(m);
    m->state_epoch++;
    const MapDims& d = m->d;
    const size_t W = (size_t)d.v_loc * d.mw;
    HIPCHK(m, hipMemsetAsync(m->s.mask, 0, sizeof(u64) * W, m->stream));
    HIPCHK(m, hipMemsetAsync(m->s.nbmask, 0, sizeof(u64) * W, m->stream));
    HIPCHK(m, hipMemsetAsync(m->s.res4, 0, sizeof(float4) * (size_t)d.v_loc, m->stream));
    HIPCHK(m, hipMemsetAsync(m->s.fut, 0, sizeof(u64) * (size_t)d.v_loc * (d.T ? d.T : 1), m->stream));
    HIPCHK(m, hipMemsetAsync(m->s.fut_stat, 0, sizeof(float) * (size_t)d.v_loc, m->stream));
    m->fut_clear_pending = false;
    HIPCHK(m, hipMemsetAsync(m->s.pyr_cnt, 0, sizeof(int) * d.np, m->stream));
    HIPCHK(m, hipMemsetAsync(&m->s.fs->vmax_bits, 0, sizeof(int), m->stream));   // (no particle, no speed)
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (m->stream3) HIPCHK(m, hipStreamSynchronize(m->stream3));
    if (m->hint_host) m->hint_host[3] = 0;   // a new state: an earlier frame's give-up on the estimator's queue is history (also: dspmap_load_checkpoint)
    m->xq_failed = false;
    m->have_last = false;
    if (m->nb_dirty) { m->nb_dirty = false; m->graph_epoch++; }
    return DSPMAP_OK;
}

extern "C" int dspmap_import_state(dspmap_t* m, int n, const int* voxel, const int* slot, const float* rec8) {
    READY(m);
    m->state_epoch++;
    if (n < 0 || (n > 0 && (!voxel || !rec8))) return DSPMAP_E_ARG;
    if (n == 0) return DSPMAP_OK;
    bool any_vz = false;
    for (int i = 0; i < n && !any_vz; i++) any_vz = rec8[8 * (size_t)i + 3] != 0.f;
    if (any_vz) { int rc = ensure_vz(m); if (rc != DSPMAP_OK) return rc; }
    bool any_nb = false;
    for (int i = 0; i < n && !any_nb; i++) any_nb = rec8[8 * (size_t)i] > 10.f;
    if (any_nb) { int rc = dspmap_mark_nb_dirty(m); if (rc != DSPMAP_OK) return rc; }
    int *dv = nullptr, *ds = nullptr, *dfail = nullptr;
    float* dr = nullptr;
    HIPCHK(m, dalloc(&dv, (size_t)n));
    HIPCHK(m, dalloc(&dr, (size_t)n * 8));
    HIPCHK(m, dalloc(&dfail, (size_t)1));
    HIPCHK(m, hipMemcpy(dv, voxel, sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(m, hipMemcpy(dr, rec8, sizeof(float) * 8 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(m, hipMemset(dfail, 0, sizeof(int)));
    if (slot) { HIPCHK(m, dalloc(&ds, (size_t)n)); HIPCHK(m, hipMemcpy(ds, slot, sizeof(int) * n, hipMemcpyHostToDevice)); }
    LaunchCtx c = dspmap_ctx_of(m);
    launch_import(c, n, dv, ds, dr, dfail);
    int nfail = 0;
    HIPCHK(m, hipMemcpyAsync(&nfail, dfail, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    (void)hipFree(dv); (void)hipFree(dr); (void)hipFree(dfail); if (ds) (void)hipFree(ds);
    if (nfail) return dspmap_fail(m, DSPMAP_E_STATE, "%d of %d records could not be placed (outside slab, bad slot, or voxel full)", nfail, n);
    return DSPMAP_OK;
}

extern "C" int dspmap_export_state(dspmap_t* m, int cap, int* voxel, int* slot, float* rec8, int* n_out) {
    READY(m);
    if (cap < 0) return DSPMAP_E_ARG;
    int *dv = nullptr, *ds = nullptr, *dc = nullptr;
    float* dr = nullptr;
    const size_t c1 = cap ? cap : 1;
    HIPCHK(m, dalloc(&dv, c1)); HIPCHK(m, dalloc(&ds, c1)); HIPCHK(m, dalloc(&dr, c1 * 8)); HIPCHK(m, dalloc(&dc, (size_t)1));
    HIPCHK(m, hipMemsetAsync(dc, 0, sizeof(int), m->stream));
    LaunchCtx c = dspmap_ctx_of(m);
    if (m->vz_frames <= 0) c.s.vz0 = nullptr;
    launch_export(c, dv, ds, dr, dc, cap);
    int n = 0;
    HIPCHK(m, hipMemcpyAsync(&n, dc, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    const int ncopy = n < cap ? n : cap;
    if (ncopy > 0) {
        if (voxel) HIPCHK(m, hipMemcpy(voxel, dv, sizeof(int) * ncopy, hipMemcpyDeviceToHost));
        if (slot) HIPCHK(m, hipMemcpy(slot, ds, sizeof(int) * ncopy, hipMemcpyDeviceToHost));
        if (rec8) HIPCHK(m, hipMemcpy(rec8, dr, sizeof(float) * 8 * (size_t)ncopy, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dv); (void)hipFree(ds); (void)hipFree(dr); (void)hipFree(dc);
    if (n_out) *n_out = n;
    return DSPMAP_OK;
}

extern "C" int dspmap_add_random_particles(dspmap_t* m, int n, float weight) {
    READY(m);
    m->state_epoch++;
    if (n < 0) return DSPMAP_E_ARG;
    int rc = ensure_vz(m);
    if (rc != DSPMAP_OK) return rc;
    rc = dspmap_mark_nb_dirty(m);
    if (rc != DSPMAP_OK) return rc;
    if ((long long)m->d.v_loc >= (1ll << 24)) return dspmap_fail(m, DSPMAP_E_ARG, "constructor pre-fill supports up to 2^24 voxels per handle");
    LaunchCtx c = dspmap_ctx_of(m);
    int* slot_of = nullptr;
    HIPCHK(m, dalloc(&slot_of, (size_t)(n > 0 ? n : 1)));
    launch_add_random(c, n, weight, slot_of);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    (void)hipFree(slot_of);
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}

extern "C" int dspmap_seed_uniform_moving(dspmap_t* m, int per_voxel, float weight, unsigned seed, float vmax) {
    READY(m);
    m->state_epoch++;
    if (per_voxel < 0 || per_voxel > m->d.slots) return dspmap_fail(m, DSPMAP_E_ARG, "per_voxel must be in [0, %d]", m->d.slots);
    if (!(vmax >= 0.f)) return dspmap_fail(m, DSPMAP_E_ARG, "vmax must be >= 0");
    LaunchCtx c = dspmap_ctx_of(m);
    launch_seed_uniform(c, per_voxel, weight, seed, vmax);
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
extern "C" int dspmap_seed_uniform(dspmap_t* m, int per_voxel, float weight, unsigned seed) {
    return dspmap_seed_uniform_moving(m, per_voxel, weight, seed, 0.f);
}

// ------------------------------------------------------------------ stages
extern "C" int dspmap_stage_bin_points(dspmap_t* m, int n, int stride, const float* pts, float qw, float qx, float qy, float qz) {
    READY(m);
    if (n < 0 || (n > 0 && (!pts || stride < 3))) return DSPMAP_E_ARG;
    m->quat[0] = qw; m->quat[1] = qx; m->quat[2] = qy; m->quat[3] = qz;
    int rc = dspmap_stage_points(m, n, stride, pts);
    if (rc != DSPMAP_OK) return rc;
    LaunchCtx c = dspmap_ctx_of(m);
    for (int i = 0; i < 4; i++) m->hp.quat[i] = m->quat[i];
    for (int i = 0; i < 3; i++) m->hp.cur_pos[i] = m->cur_pos[i];
    m->hp.n_pts = n; m->hp.n_birth = n; m->hp.static_birth = m->h_birth_valid ? 0 : 1;
    m->hp.pts = m->pts_dev; m->hp.birth = m->s.birth;
    const int nb_grid = dspmap_begin_cloud(m, n, !m->h_birth_valid);
    rc = dspmap_push_frame_params(m);
    if (rc != DSPMAP_OK) return rc;
    launch_frame_setup(c, true);
    launch_obs_bin(c, n);
    m->last_n_points = n;
    if (!m->h_birth_valid) { m->last_n_birth = nb_grid; m->last_birth_static = true; }
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
extern "C" int dspmap_set_current_position(dspmap_t* m, float x, float y, float z) {
    if (!m) return DSPMAP_E_ARG;
    m->cur_pos[0] = x; m->cur_pos[1] = y; m->cur_pos[2] = z;
    return DSPMAP_OK;
}
extern "C" int dspmap_stage_predict(dspmap_t* m, float dx, float dy, float dz, float dt) {
    READY(m);
    m->frame_parity ^= 1u;
    LaunchCtx c = dspmap_ctx_of(m);
    if (m->vz_frames <= 0) c.s.vz0 = nullptr;
    for (int i = 0; i < 4; i++) m->hp.quat[i] = m->quat[i];
    for (int i = 0; i < 3; i++) m->hp.cur_pos[i] = m->cur_pos[i];
    m->hp.od[0] = dx; m->hp.od[1] = dy; m->hp.od[2] = dz; m->hp.dt = dt;
    m->update_time += dt; m->update_counter += 1;   // :634-635
    if (!m->hp.birth) m->hp.birth = m->s.birth;
    { int rc = dspmap_push_frame_params(m); if (rc != DSPMAP_OK) return rc; }
    launch_frame_setup(c, false);
    launch_predict(c);
    launch_pyr_prepare(c);   // a full pyramid list turns its latest particles (in sweep order) away: part of the prediction (:1256-1259)
    launch_place_fix(c);     // ... and the arrivals behind a turned-away particle take the slot it hands back
    if (m->vz_frames > 0) --m->vz_frames;
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
extern "C" int dspmap_stage_update(dspmap_t* m) {
    READY(m);
    LaunchCtx c = dspmap_ctx_of(m);
    launch_ck_partial(c);
    launch_weight_update(c);
    launch_ck_finalize(c);
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
extern "C" int dspmap_stage_birth(dspmap_t* m) {
    READY(m);
    dspmap_freeze_birth_statics(m);
    int nb = m->last_n_birth;
    if (m->h_birth_valid) {
        nb = (int)m->h_birth.size();
        int rc = upload_birth(m, m->h_birth.data(), nb);
        if (rc != DSPMAP_OK) return rc;
        m->last_birth_static = false;
        m->last_n_birth = nb;
    }
    LaunchCtx c = dspmap_ctx_of(m);
    if (m->vz_frames <= 0) c.s.vz0 = nullptr;
    for (int i = 0; i < 3; i++) m->hp.cur_pos[i] = m->cur_pos[i];
    m->hp.n_birth = nb; m->hp.birth = m->s.birth;
    m->hp.static_birth = m->last_birth_static ? 1 : 0;   // a cloud supplied after the binning replaces the synthesised one
    { int rc = dspmap_push_frame_params(m); if (rc != DSPMAP_OK) return rc; }
    launch_birth(c, nb, false, false);
    HIPCHK(m, hipGetLastError());
    return dspmap_mark_nb_dirty(m);   // until a resampling turns the newborn flags into 1 (:968)
}
extern "C" int dspmap_stage_resample(dspmap_t* m) {
    READY(m);
    dspmap_flush_future_clear(m);   // a pending clear must not wipe what this stage accumulates
    LaunchCtx c = dspmap_ctx_of(m);
    if (m->vz_frames <= 0) c.s.vz0 = nullptr;
    dspmap_resample(m, c);
    if (m->nb_dirty) { m->nb_dirty = false; m->graph_epoch++; }
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}

extern "C" int dspmap_get_observations(dspmap_t* m, float* obs_out, int* count_out, float* maxlen_out, float* expected_out) {
    READY(m);
    const MapDims& d = m->d;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    std::vector<float4> o((size_t)d.np * DSP_OBS_CAP);
    std::vector<float> ck((size_t)d.np * DSP_OBS_CAP);
    std::vector<int> cnt(d.np);
    HIPCHK(m, hipMemcpy(o.data(), m->s.obs, sizeof(float4) * o.size(), hipMemcpyDeviceToHost));
    HIPCHK(m, hipMemcpy(ck.data(), m->s.obs_ckf, sizeof(float) * ck.size(), hipMemcpyDeviceToHost));
    HIPCHK(m, hipMemcpy(cnt.data(), m->s.obs_cnt, sizeof(int) * d.np, hipMemcpyDeviceToHost));
    if (count_out) memcpy(count_out, cnt.data(), sizeof(int) * d.np);
    if (maxlen_out) HIPCHK(m, hipMemcpy(maxlen_out, m->s.obs_maxlen, sizeof(float) * d.np, hipMemcpyDeviceToHost));
    if (obs_out) {
        memset(obs_out, 0, sizeof(float) * 5 * o.size());
        for (int b = 0; b < d.np; b++)
            for (int j = 0; j < cnt[b]; j++) {
                const size_t i = (size_t)b * DSP_OBS_CAP + j;
                obs_out[5 * i] = o[i].x; obs_out[5 * i + 1] = o[i].y; obs_out[5 * i + 2] = o[i].z;
                obs_out[5 * i + 3] = ck[i]; obs_out[5 * i + 4] = o[i].w;
            }
    }
    if (expected_out) {
        FrameScalars fs;
        HIPCHK(m, hipMemcpy(&fs, m->s.fs, sizeof(fs), hipMemcpyDeviceToHost));
        *expected_out = fs.has_expected_override ? fs.expected_newborn
                                                 : m->fp.nb_weight * (float)fs.n_valid * (float)m->fp.nb_num;
    }
    return DSPMAP_OK;
}
extern "C" int dspmap_set_expected_newborn(dspmap_t* m, float v) {
    READY(m);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    FrameScalars fs;
    HIPCHK(m, hipMemcpy(&fs, m->s.fs, sizeof(fs), hipMemcpyDeviceToHost));
    fs.expected_newborn = v; fs.has_expected_override = 1;
    HIPCHK(m, hipMemcpy(m->s.fs, &fs, sizeof(fs), hipMemcpyHostToDevice));
    return DSPMAP_OK;
}

extern "C" int dspmap_debug_stream(dspmap_t* m, int mode, long long* bytes_out) {
    READY(m);
    const size_t S = (((size_t)m->d.v_loc + 63) / 64) * 64 * m->d.slots;
    LaunchCtx c = dspmap_ctx_of(m);
    launch_calib(c, mode, S);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (bytes_out) *bytes_out = (long long)(mode == 0 ? S * 24 : S * 4);
    return DSPMAP_OK;
}
extern "C" int dspmap_debug_sweep_probe(dspmap_t* m, int what, int rows, int rows_per_batch, int reps, float* ms_out, long long* bytes_out) {
    READY(m);
    if (rows < 1 || rows > m->d.slots || reps < 1) return dspmap_fail(m, DSPMAP_E_ARG, "bad probe arguments");
    LaunchCtx c = dspmap_ctx_of(m);
    launch_sweep_probe(c, what, rows, rows_per_batch);   // warm-up
    HIPCHK(m, hipEventRecord(m->ev0, m->stream));
    for (int i = 0; i < reps; ++i) launch_sweep_probe(c, what, rows, rows_per_batch);
    HIPCHK(m, hipEventRecord(m->ev1, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    float ms = 0.f;
    HIPCHK(m, hipEventElapsedTime(&ms, m->ev0, m->ev1));
    if (ms_out) *ms_out = ms / (float)reps;
    const long long per_cell = ((what & 1) ? 12 : 0) + ((what & 2) ? 8 : 0) + ((what & 4) ? 4 : 0) + ((what & 8) ? 12 : 0);
    if (bytes_out) *bytes_out = (long long)m->k.ntiles * 64ll * rows * per_cell;
    return DSPMAP_OK;
}
extern "C" int dspmap_debug_tile_view(dspmap_t* m, int* out, int cap) {
    READY(m);
    if (!out || cap < m->k.ntiles) return dspmap_fail(m, DSPMAP_E_ARG, "buffer of %d ints needed", m->k.ntiles);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipMemcpy(out, m->k.tile_fov, sizeof(int) * (size_t)m->k.ntiles, hipMemcpyDeviceToHost));
    for (int i = 0; i < m->k.ntiles; ++i) out[i] = (out[i] >> 1) == m->hp.epoch ? (out[i] & 1) : -1;   // -1: not visited by the last k_predict (empty)
    return m->k.ntiles;
}
extern "C" int dspmap_debug_tile_count(dspmap_t* m) {
    READY(m);
    BENIGN(m);
    return m->k.ntiles;
}
extern "C" int dspmap_debug_tile_of_voxels(dspmap_t* m, int n, const int* voxel_global, int* tile_out) {
    READY(m);
    BENIGN(m);
    if (n < 0 || (n > 0 && (!voxel_global || !tile_out))) return DSPMAP_E_ARG;
    for (int i = 0; i < n; ++i) {
        const long long t = (long long)voxel_global[i] - m->d.v_base;
        tile_out[i] = (t >= 0 && t < m->d.v_true) ? (int)(host_lv_of_true(m->d, (size_t)t) >> 6) : -1;
    }
    return DSPMAP_OK;
}
extern "C" int dspmap_debug_tile_moving(dspmap_t* m, int* out, int cap) {
    READY(m);
    if (!out || cap < m->k.ntiles) return DSPMAP_E_ARG;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipMemcpy(out, m->s.tile_moving, sizeof(int) * m->k.ntiles, hipMemcpyDeviceToHost));
    return m->k.ntiles;
}
extern "C" int dspmap_debug_estimator_queue(dspmap_t* m, long long out[6]) {
    READY(m);
    BENIGN(m);
    if (!out) return DSPMAP_E_ARG;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (m->stream3) HIPCHK(m, hipStreamSynchronize(m->stream3));
    int w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m->xq_dev) HIPCHK(m, hipMemcpy(w, m->xq_dev, sizeof(w), hipMemcpyDeviceToHost));
    out[0] = m->xq_frames; out[1] = w[0]; out[2] = w[1]; out[3] = m->hint_host ? m->hint_host[3] : 0; out[4] = w[6]; out[5] = w[7];
    return DSPMAP_OK;
}
extern "C" int dspmap_debug_estimator_path(dspmap_t* m) {
    if (!m) return DSPMAP_E_ARG;
    return m->est_path;
}
extern "C" int dspmap_debug_rollout_paths(dspmap_t* m, long long out[3]) {
    READY(m);
    if (!out) return DSPMAP_E_ARG;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    out[0] = m->last_resample_variant; out[1] = 0; out[2] = 0;
    const int ro = m->last_resample_variant >> 1;
    if (ro == 1 || ro == 2) {   // k_rollout ran: its groups' counts
        const size_t ng = (size_t)rollout_groups(m->d, m->k.ntiles) * ((m->d.tiling && ro == 2) ? 4 : 1);   // workgroups of that launch
        std::vector<int> st(2 * ng);
        HIPCHK(m, hipMemcpy(st.data(), m->k.ro_stat, sizeof(int) * 2 * ng, hipMemcpyDeviceToHost));
        for (size_t g = 0; g < ng; ++g) { out[1] += st[2 * g]; out[2] += st[2 * g + 1]; }
    }
    return DSPMAP_OK;
}
extern "C" int dspmap_debug_frame_branches(dspmap_t* m, long long out[5]) {
    READY(m);
    BENIGN(m);
    if (!out) return DSPMAP_E_ARG;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    out[0] = m->branch_frames; out[1] = 0; out[2] = 0; out[3] = m->k.ntiles; out[4] = 0;
    std::vector<int> cls((size_t)m->k.ntiles);
    HIPCHK(m, hipMemcpy(cls.data(), m->k.tile_cls, sizeof(int) * cls.size(), hipMemcpyDeviceToHost));
    for (int c : cls) { out[1] += (c & TILE_Q) ? 1 : 0; out[2] += (c & TILE_P) ? 1 : 0; }
    int vb = 0;
    HIPCHK(m, hipMemcpy(&vb, &m->s.fs->vmax_bits, sizeof(int), hipMemcpyDeviceToHost));
    float vf; memcpy(&vf, &vb, sizeof(float));
    out[4] = (long long)(vf * 1000.f);
    return DSPMAP_OK;
}
extern "C" long long dspmap_debug_resample_split_frames(dspmap_t* m) { return m ? m->rsplit_frames : 0; }
extern "C" int dspmap_get_pyramid_counts(dspmap_t* m, int* out) {
    READY(m);
    if (!out) return DSPMAP_E_ARG;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipMemcpy(out, m->s.pyr_kept ? m->s.pyr_kept : m->s.pyr_cnt, sizeof(int) * m->d.np, hipMemcpyDeviceToHost));   // (a sharded map's global cut: what this rank keeps)
    for (int i = 0; i < m->d.np; i++) if (out[i] > m->d.capp) out[i] = m->d.capp;
    return DSPMAP_OK;
}
extern "C" int dspmap_set_profiling(dspmap_t* m, int on) {
    READY(m);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (on && !m->pev[0])
        for (int i = 0; i <= DSPMAP_N_STAGES; i++) HIPCHK(m, hipEventCreate(&m->pev[i]));
    m->prof = on != 0;
    m->prof_pending = false;
    m->prof_frames = 0;
    for (int i = 0; i < DSPMAP_N_STAGES; i++) m->stage_ms[i] = 0.0;
    if (on) {
        // what an event bracket adds to the kernel inside it (the record's own cost on the queue + the launch gaps either side):
        // a one-wave kernel that waits a KNOWN 20 us between two records, 24 times; the median excess is subtracted by callers
        // that want kernel durations from the stage brackets (bench.py's roofline; rocprofv3's kernel durations agree)
        LaunchCtx c = dspmap_ctx_of(m);
        std::vector<float> ex;
        for (int k = 0; k < 24; ++k) {
            HIPCHK(m, hipEventRecord(m->pev[0], m->stream));
            launch_spin(c, 20);
            HIPCHK(m, hipEventRecord(m->pev[1], m->stream));
            HIPCHK(m, hipEventSynchronize(m->pev[1]));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, m->pev[0], m->pev[1]) == hipSuccess) ex.push_back(ms - 0.020f);
        }
        std::sort(ex.begin(), ex.end());
        m->event_overhead_ms = ex.empty() ? 0.f : std::max(0.f, ex[ex.size() / 2]);
    }
    return DSPMAP_OK;
}
extern "C" int dspmap_get_event_overhead_ms(dspmap_t* m, float* out) {
    if (!m || !out) return DSPMAP_E_ARG;
    *out = m->event_overhead_ms;
    return DSPMAP_OK;
}
extern "C" int dspmap_get_stage_ms(dspmap_t* m, float out[DSPMAP_N_STAGES], int* n_frames) {
    READY(m);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    dspmap_prof_collect(m);
    for (int i = 0; i < DSPMAP_N_STAGES; i++) out[i] = (float)m->stage_ms[i];
    if (n_frames) *n_frames = m->prof_frames;
    return DSPMAP_OK;
}

// ------------------------------------------------- multi-GPU split-phase (see dspmap_mgpu.hip)

// ------------------------------------------------------------------ binary checkpoint
// The reference has no resume capability (its only dump is the one-shot particle CSV, :326-350).  A checkpoint
// holds what the next update() depends on: every live particle with its slot, the function statics of update()
// (:187-190) and of the birth stage (:808-811), the table cursors, the result grid and the future accumulators.
// Not saved: the Gaussian / rand() tables (regenerated from the configuration's seed, or re-injected by the
// caller) and the host velocity estimator's previous clusters (the first frame after a restore matches nothing,
// exactly like the first frame of a run).
namespace {
struct CkHeader {
    char magic[8];
    int version;
    dspmap_config cfg;
    FilterParams fp;
    int nb_frozen, have_last, vz_frames, pad;
    float last_p[3], cur_pos[3], quat[4], dt_last, p_stddev, v_stddev, voxel_filter_res;
    double last_stamp;
    int cursors[3];
    int n_particles;
    long long v_loc;
};
}  // namespace

extern "C" int dspmap_save_checkpoint(dspmap_t* m, const char* path) {
    READY(m);
    if (!path) return DSPMAP_E_ARG;
    dspmap_flush_future_clear(m);
    int n = 0;
    int rc = dspmap_export_state(m, 0, nullptr, nullptr, nullptr, &n);
    if (rc != DSPMAP_OK) return rc;
    std::vector<int> voxel((size_t)n + 1), slot((size_t)n + 1);
    std::vector<float> rec((size_t)n * 8 + 8);
    rc = dspmap_export_state(m, n, voxel.data(), slot.data(), rec.data(), &n);
    if (rc != DSPMAP_OK) return rc;
    const size_t V = (size_t)m->d.v_true, Vs = (size_t)m->d.v_loc, T = (size_t)m->d.T;
    // the future accumulators as they are: fixed-point sums of the moving particles [T][V] + the static particles' mass [V]
    std::vector<float> res(Vs * 4), fstat(Vs);
    std::vector<u64> fut(Vs * (T ? T : 1));
    HIPCHK(m, hipMemcpyAsync(res.data(), m->s.res4, sizeof(float4) * Vs, hipMemcpyDeviceToHost, m->stream));
    if (T) HIPCHK(m, hipMemcpyAsync(fut.data(), m->s.fut, sizeof(u64) * Vs * T, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipMemcpyAsync(fstat.data(), m->s.fut_stat, sizeof(float) * Vs, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (m->d.tiling) {   // cube storage -> the reference's voxel order
        std::vector<float> res2(V * 4), fstat2(V);
        std::vector<u64> fut2(V * (T ? T : 1));
        for (size_t t = 0; t < V; ++t) {
            const size_t lv = host_lv_of_true(m->d, t);
            for (int q = 0; q < 4; ++q) res2[t * 4 + q] = res[lv * 4 + q];
            fstat2[t] = fstat[lv];
            for (size_t hh = 0; hh < T; ++hh) fut2[hh * V + t] = fut[hh * Vs + lv];
        }
        res.swap(res2); fstat.swap(fstat2); fut.swap(fut2);
    }
    CkHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "DSPMAPCK", 8);
    h.version = 2;
    h.cfg = m->cfg; h.fp = m->fp;
    h.nb_frozen = m->nb_frozen; h.have_last = m->have_last; h.vz_frames = m->vz_frames;
    for (int i = 0; i < 3; i++) { h.last_p[i] = m->last_p[i]; h.cur_pos[i] = m->cur_pos[i]; }
    for (int i = 0; i < 4; i++) h.quat[i] = m->quat[i];
    h.dt_last = m->dt_last; h.p_stddev = m->p_stddev; h.v_stddev = m->v_stddev; h.voxel_filter_res = m->voxel_filter_res;
    h.last_stamp = m->last_stamp;
    rc = dspmap_get_cursors(m, &h.cursors[0], &h.cursors[1], &h.cursors[2]);
    if (rc != DSPMAP_OK) return rc;
    h.n_particles = n; h.v_loc = (long long)V;
    FILE* f = fopen(path, "wb");
    if (!f) return dspmap_fail(m, DSPMAP_E_ARG, "cannot open %s for writing", path);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && (n == 0 || (fwrite(voxel.data(), sizeof(int), n, f) == (size_t)n && fwrite(slot.data(), sizeof(int), n, f) == (size_t)n &&
                           fwrite(rec.data(), sizeof(float) * 8, n, f) == (size_t)n));
    ok = ok && fwrite(res.data(), sizeof(float) * 4, V, f) == V;
    ok = ok && (T == 0 || fwrite(fut.data(), sizeof(u64) * T, V, f) == V);
    ok = ok && fwrite(fstat.data(), sizeof(float), V, f) == V;
    ok = (fclose(f) == 0) && ok;
    if (!ok) return dspmap_fail(m, DSPMAP_E_STATE, "short write to %s", path);
    return DSPMAP_OK;
}

extern "C" int dspmap_load_checkpoint(dspmap_t* m, const char* path) {
    READY(m);
    if (!path) return DSPMAP_E_ARG;
    FILE* f = fopen(path, "rb");
    if (!f) return dspmap_fail(m, DSPMAP_E_ARG, "cannot open %s", path);
    CkHeader h;
    // version 2 (round 4): the future accumulators as they are -- u64 fixed point [T][V] + the static particles' mass [V];
    // version 1 (rounds 1-3): ONE float array in the caller's layout [V][T], static mass folded in.  Both are read.
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "DSPMAPCK", 8) != 0 || (h.version != 2 && h.version != 1)) {
        fclose(f);
        return dspmap_fail(m, DSPMAP_E_ARG, "%s is not a version-1 / version-2 dspmap checkpoint", path);
    }
    const dspmap_config &a = h.cfg, &b = m->cfg;
    bool same = a.nx == b.nx && a.ny == b.ny && a.nz == b.nz && a.voxel_resolution == b.voxel_resolution &&
                a.angle_resolution == b.angle_resolution && a.max_particle_num_voxel == b.max_particle_num_voxel &&
                a.half_fov_h == b.half_fov_h && a.half_fov_v == b.half_fov_v && a.prediction_times == b.prediction_times &&
                a.z_lo == b.z_lo && a.z_hi == b.z_hi && h.v_loc == (long long)m->d.v_true;
    for (int k = 0; same && k < a.prediction_times; k++) same = a.prediction_future_time[k] == b.prediction_future_time[k];
    same = same && a.pyramid_neighbor_n == b.pyramid_neighbor_n && a.safe_particle_factor == b.safe_particle_factor &&
           a.static_model == b.static_model;
    if (!same) { fclose(f); return dspmap_fail(m, DSPMAP_E_ARG, "checkpoint was written by a map with a different configuration"); }
    const int n = h.n_particles;
    if (n < 0 || (long long)n > (long long)m->d.v_true * m->d.slots) {
        fclose(f);
        return dspmap_fail(m, DSPMAP_E_ARG, "%s: particle count %d outside [0, %lld]", path, n, (long long)m->d.v_true * m->d.slots);
    }
    const size_t V = (size_t)m->d.v_true, Vs = (size_t)m->d.v_loc, T = (size_t)m->d.T;
    std::vector<int> voxel((size_t)n + 1), slot((size_t)n + 1);
    std::vector<float> rec((size_t)n * 8 + 8), res(V * 4), fstat(V);
    std::vector<u64> fut(V * (T ? T : 1));
    bool ok = n == 0 || (fread(voxel.data(), sizeof(int), n, f) == (size_t)n && fread(slot.data(), sizeof(int), n, f) == (size_t)n &&
                         fread(rec.data(), sizeof(float) * 8, n, f) == (size_t)n);
    ok = ok && fread(res.data(), sizeof(float) * 4, V, f) == V;
    if (h.version == 2) {
        ok = ok && (T == 0 || fread(fut.data(), sizeof(u64) * T, V, f) == V);
        ok = ok && fread(fstat.data(), sizeof(float), V, f) == V;
    } else if (T) {
        // version 1: float [V][T], the sum of both parts -> quantised onto the accumulators' grid (2^-24 per unit of weight, the
        // device's fut_quantum), horizon-major; the static part is in there already
        std::vector<float> f1(V * T);
        ok = ok && fread(f1.data(), sizeof(float) * T, V, f) == V;
        for (size_t v = 0; ok && v < V; ++v)
            for (size_t t = 0; t < T; ++t) {
                const float x = f1[v * T + t];
                fut[t * V + v] = x > 0.f ? (u64)llrint((double)x * 16777216.0) : 0ull;
            }
    }
    fclose(f);
    if (!ok) return dspmap_fail(m, DSPMAP_E_ARG, "%s is truncated", path);
    if (m->d.tiling) {   // the reference's voxel order -> cube storage (padding voxels: zero)
        std::vector<float> res2(Vs * 4, 0.f), fstat2(Vs, 0.f);
        std::vector<u64> fut2(Vs * (T ? T : 1), 0ull);
        for (size_t t = 0; t < V; ++t) {
            const size_t lv = host_lv_of_true(m->d, t);
            for (int q = 0; q < 4; ++q) res2[lv * 4 + q] = res[t * 4 + q];
            fstat2[lv] = fstat[t];
            for (size_t hh = 0; hh < T; ++hh) fut2[hh * Vs + lv] = fut[hh * V + t];
        }
        res.swap(res2); fstat.swap(fstat2); fut.swap(fut2);
    }
    int rc = dspmap_clear_state(m);
    if (rc != DSPMAP_OK) return rc;
    rc = dspmap_import_state(m, n, voxel.data(), slot.data(), rec.data());
    if (rc != DSPMAP_OK) return rc;
    HIPCHK(m, hipMemcpyAsync(m->s.res4, res.data(), sizeof(float4) * Vs, hipMemcpyHostToDevice, m->stream));
    // (on the handle's stream, behind clear_state's memsets of the same buffers and the import)
    if (T) HIPCHK(m, hipMemcpyAsync(m->s.fut, fut.data(), sizeof(u64) * Vs * T, hipMemcpyHostToDevice, m->stream));
    HIPCHK(m, hipMemcpyAsync(m->s.fut_stat, fstat.data(), sizeof(float) * Vs, hipMemcpyHostToDevice, m->stream));
    HIPCHK(m, hipMemsetAsync(m->s.fut_dirty, 1, sizeof(int) * (size_t)m->k.ntiles, m->stream));   // (any tile may hold mass now)
    HIPCHK(m, hipStreamSynchronize(m->stream));
    {   // filter parameters and the frozen birth statics come from the checkpoint; the random tables are THIS handle's
        // (regenerated from its seed or injected by its caller), so their lengths stay, and the pair-cull radius is
        // re-derived from this handle's DSPMAP_P_PAIR_CULL_SIGMAS
        FilterParams& f = m->fp;
        f.sigma_ob = h.fp.sigma_ob; f.kappa = h.fp.kappa; f.p_det = h.fp.p_det;
        f.nb_weight = h.fp.nb_weight; f.nb_num = h.fp.nb_num;
        f.min_static_nb = h.fp.min_static_nb; f.model_nb = h.fp.model_nb;
        f.occl_margin = h.fp.occl_margin;
        refresh_fp(m);
    }
    m->nb_frozen = h.nb_frozen != 0; m->have_last = h.have_last != 0; m->vz_frames = h.vz_frames;
    for (int i = 0; i < 3; i++) { m->last_p[i] = h.last_p[i]; m->cur_pos[i] = h.cur_pos[i]; }
    for (int i = 0; i < 4; i++) m->quat[i] = h.quat[i];
    m->dt_last = h.dt_last; m->p_stddev = h.p_stddev; m->v_stddev = h.v_stddev; m->voxel_filter_res = h.voxel_filter_res;
    m->last_stamp = h.last_stamp;
    m->fut_clear_pending = false;
    m->graph_epoch++;
    // cursors index THIS handle's tables
    return dspmap_set_cursors(m, h.cursors[0] % (m->fp.tab_n > 0 ? m->fp.tab_n : 1), h.cursors[1] % (m->fp.tab_n > 0 ? m->fp.tab_n : 1),
                              h.cursors[2] % (m->fp.rtab_n > 0 ? m->fp.rtab_n : 1));
}

#ifdef VE_DEBUG
extern "C" int dspmap_debug_ve(const VelEst* ve, long long* out);
extern "C" int dspmap_debug_ve_get(dspmap_t* m, long long* out) { hipDeviceSynchronize(); return dspmap_debug_ve(&m->ve, out); }
#endif
