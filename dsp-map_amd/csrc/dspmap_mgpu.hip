// dspmap_mgpu.hip -- split-phase frame for Z-slab sharding across GPUs (include/dspmap.h,
// "multi-GPU split-phase frame").  One process per GPU; the collectives between the phases are
// issued by the caller through torch.distributed / RCCL on buffers it owns and binds here.
#include <vector>

#include "dspmap_internal.h"

extern "C" int dspmap_mgpu_bind(dspmap_t* m, long long* ck_dev, int* nstatic_dev, int nstatic_cap) {
    INDEX_ORDER(m);
    READY(m);
    if (!ck_dev || !nstatic_dev || nstatic_cap <= 0) return dspmap_fail(m, DSPMAP_E_ARG, "bad buffers");
    HIPCHK(m, hipStreamSynchronize(m->stream));
    int rc = dspmap_ensure_point_cap(m, nstatic_cap);  // so that no later frame re-allocates the point buffers
    if (rc != DSPMAP_OK) return rc;
    if (!m->mgpu_bound) {
        (void)hipFree(m->s.obs_ck);
        (void)hipFree(m->s.nstatic);
    }
    m->s.obs_ck = ck_dev;
    m->s.nstatic = nstatic_dev;
    m->mgpu_bound = true;
    m->mgpu_nstatic_cap = nstatic_cap;
    HIPCHK(m, hipMemsetAsync(ck_dev, 0, sizeof(long long) * (size_t)m->d.np * DSP_OBS_CAP, m->stream));
    if (!m->k.expmask) {
        const size_t W = (size_t)m->d.v_loc * m->d.mw;
        HIPCHK(m, hipMalloc((void**)&m->k.expmask, sizeof(u64) * W));
        HIPCHK(m, hipMemsetAsync(m->k.expmask, 0, sizeof(u64) * W, m->stream));
    }
    if (!m->mgpu_count) HIPCHK(m, hipMalloc((void**)&m->mgpu_count, sizeof(int)));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    return DSPMAP_OK;
}

extern "C" int dspmap_mgpu_begin(dspmap_t* m, int n_points, const float* points_dev, int n_birth,
                                 const dspmap_vpoint* birth_dev, const float pos[3], double stamp, const float q[4]) {
    INDEX_ORDER(m);
    READY(m);
    if (!m->mgpu_bound) return dspmap_fail(m, DSPMAP_E_STATE, "call dspmap_mgpu_bind first");
    if (n_points < 0 || (n_points > 0 && !points_dev) || !pos || !q) return dspmap_fail(m, DSPMAP_E_ARG, "bad arguments");
    if (n_points > m->mgpu_nstatic_cap || n_points > m->pt_cap || (birth_dev && n_birth > m->mgpu_nstatic_cap))
        return dspmap_fail(m, DSPMAP_E_ARG, "more points than the bound capacity");
    // birth cloud: the caller's (mode 0), every point in view a static source (1), or -- DSPMAP_P_VELOCITY_ESTIMATOR = 2 -- the
    // device velocity estimator's (2).  The estimator (velocityEstimationThread :1377-1544, forked and joined by every
    // update() :297,311) works on the frame's cloud, which every rank holds, and is deterministic: every rank runs it
    // redundantly and gets the same tagged cloud bit for bit, exactly like the rank / children of the birth stage.
    bool want_est = !birth_dev && m->use_vel_est != 0 && !m->cfg.static_model;
    const bool est_host = want_est && (m->use_vel_est != 2 || n_points > m->ve.cap);
    float dp[3], dt;
    if (!dspmap_gate_and_delta(m, pos, stamp, q, dp, &dt)) return DSPMAP_REJECTED;
    if (est_host && n_points > 0) {
        // DSPMAP_P_VELOCITY_ESTIMATOR = 1, or a cloud beyond the device estimator's capacity: every rank runs the HOST stage on the
        // replicated cloud (deterministic: the same tagged cloud everywhere), like the unsharded map falls back to it
        // (dspmap_api.hip: device_frame).  One D2H copy of the cloud and a host synchronisation in such a frame.
        int rc = dspmap_pts_slot_acquire(m, n_points);
        if (rc != DSPMAP_OK) return rc;
        HIPCHK(m, hipMemcpyAsync(m->pts_pin, points_dev, sizeof(float) * 3 * (size_t)n_points, hipMemcpyDeviceToHost, m->stream));
        rc = dspmap_pts_slot_release(m);
        if (rc != DSPMAP_OK) return rc;
        rc = dspmap_ve_state_to_host(m);   // (synchronises the stream: the copy has landed)
        if (rc != DSPMAP_OK) return rc;
        HIPCHK(m, hipStreamSynchronize(m->stream));
        std::vector<float> view;
        view.reserve((size_t)n_points * 3);
        m->vel.rotate_and_filter(m->pts_pin, n_points, q, view);
        m->vel.run(view, m->cur_pos, dt, m->voxel_filter_res, m->h_birth);
        m->ve_last_at = 1;
        rc = dspmap_upload_birth(m, m->h_birth.data(), (int)m->h_birth.size());
        if (rc != DSPMAP_OK) return rc;
        birth_dev = reinterpret_cast<const dspmap_vpoint*>(m->s.birth);
        n_birth = (int)m->h_birth.size();
        want_est = false;
    } else if (est_host) {
        want_est = false;   // (an empty cloud: nothing to estimate, the view is empty either way)
    }
    if (want_est) { const int rc = dspmap_ve_state_to_device(m); if (rc != DSPMAP_OK) return rc; }
    const int nb_own = birth_dev ? n_birth : n_points;
    if (nb_own > m->mgpu_nstatic_cap) return dspmap_fail(m, DSPMAP_E_ARG, "more birth sources than the bound capacity");
    dspmap_freeze_birth_statics(m);
    m->frame_parity ^= 1u;
    LaunchCtx c = dspmap_ctx_of(m);
    if (m->vz_frames <= 0) c.s.vz0 = nullptr;
    const int mode = birth_dev ? 0 : (want_est ? 2 : 1);
    const bool static_birth = mode != 0;     // the cloud lives on the device (synthesised or estimated)
    m->mgpu_birth = static_birth ? nullptr : (BirthSrc*)birth_dev;
    for (int i = 0; i < 4; i++) m->hp.quat[i] = m->quat[i];
    for (int i = 0; i < 3; i++) { m->hp.cur_pos[i] = m->cur_pos[i]; m->hp.od[i] = -dp[i]; }
    m->hp.dt = dt;
    m->hp.res_filter = m->voxel_filter_res;
    m->hp.n_pts = n_points; m->hp.n_birth = nb_own; m->hp.static_birth = mode;
    const int nb_grid = dspmap_begin_cloud(m, n_points, static_birth);
    const int nb = static_birth ? nb_grid : nb_own;   // grid bound of the birth launches
    m->hp.pts = points_dev;
    m->hp.birth = static_birth ? m->s.birth : m->mgpu_birth;
    // the parameter block travels through the pinned ring (the frame's first kernel fetches it over the bus): a pageable H2D copy
    // makes the host wait for the stream to drain, and every kernel of the frame is then launched into an empty queue
    const FrameParams* ring = dspmap_ring_push(m);
    if (!ring) { int rcp = dspmap_push_frame_params(m); if (rcp != DSPMAP_OK) return rcp; }
    dspmap_prof_collect(m);
    HIPCHK(m, hipEventRecord(m->ev0, m->stream));
    launch_setup_and_bin(c, n_points, false, ring, DSPMAP_RING - 1);
    // the estimator runs BESIDE the prediction, like the reference's helper thread (:297,311) and like the unsharded frame's side branch: it
    // needs the binned view only; whoever needs its cloud (newborn children, the birth split) joins the side stream first (mgpu_side_pending)
    const bool est_side = mode == 2 && nb > 0 && m->stream2 && m->stream2 != m->stream;
    if (est_side) HIPCHK(m, hipEventRecord(m->ev_fork, m->stream));
    // + gather + (static tags) the birth rank; k_place follows the exchange: imported movers take part in the sweep-order placement
    launch_predict_only(c, true, nb > 0 && mode != 2);
    if (ring) dspmap_ring_pushed(m);
    m->mgpu_est_side = false;
    if (mode == 2 && nb > 0) {   // ... with the estimator the rank rides on k_ve_clusters
        if (est_side) {
            HIPCHK(m, hipStreamWaitEvent(m->stream2, m->ev_fork, 0));
            LaunchCtx c2 = c;
            c2.stream = m->stream2;
            launch_velocity_estimator(c2, true);
            HIPCHK(m, hipEventRecord(m->ev_join, m->stream2));
            m->mgpu_est_side = true; m->mgpu_side_pending = true;
        } else launch_velocity_estimator(c, true);
        m->ve_last_at = 2;
    }
    m->mgpu_all_static = mode == 1;
    m->mgpu_place_pending = true;
    m->mgpu_interior_done = false;
    m->mgpu_birth_early = false;
    m->vz_frames_at_begin = m->vz_frames;
    if (m->vz_frames > 0) --m->vz_frames;
    m->last_n_points = n_points;
    m->last_n_birth = nb;
    m->last_birth_static = static_birth;
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}

extern "C" int dspmap_mgpu_export(dspmap_t* m, int dir, float* rec_dev_out, int cap, int* n_out) {
    READY(m);
    if (!m->mgpu_bound || !rec_dev_out || cap < 0 || !n_out) return dspmap_fail(m, DSPMAP_E_ARG, "bad arguments");
    LaunchCtx c = dspmap_ctx_of(m);
    HIPCHK(m, hipMemsetAsync(m->mgpu_count, 0, sizeof(int), m->stream));
    launch_export_slab(c, dir, rec_dev_out, cap, m->mgpu_count);
    int n = 0;
    HIPCHK(m, hipMemcpyAsync(&n, m->mgpu_count, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (n > cap) return dspmap_fail(m, DSPMAP_E_STATE, "export buffer too small: %d > %d", n, cap);
    *n_out = n;
    m->last_exp[dir > 0 ? 1 : 0] = n;
    return DSPMAP_OK;
}

// Asynchronous variant for stream-ordered drivers: both directions in one call, counts stay on the device
// (counts_dev[0] = up, [1] = down) so that the caller can all-gather them without a host round trip first.
extern "C" int dspmap_mgpu_export_both(dspmap_t* m, float* up_dev_out, float* down_dev_out, int cap, int* counts_dev) {
    READY(m);
    if (!m->mgpu_bound || !up_dev_out || !down_dev_out || cap < 0 || !counts_dev) return dspmap_fail(m, DSPMAP_E_ARG, "bad arguments");
    LaunchCtx c = dspmap_ctx_of(m);
    HIPCHK(m, hipMemsetAsync(counts_dev, 0, 2 * sizeof(int), m->stream));
    launch_export_slab(c, 0, up_dev_out, cap, counts_dev, down_dev_out);   // one pass over the slab's occupancy words for both faces
    // the caller now synchronises with the host to size the exchange: the birth rank and the newborn children only need
    // the frame's birth cloud, so they fill that gap instead of sitting in dspmap_mgpu_finish
    dspmap_mgpu_birth_early(m, c);   // (the rank rode on k_predict's launch / on the estimator's)
    m->mgpu_birth_early = true;
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
// Placement of the movers in the slab's INTERIOR, before the neighbour exchange: a particle changes layer only through
// the sensor's vertical motion (vz == 0), so records from a neighbour can only land within ceil(|dz| / res) layers of a
// slab face; every tile farther inside already holds all its arrivals.  dspmap_mgpu_ck_partial then places the boundary
// tiles.  Gives the GPU work for the time the driver spends synchronising with the host to size the exchange.
extern "C" int dspmap_mgpu_place_interior(dspmap_t* m) {
    READY(m);
    if (!m->mgpu_place_pending || m->mgpu_interior_done) return DSPMAP_OK;
    if (m->vz_frames_at_begin > 0) return DSPMAP_OK;   // constructor-seeded particles still carry vz: no bound on the layer change
    const MapDims& d = m->d;
    const long long L = (long long)d.nx * d.ny;
    const int reach = (int)ceilf(fabsf(m->hp.od[2]) / d.res) + 1;     // layers an import can reach from a face (+1: rounding)
    const long long lo_v = L * reach, hi_v = (long long)d.v_loc - L * reach;
    const int lo = (int)((lo_v + 63) / 64), hi = (int)(hi_v > 0 ? hi_v / 64 : 0);
    if (hi <= lo) return DSPMAP_OK;
    LaunchCtx c = dspmap_ctx_of(m);
    c.s.vz0 = nullptr;
    launch_claim(c, 0, 1, lo, hi);
    m->mgpu_interior_done = true; m->mgpu_tile_lo = lo; m->mgpu_tile_hi = hi;
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
// bookkeeping for dspmap_get_counters once the caller has read the counts of dspmap_mgpu_export_both
extern "C" int dspmap_mgpu_set_export_counts(dspmap_t* m, int n_up, int n_down) {
    if (!m) return DSPMAP_E_ARG;
    m->last_exp[1] = n_up;
    m->last_exp[0] = n_down;
    return DSPMAP_OK;
}

extern "C" int dspmap_mgpu_import(dspmap_t* m, int n, const float* rec_dev) {
    READY(m);
    if (n < 0 || (n > 0 && !rec_dev)) return DSPMAP_E_ARG;
    LaunchCtx c = dspmap_ctx_of(m);
    launch_import_movers(c, n, rec_dev);
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}

// the newborn children (they need the frame's birth cloud only): behind the estimator on the side stream when it runs there, so that
// the main chain never waits for it before the birth split
void dspmap_mgpu_birth_early(dspmap* m, const LaunchCtx& c) {
    if (m->mgpu_est_side) {
        LaunchCtx c2 = c;
        c2.stream = m->stream2;
        launch_birth_early(c2, m->last_n_birth, false);
        (void)hipEventRecord(m->ev_join, m->stream2);
        m->mgpu_side_pending = true;
    } else launch_birth_early(c, m->last_n_birth, false);
}
// the phase in two halves, so that the C++ driver (dspmap_dist.hip) can select the pyramid lists' GLOBAL cut between them
int dspmap_mgpu_place_phase(dspmap* m) {
    READY(m);
    if (!m->mgpu_place_pending) return DSPMAP_OK;
    LaunchCtx c = dspmap_ctx_of(m);
    if (m->vz_frames_at_begin <= 0) c.s.vz0 = nullptr;
    // a large slab: only the arrivals of tiles that can see the field of view are registered in pyramids, so only their
    // placement has to precede the weight update -- the others get their slots on the side stream, beside the pair kernels
    // AND the Ck all-reduce that follows this phase (the same split as the unsharded frame, dspmap_api.hip: enqueue_frame)
    m->mgpu_split = !m->mgpu_interior_done && c.k.ntiles >= m->place_split_tiles;
    if (m->mgpu_interior_done) launch_claim(c, 0, 2, m->mgpu_tile_lo, m->mgpu_tile_hi);
    else launch_claim(c, 0, 0, 0, 0, m->mgpu_split ? 1 : -1);
    m->mgpu_interior_done = false;
    m->mgpu_place_pending = false;
    m->mgpu_placed = true;
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
int dspmap_mgpu_ck_phase(dspmap* m) {
    READY(m);
    LaunchCtx c = dspmap_ctx_of(m);   // (carries DevState::pyr_kstar when the driver selected a global cut for this frame)
    if (m->mgpu_placed) {
        m->mgpu_placed = false;
        if (m->vz_frames_at_begin <= 0) c.s.vz0 = nullptr;
        if (m->mgpu_split) {
            // (DSPMAP_P_SIDE_PLACEMENT: the side launch leaves the main chain behind the placement of the tiles with a view -- the default --
            // or behind the list preparation; the next kernel of the main chain is queued before the side launch either way)
            const bool early = m->side_fork >= 1;
            if (early) HIPCHK(m, hipEventRecord(m->ev_fork2, m->stream));
            launch_pyr_prepare(c);
            if (!early) HIPCHK(m, hipEventRecord(m->ev_fork2, m->stream));
            auto side = [&]() -> int {
                HIPCHK(m, hipStreamWaitEvent(m->stream2, m->ev_fork2, 0));
                LaunchCtx c2 = c;
                c2.stream = m->stream2;
                launch_claim(c2, 0, 0, 0, 0, 0);
                HIPCHK(m, hipEventRecord(m->ev_join, m->stream2));
                return DSPMAP_OK;
            };
            if (early) { const int rs = side(); if (rs != DSPMAP_OK) return rs; }
            launch_ck_partial(c, true);
            if (!early) { const int rs = side(); if (rs != DSPMAP_OK) return rs; }
            m->mgpu_side_pending = true;
            HIPCHK(m, hipGetLastError());
            return DSPMAP_OK;
        }
    }
    launch_ck_partial(c);
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
extern "C" int dspmap_mgpu_ck_partial(dspmap_t* m) {
    const int rc = dspmap_mgpu_place_phase(m);
    return rc != DSPMAP_OK ? rc : dspmap_mgpu_ck_phase(m);
}

extern "C" int dspmap_mgpu_weights_and_split(dspmap_t* m) {
    READY(m);
    LaunchCtx c = dspmap_ctx_of(m);
    launch_weight_update(c);
    if (m->mgpu_side_pending) { HIPCHK(m, hipStreamWaitEvent(m->stream, m->ev_join, 0)); m->mgpu_side_pending = false; }   // the side placement
    if (m->last_n_birth > 0) launch_birth_split_cksum(c, m->last_n_birth);   // split + the 1/Ck reduction in one launch
    else launch_ck_finalize(c);
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}

extern "C" int dspmap_mgpu_finish(dspmap_t* m) {
    READY(m);
    LaunchCtx c = dspmap_ctx_of(m);
    if (m->vz_frames <= 0) c.s.vz0 = nullptr;
    if (m->mgpu_side_pending) { HIPCHK(m, hipStreamWaitEvent(m->stream, m->ev_join, 0)); m->mgpu_side_pending = false; }   // (a caller that skipped the weight phase)
    if (!m->mgpu_birth_early) launch_birth_early(c, m->last_n_birth, false);   // caller used the per-direction exports
    launch_birth_finish(c, m->last_n_birth, m->mgpu_all_static);
    m->mgpu_birth_early = false;
    dspmap_resample(m, c);
    if (m->nb_dirty) { m->nb_dirty = false; m->graph_epoch++; }
    HIPCHK(m, hipEventRecord(m->ev1, m->stream));
    m->ev_valid = true;
    HIPCHK(m, hipGetLastError());
    return DSPMAP_OK;
}
