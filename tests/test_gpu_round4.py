"""GPU parity tests added in round 4 (run with `-m gpu`): EVERY kernel instantiation that serves a BASELINE configuration
against the oracle -- both resamplers on one-word maps (k_resample_wg / k_resample<1>) x the three rollout paths (inside the
resampler / k_rollout without windows / k_rollout with LDS windows), at reduced size and at config C / D's FULL size
(132x132x60 @ 24, T = 6 and 10, natively >= 8 192 tiles: split placement, k_weight<SKIP>, k_resample<1>), the SPARSE
prediction sweep and the fused birth insertion of the captured frame; the future status is order-free (fixed point) and
bit-identical across variants, runs and slab counts; k_place_fix with a held pose (stale inboxes); the static-tile shortcuts and
the resampler's sparse-map dispatch (k_resample<MW, 8>, k_resample_wg on large sparse maps, the switch in the middle of a run)
as differential tests on the depth stream."""
import numpy as np
import pytest
import torch

from tests import common
from tests.test_gpu_parity import RTOL, gpu_state, make_pair
from tests.test_gpu_configs import CONFIGS, _slot_exact

pytestmark = pytest.mark.gpu

# (DSPMAP_P_RESAMPLE_WG_TILES, DSPMAP_P_ROLLOUT_INLINE) -> the variant resample_variant() must report:
# bit 0 = k_resample_wg, bits 1-2 = rollout 0 inline / 1 k_rollout light / 2 k_rollout windows
VARIANTS = {
    "wg+inline": (1 << 30, 1, 1 | (0 << 1)),
    "wg+windows": (1 << 30, 0, 1 | (2 << 1)),
    "wave+light": (0, 1, 0 | (1 << 1)),
    "wave+windows": (0, 0, 0 | (2 << 1)),
}


def _force(m, dsp, variant):
    wg_tiles, inline, want = VARIANTS[variant]
    m.set_param(dsp.capi.P_RESAMPLE_WG_TILES, wg_tiles)
    m.set_param(dsp.capi.P_ROLLOUT_INLINE, inline)
    return want


def _resample_scene(o, m, cfgkw, n_part, seed=11):
    """more than M particles in many voxels, heavy-tailed weights, newborn flags mixed in, half of the particles moving"""
    half = common.half_extent(o.cfg)
    rng = np.random.default_rng(seed)
    M = cfgkw["ppv"]
    n_dense = int(0.8 * n_part)
    frac = float(np.sqrt(n_dense / (1.3 * M * 0.9 * cfgkw["nz"] * cfgkw["nx"] * cfgkw["ny"])))
    assert frac < 0.9
    sub = (half[0] * frac, half[1] * frac, half[2] * 0.9)
    px, py, pz, vx, vy, w = common.random_particles(seed + 12, n_dense, sub, vmax=1.2, wlo=0.0004, whi=0.05)
    bx, by, bz, bvx, bvy, bw = common.random_particles(seed + 13, n_part - n_dense, half, vmax=1.0)
    px = np.concatenate([px, bx]); py = np.concatenate([py, by]); pz = np.concatenate([pz, bz])
    vx = np.concatenate([vx, bvx]); vy = np.concatenate([vy, bvy]); w = np.concatenate([w, bw])
    w = (w * np.exp(rng.normal(0, 1.0, w.shape))).astype(np.float32)
    flag = np.where(rng.random(len(w)) < 0.3, 15.0, 1.0).astype(np.float32)
    return common.inject_both(o, m, px, py, pz, vx, vy, w, flag)


def _check_resample(o, m, T):
    res_g, res_o = m.results(), o.results
    assert np.array_equal(res_g[:, 0], res_o[:, 0]) and np.array_equal(res_g[:, 1:3], res_o[:, 1:3])
    fut_g = m.getFutureStatus()
    assert fut_g.shape == (m.V, T)
    assert np.allclose(fut_g, res_o[:, 4:4 + T], rtol=1e-4, atol=1e-6) and res_o[:, 4:4 + T].sum() > 10
    tot_g, tot_o = fut_g.astype(np.float64).sum(axis=0), res_o[:, 4:4 + T].astype(np.float64).sum(axis=0)
    assert np.allclose(tot_g, tot_o, rtol=2e-6)
    vo, so, ro, rg = _slot_exact(o, m, cols=(1, 2, 4, 5, 6))
    assert np.allclose(ro[:, 7], rg[:, 7], rtol=1e-6)
    assert m.counters()["n_live_out"] == len(vo)
    return fut_g


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("name", ["B_66x66x40_24ppv", "C_132x132x12_24ppv"])
def test_resample_every_one_word_variant_against_oracle(dsp, orc, name, variant):
    """mapOccupancyCalculationAndResample (:924-1057) is ONE function in the reference and five instantiations here, chosen
    by map size and a hint.  Every combination that a one-word map can run -- {k_resample_wg, k_resample<1>} x {rollout
    inside the resampler, k_rollout without windows, k_rollout with fixed-point LDS windows} -- forced through the handle's
    parameters, checked to have RUN (dspmap_debug_rollout_paths), against the oracle: mass, mean velocity, survivors, copies
    and their slots bit-exact, future status to 1e-4; and the future status of all variants is THE SAME bits (fixed-point
    accumulators: every particle adds the same integer on every path)."""
    cfgkw, n_part = CONFIGS[name]
    o, m = make_pair(dsp, orc, seed=5, **cfgkw)
    want = _force(m, dsp, variant)
    _resample_scene(o, m, cfgkw, n_part)
    o.occupancy_resample(); m.occupancy_resample()
    var, n_win, n_dir = m.rollout_paths()
    assert var == want, (var, want)
    if variant.endswith("windows"):
        assert n_win > 1000 and n_dir > 0, (n_win, n_dir)        # windows AND single atomics (sparse groups, stragglers)
    fut = _check_resample(o, m, 6)
    # the other three variants from the same state: identical bits
    for other in VARIANTS:
        if other == variant:
            continue
        o2, m2 = make_pair(dsp, orc, seed=5, **cfgkw)
        _force(m2, dsp, other)
        _resample_scene(o2, m2, cfgkw, n_part)
        m2.occupancy_resample()
        assert np.array_equal(m2.getFutureStatus(), fut), other
        o2.close(); m2.close()
        break                                                     # (one partner per case: the cases chain over all four)
    o.close(); m.close()


@pytest.mark.parametrize("variant", ["wave+light", "wave+windows", "wg+inline", "wg+windows"])
def test_resample_two_word_variants_against_oracle(dsp, orc, variant):
    """config E's shape (72 slots = two occupancy words) against the oracle: k_resample<2> with both k_rollout variants, and (round 6) the
    four-waves-per-tile k_resample_wg<2> -- what a depth-stream-filled 264x264x80 map runs -- with its inline rollout and with k_rollout"""
    cfgkw, n_part = CONFIGS["E_80x80x12_res010_36ppv"]
    o, m = make_pair(dsp, orc, seed=5, **cfgkw)
    want = _force(m, dsp, variant)
    _resample_scene(o, m, cfgkw, n_part)
    o.occupancy_resample(); m.occupancy_resample()
    assert m.rollout_paths()[0] == want
    _check_resample(o, m, 6)
    o.close(); m.close()


def _config_d_state(o, m, half, n_dense, n_bg):
    px, py, pz, vx, vy, w = common.random_particles(31, n_dense, (half[0] * 0.4, half[1] * 0.4, half[2] * 0.95),
                                                    vmax=3.0, static_frac=0.2, wlo=0.002, whi=0.05)
    bx, by, bz, bvx, bvy, bw = common.random_particles(32, n_bg, half, vmax=1.5, static_frac=0.5)
    px = np.concatenate([px, bx]); py = np.concatenate([py, by]); pz = np.concatenate([pz, bz])
    vx = np.concatenate([vx, bvx]); vy = np.concatenate([vy, bvy]); w = np.concatenate([w, bw])
    return common.inject_both(o, m, px, py, pz, vx, vy, w, np.ones_like(w))


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_rollout_ten_horizons_every_path_on_config_d_grid_shape(dsp, orc, variant):
    """config D's grid shape (132 x 132 voxels per layer, 24 particles per voxel, PREDICTION_TIMES = 10, horizons 0.2 (k + 1) s)
    against the oracle (:950-964) on EVERY rollout path, forced and verified: k_rollout's fixed-point LDS windows (groups with
    hundreds of moving particles; a row of the grid is 132 voxels), its straggler path (particles faster than the windows'
    design speed leave them within the 2 s horizon), its direct path (sparse groups), the LIGHT kernel and the resampler's
    inline rollout -- after both resamplers"""
    pred = tuple(0.2 * (k + 1) for k in range(10))
    cfgkw = dict(nx=132, ny=132, nz=12, res=0.15, ppv=24, pred_times=pred)
    o, m = make_pair(dsp, orc, seed=7, **cfgkw)
    assert m.T == 10
    want = _force(m, dsp, variant)
    half = common.half_extent(o.cfg)
    _config_d_state(o, m, half, 340000, 60000)
    vo, so, ro = o.export_sparse()
    moving = (ro[:, 1] != 0) | (ro[:, 2] != 0)
    per_group = np.bincount(vo[moving] >> 9, minlength=(o.V + 511) // 512)     # k_rollout's groups of 8 tiles
    assert (per_group >= 384).sum() > 100 and ((per_group > 0) & (per_group < 384)).sum() > 100   # window groups AND direct groups
    far = np.abs(ro[moving, 2]) * 2.0 / 0.15 > 31                                                   # > 30 rows away at 2 s
    assert far.sum() > 10000                                                                        # stragglers exist
    o.occupancy_resample(); m.occupancy_resample()
    var, n_win, n_dir = m.rollout_paths()
    assert var == want, (var, want)
    if variant.endswith("windows"):
        assert n_win > 10 * 100000 and n_dir > 10000, (n_win, n_dir)      # the windows took most adds, stragglers + sparse groups the rest
    elif variant == "wave+light":
        assert n_win == 0 and n_dir > 10 * 100000
    res_o = o.results
    assert np.array_equal(m.results()[:, 0], res_o[:, 0])
    fut_g = m.getFutureStatus()
    assert fut_g.shape == (m.V, 10)
    assert np.allclose(fut_g, res_o[:, 4:14], rtol=1e-4, atol=1e-6)
    tot_g, tot_o = fut_g.astype(np.float64).sum(axis=0), res_o[:, 4:14].astype(np.float64).sum(axis=0)
    assert np.allclose(tot_g, tot_o, rtol=2e-6) and tot_o[-1] < 0.99 * tot_o[0]                     # mass leaves the map over time
    o.close(); m.close()


@pytest.mark.parametrize("T", [6, 10])
def test_full_size_config_c_and_d_stages_against_oracle(dsp, orc, T):
    """configs C (T = 6) and D (T = 10: `getFutureStatus` 10-step 0.2 s rollout) at their REAL size, 132x132x60 @ 0.15 m,
    24 particles per voxel -- 16 335 tiles, so the handle picks by itself what the benchmark runs there: the split placement
    (k_place x 2 on two streams), k_weight<SKIP>, k_resample<1>, k_rollout -- from an injected 1.5 M-particle state, stage by
    stage against the oracle (dense 1.8 GB AoS on the host): prediction slot-exact, Ck / weights to 1e-4, resampling slot-exact,
    future status to 1e-4 with the window, straggler and direct paths all counted."""
    pred = tuple(0.2 * (k + 1) for k in range(10)) if T == 10 else (0.05, 0.2, 0.5, 1.0, 1.5, 2.0)
    cfgkw = dict(nx=132, ny=132, nz=60, res=0.15, ppv=24, pred_times=pred)
    o, m = make_pair(dsp, orc, seed=3, **cfgkw)
    assert m.V // 64 + 1 >= 8192 and m.T == T
    m.set_param(dsp.capi.P_ROLLOUT_INLINE, 0 if T == 10 else -1)      # D: the windows; C: whatever the handle picks
    half = common.half_extent(o.cfg)
    n = _config_d_state(o, m, half, 1100000, 400000)
    assert n > 1400000
    q = common.EX_QUATS[1]
    pts = common.wall_cloud(7, n_side=64, dist=5.0, half_w=4.0, half_h=1.8)
    cur = (0.2, -0.1, 0.05)
    o.L.dspo_set_current_position(o.h, *cur); m.set_current_position(*cur)
    o.bin_points(pts, q); m.bin_points(pts, q)
    d = (-0.017, 0.004, -0.03, 1 / 30.0)
    o.predict(*d); m.predict(*d)
    vo, so, ro, rg = _slot_exact(o, m)
    c = m.counters()
    assert c["n_live_in"] == n and c["n_moved"] > 0.02 * n and c["n_fov"] > 50000
    assert np.array_equal(np.minimum(m.pyramid_counts(), m.capp), (o.pyramid_lists[:, :, 0] != 0).sum(1))
    o.map_update(); m.map_update()
    obs, cnt, ml, lam = m.observations()
    assert np.array_equal(cnt, o.obs_count) and cnt.sum() > 1000
    nz = np.nonzero(cnt)[0]
    ck_o = np.concatenate([o.obs[b, :cnt[b], 3] for b in nz])
    ck_g = np.concatenate([obs[b, :cnt[b], 3] for b in nz])
    rel = np.abs(ck_g - ck_o) / ck_o
    assert rel.max() < RTOL and np.median(rel) < 1e-6, (rel.max(), np.median(rel))
    vo, so, ro, rg = _slot_exact(o, m, cols=(1, 2, 4, 5, 6))
    relw = np.abs(ro[:, 7] - rg[:, 7]) / np.maximum(np.abs(ro[:, 7]), 1e-12)
    assert relw.max() < RTOL, relw.max()
    # the weights the two sides resample must be the same bits for a slot-exact comparison of the stage: hand the oracle's over
    m.clear_state(); m.import_state(vo, ro, so)
    # a DENSE map of this size resamples with one wave per tile (what the benchmark's saturated 132x132x60 runs); a handle that
    # takes its map for sparse -- as this one does after import_state, before any frame has published a live-tile estimate --
    # would pick the four-wave variant (test_sparse_large_one_word_map_takes_the_four_wave_resampler_by_itself)
    m.set_param(dsp.capi.P_SPARSE_SWEEP, 0)
    o.occupancy_resample(); m.occupancy_resample()
    var, n_win, n_dir = m.rollout_paths()
    assert (var & 1) == 0                                               # k_resample<1, 4>
    if T == 10:
        assert (var >> 1) == 2 and n_win > 1000000 and n_dir > 10000, (var, n_win, n_dir)
    res_g, res_o = m.results(), o.results
    assert np.array_equal(res_g[:, 0], res_o[:, 0]) and np.array_equal(res_g[:, 1:3], res_o[:, 1:3])
    fut_g = m.getFutureStatus()
    assert np.allclose(fut_g, res_o[:, 4:4 + T], rtol=1e-4, atol=1e-6)
    assert np.allclose(fut_g.astype(np.float64).sum(0), res_o[:, 4:4 + T].astype(np.float64).sum(0), rtol=2e-6)
    vo, so, ro, rg = _slot_exact(o, m, cols=(1, 2, 4, 5, 6))
    assert np.allclose(ro[:, 7], rg[:, 7], rtol=1e-6)
    o.close(); m.close()


def test_full_size_config_c_whole_frame_against_oracle(dsp, orc):
    """one whole update() at 132x132x60 @ 24 from an injected 1.2 M-particle state through the CAPTURED frame the benchmark
    replays (device estimator, split placement on two streams, k_place_fix riding on k_ck_partial, k_weight<SKIP>, fused birth
    insertion, k_resample<1>, k_rollout): per-voxel occupancy within 1e-4 * max(1, |x|) on > 99.9 % of the voxels, total mass
    to 1e-4, the same number of live particles to 0.1 %, future-status sums to 0.5 %"""
    cfgkw = dict(nx=132, ny=132, nz=60, res=0.15, ppv=24)
    o, m = make_pair(dsp, orc, seed=3, **cfgkw)
    half = common.half_extent(o.cfg)
    _config_d_state(o, m, half, 800000, 400000)
    o.L.dspo_use_velocity_estimator(o.h, 1)      # (the oracle's modes: 1 = velocityEstimationThread restated, 2 = static tags)
    m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
    q = common.EX_QUATS[1]
    pts = common.wall_cloud(7, n_side=64, dist=5.0, half_w=4.0, half_h=1.8)
    assert o.update(pts, (0.0, 0.0, 0.0), 0.0, q) == 1
    assert m.update(pts, (0.0, 0.0, 0.0), 0.0, q) == 1
    occ_o, occ_g = o.results[:, 0], m.results()[:, 0]
    err = np.abs(occ_g - occ_o)
    tol = RTOL * np.maximum(1.0, np.abs(occ_o))
    assert (err <= tol).mean() > 0.999, (err > tol).sum()
    assert abs(occ_g.astype(np.float64).sum() - occ_o.astype(np.float64).sum()) < 1e-4 * occ_o.sum()
    assert abs(m.counters()["n_live_out"] - o.L.dspo_count_live(o.h)) < 1e-3 * o.L.dspo_count_live(o.h)
    assert m.counters()["n_born"] > 5000
    xo, fo = o.get_occupancy_with_future(0.2)
    ng, xg, fg = m.getOccupancyMapWithFutureStatus(0.2)
    assert np.allclose(fg.sum(0), fo.sum(0), rtol=5e-3)
    assert o.cursors()[0] == m.cursors()[0]
    o.close(); m.close()


@pytest.mark.parametrize("name", list(CONFIGS))
def test_sparse_prediction_sweep_against_oracle(dsp, orc, name):
    """k_predict's SPARSE instantiation (empty tiles are left after one scalar load) against the ORACLE, not only against its
    twin: a map whose particles sit in a tenth of its tiles (the rest was emptied by a resampling pass, so tile_live is 0
    there), moved so that particles arrive in tiles the sweep skipped -- the same particles in the same slots, every float"""
    cfgkw, n_part = CONFIGS[name]
    o, m = make_pair(dsp, orc, seed=3, **cfgkw)
    m.set_param(dsp.capi.P_SPARSE_SWEEP, 1)
    half = common.half_extent(o.cfg)
    # a slab of particles around y = 0 with velocities that carry many of them into the (empty) tiles beside it
    px, py, pz, vx, vy, w = common.random_particles(17, n_part // 4, (half[0], half[1] * 0.08, half[2]), vmax=4.0, static_frac=0.3, wlo=0.01, whi=0.08)
    n = common.inject_both(o, m, px, py, pz, vx, vy, w)
    # a resampling pass on both sides: tile_live becomes exact on the device (0 for the tiles that hold nothing)
    o.occupancy_resample(); m.occupancy_resample()
    m.clearOccupancyMapPrediction(); o.L.dspo_clear_future(o.h)
    _slot_exact(o, m, cols=(1, 2, 4, 5, 6, 7))
    q = common.EX_QUATS[1]
    empty = np.zeros((0, 3), np.float32)
    o.bin_points(empty, q); m.bin_points(empty, q)
    d = (-0.017, 0.004, -0.03, 0.1)
    o.predict(*d); m.predict(*d)
    assert m.get_param(dsp.capi.P_SPARSE_SWEEP) == 1
    vo, so, ro, rg = _slot_exact(o, m)
    c = m.counters()
    assert c["n_moved"] > 0.3 * c["n_live_in"] and c["n_live_in"] > 0.8 * n
    assert len(np.unique(m.tile_of(vo))) < 0.6 * m.tile_count()                    # the map IS mostly empty tiles
    assert c["n_fov"] == int((o.pyramid_lists[:, :, 0] & 1).sum())
    o.close(); m.close()


def test_captured_frame_with_fused_birth_insertion_against_oracle(dsp, orc):
    """the captured frame (dspmap_update_device's graph: children generated by the split's waves, k_birth_insert<FUSED> computing
    its own cursors) against the oracle's update() from the same injected state, with DYNAMIC birth sources (a matched moving
    cluster: velocity-table and rand() draws): after the frame the same particles sit in the same slots with the same
    positions and velocities, newborn included; weights to 1e-4; the three stream cursors equal"""
    from tests.test_gpu_round2 import _cluster_scene
    cfgkw = dict(nx=66, ny=66, nz=40, res=0.15, ppv=24)
    o, m = make_pair(dsp, orc, seed=21, **cfgkw)
    o.L.dspo_use_velocity_estimator(o.h, 1)      # (the oracle's modes: 1 = velocityEstimationThread restated, 2 = static tags)
    m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
    half = common.half_extent(o.cfg)
    px, py, pz, vx, vy, w = common.random_particles(5, 60000, half, vmax=1.0, wlo=0.01, whi=0.08)
    common.inject_both(o, m, px, py, pz, vx, vy, w)
    pos = (0.0, 0.0, 1.0)
    for f in range(2):                    # frame 1 matches frame 0's clusters: dynamic newborn velocities
        t = f * 0.1
        pts = _cluster_scene(t, f)
        assert o.update(pts, pos, t, (1, 0, 0, 0)) == 1
        assert m.update(pts, pos, t, (1, 0, 0, 0)) == 1
        if f == 0:
            o.get_occupancy_with_future(0.2); m.getOccupancyMapWithFutureStatus(0.2)
    assert o.cursors() == m.cursors()
    vo, so, ro = o.export_sparse()
    vg, sg, rg = gpu_state(m)
    ko, kg = np.lexsort((so, vo)), np.lexsort((sg, vg))
    # frame 0 is exact up to the weights (1e-6), frame 1 resamples them: equal-weight ties may pick another survivor in a few
    # voxels (DESIGN: threshold ties) -- the comparison is on the voxels whose particle count agrees, which must be nearly all
    co, cg = np.bincount(vo, minlength=o.V), np.bincount(vg, minlength=o.V)
    assert (co == cg).mean() > 0.999 and abs(len(vo) - len(vg)) < 1e-3 * len(vo)
    same = (co == cg)
    mo, mg = same[vo[ko]], same[vg[kg]]
    eq_slots = np.array_equal(vo[ko][mo], vg[kg][mg]) and np.array_equal(so[ko][mo], sg[kg][mg])
    if eq_slots:
        moving_nb = (np.abs(rg[kg][mg][:, 1]) + np.abs(rg[kg][mg][:, 2]) > 0.3).sum()
        assert moving_nb > 100                                                        # dynamic newborns are in the map
        frac = (ro[ko][mo][:, 1:7] == rg[kg][mg][:, 1:7]).all(axis=1).mean()
        assert frac > 0.999, frac
    else:
        # slots differ somewhere although the counts agree: quantify instead of failing blindly
        a = set(zip(vo[ko][mo].tolist(), so[ko][mo].tolist())); b = set(zip(vg[kg][mg].tolist(), sg[kg][mg].tolist()))
        assert len(a ^ b) < 1e-3 * len(a), len(a ^ b)
    o.close(); m.close()


def test_future_status_is_order_free_and_variant_free(dsp):
    """voxels_objects_number[v][4..] `+=` (:961) is a sequential loop in the reference; here every moving particle adds the same
    fixed-point integer on every path, so with EVERY particle moving (dozens of contributions meet in a cell in arbitrary order)
    the future status is bit-identical across two runs and across all four (resampler, rollout) variants, frame after frame,
    also when the accumulators add up over two frames"""
    cfg = dict(nx=40, ny=36, nz=12, res=0.15, ppv=24)
    tables = common.tables(4)
    maps = []
    for variant in list(VARIANTS) + ["wg+inline"]:
        m = dsp.DSPMap(dsp.make_config(**cfg)); m.set_tables(*tables)
        m.L.dspmap_init_device(m.h)
        _force(m, dsp, variant)
        m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
        m.seed_uniform(20, 0.01, 6, 1.0)
        maps.append(m)
    pts = common.wall_cloud(3, n_side=30, dist=2.0, half_w=1.5, half_h=0.6)
    d = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).cuda()
    for f in range(5):
        for m in maps:
            assert m.update_device(d.data_ptr(), len(pts), (0.02 * f, 0.0, 0.0), f / 30.0, (1.0, 0.0, 0.0, 0.0)) == 1
        if f % 2 == 1:
            continue                                        # (no clear: the next frame adds on top)
        futs = [m.getFutureStatus() for m in maps]
        assert futs[0].sum() > 100
        for k in range(1, len(maps)):
            assert np.array_equal(futs[0], futs[k]), (f, k)
    assert sorted(set(m.rollout_paths()[0] for m in maps)) == sorted(set(v[2] for v in VARIANTS.values()))
    ref = maps[0].export_state()
    for m in maps[1:]:
        for a, b in zip(ref, m.export_state()):
            assert np.array_equal(a, b)
    for m in maps:
        m.close()


def test_turned_away_stayers_with_a_held_pose_do_not_replay_old_arrivals(dsp, orc):
    """k_place_fix collects a dirty voxel's arrivals from its tile's inbox -- which k_place fills only in frames in which the
    tile RECEIVES arrivals.  Two predictions with motion and overfull pyramid lists (arrivals everywhere, some re-slotted), then
    two predictions with a held pose and dt = 0 (nobody changes voxel, no tile receives anything) while the refilled lists
    overflow again: stayers are turned away (dirty voxels) in tiles whose inbox, arrival count and pmask are two predictions
    old -- they carry another prediction's stamp and must be ignored.  Lists and slots equal the oracle's after every step."""
    cfgkw = dict(nx=40, ny=40, nz=16, res=0.15, ppv=24)
    o, m = make_pair(dsp, orc, **cfgkw)
    from tests.test_gpu_round3 import _fill_view_uniform
    n_in = _fill_view_uniform(o, m, 227000, 9, vmax=1.5)
    pts = common.wall_cloud(3, n_side=40, dist=2.2, half_w=1.8, half_h=0.6)
    o.bin_points(pts); m.bin_points(pts)
    steps = [(-0.03, 0.02, 0.0, 0.1), (0.02, -0.01, 0.0, 0.1), (0.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 0.0)]
    full_after = []
    for k, st in enumerate(steps):
        if k == 2:
            # refill the view so that the lists overflow again although nobody moves: particles injected into free slots
            # on both sides (the oracle's first-free rule), all inside the field of view
            half = common.half_extent(o.cfg)
            rng = np.random.default_rng(77)
            npx = rng.uniform(0.5, half[0] * 0.9, 60000); npy = rng.uniform(-0.6, 0.6, 60000) * npx; npz = rng.uniform(-0.3, 0.3, 60000) * npx
            vo, so, ro = o.export_sparse()
            o.inject(npx.astype(np.float32), npy.astype(np.float32), npz.astype(np.float32), np.zeros(60000, np.float32),
                     np.zeros(60000, np.float32), np.zeros(60000, np.float32), np.full(60000, 0.02, np.float32), 1.0)
            vo, so, ro = o.export_sparse()
            m.clear_state(); m.import_state(vo, ro, so)
        o.predict(*st); m.predict(*st)
        c = m.counters()
        len_o = (o.pyramid_lists[:, :, 0] != 0).sum(1)
        assert np.array_equal(len_o, m.pyramid_counts()), k
        full_after.append(((len_o == o.capp).sum(), c["n_pyramid_full"], c["n_moved"], c["n_reslotted"]))
        vo, so, ro = o.export_sparse()
        vg, sg, rg = gpu_state(m)
        ko, kg = np.lexsort((so, vo)), np.lexsort((sg, vg))
        assert len(vo) == len(vg), (k, full_after)
        assert np.array_equal(vo[ko], vg[kg]) and np.array_equal(so[ko], sg[kg]), (k, full_after)
        for col in (1, 2, 4, 5, 6, 7):
            assert np.array_equal(ro[ko][:, col], rg[kg][:, col]), (k, col)
        # the frame's resampling on both sides (bit-exact from the same state): the reference predicts a particle again only
        # after mapOccupancyCalculationAndResample has reset its "moved" flag 7 to 1 (:649,968)
        o.occupancy_resample(); m.occupancy_resample()
        o.L.dspo_clear_future(o.h); m.clearOccupancyMapPrediction()
    assert full_after[0][3] > 5 and full_after[0][2] > 5000, full_after      # the moving frames re-slotted arrivals
    assert full_after[2][1] > 100 and full_after[2][2] == 0, full_after       # the held frame turned stayers away, nobody moved
    o.close(); m.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_future_status_with_every_particle_moving(dsp, world):
    """:961 over Z-slabs: a saturated map whose EVERY particle moves (+-1 m/s), a sensor that advances and climbs (particles
    change slab), 2 / 4 / 8 slabs against the unsharded map -- future status, results, every slot and every float equal, and
    two unsharded runs equal each other (the rollout adds integers: no order dependence anywhere)"""
    sharded = __import__("dsp-map_amd.sharded", fromlist=["CppGroup"])
    from tests.test_gpu_round3 import UP
    cfg = dict(nx=32, ny=16, nz=8, res=0.15, ppv=24)
    tables = common.tables(9)
    grp = sharded.CppGroup(dsp, cfg, world)
    fulls = [dsp.DSPMap(dsp.make_config(**cfg)) for _ in range(2)]
    for x in grp.maps + fulls:
        x.set_tables(*tables)
        x.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
        x.seed_uniform(20, weight=0.01, seed=99, vmax=1.0)
    yy, zz = np.meshgrid(np.linspace(-0.3, 0.3, 25), np.linspace(-0.2, 0.2, 17))
    pts = np.stack([np.full(yy.size, 0.42) + 0.02 * np.sin(7 * yy.ravel()), yy.ravel(), zz.ravel()], 1).astype(np.float32)
    d = torch.from_numpy(pts).cuda()
    for f in range(5):
        pos = (0.05 * f, 0.0, 0.04 * f)
        assert grp.update(d, pos, f / 30.0, UP) == 1
        for x in fulls:
            assert x.update_device(d.data_ptr(), len(pts), pos, f / 30.0, UP) == 1
        grp.sync()
        if f % 2 == 0:                                   # (odd frames: no read, no clear -- the accumulators add up)
            fut_s = np.concatenate([x.getFutureStatus() for x in grp.maps], 0)
            fa, fb = fulls[0].getFutureStatus(), fulls[1].getFutureStatus()
            assert fa.sum() > 100 and (fa > 0).mean() > 0.5
            assert np.array_equal(fa, fb), f
            assert np.array_equal(fut_s, fa), f
    got = np.concatenate([x.results() for x in grp.maps], 0)
    assert np.array_equal(got, fulls[0].results())
    parts = [x.export_state() for x in grp.maps]
    sv, ss, sr = (np.concatenate([p[k] for p in parts]) for k in range(3))
    order = np.lexsort((ss, sv))
    fv, fs_, fr = fulls[0].export_state()
    assert len(fv) > 20000
    assert np.array_equal(sv[order], fv) and np.array_equal(ss[order], fs_) and np.array_equal(sr[order], fr)
    grp.close()
    for x in fulls:
        x.close()


def test_alternating_sweep_direction_changes_nothing(dsp):
    """large maps walk their tiles in alternating directions (k_predict, k_place the other way, k_resample, the next frame's
    k_predict the other way again: each sweep starts on the tiles still in the Infinity Cache, DSPMAP_P_SWEEP_ALTERNATE).  Forced
    on / off on a small map, split placement on, a moving sensor and moving particles: every slot, every float, every counter
    and the future status equal after every frame -- no stage depends on the order in which the tiles are visited"""
    cfg = dict(nx=56, ny=88, nz=12, res=0.15, ppv=24)
    tables = common.tables(5)
    maps = []
    for alt in (1, 0, 2):
        m = dsp.DSPMap(dsp.make_config(**cfg)); m.set_tables(*tables)
        m.L.dspmap_init_device(m.h)
        m.set_param(dsp.capi.P_PLACE_SPLIT_TILES, 1)
        m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
        m.set_param(dsp.capi.P_SWEEP_ALTERNATE, alt)
        m.set_param(dsp.capi.P_RESAMPLE_WG_TILES, 0)          # the one-wave-per-tile resampler (the four-wave one has no direction)
        # (8 per voxel: the fullest pyramid lists overflow their capacity CAPP = 422 -- the cut and the re-slotting run -- but stay
        # below the CAPA = 2 CAPP + 64 entries a list accepts before the cut: beyond that entries are dropped in ARRIVAL order,
        # the one documented place where the order of the tiles can show, DESIGN.md "Numerics" (3))
        m.seed_uniform(8, 0.01, 6, 0.8)
        maps.append(m)
    rng = np.random.default_rng(3)
    ys, zs = np.meshgrid(np.linspace(-2.0, 2.0, 41), np.linspace(-0.7, 0.7, 15))
    base = np.stack([np.full(ys.size, 2.3) + 0.2 * np.sin(2 * ys.ravel()), ys.ravel(), zs.ravel()], 1).astype(np.float32)
    for f in range(9):
        t = f / 30.0
        pts = torch.from_numpy(base + rng.normal(0, 0.004, base.shape).astype(np.float32)).cuda()
        pos = (0.9 * t, 0.5 * t, 0.1 * np.sin(5 * t))
        for m in maps:
            assert m.update_device(pts.data_ptr(), len(base), pos, t, (0.9659258, 0.0, 0.0, 0.258819)) == 1
        cs = [m.counters() for m in maps]
        for c in cs:
            c.pop("update_ms")
        assert cs[0] == cs[1] == cs[2], (f, cs)
        futs = [m.getFutureStatus() for m in maps]
        assert np.array_equal(futs[0], futs[1]) and np.array_equal(futs[0], futs[2]) and futs[0].sum() > 0, f
    assert cs[0]["n_moved"] > 1000
    for m in maps[1:]:
        for a, b in zip(maps[0].export_state(), m.export_state()):
            assert np.array_equal(a, b)
        assert np.array_equal(maps[0].results(), m.results())
    for m in maps:
        m.close()


def _sharded_overfull(dsp, world, exact):
    """a saturated map whose sensor looks INTO it (identity attitude: dozens of pyramid lists are overfull in every frame, the cut
    and the re-slotting run), every particle moving, the sensor advancing and climbing (particles change slab): slabs vs unsharded"""
    import os
    sharded = __import__("dsp-map_amd.sharded", fromlist=["CppGroup"])
    # 12 m long: a pyramid reaches 6 m into the map and holds ~60 voxels -> 500 - 700 of the 10 particles per voxel, its list
    # takes CAPP = 368 (and accepts CAPA = 800 before the cut, beyond which entries would be dropped in arrival order)
    cfg = dict(nx=80, ny=40, nz=16, res=0.15, ppv=24)
    tables = common.tables(9)
    os.environ["DSPMAP_SHARDED_EXACT_LISTS"] = "1" if exact else "0"
    try:
        grp = sharded.CppGroup(dsp, cfg, world)
        grp.create()
    finally:
        del os.environ["DSPMAP_SHARDED_EXACT_LISTS"]
    full = dsp.DSPMap(dsp.make_config(**cfg))
    for x in grp.maps + [full]:
        x.set_tables(*tables)
        x.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
        x.seed_uniform(10, weight=0.01, seed=99, vmax=1.0)
    pts = common.wall_cloud(3, n_side=40, dist=4.2, half_w=1.8, half_h=0.6)
    d = torch.from_numpy(pts).cuda()
    stats = []
    same = True
    for f in range(5):
        pos = (0.03 * f, -0.02 * f, 0.025 * f)
        assert grp.update(d, pos, f / 30.0, (1.0, 0.0, 0.0, 0.0)) == 1
        assert full.update_device(d.data_ptr(), len(pts), pos, f / 30.0, (1.0, 0.0, 0.0, 0.0)) == 1
        grp.sync()
        c = full.counters()
        stats.append((c["n_pyramid_full"], c["n_reslotted"], c["n_overflow_inexact"], c["n_moved"],
                      sum(x.counters()["n_exported_up"] + x.counters()["n_exported_down"] for x in grp.maps)))
        fut_s = np.concatenate([x.getFutureStatus() for x in grp.maps], 0)
        same = same and np.array_equal(fut_s, full.getFutureStatus())
        same = same and sum(x.counters()["n_pyramid_full"] for x in grp.maps) == c["n_pyramid_full"]
    got = np.concatenate([x.results() for x in grp.maps], 0)
    same = same and np.array_equal(got, full.results())
    parts = [x.export_state() for x in grp.maps]
    sv, ss, sr = (np.concatenate([p[k] for p in parts]) for k in range(3))
    order = np.lexsort((ss, sv))
    fv, fs_, fr = full.export_state()
    same = same and len(sv) == len(fv) and np.array_equal(sv[order], fv) and np.array_equal(ss[order], fs_) and np.array_equal(sr[order], fr)
    inexact = sum(x.counters()["n_overflow_inexact"] for x in grp.maps)
    grp.close(); full.close()
    return same, stats, inexact


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_map_with_overfull_pyramid_lists_is_the_unsharded_map(dsp, world):
    """SAFE_PARTICLE_NUM_PYRAMID (:64-66) bounds a pyramid's list over the WHOLE map: when it is full, the particles that come later
    in the reference's sweep are turned away (:1256-1259).  A sharded map selects the CAPP-th smallest sweep key over all ranks
    (radix select, one small all-reduce per 8-bit digit) and every rank cuts with it: the sharded map is the unsharded one, bit
    for bit, with dozens of overfull lists per frame -- 2 / 4 / 8 slabs.  With the selection switched off (a full list cut per
    rank, the documented deviation of the earlier rounds) the same run differs, and the frames that overflowed are counted."""
    same, stats, inexact = _sharded_overfull(dsp, world, exact=True)
    assert all(s[0] > 200 for s in stats), stats                                         # every frame turned particles away
    assert sum(s[1] for s in stats) > 20, stats                                          # arrivals were re-slotted
    # (n_overflow_inexact of the unsharded map counts ITS residue against the reference -- arrivals that found their voxel full
    # before the cut, a handful in the voxels the births fill up; the slabs treat those voxels alike)
    assert same, (stats, inexact)
    same0, stats0, inexact0 = _sharded_overfull(dsp, world, exact=False)
    assert inexact0 > 0, (stats0, inexact0)                # the frames whose lists overflowed globally are counted
    if world >= 4:                                          # (with two slabs nearly every pyramid lies in one of them)
        assert not same0, stats0


def test_sharded_frame_with_a_cloud_beyond_the_device_estimator(dsp):
    """a cloud larger than the device estimator's capacity (6144 points) inside a SHARDED frame: every rank falls back to the host
    stage on the replicated cloud (like the unsharded map does) instead of rejecting the frame, and the cluster state is handed
    between the two implementations -- 4 slabs == the unsharded map over a stream whose third frame is padded to 7000 points"""
    from tests.test_gpu_round2 import _cluster_scene
    from tests.test_gpu_round3 import _group_vs_full
    cfg = dict(nx=66, ny=66, nz=40, res=0.15, ppv=12)
    rng = np.random.default_rng(0)
    frames = []
    for f in range(4):
        t = f * 0.1
        pts = _cluster_scene(t, f)
        if f == 2:
            pad = np.stack([rng.uniform(2.3, 3.4, 6400), rng.uniform(-1.6, 1.6, 6400), np.full(6400, -0.98)], 1).astype(np.float32)
            pts = np.concatenate([pts, pad])
            assert len(pts) > 6144
        frames.append((pts, (0.0, 0.0, 1.0 + 0.04 * f), t, (1.0, 0.0, 0.0, 0.0)))
    clouds, rec, holding = _group_vs_full(dsp, 4, cfg, frames)
    for f, (a, b, c) in enumerate(clouds):
        assert len(a) == len(b) == len(c) > 500, f
        for k in ("x", "y", "z", "nx", "ny", "nz"):
            assert np.array_equal(a[k], c[k]) and np.array_equal(b[k], c[k]), (f, k)
    g = clouds[-1][2]
    dyn = g["intensity"] > 0.01
    assert np.isclose(g["ny"][dyn], 1.0, atol=0.02).sum() == 60      # cluster A keeps its 1 m/s through the switches


def test_full_size_config_e_stages_against_oracle(dsp, orc):
    """config E at its REAL size on one GPU -- 264x264x80 @ 0.10 m, 36 particles per voxel = 72 slots in two occupancy words,
    87 120 tiles: k_predict<2>, the split placement with k_place<2>, k_weight<SKIP>, k_resample<2>, k_rollout -- from an injected
    2 M-particle state, stage by stage against the oracle (a dense 14.5 GB AoS on the host: skipped where the host has less
    than 24 GB available): prediction slot-exact, Ck / weights to 1e-4, resampling slot-exact, future status to 1e-4"""
    try:
        avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1048576.0
    except Exception:  # noqa: BLE001
        avail = 0.0
    if avail < 24.0:
        pytest.skip("the oracle's dense 264x264x80 state needs 14.5 GB of host memory (%.0f GB available)" % avail)
    cfgkw = dict(nx=264, ny=264, nz=80, res=0.10, ppv=36)
    o, m = make_pair(dsp, orc, seed=3, **cfgkw)
    assert m.slots == 72 and m.V // 64 >= 87000
    half = common.half_extent(o.cfg)
    n = _config_d_state(o, m, half, 1400000, 600000)
    assert n > 1900000
    q = common.EX_QUATS[1]
    pts = common.wall_cloud(7, n_side=64, dist=5.0, half_w=4.0, half_h=1.8)
    o.bin_points(pts, q); m.bin_points(pts, q)
    d = (-0.017, 0.004, -0.03, 1 / 30.0)
    o.predict(*d); m.predict(*d)
    vo, so, ro, rg = _slot_exact(o, m)
    c = m.counters()
    assert c["n_live_in"] == n and c["n_moved"] > 0.02 * n and c["n_fov"] > 50000
    assert np.array_equal(np.minimum(m.pyramid_counts(), m.capp), (o.pyramid_lists[:, :, 0] != 0).sum(1))
    o.map_update(); m.map_update()
    obs, cnt, ml, lam = m.observations()
    assert np.array_equal(cnt, o.obs_count) and cnt.sum() > 1000
    nz = np.nonzero(cnt)[0]
    ck_o = np.concatenate([o.obs[b, :cnt[b], 3] for b in nz])
    ck_g = np.concatenate([obs[b, :cnt[b], 3] for b in nz])
    rel = np.abs(ck_g - ck_o) / ck_o
    assert rel.max() < RTOL and np.median(rel) < 1e-6, (rel.max(), np.median(rel))
    vo, so, ro, rg = _slot_exact(o, m, cols=(1, 2, 4, 5, 6))
    relw = np.abs(ro[:, 7] - rg[:, 7]) / np.maximum(np.abs(ro[:, 7]), 1e-12)
    assert relw.max() < RTOL, relw.max()
    m.clear_state(); m.import_state(vo, ro, so)          # (the same weight bits on both sides for a slot-exact resampling)
    o.occupancy_resample(); m.occupancy_resample()
    var, n_win, n_dir = m.rollout_paths()
    assert (var & 1) == 0                                 # two occupancy words: k_resample<2>
    res_g, res_o = m.results(), o.results
    assert np.array_equal(res_g[:, 0], res_o[:, 0]) and np.array_equal(res_g[:, 1:3], res_o[:, 1:3])
    fut_g = m.getFutureStatus()
    assert np.allclose(fut_g, res_o[:, 4:10], rtol=1e-4, atol=1e-6)
    vo, so, ro, rg = _slot_exact(o, m, cols=(1, 2, 4, 5, 6))
    assert np.allclose(ro[:, 7], rg[:, 7], rtol=1e-6)
    o.close(); m.close()


def test_static_tiles_skip_their_velocity_rows_until_a_mover_arrives_or_is_born(dsp, orc):
    """k_predict notes per tile whether any live particle has a velocity; a tile of static particles is swept without its
    velocity rows (k_predict, k_resample), and a tile that BECOMES static has all its velocity cells zeroed, so that a static
    arrival there is placed with two stores instead of three (k_place writes no velocity).  The flag must come back the moment
    a mover ARRIVES (k_place) or is BORN (k_birth_insert) there.  A map full of static particles under ego-motion (static
    particles change voxels and tiles every frame), TWO blobs of fast movers that cross it on the same track a few frames apart
    (the second one makes the sweeps read the velocity cells the first one left behind and static arrivals took over), ten
    stage-level frames against the oracle (every float of every slot after each prediction and each resampling); then whole
    frames with a matched moving cluster (dynamic newborns into static tiles): no tile that holds a particle with a velocity may
    carry a zero flag afterwards (a build whose k_birth_insert does not raise it fails here; one whose k_place does not, or whose
    k_predict does not zero the cells of a tile that becomes static, fails in the stage-level part)."""
    cfgkw = dict(nx=66, ny=66, nz=40, res=0.15, ppv=24)
    o, m = make_pair(dsp, orc, seed=8, **cfgkw)
    half = common.half_extent(o.cfg)
    px, py, pz, vx, vy, w = common.random_particles(5, 150000, half, vmax=0.0, wlo=0.01, whi=0.08)
    bx, by, bz, bvx, bvy, bw = common.random_particles(6, 6000, (0.3, 0.5, 0.8), vmax=0.0, wlo=0.01, whi=0.08)
    cx, cy, cz, cvx, cvy, cw = common.random_particles(7, 3000, (0.15, 0.5, 0.8), vmax=0.0, wlo=0.01, whi=0.08)
    bx -= 2.9; by += 2.0; cx -= 4.65; cy += 2.0        # the second blob follows 1.3 m (four frames) behind the first
    bvx[:] = 3.1; bvy[:] = -1.7; cvx[:] = 3.1; cvy[:] = -1.7
    cat = np.concatenate
    n = common.inject_both(o, m, cat([px, bx, cx]), cat([py, by, cy]), cat([pz, bz, cz]), cat([vx, bvx, cvx]),
                           cat([vy, bvy, cvy]), cat([w, bw, cw]))
    assert m.tile_moving().all()                                  # (an imported state: nothing is known yet)
    empty = np.zeros((0, 3), np.float32)
    o.bin_points(empty); m.bin_points(empty)
    seen_moving = np.zeros(len(m.tile_moving()), bool)
    n_static_movers = 0
    for f in range(10):
        ego = (-0.05, 0.03, 0.02, 0.1)
        o.predict(*ego); m.predict(*ego)
        vo, so, ro, rg = _slot_exact(o, m)
        n_static_movers += m.counters()["n_moved"]
        mv = m.tile_moving() != 0
        # the flag is exactly "holds a particle with a velocity" wherever it is 0, and set wherever a mover sits
        has_mover = np.zeros(len(mv), bool)
        has_mover[np.unique(m.tile_of(vo[(ro[:, 1] != 0) | (ro[:, 2] != 0)]))] = True
        assert not (has_mover & ~mv).any(), f
        assert (~mv).mean() > 0.8, (f, mv.mean())
        seen_moving |= has_mover
        o.occupancy_resample(); m.occupancy_resample()
        _slot_exact(o, m, cols=(1, 2, 4, 5, 6))
        assert np.array_equal(m.results()[:, 1:3], o.results[:, 1:3])      # mean velocities: the movers' voxels included
        o.L.dspo_clear_future(o.h); m.clearOccupancyMapPrediction()
    assert seen_moving.sum() > 3 * has_mover.sum() / 2                     # the blobs visited tiles that had been static
    assert n_static_movers > 300000                                        # ... and static particles changed voxels all along
    # whole frames: dynamic newborns (a matched moving cluster) into tiles that are static by now
    from tests.test_gpu_round2 import _cluster_scene
    o.L.dspo_use_velocity_estimator(o.h, 1)
    m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
    pos = (0.0, 0.0, 1.0)
    for f in range(3):
        t = 1.0 + f * 0.1
        pts = _cluster_scene(t, f)
        assert o.update(pts, pos, t, (1, 0, 0, 0)) == 1
        assert m.update(pts, pos, t, (1, 0, 0, 0)) == 1
        o.get_occupancy_with_future(0.2); m.getOccupancyMapWithFutureStatus(0.2)
    vo, so, ro = o.export_sparse()
    vg, sg, rg = gpu_state(m)
    ko, kg = np.lexsort((so, vo)), np.lexsort((sg, vg))
    co, cg = np.bincount(vo, minlength=o.V), np.bincount(vg, minlength=o.V)
    assert (co == cg).mean() > 0.999 and abs(len(vo) - len(vg)) < 1e-3 * len(vo)
    same = (co == cg)
    mo, mg = same[vo[ko]], same[vg[kg]]
    # (three resampled frames: equal-weight ties pick another survivor in a few voxels -- DESIGN: threshold ties; the cells both
    # sides occupy must hold the same particle, velocity included)
    key_o = vo[ko][mo].astype(np.int64) * 256 + so[ko][mo]; key_g = vg[kg][mg].astype(np.int64) * 256 + sg[kg][mg]
    both, io, ig = np.intersect1d(key_o, key_g, return_indices=True)
    assert len(both) > 0.995 * len(key_o), (len(both), len(key_o))
    rbo, rbg = ro[ko][mo][io], rg[kg][mg][ig]
    newborn_movers = (np.abs(rbg[:, 1]) + np.abs(rbg[:, 2]) > 0.3) & (rbg[:, 1] != np.float32(3.1))
    assert newborn_movers.sum() > 100
    # (the stream cursors may differ by a few draws by now -- a source point's static / dynamic split follows the voxel's mass, which
    # carries Ck's 1e-6 -- and every later newborn of that frame then draws other table entries: the bulk must agree, and what this
    # test is about is the flag below)
    frac = (rbo[:, 1:7] == rbg[:, 1:7]).all(axis=1).mean()
    assert frac > 0.99, frac
    mv = m.tile_moving() != 0
    has_mover = np.zeros(len(mv), bool)
    has_mover[np.unique(m.tile_of(vg[(rg[:, 1] != 0) | (rg[:, 2] != 0)]))] = True
    assert not (has_mover & ~mv).any()
    o.close(); m.close()


def test_static_tile_shortcuts_change_nothing_on_the_depth_stream(dsp):
    """DSPMAP_P_STATIC_TILE_SKIP = 0 treats every tile as if something moved in it: all velocity rows are read, every arrival's
    velocity is stored, no tile is ever zeroed.  240 frames of the metric's depth stream (moving pedestrians: matched clusters
    bear dynamic newborns, their particles cross tiles of static ones, tiles turn static and moving again all the time) through
    the captured frame with the shortcuts on and off, the SAME clouds for both: every slot, every float and the future status
    equal every 40 frames -- and the shortcuts were taken (hundreds of tiles flagged static beside thousands of movers)."""
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    cfg = dict(nx=66, ny=66, nz=40, res=0.15, ppv=24)
    sc = scene_mod.CorridorScene(66 * 0.15, 66 * 0.15, 40 * 0.15, seed=1234, device="cuda")
    frames = [sc.frame(f / 30.0) for f in range(240)]
    torch.cuda.synchronize()
    maps = []
    for skip in (1, 0):
        m = dsp.DSPMap(dsp.make_config(seed=1234, **cfg))
        m.L.dspmap_init_device(m.h)
        m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
        m.set_param(dsp.capi.P_STATIC_TILE_SKIP, skip)
        assert m.get_param(dsp.capi.P_STATIC_TILE_SKIP) == skip
        maps.append(m)
    seen_static, seen_movers = 0, 0
    for f, (pts, pos, quat) in enumerate(frames):
        for m in maps:
            assert m.update_device(pts.data_ptr(), pts.shape[0], pos, f / 30.0, quat) == 1
        if f % 40 == 39:
            a, b = maps[0].export_state(), maps[1].export_state()
            for x, y in zip(a, b):
                assert np.array_equal(x, y), f
            assert np.array_equal(maps[0].getFutureStatus(), maps[1].getFutureStatus()), f
            assert np.array_equal(maps[0].results(), maps[1].results()), f
            seen_static = max(seen_static, int((maps[0].tile_moving() == 0).sum()))
            seen_movers = max(seen_movers, int(((a[2][:, 1] != 0) | (a[2][:, 2] != 0)).sum()))
            assert (maps[1].tile_moving() != 0).all()
        for m in maps:
            m.clearOccupancyMapPrediction()
    assert seen_static > 200 and seen_movers > 1000, (seen_static, seen_movers)
    for m in maps:
        m.close()


@pytest.mark.parametrize("name", ["C_132x132x12_24ppv", "E_80x80x12_res010_36ppv"])
def test_resample_sparse_map_instantiation_against_oracle(dsp, orc, name):
    """k_resample<MW, RBK>: maps the handle takes for sparse (most tiles empty: the realistic fills of C and E) run the one-wave
    resampler with 8-row load batches, dense ones with 4-row batches -- two instantiations per occupancy-word count since round
    4.  The stage tests above leave the choice at its default (dense: RBK = 4); here DSPMAP_P_SPARSE_SWEEP = 1 forces the
    sparse launch (k_resample<1, 8> on config C's shape, k_resample<2, 8> on config E's) on the same scene, against the
    oracle: mass, mean velocity, survivors, copies and their slots bit-exact -- and the future status is the same bits as the
    dense launch's."""
    cfgkw, n_part = CONFIGS[name]
    futs = []
    for sparse in (1, 0):
        o, m = make_pair(dsp, orc, seed=5, **cfgkw)
        want = _force(m, dsp, "wave+light")
        m.set_param(dsp.capi.P_SPARSE_SWEEP, sparse)
        _resample_scene(o, m, cfgkw, n_part)
        o.occupancy_resample(); m.occupancy_resample()
        assert m.rollout_paths()[0] == want
        assert m.get_param(dsp.capi.P_SPARSE_SWEEP) == sparse     # the launch context the stage ran with
        futs.append(_check_resample(o, m, 6))
        o.close(); m.close()
    assert np.array_equal(futs[0], futs[1])


def test_sparse_large_one_word_map_takes_the_four_wave_resampler_by_itself(dsp, orc):
    """The dispatch rule of the resampling stage: a one-word map runs k_resample_wg below DSPMAP_P_RESAMPLE_WG_TILES (8 192) tiles --
    and, whatever its size, while the handle takes it for sparse (config C filled by the depth stream: 16 335 tiles of which a tenth
    hold anything).  A map of 8 712 tiles with its particles in a corner, the limit left at its default: forced sparse it runs
    k_resample_wg, forced dense k_resample<1, 4> -- both against the oracle, slot for slot, and the same future-status bits."""
    cfgkw = dict(nx=132, ny=132, nz=32, res=0.15, ppv=24)
    futs = []
    for sparse in (1, 0):
        o, m = make_pair(dsp, orc, seed=5, **cfgkw)
        assert m.V // 64 >= 8192 and m.get_param(dsp.capi.P_RESAMPLE_WG_TILES) == 8192
        m.set_param(dsp.capi.P_SPARSE_SWEEP, sparse)
        m.set_param(dsp.capi.P_ROLLOUT_INLINE, 1)
        _resample_scene(o, m, cfgkw, 300000)
        o.occupancy_resample(); m.occupancy_resample()
        var = m.rollout_paths()[0]
        assert (var & 1) == sparse, (var, sparse)                       # four waves per tile iff sparse
        assert (var >> 1) == (0 if sparse else 1)                       # rollout inside the resampler / k_rollout light
        futs.append(_check_resample(o, m, 6))
        o.close(); m.close()
    assert np.array_equal(futs[0], futs[1])


def test_resampler_variant_switching_mid_run_changes_nothing(dsp):
    """A one-word map of 8 712 tiles filled by the depth stream from empty: the handle starts on the one-wave resampler (nothing
    known about the map yet), finds the map sparse after a few frames and switches to k_resample_wg by itself -- in the middle
    of the run, on live state.  The same clouds through a handle pinned to the one-wave variant and one pinned to the four-wave
    variant: every slot, every float, every counter and the future status equal every 10 frames, and the first handle DID
    switch."""
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    cfg = dict(nx=132, ny=132, nz=32, res=0.15, ppv=24)
    sc = scene_mod.CorridorScene(132 * 0.15, 132 * 0.15, 32 * 0.15, seed=1234, device="cuda")
    frames = [sc.frame(f / 30.0) for f in range(60)]
    torch.cuda.synchronize()
    maps = []
    for limit in (None, 0, 1 << 30):
        m = dsp.DSPMap(dsp.make_config(seed=1234, **cfg))
        m.L.dspmap_init_device(m.h)
        m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
        if limit is not None:
            m.set_param(dsp.capi.P_RESAMPLE_WG_TILES, limit)
        maps.append(m)
    assert maps[0].V // 64 >= 8192
    seen = set()
    for f, (pts, pos, quat) in enumerate(frames):
        for m in maps:
            assert m.update_device(pts.data_ptr(), pts.shape[0], pos, f / 30.0, quat) == 1
        seen.add(maps[0].rollout_paths()[0] & 1)
        assert maps[1].rollout_paths()[0] & 1 == 0 and maps[2].rollout_paths()[0] & 1 == 1
        if f % 10 == 9:
            ref = maps[0].export_state()
            assert len(ref[0]) > 1000
            futs = [m.getFutureStatus() for m in maps]      # (ONE read per map: like the reference's getter, a read clears, :397-400)
            assert futs[0].sum() > 100
            for m, fut in zip(maps[1:], futs[1:]):
                for a, b in zip(ref, m.export_state()):
                    assert np.array_equal(a, b), f
                assert np.array_equal(futs[0], fut), f
                assert np.array_equal(maps[0].results(), m.results()), f
            cs = [m.counters() for m in maps]
            for c in cs:
                c.pop("update_ms")
            assert cs[0] == cs[1] == cs[2], f
        for m in maps:
            m.clearOccupancyMapPrediction()
    assert seen == {0, 1}, seen                     # one wave per tile at first, four once the handle knew the map to be sparse
    assert maps[0].get_param(dsp.capi.P_SPARSE_SWEEP) == 1
    for m in maps:
        m.close()
