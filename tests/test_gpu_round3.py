"""GPU parity tests added in round 3 (run with `-m gpu`): heavy mover traffic through k_place, the device velocity
estimator inside the sharded C++ frame (the 10-horizon rollout on config D's grid shape moved to tests/test_gpu_round4.py,
where every rollout path is forced and verified to have run)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import common
from tests.test_gpu_parity import gpu_state, make_pair

pytestmark = pytest.mark.gpu


# sensor pitched 90 degrees: the field of view leaves these flat maps through the top face after ~0.4 m, so that no pyramid
# list overflows (a full LIST is a different rule, :1256-1259, tested on its own)
UP = (0.70710678, 0.0, -0.70710678, 0.0)


def _saturated(o, m, ppv, seed, vmax=0.0, moving_frac=0.0):
    """every voxel holds exactly `ppv` particles (slots 0..ppv-1 on both sides), positions uniform inside the voxel and
    2 % away from its faces"""
    cfg = o.cfg
    nx, ny, nz, res = cfg.nx, cfg.ny, cfg.nz, np.float32(cfg.voxel_resolution)
    half = common.half_extent(cfg)
    rng = np.random.default_rng(seed)
    V = nx * ny * nz
    idx = np.repeat(np.arange(V), ppv)
    zi, rest = idx // (nx * ny), idx % (nx * ny)
    yi, xi = rest // nx, rest % nx
    u = 0.02 + 0.96 * rng.random((3, len(idx)))
    px = ((xi + u[0]) * res - half[0]).astype(np.float32)
    py = ((yi + u[1]) * res - half[1]).astype(np.float32)
    pz = ((zi + u[2]) * res - half[2]).astype(np.float32)
    mv = rng.random(len(idx)) < moving_frac
    vx = (rng.uniform(-vmax, vmax, len(idx)) * mv).astype(np.float32)
    vy = (rng.uniform(-vmax, vmax, len(idx)) * mv).astype(np.float32)
    w = rng.uniform(0.005, 0.02, len(idx)).astype(np.float32)
    n = common.inject_both(o, m, px, py, pz, vx, vy, w)
    assert n == V * ppv
    return n


@pytest.mark.parametrize("case", ["shift_x", "shift_back_diag", "mixed_overflow", "two_words"])
def test_heavy_mover_traffic_is_slot_exact(dsp, orc, case):
    """moveParticle's first-free-in-sweep-order rule (:1209-1230) when a 64-voxel tile receives far more arrivals than the
    1 024 whose keys fit k_place's LDS table (a saturated 24-particles-per-voxel tile holds 1 536): an ego step of one whole
    voxel moves EVERY particle; the same particles must end up in the same slots as in the oracle, two runs must agree bit
    for bit."""
    ppv = 36 if case == "two_words" else 24
    cfgkw = dict(nx=32, ny=16, nz=4, res=0.15 if ppv == 24 else 0.10, ppv=ppv)
    res = cfgkw["res"]
    step = {"shift_x": (res, 0.0, 0.0, 1 / 30.0),                      # everybody one voxel up in x: forward arrivals only
            "shift_back_diag": (-0.7 * res, -0.6 * res, 0.0, 1 / 30.0),  # most move, to LOWER voxel indices: backward arrivals
            "mixed_overflow": (0.4 * res, 0.0, 0.3 * res, 0.1),          # + random velocities: both directions, full voxels
            "two_words": (res, res, 0.0, 1 / 30.0)}[case]
    runs = []
    for rep in range(2):
        o, m = make_pair(dsp, orc, **cfgkw)
        n = _saturated(o, m, ppv, 11, vmax=6.0 if case == "mixed_overflow" else 0.0,
                       moving_frac=0.7 if case == "mixed_overflow" else 0.0)
        empty = np.zeros((0, 3), np.float32)
        m.bin_points(empty, UP)
        m.predict(*step)
        vg, sg, rg = gpu_state(m)
        c = m.counters()
        kg = np.lexsort((sg, vg))
        runs.append((vg[kg], sg[kg], rg[kg]))
        if rep == 0:
            o.bin_points(empty, UP)
            o.predict(*step)
            vo, so, ro = o.export_sparse()
            ko = np.lexsort((so, vo))
            assert c["n_live_in"] == n
            assert c["n_moved"] > 0.55 * n, c
            # arrivals per 64-voxel tile: well beyond the LDS table
            assert c["n_moved"] / (m.V / 64) > 1100 or case == "mixed_overflow", c
            if case == "mixed_overflow":
                assert c["n_voxel_full"] > 10, c
            assert c["n_pyramid_full"] == 0 and c["n_fov"] > 0, c
            assert len(vg) == len(vo) == n - c["n_out_of_map"] - c["n_voxel_full"]
            assert np.array_equal(vo[ko], vg[kg]) and np.array_equal(so[ko], sg[kg])
            for col in (1, 2, 4, 5, 6, 7):
                assert np.array_equal(ro[ko][:, col], rg[kg][:, col]), col
        o.close(); m.close()
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)


def _group_vs_full(dsp, world, cfg, frames, seed=3, sparse=None):
    """the C++ frame driver over `world` slabs in one process and the unsharded map, both with the DEVICE velocity
    estimator in the frame, fed the same frames: every slot and every float must be equal"""
    sharded = __import__("dsp-map_amd.sharded", fromlist=["CppGroup"])
    tables = common.tables(seed)
    grp = sharded.CppGroup(dsp, cfg, world)
    full = dsp.DSPMap(dsp.make_config(**cfg))
    for x in grp.maps + [full]:
        x.set_tables(*tables)
        x.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
    if sparse is not None:          # the slabs run k_predict's SPARSE variant, the unsharded map the other one
        for x in grp.maps:
            x.set_param(dsp.capi.P_SPARSE_SWEEP, sparse)
        full.set_param(dsp.capi.P_SPARSE_SWEEP, 1 - sparse)
    clouds = []
    for pts, pos, t, q in frames:
        d = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).cuda()
        assert grp.update(d, pos, t, q) == 1
        assert full.update_device(d.data_ptr(), len(pts), pos, t, q) == 1
        grp.sync()
        clouds.append([x.get_birth_cloud() for x in (grp.maps[0], grp.maps[-1], full)])
        # the future status (fixed-point accumulators: order-free) of the slabs is the unsharded map's, bit for bit; reading it
        # also clears it on every map alike (:420-424)
        fut_s = np.concatenate([x.getFutureStatus() for x in grp.maps], 0)
        assert np.array_equal(fut_s, full.getFutureStatus()), len(clouds)
    got = np.concatenate([x.results() for x in grp.maps], 0)
    assert np.array_equal(got, full.results())
    parts = [x.export_state() for x in grp.maps]
    sv, ss, sr = (np.concatenate([p[k] for p in parts]) for k in range(3))
    order = np.lexsort((ss, sv))
    fv, fs_, fr = full.export_state()
    assert np.array_equal(sv[order], fv) and np.array_equal(ss[order], fs_) and np.array_equal(sr[order], fr)
    assert sum(x.counters()["n_live_out"] for x in grp.maps) == full.counters()["n_live_out"]
    cur = [x.cursors() for x in grp.maps + [full]]
    assert all(c == cur[0] for c in cur)                    # every rank consumed the three random streams alike
    holding = sum(1 for p in parts if len(p[0]) > 0)         # slabs that hold particles
    grp.close(); full.close()
    return clouds, fr, holding


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_frame_runs_the_velocity_estimator_multi_cluster(dsp, world):
    """the reference forks / joins velocityEstimationThread in EVERY update() (:297,311,1377-1544): a sharded frame
    (dspmap_mgpu_update's phases) runs the device estimator redundantly on every slab -- same cloud, deterministic
    kernels, same tagged birth cloud -- so that a sharded map is the same filter as the unsharded one: bit-identical on
    the seven-group scene (moving, too fast, gated, static clusters), dynamic newborn velocities included"""
    from tests.test_gpu_round2 import _cluster_scene
    cfg = dict(nx=66, ny=66, nz=40, res=0.15, ppv=12)
    frames = []
    for f in range(4):
        t = f * 0.1
        frames.append((_cluster_scene(t, f), (0.0, 0.0, 1.0 + 0.04 * f), t, (1.0, 0.0, 0.0, 0.0)))
    clouds, rec, holding = _group_vs_full(dsp, world, cfg, frames)
    for f, (a, b, c) in enumerate(clouds):
        assert len(a) == len(b) == len(c) > 500
        for k in ("x", "y", "z", "nx", "ny", "nz", "intensity"):
            assert np.array_equal(a[k], c[k]) and np.array_equal(b[k], c[k]), (f, k)
    g = clouds[-1][2]
    dyn = g["intensity"] > 0.01
    assert dyn.sum() > 200 and np.isclose(g["ny"][dyn], 1.0, atol=0.02).sum() == 60      # cluster A matched at 1 m/s
    assert (np.abs(rec[:, 1]) + np.abs(rec[:, 2]) > 0.3).sum() > 100                     # moving newborns in the map
    assert holding >= 2                                                                  # the scene spans several slabs


def test_sharded_frame_runs_the_velocity_estimator_depth_stream(dsp):
    """the benchmark's depth stream (pedestrians, boxes, ~5000 points per frame) through 4 slabs with the device estimator
    in the frame, incl. a frame with an empty view (the previous birth cloud is re-used, :1379-1381): == unsharded"""
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    cfg = dict(nx=66, ny=66, nz=40, res=0.15, ppv=24)
    sc = scene_mod.CorridorScene(66 * 0.15, 66 * 0.15, 40 * 0.15, seed=1234, device="cuda")
    frames = []
    for f in range(7):
        t = f / 30.0
        pts_t, pos, quat = sc.frame(t)
        pts = pts_t.cpu().numpy().copy()
        if f == 4:
            pts[:, 0] *= -1.0
        frames.append((pts, pos, t, quat))
    clouds, rec, holding = _group_vs_full(dsp, 4, cfg, frames, seed=5)
    assert sum(int((c[2]["intensity"] > 0.01).sum()) for c in clouds) > 200
    assert len(clouds[4][2]) == len(clouds[3][2])            # empty view: the previous cloud
    assert (np.abs(rec[:, 1]) + np.abs(rec[:, 2]) > 0.3).sum() > 100


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_saturated_map_stepping_a_voxel_per_frame(dsp, world):
    """a saturated 24-particles-per-voxel map whose sensor advances one whole voxel per frame (every particle changes
    voxel: ~1 500 arrivals per tile, beyond k_place's LDS table) and climbs a fifth of a voxel per frame (a layer of
    particles changes slab every few frames): the sharded map is the unsharded one, slot for slot and bit for bit.  (The
    sensor looks straight up: its field of view leaves this flat map after 0.6 m and no pyramid list overflows -- a full
    list is cut per slab, the one documented difference between a sharded and an unsharded map)"""
    sharded = __import__("dsp-map_amd.sharded", fromlist=["CppGroup"])
    cfg = dict(nx=32, ny=16, nz=8, res=0.15, ppv=24)
    tables = common.tables(9)
    grp = sharded.CppGroup(dsp, cfg, world)
    full = dsp.DSPMap(dsp.make_config(**cfg))
    for x in grp.maps + [full]:
        x.set_tables(*tables)
        x.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
        x.seed_uniform(24, weight=0.01, seed=99)
    yy, zz = np.meshgrid(np.linspace(-0.3, 0.3, 25), np.linspace(-0.2, 0.2, 17))
    pts = np.stack([np.full(yy.size, 0.42) + 0.02 * np.sin(7 * yy.ravel()), yy.ravel(), zz.ravel()], 1).astype(np.float32)   # a patch 0.4 m in front of (= above) the sensor
    d = torch.from_numpy(pts).cuda()
    moved = 0
    for f in range(5):
        pos = (0.15 * f, 0.0, 0.03 * f)
        assert grp.update(d, pos, f / 30.0, UP) == 1
        assert full.update_device(d.data_ptr(), len(pts), pos, f / 30.0, UP) == 1
        grp.sync()
        moved = max(moved, full.counters()["n_moved"])
        for x in grp.maps + [full]:
            x.clearOccupancyMapPrediction()
    assert moved > 0.8 * 24 * 31 * 16 * 8 * 0.9, moved          # (nearly) every particle changed voxel in a frame
    got = np.concatenate([x.results() for x in grp.maps], 0)
    assert np.array_equal(got, full.results())
    parts = [x.export_state() for x in grp.maps]
    sv, ss, sr = (np.concatenate([p[k] for p in parts]) for k in range(3))
    order = np.lexsort((ss, sv))
    fv, fs_, fr = full.export_state()
    assert len(fv) > 20000 and full.counters()["n_pyramid_full"] == 0
    assert np.array_equal(sv[order], fv) and np.array_equal(ss[order], fs_) and np.array_equal(sr[order], fr)
    grp.close(); full.close()


@pytest.mark.parametrize("res", [0.15, 0.10, 0.2, 0.073, 1.0, 2.5])
def test_fast_voxel_division_is_the_ieee_division(dsp, res):
    """(int)((p + half) / res) (:1062-1088) is computed as reciprocal + two FMAs only after a kernel has compared that
    quotient with the IEEE division bit for bit (all 2^23 floats of a binade in which every quotient is >= 1, also for
    resolutions of a metre and more -- the sequence commutes with scaling by powers of two -- plus a sample of the whole range): DSPMAP_P_FAST_DIVISION reports it; forcing the IEEE division
    gives the same map, slot for slot, after a prediction that moves most particles"""
    cfgkw = dict(nx=48, ny=40, nz=12, res=res, ppv=12)
    outs = []
    for force_ieee in (False, True):
        m = dsp.DSPMap(dsp.make_config(**cfgkw))
        m._chk(m.L.dspmap_init_device(m.h))
        assert m.L.dspmap_get_param(m.h, dsp.capi.P_FAST_DIVISION) == 1.0     # verified for this resolution
        if force_ieee:
            m.set_param(dsp.capi.P_FAST_DIVISION, 0)
            assert m.L.dspmap_get_param(m.h, dsp.capi.P_FAST_DIVISION) == 0.0
        m.seed_uniform(10, weight=0.01, seed=5, vmax=2.0)
        m.bin_points(np.zeros((0, 3), np.float32), common.EX_QUATS[1])
        m.predict(0.37 * res, -0.81 * res, 0.23 * res, 0.11)
        c = m.counters()
        assert c["n_moved"] > 0.5 * c["n_live_in"]
        v, s, r = m.export_state()
        k = np.lexsort((s, v))
        outs.append((v[k], s[k], r[k]))
        m.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def _fill_view_uniform(o, m, n, seed, vmax):
    """n particles uniform over the map box, those inside the field of view kept (uniform density: no pile-up in the voxels
    next to the sensor), all with a velocity"""
    half = common.half_extent(o.cfg)
    rng = np.random.default_rng(seed)
    px = rng.uniform(0.3, half[0] * 0.97, n); py = rng.uniform(-half[1] * 0.97, half[1] * 0.97, n); pz = rng.uniform(-half[2] * 0.97, half[2] * 0.97, n)
    keep = (np.abs(np.degrees(np.arctan2(py, px))) < 40.0) & (np.abs(np.degrees(np.arctan2(pz, px))) < 22.0)
    px, py, pz = (a[keep].astype(np.float32) for a in (px, py, pz))
    vx = (rng.uniform(-vmax, vmax, len(px))).astype(np.float32)
    vy = (rng.uniform(-vmax, vmax, len(px))).astype(np.float32)
    w = rng.uniform(0.01, 0.05, len(px)).astype(np.float32)
    return common.inject_both(o, m, px, py, pz, vx, vy, w)


@pytest.mark.parametrize("ppv,n,seed", [(24, 227000, 9), (36, 303000, 4)])
def test_turned_away_movers_hand_their_slot_back_at_once(dsp, orc, ppv, n, seed):
    """a particle that changes voxel and finds its pyramid's list full takes a slot and hands it back at once (:1256-1259):
    the arrivals that the sweep serves after it use that slot; likewise the slot of a turned-away particle that stayed in
    its voxel is free for the arrivals from higher voxel indices.  k_place gives the slots out before the lists are cut;
    k_place_fix re-slots the arrivals of the voxels concerned: lists AND slots equal the oracle's sequential sweep exactly --
    one and two occupancy words.  (Not treated, and counted in n_overflow_inexact: an arrival that found its voxel full
    before the lists were cut and would fit afterwards -- see test_pyramid_list_overflow_in_sweep_order.)"""
    cfgkw = dict(nx=40, ny=40, nz=16, res=0.15 if ppv == 24 else 0.10, ppv=ppv)
    o, m = make_pair(dsp, orc, **cfgkw)
    n_in = _fill_view_uniform(o, m, n, seed, vmax=1.5)
    pts = common.wall_cloud(3, n_side=40, dist=2.2 if ppv == 24 else 1.5, half_w=1.8 if ppv == 24 else 1.2, half_h=0.6)
    o.bin_points(pts); m.bin_points(pts)
    o.predict(-0.03, 0.02, 0.0, 0.1); m.predict(-0.03, 0.02, 0.0, 0.1)
    len_o = (o.pyramid_lists[:, :, 0] != 0).sum(1)
    c = m.counters()
    assert c["n_moved"] > 5000 and c["n_reslotted"] > 5 and c["n_overflow_inexact"] == 0, c
    assert (len_o == o.capp).sum() >= 10 and c["n_pyramid_full"] > 100, ((len_o == o.capp).sum(), c)
    assert np.array_equal(len_o, m.pyramid_counts())
    vo, so, ro = o.export_sparse()
    vg, sg, rg = gpu_state(m)
    assert len(vo) == len(vg) == n_in - c["n_out_of_map"] - c["n_pyramid_full"] - c["n_voxel_full"]
    ko, kg = np.lexsort((so, vo)), np.lexsort((sg, vg))
    assert np.array_equal(vo[ko], vg[kg]) and np.array_equal(so[ko], sg[kg])      # the same particles in the SAME SLOTS
    for col in (1, 2, 4, 5, 6, 7):
        assert np.array_equal(ro[ko][:, col], rg[kg][:, col]), col
    # the weight update writes through the re-pointed list entries: weights against the oracle after mapUpdate
    o.map_update(); m.map_update()
    vo, so, ro = o.export_sparse()
    vg, sg, rg = gpu_state(m)
    ko, kg = np.lexsort((so, vo)), np.lexsort((sg, vg))
    assert np.array_equal(vo[ko], vg[kg]) and np.array_equal(so[ko], sg[kg])
    assert np.allclose(ro[ko][:, 7], rg[kg][:, 7], rtol=1e-4, atol=1e-9)
    o.close(); m.close()


def test_estimator_keeps_one_last_state_across_host_and_device(dsp):
    """the reference matches every frame's clusters against ONE `clusters_feature_vector_dynamic_last` (a function static,
    :1401,1542).  A cloud beyond the device estimator's capacity (6144 points) is clustered by the host stage: the two
    implementations hand that state to each other, so a stream whose cloud sizes straddle the capacity gets the same
    velocity tags as the host stage alone (frame 2 here is padded with ground points to 7000 points)"""
    from tests.test_gpu_round2 import _cluster_scene
    cfgkw = dict(nx=66, ny=66, nz=40, ppv=9)
    maps = []
    for mode in (2, 1):
        m = dsp.DSPMap(dsp.make_config(**cfgkw))
        m.set_tables(*common.tables(1))
        m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, mode)
        maps.append(m)
    rng = np.random.default_rng(0)
    pos = (0.0, 0.0, 1.0)
    for f in range(4):
        t = f * 0.1
        pts = _cluster_scene(t, f)
        if f == 2:
            pad = np.stack([rng.uniform(2.3, 3.4, 6400), rng.uniform(-1.6, 1.6, 6400), np.full(6400, -0.98)], 1).astype(np.float32)
            pts = np.concatenate([pts, pad])
            assert len(pts) > 6144
        clouds = []
        for m in maps:
            assert m.update(pts, pos, t, (1, 0, 0, 0)) == 1
            clouds.append(m.get_birth_cloud())
            m.getOccupancyMapWithFutureStatus(0.2)
        g, w = clouds
        assert len(g) == len(w), f
        for k in ("x", "y", "z", "nx", "ny", "nz"):
            assert np.array_equal(g[k], w[k]), (f, k)
        assert np.array_equal(g["intensity"] > 0.01, w["intensity"] > 0.01)
        if f >= 1:
            dyn = g["intensity"] > 0.01
            assert np.isclose(g["ny"][dyn], 1.0, atol=0.02).sum() == 60, f      # cluster A keeps its 1 m/s through the switches
    for m in maps:
        m.close()


def test_sharded_frame_with_split_placement_matches_unsharded(dsp):
    """large slabs place the arrivals of the tiles that cannot see the field of view on a side stream, beside the pair kernels
    and the Ck all-reduce (dspmap_mgpu_ck_partial); forced on here for a small map: 4 slabs == the unsharded map"""
    sharded = __import__("dsp-map_amd.sharded", fromlist=["CppGroup"])
    from tests.test_gpu_sharded import _stream
    cfg = dict(nx=40, ny=40, nz=24, res=0.15, ppv=12)
    tables = common.tables(3)
    grp = sharded.CppGroup(dsp, cfg, 4)
    full = dsp.DSPMap(dsp.make_config(**cfg))
    for x in grp.maps + [full]:
        x.set_tables(*tables)
        x.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
    for x in grp.maps:
        x.set_param(dsp.capi.P_PLACE_SPLIT_TILES, 1)
    for pts, pos, t, q in _stream(8):
        d = torch.from_numpy(pts).cuda()
        assert grp.update(d, pos, t, q) == 1
        assert full.update_device(d.data_ptr(), len(pts), pos, t, q) == 1
        grp.sync()
        for x in grp.maps + [full]:
            x.clearOccupancyMapPrediction()
    got = np.concatenate([x.results() for x in grp.maps], 0)
    assert np.array_equal(got, full.results())
    parts = [x.export_state() for x in grp.maps]
    sv, ss, sr = (np.concatenate([p[k] for p in parts]) for k in range(3))
    order = np.lexsort((ss, sv))
    fv, fs_, fr = full.export_state()
    assert len(fv) > 3000
    assert np.array_equal(sv[order], fv) and np.array_equal(ss[order], fs_) and np.array_equal(sr[order], fr)
    grp.close(); full.close()


@pytest.mark.gpu
@pytest.mark.parametrize("quat", [(1.0, 0.0, 0.0, 0.0), (0.9659258, 0.0, 0.0, 0.258819)])
def test_sparse_sweep_variant_changes_nothing(dsp, quat):
    """A map filled by the depth stream is mostly empty tiles; k_predict's SPARSE variant leaves such a tile after one scalar
    load (no view test, no zeroing of future accumulators nobody added to: DevState::fut_dirty), and k_place tests the view of
    a skipped tile that receives arrivals itself.  Forced on / forced off / chosen by the handle (DSPMAP_P_SPARSE_SWEEP), with
    the split placement forced on, from an EMPTY map under a moving sensor (particles keep arriving in tiles that were empty),
    with clearOccupancyMapPrediction skipped on every third frame (the accumulators then add up over two frames): every slot,
    every float, every counter and every future-status value equal, frame after frame."""
    cfg = dict(nx=56, ny=88, nz=12, res=0.15, ppv=24)
    tables = common.tables(5)
    maps = []
    for force in (1, 0, -1):
        m = dsp.DSPMap(dsp.make_config(**cfg)); m.set_tables(*tables)
        m.L.dspmap_init_device(m.h)
        m.set_param(dsp.capi.P_PLACE_SPLIT_TILES, 1)
        m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, 2)
        m.set_param(dsp.capi.P_SPARSE_SWEEP, force)
        maps.append(m)
    rng = np.random.default_rng(3)
    ys, zs = np.meshgrid(np.linspace(-2.0, 2.0, 41), np.linspace(-0.7, 0.7, 15))
    base = np.stack([np.full(ys.size, 2.3) + 0.2 * np.sin(2 * ys.ravel()), ys.ravel(), zs.ravel()], 1).astype(np.float32)
    moved = 0
    for f in range(36):
        t = f / 30.0
        pts = torch.from_numpy(base + rng.normal(0, 0.004, base.shape).astype(np.float32)).cuda()
        pos = (0.9 * t, 0.5 * t, 0.1 * np.sin(5 * t))
        for m in maps:
            assert m.update_device(pts.data_ptr(), len(base), pos, t, quat) == 1
        cs = [m.counters() for m in maps]
        for c in cs:
            c.pop("update_ms")
        assert cs[0] == cs[1] == cs[2], (f, cs)
        moved += cs[0]["n_moved"]
        if f % 6 == 5:
            futs = [m.getFutureStatus() for m in maps]
            assert np.array_equal(futs[0], futs[1]) and np.array_equal(futs[0], futs[2]), f
            assert futs[0].sum() > 0
        if f % 3 != 2:
            for m in maps:
                m.clearOccupancyMapPrediction()
    assert moved > 2000
    assert maps[0].get_param(dsp.capi.P_SPARSE_SWEEP) == 1 and maps[1].get_param(dsp.capi.P_SPARSE_SWEEP) == 0
    live_tiles = maps[0].tile_of(maps[0].export_state()[0])
    assert len(np.unique(live_tiles)) < 0.5 * maps[0].tile_count()    # the map IS mostly empty tiles
    ref = maps[0].export_state()
    for m in maps[1:]:
        for a, b in zip(ref, m.export_state()):
            assert np.array_equal(a, b)
        assert np.array_equal(maps[0].results(), m.results())
    for m in maps:
        m.close()


@pytest.mark.gpu
def test_sharded_frame_with_the_sparse_sweep_variant(dsp):
    """the slabs of a sharded map run k_predict's SPARSE variant (empty tiles are left after one scalar load; arrivals into
    skipped tiles -- also those imported from the neighbouring slab -- have their view tested by k_place), the unsharded map
    the dense one: the depth stream from an empty map through 4 slabs, 10 frames, == unsharded, bit for bit"""
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    cfg = dict(nx=66, ny=66, nz=40, res=0.15, ppv=24)
    sc = scene_mod.CorridorScene(66 * 0.15, 66 * 0.15, 40 * 0.15, seed=77, device="cuda")
    frames = []
    for f in range(10):
        t = f / 30.0
        pts_t, pos, quat = sc.frame(t)
        frames.append((pts_t.cpu().numpy().copy(), pos, t, quat))
    clouds, rec, holding = _group_vs_full(dsp, 4, cfg, frames, seed=9, sparse=1)
    assert holding >= 2 and len(rec) > 20000


@pytest.mark.gpu
def test_inline_rollout_equals_k_rollout(dsp):
    """maps small enough for the four-waves-per-tile resampler add the future status of their moving particles from inside
    k_resample_wg (no k_rollout launch) unless many tiles hold hundreds of moving particles (DSPMAP_P_ROLLOUT_INLINE: the
    handle's choice from last frame's count): forced on / forced off / chosen, with every seeded particle moving -- the same
    particles in the same slots, the same future status BIT FOR BIT (fixed-point accumulators: every particle adds the same
    integer on every path), and the handle's own choice ends up at k_rollout on this (pathological) fill"""
    cfg = dict(nx=40, ny=36, nz=12, res=0.15, ppv=24)
    tables = common.tables(4)
    maps = []
    for force in (1, 0, -1):
        m = dsp.DSPMap(dsp.make_config(**cfg)); m.set_tables(*tables)
        m.L.dspmap_init_device(m.h)
        m.set_param(dsp.capi.P_ROLLOUT_INLINE, force)
        m.seed_uniform(20, 0.01, 6, 1.0)
        maps.append(m)
    pts = common.wall_cloud(3, n_side=30, dist=2.0, half_w=1.5, half_h=0.6)
    d = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).cuda()
    for f in range(4):
        for m in maps:
            assert m.update_device(d.data_ptr(), len(pts), (0.02 * f, 0.0, 0.0), f / 30.0, (1.0, 0.0, 0.0, 0.0)) == 1
        futs = [m.getFutureStatus() for m in maps]
        assert futs[0].sum() > 100
        assert np.array_equal(futs[0], futs[1]) and np.array_equal(futs[2], futs[1]), f
        for m in maps:
            m.clearOccupancyMapPrediction()
    assert [int(m.get_param(dsp.capi.P_ROLLOUT_INLINE)) for m in maps] == [1, 0, 0]
    ref = maps[0].export_state()
    for m in maps[1:]:
        for a, b in zip(ref, m.export_state()):
            assert np.array_equal(a, b)
    for m in maps:
        m.close()
