"""GPU parity tests added in round 6 (run with `-m gpu`): CUBE STORAGE (DSPMAP_P_TILING: a tile of the particle store is a cube of
4 x 4 x 4 voxels instead of a run of 64 voxel indices) and the TWO-BRANCH frame on top of it (DSPMAP_P_FRAME_BRANCHES: the part of the
map the sensor can see runs prediction -> placement -> mapUpdate -> births -> resampling on the main stream, the rest of the map
prediction -> placement -> resampling beside it on a second stream) change nothing -- small maps that force them against the
index-order serial frame of rounds 1-5, every slot and every float, and the 132x132x60 map at its full size."""
import numpy as np
import pytest
import torch

from tests import common


def _snapshot(m):
    """everything a map reports after a frame, read ONCE (reading the future status clears it, :420-424)"""
    c = m.counters()
    return (m.export_state(), m.results(), m.getFutureStatus(),
            {k: c[k] for k in ("n_live_in", "n_moved", "n_out_of_map", "n_voxel_full", "n_pyramid_full", "n_born", "n_live_out", "n_reslotted", "n_fov")}, c)


def _same(a, b, what):
    for x, y in zip(a[0], b[0]):
        assert np.array_equal(x, y), what              # every particle: voxel, slot, the eight floats of its record
    assert np.array_equal(a[1], b[1]), what            # voxels_objects_number[v][0..3]
    assert np.array_equal(a[2], b[2]), what            # the future status
    assert a[3] == b[3], (what, a[3], b[3])

pytestmark = pytest.mark.gpu

UP = (0.70710678, 0.0, -0.70710678, 0.0)


def _trio(dsp, cfg, estimator=0, extra=None):
    """three maps of one configuration: [0] cube storage, frames forced to run as two branches; [1] cube storage, the serial frame;
    [2] index-order storage (rounds 1-5), the serial frame.  All on the one-wave-per-tile resampler (the four-waves-per-tile variant
    of small maps has no class filter: a map that runs it keeps the serial frame)"""
    maps = []
    # ... and [3] cube storage, the serial frame with its placement split (forced: DSPMAP_P_PLACE_SPLIT_TILES = 1), the side launch
    # leaving the main chain behind the placement of the tiles with a view, and the RESAMPLING split (DSPMAP_P_RESAMPLE_SPLIT: the tiles
    # no newborn can reach are resampled on the side stream beside the weight update and the births)
    for tiling, br in ((1, 1), (1, 0), (0, 0), (1, 2)):
        m = dsp.DSPMap(dsp.make_config(seed=1234, **cfg))
        m.set_param(dsp.capi.P_TILING, tiling)            # (before the device state exists)
        m.L.dspmap_init_device(m.h)
        assert m.get_param(dsp.capi.P_TILING) == tiling
        m.set_param(dsp.capi.P_RESAMPLE_WG_TILES, 0)
        if br == 2:
            br = 0
            m.set_param(dsp.capi.P_PLACE_SPLIT_TILES, 1)
            m.set_param(dsp.capi.P_SIDE_PLACEMENT, 16 + 3)
            m.set_param(dsp.capi.P_RESAMPLE_SPLIT, 1)
            assert m.get_param(dsp.capi.P_RESAMPLE_SPLIT) == 1
        else:
            m.set_param(dsp.capi.P_RESAMPLE_SPLIT, 0)
        # the index-order map is the frame of rounds 1-5 in this respect too: every workgroup of a sparse sweep loads its tile's own flags;
        # the others find their empty tiles in the tile bitmaps (DSPMAP_P_TILE_BITMAPS, in use where the sparse sweep is)
        m.set_param(dsp.capi.P_TILE_BITMAPS, 0 if tiling == 0 else 1)
        assert m.get_param(dsp.capi.P_TILE_BITMAPS) == (0 if tiling == 0 else 1)
        m.set_param(dsp.capi.P_FRAME_BRANCHES, br)
        assert m.get_param(dsp.capi.P_FRAME_BRANCHES) == br
        if estimator:
            m.set_param(dsp.capi.P_VELOCITY_ESTIMATOR, estimator)
        for k, v in (extra or {}).items():
            m.set_param(k, v)
        maps.append(m)
    return maps


@pytest.mark.parametrize("case", ["saturated_step", "nearly_full_voxels", "depth_stream_estimator", "two_words_overfull_lists",
                                  "moving_fill_turning", "depth_stream_static_tags", "sparse_sweep_variant", "empty_view_frames", "sparse_bitmaps_moving"])
def test_cube_storage_and_two_branch_frame_change_nothing(dsp, case):
    """DSPMAP_P_TILING + DSPMAP_P_FRAME_BRANCHES (round 6).  Storage: the device arrays are indexed tile by tile; with cubes for tiles the
    voxel -> (tile, lane) map changes, the reference's voxel index (:1081) stays what orders sweeps (source keys) and what every result
    and state record carries.  Frame:  The reference's frame is four sweeps over every voxel (:300-322).  Here k_tile_class cuts the
    tiles of a dense large map into Q -- a newborn of this frame can land there (:871-873: the observation's voxel row, grown by the
    position table's largest value, touches the field of view), which includes every tile in which a particle can be registered in a
    pyramid -- and P -- a particle of the tile can reach a Q tile this frame (:665-667: |od| + dt * the largest speed the map has ever
    seen, in rows and layers).  The main stream runs predict(P) -> place(Q) -> lists -> Ck -> weights -> births -> resample(Q), a forked
    branch predict(not P) -> [predict(P) done] -> place(not Q) -> resample(not Q); one rollout behind both.  Forced on for maps small
    enough to test, against the same map with the serial frame, frame by frame: every slot, every float, results, future status and
    counters equal --
      saturated_step        every particle changes voxel every frame (24 arrivals per voxel, across the class borders too), lists beyond CAPP
      nearly_full_voxels    voxels with 44 of 48 slots taken receive 10 arrivals each: arrivals in view find their voxel full
      depth_stream_estimator  the metric's stream, the estimator a third branch; particles born with velocities: the halo follows vmax
      two_words_overfull_lists  72 slots (two occupancy words), every list beyond CAPP: turned-away arrivals hand their slots back
      moving_fill_turning   every voxel seeded with particles at up to 2 m/s, the sensor advances, climbs and turns: movers cross the
                            Q / P borders in every direction, P is several rows wide
      depth_stream_static_tags  the stream with every point in view a static source (the saturated benchmark's birth mode)
      sparse_sweep_variant  the same with the prediction's SPARSE variant and the resampler's 8-row batches forced
      empty_view_frames     the stream, but in every third frame all points lie BEHIND the sensor while it turns: the view is empty, the
                            birth stage re-uses the cloud of the last non-empty view (:1379-1381) and its newborns land where that
                            frame's field of view was -- outside this frame's Q: the split resampling takes every tile behind the births
      sparse_bitmaps_moving the stream with the device estimator (newborns with velocities: their rollout dirties the accumulators of EMPTY tiles,
                            tiles fill and empty as the sensor turns), sparse sweeps forced, the future status read in some frames, cleared in
                            others and left to accumulate in the rest: maps 1 and 3 sweep by the tile bitmaps, map 2 by the tiles' own flags
    A FOURTH map runs the serial frame with its placement split and the resampling stage split (DSPMAP_P_RESAMPLE_SPLIT) in every case."""
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    quat = (1.0, 0.0, 0.0, 0.0)
    extra = {}
    if case == "two_words_overfull_lists":
        cfg = dict(nx=48, ny=48, nz=16, res=0.10, ppv=36)
    elif case == "nearly_full_voxels":
        cfg = dict(nx=64, ny=48, nz=12, res=0.15, ppv=24)
        quat = UP
    else:
        cfg = dict(nx=66, ny=66, nz=40, res=0.15, ppv=24)
    if case in ("sparse_sweep_variant", "sparse_bitmaps_moving"):
        extra[dsp.capi.P_SPARSE_SWEEP] = 1
    maps = _trio(dsp, cfg, estimator=2 if case in ("depth_stream_estimator", "sparse_bitmaps_moving") else 0, extra=extra)
    res = cfg["res"]
    if case in ("depth_stream_estimator", "depth_stream_static_tags", "sparse_sweep_variant"):
        sc = scene_mod.CorridorScene(66 * 0.15, 66 * 0.15, 40 * 0.15, seed=1234, device="cuda")
        frames = [sc.frame(f / 30.0) for f in range(90)]
        every = 15
    elif case == "sparse_bitmaps_moving":
        sc = scene_mod.CorridorScene(66 * 0.15, 66 * 0.15, 40 * 0.15, seed=1234, device="cuda")
        frames = []
        for f in range(60):
            pts, pos, _ = sc.frame(f / 30.0)
            yaw = 0.06 * f                                  # a steady turn: tiles enter and leave the view, fill and empty
            frames.append((pts, pos, (float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2)))))
        every = 5
    elif case == "empty_view_frames":
        sc = scene_mod.CorridorScene(66 * 0.15, 66 * 0.15, 40 * 0.15, seed=1234, device="cuda")
        frames = []
        for f in range(18):
            pts, pos, _ = sc.frame(f / 30.0)
            yaw = 0.5 * f                                   # (29 degrees per frame: the stale cloud's newborns land far outside the new view)
            q = (float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2)))
            if f % 3 == 2:
                pts = pts.clone(); pts[:, 0] = -pts[:, 0].abs() - 0.5   # every point behind the sensor (sensor frame): an empty view
            frames.append((pts, pos, q))
        every = 1
    elif case == "moving_fill_turning":
        for m in maps:
            m.seed_uniform(10, 0.02, 77, 2.0)
        pts = torch.from_numpy(common.wall_cloud(5, n_side=50, dist=2.4, half_w=2.0, half_h=1.0)).cuda()
        frames = []
        for f in range(8):
            yaw = 0.35 * f
            frames.append((pts, (0.05 * f, -0.03 * f, 0.04 * f), (float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2)))))
        every = 1
    else:
        per = {"saturated_step": 24, "nearly_full_voxels": 44, "two_words_overfull_lists": 24}[case]
        for m in maps:
            if case != "nearly_full_voxels":   # (that case starts empty: its state is written after the first frame)
                m.seed_uniform(per, 0.01, 99)
        if case == "nearly_full_voxels":
            yy, zz = np.meshgrid(np.linspace(-0.5, 0.5, 41), np.linspace(-0.3, 0.3, 25))
            pts = np.stack([np.full(yy.size, 0.45) + 0.02 * np.sin(7 * yy.ravel()), yy.ravel(), zz.ravel()], 1).astype(np.float32)   # a patch above the sensor
            pts = torch.from_numpy(pts).cuda()
        else:
            small = cfg["nx"] != 66
            pts = torch.from_numpy(common.wall_cloud(5, n_side=50, dist=1.6 if small else 2.4, half_w=1.2 if small else 2.0,
                                                     half_h=0.5 if small else 1.0)).cuda()
        # the sensor advances a whole voxel per frame along x (and a third of one along z): every particle changes voxel
        frames = [(pts, (res * f, 0.0, 0.34 * res * f), quat) for f in range(6)]
        if case == "nearly_full_voxels":
            frames = [(pts, (0.0, 0.0, 0.0), quat) for f in range(4)]   # the sensor rests; the state below arrives after the first frame
        every = 1
    torch.cuda.synchronize()
    tot = dict(n_voxel_full=0, n_pyramid_full=0, n_moved=0, n_reslotted=0, n_fov=0, n_born=0)
    for f, (pts, pos, q) in enumerate(frames):
        if case == "nearly_full_voxels" and f == 1:
            # a block of voxels above the sensor (in view, it looks straight up) holds 36 static particles each and 10 that move into the
            # next voxel in x within this frame's dt: they come from a LOWER voxel index, the destination's own particles -- its 10 leavers
            # too -- still hold their slots (:1214-1215): 2 fit, 8 find the voxel full
            rng = np.random.default_rng(3)
            nx, ny, nz = cfg["nx"], cfg["ny"], cfg["nz"]
            hx, hy, hz = (np.float32(res) * np.float32(n) * np.float32(0.5) for n in (nx, ny, nz))
            xs, ys, zs = np.meshgrid(np.arange(nx // 2 - 6, nx // 2 + 6), np.arange(ny // 2 - 6, ny // 2 + 6), np.arange(nz // 2 + 2, nz // 2 + 5), indexing="ij")
            vox = (zs * ny * nx + ys * nx + xs).ravel()

            def cloud(vx_idx, n_per, vel):
                v = np.repeat(vx_idx, n_per)
                zi, yi, xi = v // (ny * nx), (v // nx) % ny, v % nx
                u = rng.uniform(0.2, 0.8, (len(v), 3))
                px = (xi + u[:, 0]) * res - hx; py = (yi + u[:, 1]) * res - hy; pz = (zi + u[:, 2]) * res - hz
                rec = np.zeros((len(v), 8), np.float32)
                rec[:, 0] = 1.0; rec[:, 1] = vel; rec[:, 4] = px; rec[:, 5] = py; rec[:, 6] = pz; rec[:, 7] = 0.02
                return v.astype(np.int32), rec
            dt = 1.0 / 30.0
            v1, r1 = cloud(vox, 36, 0.0)
            v2, r2 = cloud(vox, 10, np.float32(res / dt))            # a voxel per frame towards +x: into the block's next voxel
            vv, rr = np.concatenate([v1, v2]), np.concatenate([r1, r2])
            sl = np.concatenate([np.tile(np.arange(36), len(vox)), np.tile(np.arange(36, 46), len(vox))]).astype(np.int32)   # explicit slots: the same state in both maps
            for m in maps:
                m.import_state(vv, rr, sl)
        for m in maps:
            npts = 0 if (case == "nearly_full_voxels" and f == 0) else pts.shape[0]   # (that case: no births before its state is written)
            assert m.update_device(pts.data_ptr(), npts, pos, f / 30.0, q) == 1
        if f % every == every - 1:
            snaps = [_snapshot(m) for m in maps]
            _same(snaps[1], snaps[2], (f, "cube storage against index-order storage"))
            _same(snaps[0], snaps[2], (f, "... and the two-branch frame on top of it"))
            _same(snaps[3], snaps[2], (f, "... and the serial frame with its placement and its resampling split"))
            if case == "empty_view_frames" and f % 3 == 2:
                assert snaps[2][4]["n_obs"] == 0 and snaps[2][4]["n_born"] > 0, (f, snaps[2][4])   # an empty view, births from the stale cloud
            ca = snaps[0][4]
            for k in tot:
                tot[k] += ca[k]
        if f % every != every - 1 and not (case == "sparse_bitmaps_moving" and f % 5 in (1, 2)):   # (that case: two frames in five leave the accumulators alone)
            for m in maps:
                m.clearOccupancyMapPrediction()
    br = [m.frame_branches() for m in maps]
    print(case, tot, "branches", br)
    assert br[0][0] == len(frames) and br[1][0] == 0 and br[2][0] == 0 and br[3][0] == 0, br   # every frame of map 0 ran as two branches, none of the others
    rs = [m.resample_split_frames() for m in maps]
    assert rs[3] == len(frames) and rs[0] == 0 and rs[1] == 0 and rs[2] == 0, rs   # every frame of map 3 split its resampling
    assert 0 < br[0][1] <= br[0][2] <= br[0][3], br                # Q inside P inside the map
    assert tot["n_moved"] > 2000 and tot["n_born"] > 100, tot
    if case == "saturated_step":
        assert tot["n_moved"] > 5 * 4000000 * 0.5 and tot["n_pyramid_full"] > 1000, tot
        assert br[0][2] < br[0][3], br                             # (no particle has a velocity: the halo is the ego-motion's)
    if case == "nearly_full_voxels":
        assert tot["n_voxel_full"] > 200 and tot["n_pyramid_full"] == 0 and tot["n_fov"] > 1000, tot
        assert br[0][4] > 4000, br                                  # the imported movers' 4.5 m/s were noted
    if case == "two_words_overfull_lists":
        assert maps[0].slots == 72 and tot["n_pyramid_full"] > 1000 and tot["n_reslotted"] > 0, tot
    if case == "moving_fill_turning":
        assert 1900 <= br[0][4] <= 2000 and tot["n_moved"] > 100000, (br, tot)
    if case == "depth_stream_estimator":
        assert br[0][4] > 0, br                                     # newborns of matched clusters / random velocities were noted
    if case in ("sparse_sweep_variant", "sparse_bitmaps_moving"):
        assert all(m.get_param(dsp.capi.P_SPARSE_SWEEP) == 1 for m in maps)
    if case == "sparse_bitmaps_moving":
        assert br[0][4] > 0, br                                     # velocities were given: the rollout ran
    for m in maps:
        m.close()


def test_two_branch_frame_changes_nothing_at_config_c_full_size(dsp):
    """the same at the size the branches are built for: 132x132x60 @ 0.15 m, every voxel seeded with 24 particles (the benchmark's
    C_sat), the depth stream's clouds and poses; map 0 runs what the handle chooses by itself (cube storage, two branches), map 1
    index-order storage and the serial frame.  Results, future status and counters every frame, every slot and float of the ~17 M particles after the last one."""
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    cfg = dict(nx=132, ny=132, nz=60, res=0.15, ppv=24)
    maps = []
    # (map 2: what the handle runs by default since the split resampling exists -- cube storage, the serial frame with its placement split,
    # the side launch behind the placement of the tiles with a view, the tiles no newborn can reach resampled on the side stream)
    for tiling, br, rs in ((-1, -1, 0), (0, 0, 0), (-1, 0, 1)):
        m = dsp.DSPMap(dsp.make_config(seed=1234, **cfg))
        m.set_param(dsp.capi.P_TILING, tiling)
        m.L.dspmap_init_device(m.h)
        m.set_param(dsp.capi.P_FRAME_BRANCHES, br)
        m.set_param(dsp.capi.P_RESAMPLE_SPLIT, rs)
        if rs:
            m.set_param(dsp.capi.P_SIDE_PLACEMENT, 16 + 3)
        m.seed_uniform(24, 0.01, 99)
        maps.append(m)
    assert maps[0].get_param(dsp.capi.P_TILING) == 1 and maps[1].get_param(dsp.capi.P_TILING) == 0
    sc = scene_mod.CorridorScene(132 * 0.15, 132 * 0.15, 60 * 0.15, seed=1234, device="cuda")
    frames = [sc.frame(f / 30.0) for f in range(5)]
    torch.cuda.synchronize()
    for f, (pts, pos, q) in enumerate(frames):
        for m in maps:
            assert m.update_device(pts.data_ptr(), pts.shape[0], pos, f / 30.0, q) == 1
        r1, f1, cb = maps[1].results(), maps[1].getFutureStatus(), maps[1].counters()
        for mi in (0, 2):
            assert np.array_equal(maps[mi].results(), r1), (f, mi)
            assert np.array_equal(maps[mi].getFutureStatus(), f1), (f, mi)
            ca = maps[mi].counters()
            for k in ("n_live_in", "n_moved", "n_out_of_map", "n_voxel_full", "n_pyramid_full", "n_born", "n_live_out", "n_reslotted", "n_fov"):
                assert ca[k] == cb[k], (f, mi, k, ca[k], cb[k])
        for m in maps:
            m.clearOccupancyMapPrediction()
    br = [m.frame_branches() for m in maps]
    print("C full size: branches", br, "moved", ca["n_moved"], "fov", ca["n_fov"], "born", ca["n_born"])
    assert br[0][0] == len(frames) and br[1][0] == 0, br
    assert 0 < br[0][1] <= br[0][2] < br[0][3] // 2, br            # the in-view branch is the smaller part of the map
    assert ca["n_moved"] > 100000 and ca["n_fov"] > 100000 and ca["n_born"] > 10000, ca
    assert maps[2].resample_split_frames() == len(frames) and maps[0].resample_split_frames() == 0 and maps[1].resample_split_frames() == 0
    sb = maps[1].export_state()
    for mi in (0, 2):
        sa = maps[mi].export_state()
        for x, y in zip(sa, sb):
            assert np.array_equal(x, y), mi
    assert len(sa[0]) > 15_000_000
    for m in maps:
        m.close()
