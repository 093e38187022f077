"""GPU tests of the Z-slab path through the C ABI (dspmap_mgpu_*): several slabs on ONE GPU in one
process (LocalComm) against the unsharded HIP map and the oracle, and the drop-in C++ example."""
import os
import subprocess

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(n_frames, seed=5):
    rng = np.random.default_rng(seed)
    ys, zs = np.meshgrid(np.linspace(-2.0, 2.0, 41), np.linspace(-1.0, 1.0, 21))
    base = np.stack([np.full(ys.size, 2.2) + 0.2 * np.sin(2 * ys.ravel()), ys.ravel(), zs.ravel()], 1).astype(np.float32)
    out = []
    for f in range(n_frames):
        t = f / 30.0
        pts = base + rng.normal(0, 0.005, base.shape).astype(np.float32)
        out.append((pts, (0.4 * t, 0.0, 0.1 * np.sin(5 * t)), t, (1.0, 0.0, 0.0, 0.0)))
    return out


@pytest.mark.parametrize("world,ppv", [(2, 12), (4, 36)])
def test_slabs_on_one_gpu_match_unsharded(dsp, orc, world, ppv):
    sharded = __import__("dsp-map_amd.sharded", fromlist=["ShardedDSPMap"])
    cfg = dict(nx=40, ny=40, nz=24, res=0.15, ppv=ppv)
    tables = common.tables(3)
    slabs = []
    for (z_lo, z_hi) in sharded.slab_ranges(cfg["nz"], world):
        s = sharded.HipSlab(dsp, cfg, z_lo, z_hi, 0)
        s.map.set_tables(*tables)
        slabs.append(s)
    sm = sharded.ShardedDSPMap(slabs, sharded.LocalComm())
    full = dsp.DSPMap(dsp.make_config(**cfg))
    full.set_tables(*tables)
    o = orc.Oracle(orc.make_config(**cfg))
    o.set_tables(*tables)
    o.L.dspo_use_velocity_estimator(o.h, 2)
    crossed = 0
    for pts, pos, t, q in _stream(8):
        d = torch.from_numpy(pts).cuda()
        assert sm.update(d, pos, t, q) == 1
        assert full.update(pts, pos, t, q) == 1
        assert o.update(pts, pos, t, q) == 1
        sm.sync()
        crossed += sum(s.map.counters()["n_exported_up"] + s.map.counters()["n_exported_down"] for s in slabs)
    got = np.concatenate([s.results() for s in slabs], 0)
    want = full.results()
    ref = o.results[:, :4]
    # against the unsharded HIP map: BIT-IDENTICAL.  Ck is summed on a fixed-point grid (the all-reduce over int64 is
    # associative), exported movers carry their source key and are placed by the receiving slab's k_place together with
    # its own movers, n_static is an exact MAX: nothing in a sharded frame depends on where the slabs are cut.
    assert crossed > 0
    assert np.array_equal(got, want)
    parts = [s.map.export_state() for s in slabs]
    sv, ss, sr = (np.concatenate([p[k] for p in parts]) for k in range(3))
    order = np.lexsort((ss, sv))
    fv, fs_, fr = full.export_state()
    assert np.array_equal(sv[order], fv) and np.array_equal(ss[order], fs_) and np.array_equal(sr[order], fr)
    for other, name in ((ref, "oracle"),):
        m_o, m_g = other[:, 0].astype(np.float64).sum(), got[:, 0].astype(np.float64).sum()
        assert abs(m_g - m_o) < 5e-3 * m_o, name
        close = np.abs(got[:, 0] - other[:, 0]) <= 1e-3 * np.maximum(1.0, np.abs(other[:, 0]))
        assert close.mean() > 0.97, (name, close.mean())
    live = sum(s.map.counters()["n_live_out"] for s in slabs)
    assert live == full.counters()["n_live_out"]
    # every slab only holds particles of its own layers
    for s in slabs:
        v, _, _ = s.map.export_state()
        if len(v):
            lay = v // (cfg["nx"] * cfg["ny"])
            assert lay.min() >= s.z_lo and lay.max() < s.z_hi
    o.close(); full.close()


def test_slabs_with_dynamic_birth_cloud_match_unsharded(dsp):
    """a caller-supplied birth cloud with dynamic sources (velocity-table and rand() cursors, k_birth_cursors) through
    the split-phase slab path: 3 slabs == the unsharded captured frame, every slot and every bit"""
    sharded = __import__("dsp-map_amd.sharded", fromlist=["ShardedDSPMap"])
    cfg = dict(nx=40, ny=40, nz=24, res=0.15, ppv=12)
    tables = common.tables(11)
    slabs = []
    for (z_lo, z_hi) in sharded.slab_ranges(cfg["nz"], 3):
        s = sharded.HipSlab(dsp, cfg, z_lo, z_hi, 0)
        s.map.set_tables(*tables)
        slabs.append(s)
    sm = sharded.ShardedDSPMap(slabs, sharded.LocalComm())
    full = dsp.DSPMap(dsp.make_config(**cfg))
    full.set_tables(*tables)
    rng = np.random.default_rng(3)
    for pts, pos, t, q in _stream(6):
        src = np.zeros(len(pts), dsp.VPOINT_DTYPE)
        src["x"] = pts[:, 0] + np.float32(pos[0]); src["y"] = pts[:, 1] + np.float32(pos[1]); src["z"] = pts[:, 2] + np.float32(pos[2])
        dyn = rng.choice(len(src), 80, replace=False)
        src["intensity"][dyn] = rng.uniform(0.1, 1.0, 80)
        src["nx"][dyn] = rng.uniform(-1, 1, 80); src["ny"][dyn] = rng.uniform(-1, 1, 80)
        src["nx"][dyn[:25]] = -10000; src["ny"][dyn[:25]] = -10000; src["nz"][dyn[:25]] = -10000
        d_pts = torch.from_numpy(pts).cuda()
        d_src = torch.from_numpy(src.view(np.float32).reshape(-1, 7).copy()).cuda()
        assert sm.update(d_pts, pos, t, q, birth=d_src) == 1
        assert full.update_device(d_pts.data_ptr(), len(pts), pos, t, q, birth_dev_ptr=d_src.data_ptr(), n_birth=len(src)) == 1
        sm.sync()
    parts = [s.map.export_state() for s in slabs]
    sv, ss, sr = (np.concatenate([p[k] for p in parts]) for k in range(3))
    order = np.lexsort((ss, sv))
    fv, fs_, fr = full.export_state()
    assert np.array_equal(sv[order], fv) and np.array_equal(ss[order], fs_) and np.array_equal(sr[order], fr)
    assert (fr[:, 1] != 0).sum() > 50
    assert all(s.map.cursors() == full.cursors() for s in slabs)
    full.close()


def test_first_frame_slabs_equal_unsharded_exactly(dsp):
    """frame 0 has no exchange and identical Ck -> every slab must hold exactly the unsharded result"""
    sharded = __import__("dsp-map_amd.sharded", fromlist=["ShardedDSPMap"])
    cfg = dict(nx=40, ny=40, nz=24, res=0.15, ppv=12)
    tables = common.tables(4)
    slabs = []
    for (z_lo, z_hi) in sharded.slab_ranges(cfg["nz"], 3):
        s = sharded.HipSlab(dsp, cfg, z_lo, z_hi, 0)
        s.map.set_tables(*tables)
        slabs.append(s)
    sm = sharded.ShardedDSPMap(slabs, sharded.LocalComm())
    full = dsp.DSPMap(dsp.make_config(**cfg))
    full.set_tables(*tables)
    pts, pos, t, q = _stream(1)[0]
    assert sm.update(torch.from_numpy(pts).cuda(), pos, t, q) == 1
    assert full.update(pts, pos, t, q) == 1
    got = np.concatenate([s.results() for s in slabs], 0)
    want = full.results()
    assert np.array_equal(got, want)
    a = np.concatenate([np.column_stack(s.map.export_state()[0:2]) for s in slabs])
    b = np.column_stack(full.export_state()[0:2])
    assert np.array_equal(a[np.lexsort((a[:, 1], a[:, 0]))], b)  # same particles in the same slots
    full.close()


def test_rccl_driver_single_rank_stream_ordered(dsp):
    """the torch.distributed (RCCL) driver in its stream-ordered mode -- slab kernels, torch ops and collectives on
    ONE torch stream, a single host synchronisation per frame -- with world_size 1 (all this box has): the one
    full-height slab must reproduce the unsharded map frame by frame (same kernels, same order of operations)"""
    import torch.distributed as dist
    sharded = __import__("dsp-map_amd.sharded", fromlist=["ShardedDSPMap"])
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        cfg = dict(nx=40, ny=40, nz=24, res=0.15, ppv=12)
        tables = common.tables(6)
        slab = sharded.HipSlab(dsp, cfg, 0, cfg["nz"], 0)
        slab.map.set_tables(*tables)
        sm = sharded.ShardedDSPMap([slab], sharded.TorchDistComm(torch.device("cuda", 0)))
        assert sm.stream is not None and slab.ordered
        full = dsp.DSPMap(dsp.make_config(**cfg))
        full.set_tables(*tables)
        for pts, pos, t, q in _stream(6):
            d = torch.from_numpy(pts).cuda()
            assert sm.update(d, pos, t, q) == 1
            assert full.update(pts, pos, t, q) == 1
            slab.map.clearOccupancyMapPrediction(); full.clearOccupancyMapPrediction()
        sm.sync()
        got, want = slab.results(), full.results()
        # Ck is accumulated on a fixed-point grid (order-independent), every other stage is slot-exact: bit-identical
        assert np.array_equal(got, want)
        for a, b in zip(slab.map.export_state(), full.export_state()):
            assert np.array_equal(a, b)
        assert slab.map.counters()["n_live_out"] == full.counters()["n_live_out"]
        full.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_dropin_example_runs(dsp):
    exe = os.path.join(ROOT, "examples", "map_example")
    subprocess.check_call(["g++", "-std=c++14", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "map_example.cpp"),
                           "-L" + os.path.join(ROOT, "dsp-map_amd", "lib"), "-ldspmap_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "dsp-map_amd", "lib"), "-o", exe])
    out = subprocess.check_output([exe, "12"]).decode()
    assert "Map is ready to update!" in out and "occupied" in out
    occ = int(out.split("occupied")[1].split()[0])
    assert occ > 50


@pytest.mark.parametrize("world,ppv,jump", [(2, 12, 0.0), (4, 36, 0.0), (8, 12, 0.7)])
def test_cpp_driver_group_matches_unsharded(dsp, world, ppv, jump):
    """the C++ frame driver (dspmap_dist.hip: header-carrying fixed-size exchange messages, forwarding rounds, the two
    reductions) over several slabs in one process: BIT-IDENTICAL to the unsharded map, slot for slot.  jump > 0: the sensor
    steps 0.7 m vertically between two frames -- more than a slab is high (3 layers = 0.45 m) -- so particles cross more
    than one slab face and are forwarded in a second round."""
    sharded = __import__("dsp-map_amd.sharded", fromlist=["CppGroup"])
    cfg = dict(nx=40, ny=40, nz=24, res=0.15, ppv=ppv)
    tables = common.tables(3)
    grp = sharded.CppGroup(dsp, cfg, world)
    for m in grp.maps:
        m.set_tables(*tables)
    full = dsp.DSPMap(dsp.make_config(**cfg))
    full.set_tables(*tables)
    crossed = 0
    for f, (pts, pos, t, q) in enumerate(_stream(8)):
        if jump and f >= 4:
            pos = (pos[0], pos[1], pos[2] + jump)
        d = torch.from_numpy(pts).cuda()
        assert grp.update(d, pos, t, q) == 1
        assert full.update_device(d.data_ptr(), len(pts), pos, t, q) == 1
        grp.sync()
        crossed += sum(m.counters()["n_moved"] for m in grp.maps)
        for m in grp.maps + [full]:
            m.clearOccupancyMapPrediction()
    got = np.concatenate([m.results() for m in grp.maps], 0)
    assert np.array_equal(got, full.results())
    parts = [m.export_state() for m in grp.maps]
    sv, ss, sr = (np.concatenate([p[k] for p in parts]) for k in range(3))
    order = np.lexsort((ss, sv))
    fv, fs_, fr = full.export_state()
    assert len(fv) > 3000
    assert np.array_equal(sv[order], fv) and np.array_equal(ss[order], fs_) and np.array_equal(sr[order], fr)
    assert sum(m.counters()["n_live_out"] for m in grp.maps) == full.counters()["n_live_out"]
    if jump:
        assert max(m.L.dspmap_mgpu_message_records(m.h) for m in grp.maps) >= 4096
    grp.close(); full.close()


def test_cpp_driver_rccl_single_rank(dsp):
    """dspmap_mgpu_update on a one-rank RCCL communicator (the library dlopens librccl.so, creates the communicator from a
    unique id and issues both all-reduces on its own stream): same map as the unsharded device-resident frame"""
    sharded = __import__("dsp-map_amd.sharded", fromlist=["CppShardedRank"])
    cfg = dict(nx=40, ny=40, nz=24, res=0.15, ppv=12)
    tables = common.tables(3)
    rk = sharded.CppShardedRank(dsp, cfg, 1, 0)
    rk.map.set_tables(*tables)
    full = dsp.DSPMap(dsp.make_config(**cfg))
    full.set_tables(*tables)
    for pts, pos, t, q in _stream(6):
        d = torch.from_numpy(pts).cuda()
        assert rk.update(d, pos, t, q) == 1
        assert full.update_device(d.data_ptr(), len(pts), pos, t, q) == 1
        rk.map.clearOccupancyMapPrediction(); full.clearOccupancyMapPrediction()
    rk.sync()
    for a, b in zip(rk.map.export_state(), full.export_state()):
        assert np.array_equal(a, b)
    assert np.array_equal(rk.map.results(), full.results())
    assert "librccl" in open("/proc/self/maps").read()
    rk.map._chk(rk.map.L.dspmap_mgpu_comm_destroy(rk.map.h))
    rk.map.close(); full.close()
