"""CPU tests: pin the oracle against every known answer SURVEY.md records from
the real reference ([probe] values), plus internal invariants of the restated
stages.  The reference ships no tests or golden vectors
and cannot be compiled here (Eigen/PCL/munkres absent) -> "parity unpinned"
beyond these known answers."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import common

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_sizes_match_reference_probe(orc):
    # SURVEY Appendix B, rows marked [probe ok]: CAPP 462 / 1188 / 6996, V, SLOTS
    o = orc.Oracle()
    assert (o.V, o.slots, o.NP, o.capp, o.rdim) == (174240, 18, 448, 462, 10)
    o.close()
    o = orc.Oracle(orc.make_config(ppv=24))
    assert (o.slots, o.capp) == (48, 1188)
    o.close()


def test_capp_large_config(orc):
    cfg = orc.make_config(nx=132, ny=132, nz=12, ppv=24)  # reduced z so the dense arrays stay small
    o = orc.Oracle(cfg)
    assert o.capp == (132 * 132 * 12 * 24 + 100000) // 7200 * 2
    o.close()
    # the formula at the full 132x132x60 size (SURVEY: 6996 [probe ok])
    assert (132 * 132 * 60 * 24 + 100000) // 7200 * 2 == 6996


def test_pdf_lut_known_answers(orc):
    o = orc.Oracle()
    lut = o.pdf_lut
    # SURVEY 8a/a9 [probe]: pdf[10000] = 0.5641896 (1/sqrt(pi), not 1/sqrt(2 pi))
    assert abs(float(lut[10000]) - 0.5641896) < 1e-7
    assert abs(float(lut[10000]) - 1.0 / np.sqrt(np.pi)) < 1e-7
    # clamp +-9.9 -> floor value ~1.1e-22, never exactly 0 (Appendix A-1)
    lo = o.L.dspo_query_normal_pdf(o.h, 100.0, 0.0, 0.1)
    assert 0 < lo < 1e-20 and abs(lo - float(lut[19900])) == 0
    assert o.L.dspo_query_normal_pdf(o.h, -100.0, 0.0, 0.1) == float(lut[100])
    # truncating index, step 1e-3: z in [0.0120, 0.0129] all map to LUT[10012]
    v = o.L.dspo_query_normal_pdf(o.h, 0.00125, 0.0, 0.1)
    assert v == float(lut[10012])
    # no 1/sigma factor: value at x==mu is the LUT centre for any sigma
    assert o.L.dspo_query_normal_pdf(o.h, 1.0, 1.0, 0.37) == float(lut[10000])
    # symmetric table
    assert np.allclose(lut[10000 - 3000], lut[10000 + 3000], rtol=1e-6)
    o.close()


def test_neighbor_table_known_answers(orc):
    o = orc.Oracle()
    nb = o.neighbors
    # SURVEY a10 [probe]: pyr 17 -> 0 1 2 16 17 18 32 33 34 ; corner -> 4 ; edge -> 6
    assert nb[17].tolist() == [9, 0, 1, 2, 16, 17, 18, 32, 33, 34]
    assert nb[0, 0] == 4 and nb[0, 1:5].tolist() == [0, 1, 16, 17]
    assert nb[1, 0] == 6
    assert nb[447, 0] == 4
    # symmetry of the neighbourhood relation (used by the GPU pass-1 tiling)
    sets = [set(nb[b, 1:1 + nb[b, 0]].tolist()) for b in range(o.NP)]
    for b in range(o.NP):
        for c in sets[b]:
            assert b in sets[c]
    o.close()


def test_pyramid_index_formula(orc):
    # SURVEY a4 [probe], identity attitude: h = ceil(13 + az/3deg), v = ceil(7 - el/3deg),
    # az = atan2(y,x), el = atan2(z,x)
    o = orc.Oracle()
    o.bin_points(np.zeros((0, 3), np.float32))
    rng = np.random.default_rng(0)
    for _ in range(5000):
        az, el = rng.uniform(-41.9, 41.9), rng.uniform(-23.9, 23.9)
        x = np.float32(rng.uniform(0.5, 9.0))
        y = np.float32(x * np.tan(np.radians(az)))
        z = np.float32(x * np.tan(np.radians(el)))
        if abs(13 + az / 3 - round(13 + az / 3)) < 1e-3 or abs(7 - el / 3 - round(7 - el / 3)) < 1e-3:
            continue
        assert o.L.dspo_in_pyramids_area(o.h, x, y, z) == 1
        assert o.L.dspo_pyramid_h(o.h, x, y, z) == int(np.ceil(13 + az / 3))
        assert o.L.dspo_pyramid_v(o.h, x, y, z) == int(np.ceil(7 - el / 3))
    # outside the 84 x 48 degree wedge
    assert o.L.dspo_in_pyramids_area(o.h, 1.0, 1.0, 0.0) == 0
    assert o.L.dspo_in_pyramids_area(o.h, 1.0, 0.0, 0.5) == 0
    assert o.L.dspo_in_pyramids_area(o.h, -1.0, 0.0, 0.0) == 0
    o.close()


def test_voxel_index_boundary_rules(orc):
    # Appendix A-10: closed boundary (|p| >= half is outside), idx = z*ny*nx + y*nx + x
    o = orc.Oracle()
    idx = C.c_int()
    hx, hy, hz = common.half_extent(o.cfg)
    assert o.L.dspo_voxel_index(o.h, hx, 0.0, 0.0, C.byref(idx)) == 0
    assert o.L.dspo_voxel_index(o.h, -hx, 0.0, 0.0, C.byref(idx)) == 0
    assert o.L.dspo_voxel_index(o.h, 0.0, 0.0, hz, C.byref(idx)) == 0
    assert o.L.dspo_voxel_index(o.h, np.nextafter(np.float32(hx), np.float32(0)), 0.0, 0.0, C.byref(idx)) == 1
    assert idx.value % 66 == 65
    assert o.L.dspo_voxel_index(o.h, np.nextafter(np.float32(-hx), np.float32(0)),
                                np.nextafter(np.float32(-hy), np.float32(0)),
                                np.nextafter(np.float32(-hz), np.float32(0)), C.byref(idx)) == 1
    assert idx.value == 0
    x, y, z = C.c_float(), C.c_float(), C.c_float()
    o.L.dspo_voxel_center(o.h, 0, C.byref(x), C.byref(y), C.byref(z))
    assert abs(x.value - (-hx + 0.075)) < 1e-6 and abs(z.value - (-hz + 0.075)) < 1e-6
    # centre -> index round trip over a sample of voxels
    for v in [0, 65, 66, 4355, 4356, 100000, o.V - 1]:
        o.L.dspo_voxel_center(o.h, v, C.byref(x), C.byref(y), C.byref(z))
        assert o.L.dspo_voxel_index(o.h, x.value, y.value, z.value, C.byref(idx)) == 1 and idx.value == v
    o.close()


def test_observation_overflow_rule(orc):
    # Appendix A-5: count saturates at 99, later points still count as valid and raise max range
    o = orc.Oracle()
    n = 150
    pts = np.zeros((n, 3), np.float32)
    pts[:, 0] = np.linspace(2.0, 5.0, n)
    pts[:, 1] = 0.01
    pts[:, 2] = 0.01
    valid = o.bin_points(pts)
    assert valid == n
    b = np.nonzero(o.obs_count)[0]
    assert b.size == 1 and o.obs_count[b[0]] == 99
    assert abs(o.obs_max_length[b[0]] - np.sqrt(25 + 2e-4)) < 1e-4
    assert np.allclose(o.obs[b[0], :99, 0], pts[:99, 0])
    lam = o.L.dspo_expected_newborn(o.h)
    assert abs(lam - 1e-4 * n * 20) < 1e-6
    o.close()


def test_empty_voxel_static_split_is_three(orc):
    # Appendix A-8: empty voxel -> 0/0 -> (int)NaN -> max(3, .) = 3 static children out of 20
    o = orc.Oracle()
    p, v, r = common.tables(5)
    o.set_tables(p * 0, v, r)  # zero position noise: children sit on the source point
    o.L.dspo_set_current_position(o.h, 0, 0, 0)
    pts = np.array([[3.0, 0.2, 0.1]], np.float32)
    o.bin_points(pts)
    o.map_update()
    src = np.zeros(1, orc.VPOINT_DTYPE)
    src["x"], src["y"], src["z"] = 3.0, 0.2, 0.1
    src["nx"], src["ny"], src["nz"] = 0.5, 0.0, 0.0
    src["intensity"] = 0.5
    o.set_birth_cloud(src)
    o.add_newborn()
    _, _, rec = o.export_sparse()
    assert len(rec) == 18  # 20 children, voxel capacity 18 (2*9)
    assert (rec[:, 0] == 15).all()
    assert (rec[:3, 1:4] == 0).all()             # first 3 static
    assert (np.abs(rec[3:16, 1]) > 0).all()      # children 3..15: cluster velocity + 4*N(0,sigma_v)
    assert (rec[:, 3] == 0).all()                # vz forced to 0
    assert o.cursors()[0] == 60                  # 3 position draws per child, always consumed
    o.close()


def test_stage_invariants_small_trajectory(orc):
    """Appendix C invariants on a 12-frame run: flags, voxel membership, mass bookkeeping."""
    cfg = orc.make_config(nx=40, ny=40, nz=20, ppv=12)
    o = orc.Oracle(cfg)
    p, v, r = common.tables(11)
    o.set_tables(p, v, r)
    o.L.dspo_use_velocity_estimator(o.h, 2)
    pts = common.wall_cloud(3, n_side=40, dist=2.2, half_w=1.8, half_h=0.9)
    idx = C.c_int()
    for f in range(12):
        assert o.update(pts, (0.01 * f, 0.0, 0.002 * f), f / 30.0, (1, 0, 0, 0)) == 1
        voxel, slot, rec = o.export_sparse()
        assert set(np.unique(rec[:, 0]).tolist()) <= {np.float32(0.6), np.float32(1.0)}
        assert (rec[:, 3] == 0).all()
        for k in range(0, len(voxel), 97):
            assert o.L.dspo_voxel_index(o.h, rec[k, 4], rec[k, 5], rec[k, 6], C.byref(idx)) == 1
            assert idx.value == voxel[k]
        occ = o.results[:, 0].astype(np.float64).sum()
        assert abs(occ - rec[:, 7].astype(np.float64).sum()) < 1e-3 * max(1.0, occ)
        counts = np.bincount(voxel, minlength=o.V)
        assert counts.max() <= o.slots
        xyz, fut = o.get_occupancy_with_future(0.2)
        assert fut.shape == (o.V, 6) and (fut >= 0).all()
        assert o.results[:, 4:].sum() == 0
    assert o.L.dspo_count_live(o.h) > 1000
    # rejected frames leave state untouched (update() :193-208)
    before = o.particles.copy()
    assert o.update(pts, (0.0, 0.0, 0.0), 13 / 30.0, (1.5, 0, 0, 0)) == 0
    assert o.update(pts, (50.0, 0.0, 0.0), 13 / 30.0, (1, 0, 0, 0)) == 0
    assert o.update(pts, (0.1, 0.0, 0.0), 0.0, (1, 0, 0, 0)) == 0  # time going backwards
    assert np.array_equal(before, o.particles)
    o.close()


def test_resample_mass_conservation_and_counts(orc):
    cfg = orc.make_config(nx=10, ny=10, nz=6, ppv=8)
    o = orc.Oracle(cfg)
    rng = np.random.default_rng(2)
    half = common.half_extent(cfg)
    px, py, pz, vx, vy, w = common.random_particles(7, 3000, half, wlo=0.0005, whi=0.2)
    o.inject(px, py, pz, vx, vy, np.zeros_like(vx), w, 1.0)
    voxel0, _, rec0 = o.export_sparse()
    o.occupancy_resample()
    voxel1, _, rec1 = o.export_sparse()
    M = cfg.max_particle_num_voxel
    for v in np.unique(voxel0):
        a = rec0[voxel0 == v]
        alive = a[a[:, 7] >= np.float32(1e-3)]
        b = rec1[voxel1 == v]
        mass = alive[:, 7].astype(np.float64).sum()
        assert abs(mass - o.results[v, 0]) < 1e-5 * max(1, mass)
        assert abs(b[:, 7].astype(np.float64).sum() - mass) < 1e-4 * max(1, mass)
        if len(alive) < 5:
            assert len(b) == len(alive)
        else:
            assert len(b) <= min(len(alive), M)
    o.close()


def test_golden_regression_vectors(orc):
    """tests/golden/oracle_regression.json: outputs of THIS oracle on fixed seeds, committed so that
    later edits of the oracle cannot drift silently (generated by tests/golden/make_golden.py).
    They are regression vectors, not reference outputs."""
    path = os.path.join(GOLD, "oracle_regression.json")
    if not os.path.exists(path):
        pytest.skip("golden file not generated yet")
    from tests.golden import make_golden
    got = make_golden.compute(orc)
    want = json.load(open(path))
    for k in want:
        assert np.allclose(got[k], want[k], rtol=2e-5, atol=1e-6), k


def test_scene_generator_capture():
    """tests/golden/scene_capture.json: the benchmark's synthetic depth stream (point count, centroid, pose) on the
    CPU for three seeds -- the build's own generator is pinned (SURVEY 8(c) golden vectors, item 4)"""
    path = os.path.join(GOLD, "scene_capture.json")
    if not os.path.exists(path):
        pytest.skip("golden file not generated yet")
    from tests.golden import make_golden
    got, want = make_golden.scene_capture(), json.load(open(path))
    for seed, w in want.items():
        g = got[seed]
        assert abs(g["n"] - w["n"]) <= max(2, w["n"] // 500)      # libm differences may move a point across a leaf face
        assert np.allclose(g["mean"], w["mean"], atol=2e-3) and np.allclose(g["pos"], w["pos"]) and np.allclose(g["quat"], w["quat"])


def test_preprocess_voxel_grid_properties(orc):
    """cloudCallback pre-processing (src/map_sim_example.cpp:309-336), pcl::VoxelGrid restated: every output is the
    mean of the input points of one leaf, leaves come in ascending lattice order, crop is an open interval, the cap
    keeps the first points of that order"""
    rng = np.random.default_rng(3)
    pts = (rng.random((20000, 3), dtype=np.float32) - 0.5) * np.array([8.0, 5.0, 9.0], np.float32)
    pts[::97] = np.nan                                   # non-finite points are skipped
    leaf, half = 0.1, (4.95, 4.95, 3.0)
    out, leaves = orc.preprocess_cloud(pts, leaf, half, max_points=100000, swap_axes=True)
    fin = pts[np.isfinite(pts).all(1)]
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(fin * inv).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    key = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    uniq, inv_idx = np.unique(key, return_inverse=True)
    assert leaves == len(uniq)
    cen = np.zeros((len(uniq), 3), np.float64)
    np.add.at(cen, inv_idx, fin.astype(np.float64))
    cen /= np.bincount(inv_idx)[:, None]
    sw = np.stack([cen[:, 2], -cen[:, 0], -cen[:, 1]], 1)            # x = z, y = -x, z = -y (:321-323)
    keep = (np.abs(sw) < np.array(half)).all(1)
    assert len(out) == int(keep.sum())
    assert np.allclose(out, sw[keep], atol=2e-6)                      # same leaves in the same (ascending) order
    capped, _ = orc.preprocess_cloud(pts, leaf, half, max_points=500, swap_axes=True)
    assert len(capped) == 500 and np.array_equal(capped, out[:500])   # :332 stops at the cap
    empty, nl = orc.preprocess_cloud(np.zeros((0, 3), np.float32), leaf, half)
    assert len(empty) == 0 and nl == 0
    # a point exactly on the box face is outside (inRange uses > and <, :190-197)
    face = np.array([[0.0, 0.0, 4.95]], np.float32)                   # z_cam -> x = 4.95 = x_max
    o2, _ = orc.preprocess_cloud(face, leaf, half, swap_axes=True)
    assert len(o2) == 0


def test_workload_statistics_match_the_reference_probe(orc):
    """SURVEY 6.1 records workload statistics of the REAL reference header (66x66x40, M = 24, wavy wall ~3 m ahead filling
    the field of view, ~1300 observations, sensor 0.5 m/s forward with a +-5 cm bob, steady state after ~20 frames):
    ~39 k live particles, ~14 % of them change voxel per prediction, fullest voxel = 24 (the cap :993-994), resample copies
    of the order of 15 % of the live set, no particle lost to overflow.  The probe's exact scene is not recorded, so this is
    a SOFT pin: the oracle on a scene built from that description must land in the same regime (bounds below), and hit the
    exact invariants (fullest voxel == M, nothing lost)."""
    o = orc.Oracle(orc.make_config(nx=66, ny=66, nz=40, res=0.15, ppv=24))
    p, v, r = common.tables(7, n=2000003)
    o.set_tables(p, v, r)
    ys = np.arange(-3.2, 3.2, 0.1); zs = np.arange(-1.6, 1.6, 0.1)
    Y, Z = np.meshgrid(ys, zs)
    pts = np.stack([(3.0 + 0.2 * np.sin(2 * Y)).ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    stats, last = [], None
    for f in range(32):
        t = f / 30.0
        pos = (0.5 * t, 0.0, 0.05 * np.sin(2 * np.pi * t))
        last = last or pos
        dp = [np.float32(pos[i]) - np.float32(last[i]) for i in range(3)]
        last = pos
        o.L.dspo_set_current_position(o.h, *[float(x) for x in pos])
        n_obs = o.bin_points(pts)
        o.L.dspo_static_birth_cloud(o.h)
        o.predict(float(-dp[0]), float(-dp[1]), float(-dp[2]), 1 / 30.0 if f else 0.0)
        fl = o.particles[:, :, 0]
        live_in, moved = int((fl > 0.1).sum()), int((np.abs(fl - 7) < 0.1).sum())
        o.map_update(); o.add_newborn(); o.occupancy_resample()
        fl = o.particles[:, :, 0]
        stats.append((n_obs, live_in, moved, int((np.abs(fl - 0.6) < 0.05).sum()), int((fl > 0.1).sum()), int((fl > 0.1).sum(1).max())))
        o.L.dspo_clear_future(o.h)
    s = np.array(stats[22:], np.float64)
    assert 1000 < s[:, 0].mean() < 1700                       # ~1300 observations
    assert 25e3 < s[:, 1].mean() < 60e3                       # ~39 k live particles
    assert 0.09 < (s[:, 2] / s[:, 1]).mean() < 0.22           # ~14 % change voxel per step
    assert 0.05 < (s[:, 3] / s[:, 4]).mean() < 0.30           # copies: of the order of 15 % of the live set
    assert s[:, 5].max() == 24                                # fullest voxel after resampling = MAX_PARTICLE_NUM_VOXEL
    o.close()


# ---- the third-party algorithms of the velocity estimator (absent from /root/reference: PCL, munkres-cpp, Eigen) ----
# The oracle restates their PUBLISHED algorithms; these tests check the restatements against independent implementations
# that ship with the image (scipy), which is as far as they can be pinned without the libraries themselves.

def test_restated_munkres_is_a_minimum_cost_assignment(orc):
    """saebyn/munkres-cpp as used at include/dsp_dynamic.h:1474-1481 (only WHICH cells are assigned is read): the restated
    Hungarian algorithm assigns min(rows, cols) distinct cells at the minimum total cost -- the same total as
    scipy.optimize.linear_sum_assignment on square, wide, tall and tie-ridden matrices."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(7)
    shapes = [(1, 1), (1, 6), (6, 1), (4, 4), (3, 9), (9, 3), (17, 17), (12, 30), (30, 12), (64, 64)]
    for nr, nc in shapes:
        for kind in ("uniform", "ties", "distances"):
            if kind == "uniform":
                cost = rng.uniform(0, 10, (nr, nc)).astype(np.float32)
            elif kind == "ties":
                cost = rng.integers(0, 4, (nr, nc)).astype(np.float32)
            else:   # what the estimator feeds it: centre distances of clusters, gated pairs at a large constant (:1459-1472)
                a, b = rng.uniform(-5, 5, (nr, 3)), rng.uniform(-5, 5, (nc, 3))
                cost = np.linalg.norm(a[:, None] - b[None], axis=2).astype(np.float32)
                cost[cost > 6] = 1000.0
            assign = orc.hungarian(cost)
            rows = np.nonzero(assign >= 0)[0]
            assert len(rows) == min(nr, nc), (nr, nc, kind)
            assert len(set(assign[rows].tolist())) == len(rows)          # one row per column
            assert (assign[rows] < nc).all()
            r, c = linear_sum_assignment(cost.astype(np.float64))
            want = cost.astype(np.float64)[r, c].sum()
            got = cost.astype(np.float64)[rows, assign[rows]].sum()
            assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), (nr, nc, kind, got, want)


def test_restated_euclidean_cluster_extraction_is_the_radius_graphs_components(orc):
    """pcl::EuclideanClusterExtraction as used at :1406-1417 (tolerance, min 5, max 10000 points): the region growing of the
    restatement yields exactly the connected components of the graph "distance <= tolerance" whose size is within the limits,
    largest first -- checked against scipy.sparse.csgraph.connected_components on blobs, chains and scattered points."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    rng = np.random.default_rng(11)
    for case in range(6):
        blobs = [rng.normal(rng.uniform(-4, 4, 3), rng.uniform(0.05, 0.3), (int(rng.integers(3, 120)), 3)) for _ in range(8)]
        chain = np.stack([np.linspace(-3, 3, 40), np.full(40, 5.0), np.zeros(40)], 1)    # neighbours 0.154 apart: one component at 0.2
        lone = rng.uniform(-6, 6, (60, 3))
        pts = np.concatenate(blobs + [chain, lone]).astype(np.float32)
        pts = pts[rng.permutation(len(pts))]
        tol = np.float32(0.2)
        for lo, hi in ((5, 10000), (5, 60), (1, 10000)):
            label, n = orc.euclidean_clusters(pts, tol, lo, hi)
            d = pts[:, None, :] - pts[None, :, :]
            d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]          # fp32, as the radius search
            i, j = np.nonzero(d2 <= tol * tol)
            _, comp = connected_components(coo_matrix((np.ones(len(i)), (i, j)), shape=(len(pts),) * 2), directed=False)
            sizes = np.bincount(comp)
            keep = (sizes[comp] >= lo) & (sizes[comp] <= hi)
            assert np.array_equal(label >= 0, keep), (case, lo, hi)
            assert n == int(((sizes >= lo) & (sizes <= hi)).sum())
            # same partition: a label maps to one component and back
            pairs = set(zip(label[keep].tolist(), comp[keep].tolist()))
            assert len(pairs) == n and len({a for a, _ in pairs}) == n and len({b for _, b in pairs}) == n
            # largest first
            got_sizes = [int((label == c).sum()) for c in range(n)]
            assert got_sizes == sorted(got_sizes, reverse=True)


def test_restated_quaternion_rotation_is_the_rotation(orc):
    """rotateVectorByQuaternion :1303-1322 (Eigen's q * p * q^-1): equal to scipy's Rotation for unit quaternions (1e-5),
    and the identity / the half turns are exact."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    q = rng.normal(size=(200, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.uniform(-10, 10, (200, 3)).astype(np.float32)
    out = np.zeros(3, np.float32)
    for k in range(200):
        qw = q[k].astype(np.float32)     # (w, x, y, z)
        orc.lib().dspo_rotate_vector(v[k].ctypes.data_as(C.c_void_p), qw.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        want = Rotation.from_quat([qw[1], qw[2], qw[3], qw[0]]).apply(v[k].astype(np.float64))
        assert np.allclose(out, want, atol=2e-5 * max(1.0, np.abs(v[k]).max())), (k, out, want)
    for qq, f in (((1, 0, 0, 0), lambda a: a), ((0, 0, 0, 1), lambda a: np.array([-a[0], -a[1], a[2]])),
                  ((0, 1, 0, 0), lambda a: np.array([a[0], -a[1], -a[2]]))):
        qw = np.array(qq, np.float32)
        orc.lib().dspo_rotate_vector(v[0].ctypes.data_as(C.c_void_p), qw.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, f(v[0]).astype(np.float32))


def test_strict_and_fastmath_builds_of_the_oracle_span_an_envelope(orc):
    """the reference's own -O2 and -O3 -ffast-math builds drift apart (1.2e-4 in mass after one frame, 1.8e-3 after
    twelve [probe]); the oracle is built both ways and the GPU trajectory test (test_gpu_round5.py) requires the HIP map to sit
    inside what the two builds -- and a one-ulp nudge of the newborn weight -- span.  Here, without a GPU: the table exists for
    both scenes, the two builds agree on the first frame to the order the probe saw, and stay within the envelope SURVEY states
    for twelve frames (mass 0.5 %, Jaccard 0.98) through frame 10."""
    from tests import envelope
    for scene in envelope.SCENES:
        tab = envelope.run(orc, scene)
        checks = envelope.SCENES[scene]["checks"]
        assert sorted(tab) == list(checks)
        first = tab[checks[0]]
        assert first["strict_vs_fast"]["mass_rel"] < 1e-3 and first["strict_vs_fast"]["jaccard"] > 0.995, (scene, first)
        assert first["n_live"]["strict"] == first["n_live"]["fast"] > 1000        # [probe]: N_live equal at frames 0-1
        for fr in checks:
            row = tab[fr]
            print(scene, "frame", fr, {k: (round(v["mass_rel"], 6), round(v["jaccard"], 4)) for k, v in row.items() if k != "n_live"},
                  row["n_live"])
            if fr <= 10:
                assert row["strict_vs_fast"]["mass_rel"] < 5e-3 and row["strict_vs_fast"]["jaccard"] >= 0.97, (scene, fr, row)
            b = envelope.bars(row, fr == checks[0])
            assert b["mass_rel"] >= envelope.STATED["mass"] and b["jaccard"] <= 0.999
