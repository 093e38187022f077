"""Generates tests/golden/oracle_regression.json from the CPU oracle (run in the build container:
`python -m tests.golden.make_golden`).  The reference itself cannot be built here, so these are
regression vectors of the restatement on seeded inputs -- see DESIGN.md 'oracle'."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common  # noqa: E402


def compute(orc):
    out = {}
    cfg = orc.make_config(nx=40, ny=40, nz=20, ppv=12)
    o = orc.Oracle(cfg)
    p, v, r = common.tables(11)
    o.set_tables(p, v, r)
    o.L.dspo_use_velocity_estimator(o.h, 2)
    pts = common.wall_cloud(3, n_side=40, dist=2.2, half_w=1.8, half_h=0.9)
    mass, live, occ = [], [], []
    for f in range(8):
        o.update(pts, (0.01 * f, 0.0, 0.002 * f), f / 30.0, common.EX_QUATS[f % 3])
        mass.append(float(o.results[:, 0].astype(np.float64).sum()))
        live.append(int(o.L.dspo_count_live(o.h)))
        xyz, fut = o.get_occupancy_with_future(0.2)
        occ.append(len(xyz))
    out["traj_mass"] = mass
    out["traj_live"] = live
    out["traj_occupied"] = occ
    out["traj_future_sum"] = [float(x) for x in fut.astype(np.float64).sum(0)]
    out["cursors"] = list(o.cursors())
    o.close()
    o = orc.Oracle()
    zs = np.linspace(-12, 12, 49).astype(np.float32)
    out["pdf_samples"] = [float(o.L.dspo_query_normal_pdf(o.h, float(z) * 0.1, 0.0, 0.1)) for z in zs]
    o.close()
    # the reference's two other headers as parameter sets
    for tag, kw in (("variant_multiple_neighbors", dict(nx=30, ny=30, nz=16, res=0.2, ppv=12, angle=1, half_fov_v=27, neighbor_n=2)),
                    ("variant_static", dict(nx=30, ny=30, nz=16, res=0.2, ppv=10, half_fov_v=27, pred_times=(0.05,),
                                            safe_factor=5, static_model=1))):
        o = orc.Oracle(orc.make_config(**kw))
        o.set_tables(p, v, r)
        o.L.dspo_use_velocity_estimator(o.h, 2)
        o.L.dspo_set_occlusion_margin(o.h, 0.2)
        pts2 = common.wall_cloud(5, n_side=40, dist=2.0, half_w=1.6, half_h=0.8)
        m2, l2 = [], []
        for f in range(4):
            o.update(pts2, (0.01 * f, 0.0, 0.0), f / 30.0, (1.0, 0.0, 0.0, 0.0))
            m2.append(float(o.results[:, 0].astype(np.float64).sum()))
            l2.append(int(o.L.dspo_count_live(o.h)))
        out[tag + "_mass"] = m2
        out[tag + "_live"] = l2
        o.close()
    # caller-side pre-processing (voxel-grid filter, axis swap, crop, cap)
    rng = np.random.default_rng(17)
    raw = ((rng.random((30000, 3)) - 0.5) * np.array([9.0, 5.0, 11.0])).astype(np.float32)
    flt, leaves = orc.preprocess_cloud(raw, 0.1, (4.95, 4.95, 3.0), max_points=5000, swap_axes=True)
    out["preprocess_counts"] = [int(len(flt)), int(leaves)]
    out["preprocess_sum"] = [float(x) for x in flt.astype(np.float64).sum(0)]
    out["preprocess_first"] = [float(x) for x in flt[:3].ravel()]
    return out


def scene_capture():
    """the synthetic depth stream of bench.py (dsp-map_amd/scene.py) on the CPU for 3 seeds: pins the generator"""
    import importlib
    scene = importlib.import_module("dsp-map_amd.scene")
    out = {}
    for seed in (1234, 1235, 1236):
        sc = scene.CorridorScene(9.9, 9.9, 6.0, seed=seed, device="cpu")
        pts, pos, quat = sc.frame(0.5)
        a = pts.numpy().astype(np.float64)
        out[str(seed)] = {"n": int(a.shape[0]), "mean": [float(x) for x in a.mean(0)], "pos": [float(x) for x in pos],
                          "quat": [float(x) for x in quat]}
    return out


if __name__ == "__main__":
    from oracle import oracle_py
    res = compute(oracle_py)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.json")
    json.dump(res, open(path, "w"), indent=1)
    print("wrote", path)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_capture.json")
    json.dump(scene_capture(), open(path, "w"), indent=1)
    print("wrote", path)
