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
    return out


if __name__ == "__main__":
    from oracle import oracle_py
    res = compute(oracle_py)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.json")
    json.dump(res, open(path, "w"), indent=1)
    print("wrote", path)
