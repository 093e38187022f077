import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import dsp_map_amd as dsp
from oracle import oracle_py as orc
from tests import common
from tests.test_gpu_parity import make_pair, gpu_state

def slab72():
    cfgkw = dict(nx=24, ny=24, nz=10, res=0.10, ppv=36)
    o, m = make_pair(dsp, orc, **cfgkw)
    half = common.half_extent(o.cfg)
    px, py, pz, vx, vy, w = common.random_particles(8, 60000, (half[0] * 0.5, half[1] * 0.5, half[2] * 0.9), wlo=0.0005, whi=0.06)
    n = common.inject_both(o, m, px, py, pz, vx, vy, w)
    vo, so, ro = o.export_sparse(); vg, sg, rg = gpu_state(m)
    print("import equal:", np.array_equal(vo, vg), np.array_equal(so, sg), np.array_equal(ro[:,1:], rg[:,1:]))
    o.occupancy_resample(); m.occupancy_resample()
    a = o.results[:,0]; b = m.results()[:,0]
    bad = np.nonzero(~np.isclose(a, b, rtol=1e-5, atol=1e-7))[0]
    print("resample-only mismatches:", len(bad), "of", (a>0).sum())
    cnt = np.bincount(vo, minlength=o.V)
    for v in bad[:8]:
        print(v, cnt[v], a[v], b[v])
    if len(bad):
        print("counts of bad voxels: min", cnt[bad].min(), "max", cnt[bad].max(), " good max", cnt[np.setdiff1d(np.nonzero(a>0)[0], bad)].max())

def traj():
    cfgkw = dict(nx=66, ny=66, nz=40, ppv=9)
    o, m = make_pair(dsp, orc, seed=9, **cfgkw)
    o2, m2 = make_pair(dsp, orc, seed=9, **cfgkw)
    o.L.dspo_use_velocity_estimator(o.h, 2)
    base = common.wall_cloud(77, n_side=50, dist=2.8, half_w=2.2, half_h=1.1)
    for f in range(30):
        t = f / 30.0
        pos = (0.5 * t, 0.05 * np.sin(t), 0.03 * np.sin(2 * t))
        yaw = np.radians(10.0) * np.sin(0.5 * t)
        q = (float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2)))
        pts = base.copy(); pts[:, 0] -= np.float32(0.5 * t)
        o.update(pts, pos, t, q); m.update(pts, pos, t, q); m2.update(pts, pos, t, q)
        a = o.results[:,0].astype(np.float64); b = m.results()[:,0].astype(np.float64); b2 = m2.results()[:,0].astype(np.float64)
        occd = (a>0.02)|(b>0.02)
        c = m.counters()
        so_, sg_ = a > 0.2, b > 0.2
        jac = (so_ & sg_).sum() / max(1, (so_ | sg_).sum())
        print("jac %.4f nocc %d" % (jac, so_.sum()), end=" ")
        print("f%2d mass o %.3f g %.3f g2 %.3f | live o %d g %d | frac|d|>0.02: o-g %.4f g-g2 %.4f (of occupied %.3f / %.3f) max %.3f | born %d dropped %d vfull %d pfull %d" % (
            f, a.sum(), b.sum(), b2.sum(), o.L.dspo_count_live(o.h), c["n_live_out"],
            (np.abs(a-b)>0.02).mean(), (np.abs(b-b2)>0.02).mean(), (np.abs(a-b)>0.02)[occd].mean(), (np.abs(b-b2)>0.02)[occd].mean(), np.abs(a-b).max(),
            c["n_born"], c["n_born_dropped"], c["n_voxel_full"], c["n_pyramid_full"]))
        o.get_occupancy_with_future(0.2); m.getOccupancyMapWithFutureStatus(0.2); m2.getOccupancyMapWithFutureStatus(0.2)

traj()
