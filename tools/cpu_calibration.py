"""BASELINE.md section 3, calibration of the CPU column: the oracle (oracle/dsp_oracle.c, the reference's flags, one thread)
timed in THIS container on a scene built from the description of SURVEY 6's [probe] of the real reference header (66x66x40,
24 particles per voxel, wavy wall ~3 m ahead, ~1300 observations, sensor 0.5 m/s forward), next to the probe's own number
(100 ms per update() on the survey host, a Xeon at 2.1 GHz; the probe's exact cloud is not recorded).  The reference itself
cannot be rebuilt here (Eigen / PCL / munkres-cpp are absent and stand-ins are not allowed), so this relates the PORT's time
to the REFERENCE's on the same class of machine; bench.py's `cpu_baseline` is the same port on the GPU box's host cores."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as orc  # noqa: E402
from tests import common  # noqa: E402


def main():
    orc.build()
    o = orc.Oracle(orc.make_config(nx=66, ny=66, nz=40, res=0.15, ppv=24), fast=True)
    p, v, r = common.tables(7, n=2000003)
    o.set_tables(p, v, r)
    o.L.dspo_use_velocity_estimator(o.h, 1)
    ys = np.arange(-3.2, 3.2, 0.1); zs = np.arange(-1.6, 1.6, 0.1)
    Y, Z = np.meshgrid(ys, zs)
    pts = np.stack([(3.0 + 0.2 * np.sin(2 * Y)).ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    ts = []
    for f in range(60):
        t = f / 30.0
        pos = (0.5 * t, 0.0, 0.05 * np.sin(2 * np.pi * t))
        t0 = time.perf_counter()
        assert o.update(pts, pos, t, (1.0, 0.0, 0.0, 0.0)) == 1
        ts.append(time.perf_counter() - t0)
        o.get_occupancy_with_future(0.2)
    live = int((o.particles[:, :, 0] > 0.1).sum())
    ms = np.array(ts[25:]) * 1e3
    cpu = [line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name")][:1]
    print("cpu:", cpu[0] if cpu else "?", "| points %d, live particles %d" % (len(pts), live))
    print("oracle port, reference flags, 1 thread: update() mean %.1f ms, median %.1f ms, p95 %.1f ms -> %.1f frames/s"
          % (ms.mean(), np.median(ms), np.percentile(ms, 95), 1e3 / ms.mean()))
    print("SURVEY 6 [probe] of the real header on a scene of this description: 100 ms (10.0 frames/s); ratio port / reference = %.2f"
          % (ms.mean() / 100.0))


if __name__ == "__main__":
    main()
