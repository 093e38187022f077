"""Where a tile's workgroup of k_place spends its cycles (build with DSPMAP_EXTRA_FLAGS=-DPLACE_PROF; run on the MI355X box):
   python tools/prof/place_prof.py [C_sat|E_sat]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dsp-map_amd"))
import build_ext
build_ext.build()
import dsp_map_amd as D
import bench
wn = sys.argv[1] if len(sys.argv) > 1 else "C_sat"
w = bench.WORKLOADS[wn]
scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
m = D.DSPMap(D.make_config(nx=w["nx"], ny=w["ny"], nz=w["nz"], res=w["res"], ppv=w["ppv"], seed=1234))
m.L.dspmap_init_device(m.h)
m.seed_uniform(w["ppv"], 0.01, 99)
sc = scene_mod.CorridorScene(w["nx"] * w["res"], w["ny"] * w["res"], w["nz"] * w["res"], seed=1234, device="cuda", scale=1.0 if w["res"] >= 0.15 else 1.33)
for f in range(8):
    pts, pos, quat = sc.frame(f / 30.0)
    assert m.update_device(pts.data_ptr(), pts.shape[0], pos, f / 30.0, quat) == 1
    m.clearOccupancyMapPrediction()
m.sync()
nt = min(m.tile_count(), 131072)
buf = np.zeros(nt * 8, np.int64)
m.L.dspmap_debug_place_prof.restype = C.c_int
m.L.dspmap_debug_place_prof(buf.ctypes.data_as(C.c_void_p), C.c_int(nt * 8))
t = buf.reshape(nt, 8)
ok = (t[:, 5] > t[:, 0]) & (t[:, 6] > 0)
d = np.diff(t[ok][:, :6], axis=1).astype(np.float64)
names = ["loads + occupancy words (round trip)", "count / scan / bucket keys (LDS)", "rank, slot, stores, pyramid registration", "counter reduction", "barrier + mask write-back", ]
print(wn, "tiles with arrivals:", int(ok.sum()), "of", nt, " mean arrivals per tile:", float(t[ok][:, 6].mean()))
for k, nm in enumerate(names):
    print("  %-45s mean %8.0f cycles  median %8.0f  p95 %8.0f" % (nm, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 95)))
print("  %-45s mean %8.0f cycles (%.2f us at 2.1 GHz)" % ("whole tile", d.sum(1).mean(), d.sum(1).mean() / 2100.0))
