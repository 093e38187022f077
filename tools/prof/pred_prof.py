"""Where a tile's workgroup of k_predict spends its time.  The stamps are not in the tree (the kernel sources carry the fingerprint of the
committed PMC figures): `git apply tools/prof/pred_prof.patch`, build with DSPMAP_EXTRA_FLAGS=-DPRED_PROF, run on the MI355X box:
   python tools/prof/pred_prof.py [B|C_sat] [frames]      (stamps: wall clock, 100 MHz)"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dsp-map_amd"))
import build_ext
build_ext.build()
import torch  # noqa: E402
import dsp_map_amd as D  # noqa: E402
import bench  # noqa: E402
wn = sys.argv[1] if len(sys.argv) > 1 else "B"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 330
w = bench.WORKLOADS[wn]
scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
m = D.DSPMap(D.make_config(nx=w["nx"], ny=w["ny"], nz=w["nz"], res=w["res"], ppv=w["ppv"], seed=1234))
m.L.dspmap_init_device(m.h)
if w["sat"]:
    m.seed_uniform(w["ppv"], 0.01, 99)
else:
    m.set_param(D.capi.P_VELOCITY_ESTIMATOR, 2)
sc = scene_mod.CorridorScene(w["nx"] * w["res"], w["ny"] * w["res"], w["nz"] * w["res"], seed=1234, device="cuda", scale=1.0 if w["res"] >= 0.15 else 1.33)
for f in range(nf):
    pts, pos, quat = sc.frame(f / 30.0)
    assert m.update_device(pts.data_ptr(), pts.shape[0], pos, f / 30.0, quat) == 1
    m.clearOccupancyMapPrediction()
m.sync()
nt = min(m.tile_count(), 131072)
buf = np.zeros(nt * 8, np.int64)
m.L.dspmap_debug_pred_prof.restype = C.c_int
m.L.dspmap_debug_pred_prof(buf.ctypes.data_as(C.c_void_p), C.c_int(nt * 8))
t = buf.reshape(nt, 8).astype(np.float64)
t0 = t[:, 0][t[:, 0] > 0].min()
us = lambda x: (x - t0) / 100.0
full = t[:, 6] > 0            # tiles that ran to the end
part = (t[:, 0] > 0) & ~full
print(wn, "tiles", nt, "ran to the end", int(full.sum()), "left early", int(part.sum()))
print("  first workgroup enters at 0; last one enters at %.1f us; last stamp of the launch at %.1f us" % (us(t[:, 0].max()), us(t[:, 1:7].max())))
names = ["flags (scalar round trip)", "occupancy words + planes + barrier", "rows / cells (loads, advance, stores, notes)", "tail 1: pyramid registration", "tail 2: movers to inboxes", "statistics, mask write-back"]
d = np.diff(t[full][:, :7], axis=1) / 100.0
for k, nm in enumerate(names):
    print("  %-48s mean %6.2f us  median %6.2f  p95 %6.2f  max %6.2f" % (nm, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 95), d[:, k].max()))
print("  %-48s mean %6.2f us  max %6.2f" % ("whole tile", d.sum(1).mean(), d.sum(1).max()))
e = us(t[full][:, 0]); x = us(t[full][:, 6])
print("  full tiles enter between %.1f and %.1f us, end between %.1f and %.1f us" % (e.min(), e.max(), x.min(), x.max()))
late = np.argsort(-x)[:5]
info = buf.reshape(nt, 8)[full][:, 7]
print("  the five tiles that end last: enter", np.round(e[late], 1), "end", np.round(x[late], 1), "grid index", (info[late] & 0xffffff), "compact-cell path", ((info[late] >> 24) & 1), "live (wave 0's share)", (info[late] >> 32))
dn = ((info >> 24) & 1) == 1
rows = d[:, 2]
for nm, sel in (("compact-cell path", dn), ("row path", ~dn)):
    if sel.any():
        print("  %-18s %5d tiles: rows / cells phase mean %.2f us  p95 %.2f  max %.2f; wave 0's live count mean %.0f max %d" % (nm, int(sel.sum()), rows[sel].mean(), np.percentile(rows[sel], 95), rows[sel].max(), (info[sel] >> 32).mean(), int((info[sel] >> 32).max())))
