"""How far up the slot rows the particles of a saturated map sit (what k_predict's position write-back pays for): live particles per
voxel against occupied ROWS per tile.  A row of a tile (slot r of its 64 voxels, 768 bytes of positions) is written back in whole 64-byte
chunks wherever one of the ~5 lanes of a chunk is live, so the sweep's write traffic follows the rows in use, not the particles.
   python tools/slot_spread.py [C_sat] [frames]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import dsp_map_amd as D  # noqa: E402
wn = sys.argv[1] if len(sys.argv) > 1 else "C_sat"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 13
w = bench.WORKLOADS[wn]
scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
m = D.DSPMap(D.make_config(nx=w["nx"], ny=w["ny"], nz=w["nz"], res=w["res"], ppv=w["ppv"], seed=1234))
m.L.dspmap_init_device(m.h)
m.seed_uniform(w["ppv"], 0.01, 99)
sc = scene_mod.CorridorScene(w["nx"] * w["res"], w["ny"] * w["res"], w["nz"] * w["res"], seed=1234, device="cuda", scale=1.0 if w["res"] >= 0.15 else 1.33)
for f in range(nf):
    pts, pos, quat = sc.frame(f / 30.0)
    assert m.update_device(pts.data_ptr(), pts.shape[0], pos, f / 30.0, quat) == 1
    m.clearOccupancyMapPrediction()
    if f in (0, nf - 1):
        vox, slot, rec = m.export_state()
        nx, ny, nz = w["nx"], w["ny"], w["nz"]
        x, y, z = vox % nx, (vox // nx) % ny, vox // (nx * ny)
        tile = ((z >> 2) * ((ny + 3) // 4) + (y >> 2)) * ((nx + 3) // 4) + (x >> 2)
        ntiles = ((nx + 3) // 4) * ((ny + 3) // 4) * ((nz + 3) // 4)
        rows = np.unique(tile.astype(np.int64) * 256 + slot).size
        chunks = np.unique((tile.astype(np.int64) * 256 + slot) * 16 + (((z & 3) * 16 + (y & 3) * 4 + (x & 3)) * 12 // 64)).size
        per_voxel = len(vox) / (nx * ny * nz)
        print("%s after frame %d: %d live, %.2f per voxel; occupied rows per tile %.2f (a bottom-packed map would use %.2f); 64-byte position chunks "
              "with a live lane %.2f M = %.1f MB against %.1f MB of live positions (x %.2f); particles in slots >= %d: %.1f %%; highest slot in use %d"
              % (wn, f, len(vox), per_voxel, rows / ntiles, per_voxel, chunks / 1e6, chunks * 64 / 1e6, len(vox) * 12 / 1e6, chunks * 64 / (len(vox) * 12.0),
                 w["ppv"], 100.0 * (slot >= w["ppv"]).mean(), int(slot.max())))
        hist = np.bincount(slot, minlength=2 * w["ppv"])
        print("   particles by slot:", " ".join("%d" % (h // 1000) for h in hist), "(thousands)")
m.close()
