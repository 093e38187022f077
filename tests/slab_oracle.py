"""Oracle-backed Z-slab backend (TEST INFRASTRUCTURE): emulates one slab of the sharded map with
a full-grid CPU oracle that only ever holds the particles of its own layers.  Used by the gloo
world_size-2 test of dsp-map_amd/sharded.py's orchestration."""
import ctypes as C

import numpy as np
import torch


class OracleSlab:
    def __init__(self, orc, cfg_kwargs, z_lo, z_hi, tables):
        self.O = orc
        self.o = orc.Oracle(orc.make_config(**cfg_kwargs))
        self.o.set_tables(*tables)
        self.o.L.dspo_use_velocity_estimator(self.o.h, 0)
        self.z_lo, self.z_hi = z_lo, z_hi
        self.layer = self.o.cfg.nx * self.o.cfg.ny
        self.have_last = False
        self.nstatic = None
        self._ns_keep = None

    # -- helpers
    def _layer_of(self, voxel):
        return voxel // self.layer

    def _remove(self, voxel, slot):
        o = self.o
        p = o.particles
        p[voxel, slot, 0] = 0.0
        pl = o.pyramid_lists
        key = pl[:, :, 1].astype(np.int64) * o.slots + pl[:, :, 2]
        gone = np.isin(key, voxel.astype(np.int64) * o.slots + slot) & ((pl[:, :, 0] & 1) == 1)
        pl[:, :, 0][gone] = 0

    # -- split-phase frame
    def begin(self, pts, pos, stamp, quat, birth=None):
        o = self.o
        pts = np.ascontiguousarray(pts.cpu().numpy() if isinstance(pts, torch.Tensor) else pts, np.float32)
        if not self.have_last:
            self.last_p, self.last_t, self.have_last = tuple(pos), stamp, True
        dp = [np.float32(pos[i]) - np.float32(self.last_p[i]) for i in range(3)]
        dt = np.float32(stamp - self.last_t)
        if any(abs(q) > 1.001 for q in quat) or any(abs(x) > 10 for x in dp) or dt < 0 or dt > 10:
            return 0
        self.last_p, self.last_t = tuple(pos), stamp
        o.L.dspo_set_current_position(o.h, *[float(np.float32(x)) for x in pos])
        o.bin_points(pts, quat)
        o.L.dspo_static_birth_cloud(o.h)  # every rank: all in-FOV points are static birth sources
        self.n_birth = len(o.get_birth_cloud())
        o.predict(float(-dp[0]), float(-dp[1]), float(-dp[2]), float(dt))
        return 1

    def export(self, direction):
        o = self.o
        voxel, slot, rec = o.export_sparse()
        lay = self._layer_of(voxel)
        sel = lay >= self.z_hi if direction > 0 else lay < self.z_lo
        out = np.zeros((int(sel.sum()), 8), np.float32)
        if sel.any():
            out[:, 0] = voxel[sel].astype(np.int32).view(np.float32)
            out[:, 1:3] = rec[sel][:, 1:3]
            out[:, 3:6] = rec[sel][:, 4:7]
            out[:, 6] = rec[sel][:, 7]
            self._remove(voxel[sel], slot[sel])
        return torch.from_numpy(out)

    def import_(self, rec):
        o = self.o
        rec = rec.cpu().numpy() if isinstance(rec, torch.Tensor) else rec
        p = o.particles
        pl = o.pyramid_lists
        for r in rec:
            v = int(np.float32(r[0]).view(np.int32))
            if not (self.z_lo <= v // self.layer < self.z_hi):
                continue
            free = np.nonzero(p[v, :, 0] < 0.1)[0]
            if free.size == 0:
                continue
            s = int(free[0])
            p[v, s, :8] = (7.0, r[1], r[2], 0.0, r[3], r[4], r[5], r[6])
            x, y, z = float(r[3]), float(r[4]), float(r[5])
            if o.L.dspo_in_pyramids_area(o.h, x, y, z):
                b = o.L.dspo_pyramid_h(o.h, x, y, z) * (o.cfg.half_fov_v * 2 // o.cfg.angle_resolution) + \
                    o.L.dspo_pyramid_v(o.h, x, y, z)
                f = np.nonzero(pl[b, :, 0] == 0)[0]
                if f.size:
                    pl[b, f[0]] = (1, v, s)
                else:
                    p[v, s, 0] = 0.0

    def ck_partial(self):
        self.o.L.dspo_map_update_ck(self.o.h)
        self.ck = torch.from_numpy(self.o.obs[:, :, 3].copy().reshape(-1))
        return self.ck

    def weights_and_split(self):
        o = self.o
        o.obs[:, :, 3] = self.ck.numpy().reshape(o.NP, 100)
        o.L.dspo_map_update_weights(o.h)
        ns = np.zeros(max(self.n_birth, 1), np.int32)
        o.L.dspo_compute_nstatic(o.h, ns.ctypes.data_as(C.c_void_p))
        # a source whose voxel belongs to another slab is "empty" here: contribute 0, the owner decides
        src = o.get_birth_cloud()
        idx = C.c_int()
        cur = self.last_p
        for i in range(self.n_birth):
            ok = o.L.dspo_voxel_index(o.h, float(np.float32(src["x"][i]) - np.float32(cur[0])),
                                      float(np.float32(src["y"][i]) - np.float32(cur[1])),
                                      float(np.float32(src["z"][i]) - np.float32(cur[2])), C.byref(idx))
            if not ok or not (self.z_lo <= idx.value // self.layer < self.z_hi):
                ns[i] = 0
        self.nstatic = torch.from_numpy(ns[:self.n_birth])
        return self.nstatic

    def finish(self):
        o = self.o
        self._ns_keep = np.ascontiguousarray(self.nstatic.numpy(), np.int32)
        o.L.dspo_set_nstatic_override(o.h, self._ns_keep.ctypes.data_as(C.c_void_p))
        o.add_newborn()
        o.L.dspo_set_nstatic_override(o.h, None)
        voxel, slot, rec = o.export_sparse()
        lay = self._layer_of(voxel)
        out = (lay < self.z_lo) | (lay >= self.z_hi)
        if out.any():
            o.particles[voxel[out], slot[out], 0] = 0.0  # children that landed in another slab
        o.occupancy_resample()

    def results(self):
        r = self.o.results
        return r[self.z_lo * self.layer:self.z_hi * self.layer, :4].copy()

    def sync(self):
        pass
