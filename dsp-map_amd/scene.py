"""Synthetic depth-camera stream for benchmarks and tests.

The reference's only data set (street.bag) is an external download that is not
available offline, so the stream is generated: a 640x480 pinhole depth camera
(HFOV 90 deg, VFOV 60 deg, 30 Hz) flies at 0.5 m/s through a corridor world
(ground plane, two side walls, static boxes and walking pedestrians = vertical
cylinders; the world repeats every `period` metres along x so a run can be as
long as needed), with depth noise N(0, 0.01 d).  The pre-processing mirrors the
reference's caller, src/map_sim_example.cpp:309-336: back-projection, 0.1 m
voxel-grid centroid filter, camera->body axis swap (x = z_cam, y = -x_cam,
z = -y_cam; done implicitly by casting rays in the body frame), crop to the map
box, cap at 5000 points.

Everything is torch so it runs on the GPU when there is one (it is benchmark
scaffolding, outside every timed region) and on the CPU in tests.
"""
import math

import torch


class CorridorScene:
    def __init__(self, lx, ly, lz, seed=1234, period=8.0, width=640, height=480, hfov=90.0, vfov=60.0,
                 voxel_filter=0.1, max_points=5000, device=None, scale=1.0):
        self.lx, self.ly, self.lz = lx, ly, lz
        self.period = period * scale
        self.dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
        self.voxel_filter = voxel_filter
        self.max_points = max_points
        g = torch.Generator().manual_seed(seed)
        self.gen = torch.Generator(device=self.dev).manual_seed(seed + 1)
        # rays in the body frame (x forward, y left, z up)
        fx = (width / 2) / math.tan(math.radians(hfov / 2))
        fy = (height / 2) / math.tan(math.radians(vfov / 2))
        u = (torch.arange(width, dtype=torch.float32) + 0.5 - width / 2) / fx
        v = (torch.arange(height, dtype=torch.float32) + 0.5 - height / 2) / fy
        V, U = torch.meshgrid(v, u, indexing="ij")
        d = torch.stack([torch.ones_like(U), -U, -V], -1).reshape(-1, 3)
        self.dir_body = d.to(self.dev)           # not normalised: depth = x-distance (a depth image)
        self.wall_y = 0.45 * ly
        # 6 static boxes (0.5-1.5 m) per period, 3 pedestrians per period
        self.boxes = []
        for _ in range(6):
            sx, sy, sz = (0.5 + torch.rand(3, generator=g)).tolist()
            cx = float(torch.rand(1, generator=g)) * self.period
            cy = (float(torch.rand(1, generator=g)) * 2 - 1) * (self.wall_y - 0.8)
            if abs(cy) < 0.9:
                cy = math.copysign(0.9 + abs(cy), cy if cy != 0 else 1.0)  # keep the flight lane free
            self.boxes.append((cx * 1.0, cy, sx * scale, sy * scale, sz * scale))
        self.peds = []
        for _ in range(3):
            x0 = float(torch.rand(1, generator=g)) * self.period
            y0 = (float(torch.rand(1, generator=g)) * 2 - 1) * (self.wall_y - 1.0)
            sp = 1.0 + 0.5 * float(torch.rand(1, generator=g))
            ang = float(torch.rand(1, generator=g)) * 2 * math.pi
            self.peds.append((x0, y0, sp * math.cos(ang), sp * math.sin(ang)))
        self.ped_r, self.ped_h = 0.25 * scale, 1.7 * scale

    # ---- pose of SURVEY 8(d): z = 1.2 m, 0.5 m/s forward, yaw 10deg*sin(0.5t), bob 0.05*sin(t)
    @staticmethod
    def pose(t):
        yaw = math.radians(10.0) * math.sin(0.5 * t)
        pos = (0.5 * t, 0.0, 1.2 + 0.05 * math.sin(t))
        quat = (math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2))
        return pos, quat, yaw

    def _depth(self, t):
        """ray-cast the world; returns the ray parameter (= depth along body x) per pixel, inf if no hit"""
        pos, quat, yaw = self.pose(t)
        c, s = math.cos(yaw), math.sin(yaw)
        d = self.dir_body
        dx = c * d[:, 0] - s * d[:, 1]
        dy = s * d[:, 0] + c * d[:, 1]
        dz = d[:, 2]
        ox, oy, oz = pos
        inf = torch.full_like(dx, float("inf"))
        best = inf.clone()

        def upd(tt, ok):
            nonlocal best
            tt = torch.where(ok & (tt > 0.05), tt, inf)
            best = torch.minimum(best, tt)

        upd((0.0 - oz) / dz, dz < 0)                                   # ground z = 0
        upd((self.wall_y - oy) / dy, dy > 0)                            # side walls
        upd((-self.wall_y - oy) / dy, dy < 0)
        P = self.period
        cell = math.floor(ox / P)
        eps = 1e-9
        for (cx, cy, sx, sy, sz) in self.boxes:                         # axis-aligned boxes standing on the ground
            for k in (cell - 1, cell, cell + 1, cell + 2):
                lo = (cx + k * P - sx / 2, cy - sy / 2, 0.0)
                hi = (cx + k * P + sx / 2, cy + sy / 2, sz)
                t1x = (lo[0] - ox) / (dx + eps); t2x = (hi[0] - ox) / (dx + eps)
                t1y = (lo[1] - oy) / (dy + eps); t2y = (hi[1] - oy) / (dy + eps)
                t1z = (lo[2] - oz) / (dz + eps); t2z = (hi[2] - oz) / (dz + eps)
                tn = torch.maximum(torch.maximum(torch.minimum(t1x, t2x), torch.minimum(t1y, t2y)), torch.minimum(t1z, t2z))
                tf = torch.minimum(torch.minimum(torch.maximum(t1x, t2x), torch.maximum(t1y, t2y)), torch.maximum(t1z, t2z))
                upd(tn, tn <= tf)
        for (x0, y0, vx, vy) in self.peds:                              # pedestrians: vertical cylinders
            px = (x0 + vx * t) % P
            yspan = 2 * (self.wall_y - 0.6)
            py = ((y0 + vy * t + self.wall_y - 0.6) % (2 * yspan))
            py = (py if py < yspan else 2 * yspan - py) - (self.wall_y - 0.6)  # bounce between the walls
            for k in (cell - 1, cell, cell + 1, cell + 2):
                fx_, fy_ = ox - (px + k * P), oy - py
                a = dx * dx + dy * dy
                b = 2 * (fx_ * dx + fy_ * dy)
                cc = fx_ * fx_ + fy_ * fy_ - self.ped_r ** 2
                disc = b * b - 4 * a * cc
                ok = disc > 0
                tt = (-b - torch.sqrt(torch.clamp(disc, min=0))) / (2 * a + eps)
                zz = oz + tt * dz
                upd(tt, ok & (zz > 0) & (zz < self.ped_h))
        return best, pos, quat

    def raw(self, t):
        """-> (back-projected cloud (n,3) float32 in the SENSOR frame, before any filtering; pos, quat)"""
        depth, pos, quat = self._depth(t)
        ok = torch.isfinite(depth)
        noise = 1.0 + 0.01 * torch.randn(depth.shape, device=self.dev, generator=self.gen)
        p = self.dir_body[ok] * (depth[ok] * noise[ok]).unsqueeze(1)
        return p.contiguous().float(), pos, quat

    def frame(self, t):
        """-> (points (n,3) float32 on self.dev in the SENSOR frame, pos, quat)"""
        p, pos, quat = self.raw(t)
        # voxel-grid centroid filter (pcl::VoxelGrid, src/map_sim_example.cpp:313-317)
        cellf = torch.floor(p / self.voxel_filter).to(torch.int64)
        cellf = cellf - cellf.min(0).values
        dims = cellf.max(0).values + 1
        key = (cellf[:, 2] * dims[1] + cellf[:, 1]) * dims[0] + cellf[:, 0]
        uniq, inv = torch.unique(key, return_inverse=True)
        cnt = torch.zeros(uniq.numel(), device=self.dev).index_add_(0, inv, torch.ones_like(inv, dtype=torch.float32))
        cen = torch.zeros(uniq.numel(), 3, device=self.dev).index_add_(0, inv, p) / cnt.unsqueeze(1)
        # crop to the map box in sensor-frame coordinates (:321-326) and cap (:332)
        hx, hy, hz = self.lx / 2, self.ly / 2, self.lz / 2
        inside = (cen[:, 0] > -hx) & (cen[:, 0] < hx) & (cen[:, 1] > -hy) & (cen[:, 1] < hy) & \
                 (cen[:, 2] > -hz) & (cen[:, 2] < hz)
        cen = cen[inside][: self.max_points].contiguous().float()
        return cen, pos, quat
