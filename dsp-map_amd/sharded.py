"""Z-slab sharding of the map across the GPUs of one node.

One process per GPU.  Rank g owns the voxel layers [z_lo, z_hi) -- a contiguous
index range because the voxel index is z-major (reference include/dsp_dynamic.h:1081).
Every rank is fed the same cloud + pose.  Per frame (see include/dspmap.h,
"multi-GPU split-phase frame"):

    begin                      binning, prediction, local re-binning
    exchange                   particles that crossed a slab face -> neighbour rank (send/recv)
    ck_partial + all-reduce    per-observation sums Ck (pyramids cut across slabs)     [SUM, <= 177 kB]
    weights_and_split + a-r    n_static of each birth source, known to the owner only  [MAX, <= 20 kB]
    finish                     births (each rank keeps the children landing in its slab), resampling

With vz == 0 (LIMIT_MOVEMENT_IN_XY_PLANE, :661-663) only the sensor's own vertical motion moves
particles across layers, so the exchange is a thin boundary layer to the two neighbours; both
all-reduces are latency-bound, not link-bound.

The driver is written against two small interfaces so that the same orchestration runs on
RCCL (torch.distributed "nccl"), on gloo (CPU tests) and inside one process (several slabs on
one GPU, for tests):  a *slab backend* (HipSlab below; tests provide an oracle-based one) and a
*communicator* (TorchDistComm / LocalComm).
"""
import ctypes as C

import numpy as np
import torch


def slab_ranges(nz, world):
    """contiguous z-layer ranges, as even as possible"""
    base, rem = divmod(nz, world)
    out, z = [], 0
    for r in range(world):
        h = base + (1 if r < rem else 0)
        out.append((z, z + h))
        z += h
    return out


# --------------------------------------------------------------------------- communicators
class TorchDistComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests)"""

    def __init__(self, device):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self._into_tensor = dist.get_backend() == "nccl"

    ordered = False   # True: the slab runs on the torch stream the collectives are issued from (no host syncs needed)
    _side = None      # side stream + pinned buffer for the one host read of a frame (the gathered export counts)
    _host_counts = None

    def exchange(self, up, down, counts, overlap=None, cap=None):
        """send the first counts[0] rows of `up` to rank+1 and the first counts[1] rows of `down` to rank-1.
        `counts` is an int32[2] tensor on the communicator's device (it may still be in flight on the stream).
        Returns (from_below, from_above, (n_up, n_down)).  ONE host synchronisation: reading the gathered counts.
        `overlap()` is called once the gather is queued and before the host waits for it: work the caller enqueues there
        (on the current stream) runs while the counts travel to the host on a side stream."""
        dist = self.dist
        r, w = self.rank, self.world
        if self._into_tensor:
            stacked = torch.empty((w, 2), dtype=counts.dtype, device=counts.device)
            dist.all_gather_into_tensor(stacked.view(-1), counts)      # one collective, no stacking kernel
        else:
            allc = [torch.zeros_like(counts) for _ in range(w)]
            dist.all_gather(allc, counts)
            stacked = torch.stack(allc)
        if self.device.type == "cuda" and self.ordered:
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
                self._host_counts = torch.empty((w, 2), dtype=torch.int32).pin_memory()
            ready = torch.cuda.Event()
            ready.record()                                   # the gather + stack on the frame's stream
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                self._host_counts.copy_(stacked, non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(self._side)
            stacked.record_stream(self._side)
            if overlap is not None:
                overlap()
            landed.synchronize()
            host = self._host_counts.tolist()
        else:
            if overlap is not None:
                overlap()
            host = stacked.cpu().tolist()
        # every rank sees the same gathered counts: an export that overflowed its buffer (the surplus records were dropped
        # on the device) is detected by ALL ranks here, before any rank posts a receive its peer cannot fill
        # (`cap`: the export buffers' capacity in records, the same on every rank since the slabs share one configuration)
        worst = max(max(int(c[0]), int(c[1])) for c in host)
        if cap is not None and worst > cap:
            raise RuntimeError("slab export buffer too small: a rank exported %d records, capacity %d" % (worst, cap))
        n_up, n_down = int(host[r][0]), int(host[r][1])
        n_from_below = int(host[r - 1][0]) if r > 0 else 0       # what rank-1 sends up
        n_from_above = int(host[r + 1][1]) if r < w - 1 else 0   # what rank+1 sends down
        from_below = torch.empty((n_from_below, 8), dtype=torch.float32, device=self.device)
        from_above = torch.empty((n_from_above, 8), dtype=torch.float32, device=self.device)
        ops = []
        if r < w - 1 and n_up > 0:
            ops.append(dist.P2POp(dist.isend, up[:n_up].contiguous(), r + 1))
        if r > 0 and n_down > 0:
            ops.append(dist.P2POp(dist.isend, down[:n_down].contiguous(), r - 1))
        if n_from_below > 0:
            ops.append(dist.P2POp(dist.irecv, from_below, r - 1))
        if n_from_above > 0:
            ops.append(dist.P2POp(dist.irecv, from_above, r + 1))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            self._done()
        return from_below, from_above, (n_up, n_down)

    def _done(self):
        # slab on its own stream: make sure the collective has landed before the library reads its result.
        # ordered mode: the library runs on the stream the collective was issued from -- stream order suffices.
        if self.device.type == "cuda" and not self.ordered:
            torch.cuda.current_stream(self.device).synchronize()

    def allreduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self._done()

    def allreduce_max(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self._done()


class LocalComm:
    """all slabs live in this process: ShardedDSPMap then drives every slab itself"""
    rank, world = 0, 1


# --------------------------------------------------------------------------- HIP slab backend
class HipSlab:
    """one Z-slab on one GPU through the C ABI (dspmap_mgpu_*)"""

    def __init__(self, dsp, cfg_kwargs, z_lo, z_hi, device_index=0, point_cap=8192, example_params=True):
        self.D = dsp
        self.dev = torch.device("cuda", device_index)
        cfg = dsp.make_config(z_lo=z_lo, z_hi=z_hi, device=device_index, **cfg_kwargs)
        self.map = dsp.DSPMap(cfg, example_params=example_params)
        m = self.map
        self.z_lo, self.z_hi = z_lo, z_hi
        self.ck = torch.zeros(m.NP * 100, dtype=torch.int64, device=self.dev)   # fixed-point Ck sums (include/dspmap.h)
        self.nstatic = torch.zeros(point_cap, dtype=torch.int32, device=self.dev)
        self.point_cap = point_cap
        layer = cfg.nx * cfg.ny * m.slots
        self.exp_cap = max(4096, 2 * layer)
        self.exp_buf = {+1: torch.empty((self.exp_cap, 8), dtype=torch.float32, device=self.dev),
                        -1: torch.empty((self.exp_cap, 8), dtype=torch.float32, device=self.dev)}
        m._chk(m.L.dspmap_mgpu_bind(m.h, self.ck.data_ptr(), self.nstatic.data_ptr(), point_cap))
        self.n_birth = 0
        self.counts = torch.zeros(2, dtype=torch.int32, device=self.dev)
        self.ordered = False

    def adopt_stream(self, stream):
        """run the library on `stream` (a torch.cuda.Stream): everything the driver enqueues on that stream -- kernels
        of this slab, torch ops, RCCL collectives -- is then ordered by the stream and needs no host synchronisation"""
        self.map.sync()
        self.map._chk(self.map.L.dspmap_set_stream(self.map.h, stream.cuda_stream))
        self.ordered = True

    def export_both(self):
        """both exports without a host round trip: (up buffer, down buffer, device int32[2] counts)"""
        m = self.map
        m._chk(m.L.dspmap_mgpu_export_both(m.h, self.exp_buf[+1].data_ptr(), self.exp_buf[-1].data_ptr(), self.exp_cap,
                                           self.counts.data_ptr()))
        if not self.ordered:
            m.sync()
        return self.exp_buf[+1], self.exp_buf[-1], self.counts

    def note_exports(self, n_up, n_down):
        if n_up > self.exp_cap or n_down > self.exp_cap:
            raise RuntimeError("slab export buffer too small: %d / %d > %d" % (n_up, n_down, self.exp_cap))
        self.map._chk(self.map.L.dspmap_mgpu_set_export_counts(self.map.h, n_up, n_down))

    def begin(self, pts, pos, stamp, quat, birth=None):
        m = self.map
        pos_a = (C.c_float * 3)(*pos)
        q_a = (C.c_float * 4)(*quat)
        n = int(pts.shape[0])
        self.n_birth = n if birth is None else int(birth.shape[0])
        if birth is None:
            # a frame with an empty view re-uses the cloud of the last non-empty one (reference :1379-1381), which may be
            # longer than this frame's point count: the n_static exchange covers the longest synthesised cloud so far
            self.n_birth_hi = max(getattr(self, "n_birth_hi", 0), n)
            self.n_birth = self.n_birth_hi
        bptr = None if birth is None else birth.data_ptr()
        return m._chk(m.L.dspmap_mgpu_begin(m.h, n, pts.data_ptr(), self.n_birth, bptr, C.cast(pos_a, C.c_void_p),
                                            float(stamp), C.cast(q_a, C.c_void_p)))

    def export(self, direction):
        m = self.map
        n = C.c_int()
        buf = self.exp_buf[direction]
        m._chk(m.L.dspmap_mgpu_export(m.h, direction, buf.data_ptr(), self.exp_cap, C.byref(n)))
        return buf[:n.value]

    def import_(self, rec):
        if rec.shape[0]:
            rec = rec.contiguous()
            self.map._chk(self.map.L.dspmap_mgpu_import(self.map.h, int(rec.shape[0]), rec.data_ptr()))
            if not self.ordered:
                self.map.sync()  # `rec` may be a temporary (ordered mode: same stream as its allocation)

    def place_interior(self):
        """movers of the slab's interior are placed before the exchange (work for the GPU while the host sizes it)"""
        self.map._chk(self.map.L.dspmap_mgpu_place_interior(self.map.h))

    def ck_partial(self):
        self.map._chk(self.map.L.dspmap_mgpu_ck_partial(self.map.h))
        if not self.ordered:
            self.map.sync()
        return self.ck

    def weights_and_split(self):
        self.map._chk(self.map.L.dspmap_mgpu_weights_and_split(self.map.h))
        if not self.ordered:
            self.map.sync()
        return self.nstatic[:self.n_birth]

    def finish(self):
        self.map._chk(self.map.L.dspmap_mgpu_finish(self.map.h))

    def results(self):
        return self.map.results()

    def sync(self):
        self.map.sync()


# --------------------------------------------------------------------------- driver
class ShardedDSPMap:
    """update() over the slabs this process owns (one per rank with TorchDistComm; all with LocalComm)"""

    def __init__(self, slabs, comm):
        self.slabs = list(slabs)
        self.comm = comm
        self.local = isinstance(comm, LocalComm)
        self.stream = None
        if not self.local:
            assert len(self.slabs) == 1
            dev = getattr(comm, "device", None)
            if dev is not None and dev.type == "cuda" and hasattr(self.slabs[0], "adopt_stream"):
                # one stream for the slab's kernels, the torch ops and the collectives: no host syncs between phases
                self.stream = torch.cuda.Stream(dev)
                self.slabs[0].adopt_stream(self.stream)
                comm.ordered = True

    def update(self, pts, pos, stamp, quat, birth=None):
        if self.stream is None:
            return self._update(pts, pos, stamp, quat, birth)
        self.stream.wait_stream(torch.cuda.current_stream(self.stream.device))   # inputs produced on the caller's stream
        with torch.cuda.stream(self.stream):
            return self._update(pts, pos, stamp, quat, birth)

    @staticmethod
    def _export_both(slab):
        if hasattr(slab, "export_both"):
            return slab.export_both()
        up, down = slab.export(+1), slab.export(-1)
        return up, down, torch.tensor([up.shape[0], down.shape[0]], dtype=torch.int32, device=up.device)

    def _update(self, pts, pos, stamp, quat, birth=None):
        rcs = [s.begin(pts, pos, stamp, quat, birth) for s in self.slabs]
        if any(rc == 0 for rc in rcs):
            return 0
        # (1) particles that crossed a slab face
        if self.local:
            ups = [s.export(+1) for s in self.slabs]
            downs = [s.export(-1) for s in self.slabs]
            for s in self.slabs:
                if hasattr(s, "place_interior"):
                    s.place_interior()
            for i, s in enumerate(self.slabs):
                if i > 0:
                    s.import_(ups[i - 1])
                if i < len(self.slabs) - 1:
                    s.import_(downs[i + 1])
        else:
            slab = self.slabs[0]
            up, down, counts = self._export_both(slab)
            overlap = slab.place_interior if hasattr(slab, "place_interior") else None
            below, above, (n_up, n_down) = self.comm.exchange(up, down, counts, overlap, getattr(slab, "exp_cap", None))
            if hasattr(slab, "note_exports"):
                slab.note_exports(n_up, n_down)
            slab.import_(below)
            slab.import_(above)
        # (2) per-observation sums
        cks = [s.ck_partial() for s in self.slabs]
        if self.local:
            tot = cks[0].clone()
            for c in cks[1:]:
                tot += c
            for c in cks:
                c.copy_(tot)
        else:
            self.comm.allreduce_sum(cks[0])
        # (3) n_static of the birth sources
        ns = [s.weights_and_split() for s in self.slabs]
        if self.local:
            mx = ns[0].clone()
            for t in ns[1:]:
                mx = torch.maximum(mx, t)
            for t in ns:
                t.copy_(mx)
        else:
            self.comm.allreduce_max(ns[0])
        for s in self.slabs:
            s.finish()
        return 1

    def sync(self):
        for s in self.slabs:
            s.sync()


# --------------------------------------------------------------------------- the C++ driver (dspmap_dist.hip)
class CppShardedRank:
    """one rank of the C++ RCCL driver: dspmap_mgpu_update() issues every collective itself (ncclSend / ncclRecv pairs
    with rank +- 1, two ncclAllReduce) on the library's stream.  torch.distributed is only used ONCE, to hand rank 0's
    RCCL unique id to the other ranks."""

    def __init__(self, dsp, cfg_kwargs, world, rank, device_index=0, example_params=True, broadcast=None):
        self.D = dsp
        self.world, self.rank = world, rank
        z_lo, z_hi = slab_ranges(cfg_kwargs["nz"], world)[rank]
        cfg = dsp.make_config(z_lo=z_lo if world > 1 else 0, z_hi=z_hi if world > 1 else 0, device=device_index, **cfg_kwargs)
        self.map = dsp.DSPMap(cfg, example_params=example_params)
        m = self.map
        m.set_param(dsp.capi.P_TILING, 0)   # (the sharded frame's kernels take the slab in index order; a one-rank "slab" is the whole map)
        m._chk(m.L.dspmap_init_device(m.h))
        idb = (C.c_char * 128)()
        if rank == 0:
            m._chk(m.L.dspmap_mgpu_get_unique_id(C.cast(idb, C.c_void_p)))
        if world > 1:
            t = torch.frombuffer(bytearray(idb.raw), dtype=torch.uint8).clone()
            t = broadcast(t)          # caller-supplied: returns rank 0's tensor on every rank
            idb = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        m._chk(m.L.dspmap_mgpu_comm_init(m.h, world, rank, C.cast(idb, C.c_void_p)))

    def update(self, pts_dev, pos, stamp, quat, birth=None):
        m = self.map
        pos_a = (C.c_float * 3)(*pos)
        q_a = (C.c_float * 4)(*quat)
        nb = 0 if birth is None else int(birth.shape[0])
        bptr = None if birth is None else birth.data_ptr()
        return m._chk(m.L.dspmap_mgpu_update(m.h, int(pts_dev.shape[0]), pts_dev.data_ptr(), nb, bptr, C.cast(pos_a, C.c_void_p),
                                             float(stamp), C.cast(q_a, C.c_void_p)))

    def sync(self):
        self.map.sync()


class CppGroup:
    """several slabs in ONE process driven by the same C++ frame driver (dspmap_mgpu_group_update): device-to-device
    copies and small reduction kernels stand in for the collectives"""

    def __init__(self, dsp, cfg_kwargs, world, device_index=0, example_params=True, ranges=None):
        self.maps = []
        self.ranges = list(ranges) if ranges is not None else slab_ranges(cfg_kwargs["nz"], world)   # (ranges: slabs of unequal height)
        assert len(self.ranges) == world and self.ranges[0][0] == 0 and self.ranges[-1][1] == cfg_kwargs["nz"]
        for (z_lo, z_hi) in self.ranges:
            cfg = dsp.make_config(z_lo=z_lo, z_hi=z_hi, device=device_index, **cfg_kwargs)
            m = dsp.DSPMap(cfg, example_params=example_params)
            m.set_param(dsp.capi.P_TILING, 0)   # (index-order storage: see CppShardedRank)
            m._chk(m.L.dspmap_init_device(m.h))
            self.maps.append(m)
        self.L = self.maps[0].L
        self.handles = (C.c_void_p * world)(*[m.h for m in self.maps])
        self.created = False

    def create(self):
        """after the tables / parameters of the members are set"""
        rc = self.L.dspmap_mgpu_group_create(C.cast(self.handles, C.c_void_p), len(self.maps))
        if rc < 0:
            raise RuntimeError(self.L.dspmap_last_error(self.maps[0].h).decode())
        self.created = True

    def update(self, pts_dev, pos, stamp, quat):
        if not self.created:
            self.create()
        pos_a = (C.c_float * 3)(*pos)
        q_a = (C.c_float * 4)(*quat)
        rc = self.L.dspmap_mgpu_group_update(C.cast(self.handles, C.c_void_p), len(self.maps), int(pts_dev.shape[0]), pts_dev.data_ptr(),
                                             0, None, C.cast(pos_a, C.c_void_p), float(stamp), C.cast(q_a, C.c_void_p))
        if rc < 0:
            raise RuntimeError(self.L.dspmap_last_error(self.maps[0].h).decode())
        return rc

    GROUP_PHASES = ("begin", "exchange+import", "place", "select", "prepare+ck", "weights+split", "births+resample")

    def set_profiling(self, on):
        if not self.created:
            self.create()
        rc = self.L.dspmap_mgpu_group_set_profiling(C.cast(self.handles, C.c_void_p), len(self.maps), 1 if on else 0)
        if rc < 0:
            raise RuntimeError(self.L.dspmap_last_error(self.maps[0].h).decode())

    def phase_ms(self):
        """([slab or 'collectives'][phase] mean ms per frame, frames): HIP events around every phase of every slab"""
        n = len(self.maps)
        out = (C.c_float * ((n + 1) * len(self.GROUP_PHASES)))()
        nf = C.c_int()
        rc = self.L.dspmap_mgpu_group_get_phase_ms(C.cast(self.handles, C.c_void_p), n, out, C.byref(nf))
        if rc < 0:
            raise RuntimeError(self.L.dspmap_last_error(self.maps[0].h).decode())
        k = len(self.GROUP_PHASES)
        f = max(nf.value, 1)
        return [[out[i * k + p] / f for p in range(k)] for i in range(n + 1)], nf.value

    def sync(self):
        for m in self.maps:
            m.sync()

    def close(self):
        for m in self.maps:
            m.close()
