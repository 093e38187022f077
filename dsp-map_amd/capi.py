"""ctypes binding of libdspmap_hip.so (include/dspmap.h).

Host-side mirror of the reference's `class DSPMap` surface
(include/dsp_dynamic.h:142-446,1550-1584 of g-ch/DSP-map): same method names,
argument meaning and return contract, forwarding to the C ABI.  There is no
CPU path in here: if the shared library is missing, or no HIP device is
usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdspmap_hip.so")

MAX_PRED = 16
OK, REJECTED = 1, 0

P_POSITION_STDDEV, P_VELOCITY_STDDEV, P_OBSERVATION_STDDEV, P_NEWBORN_WEIGHT, P_NEWBORN_NUMBER, \
    P_VOXEL_FILTER_RES, P_KAPPA, P_DETECTION, P_VELOCITY_ESTIMATOR, P_REGENERATE_TABLES, P_USE_GRAPH, P_OCCLUSION_MARGIN, \
    P_PAIR_CULL_SIGMAS, P_UPDATE_TIME, P_UPDATE_COUNTER, P_PLACE_SPLIT_TILES, P_FAST_DIVISION, P_SPARSE_SWEEP, P_ROLLOUT_INLINE, \
    P_RESAMPLE_WG_TILES, P_SWEEP_ALTERNATE, P_STATIC_TILE_SKIP, P_HOST_CLOUD_DIRECT, _P_REMOVED_24, P_ESTIMATOR_QUEUE, P_FRAME_BRANCHES, P_TILING, P_SIDE_PLACEMENT, P_RESAMPLE_SPLIT, P_TILE_BITMAPS = range(1, 31)


class Config(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("voxel_resolution", C.c_float),
        ("angle_resolution", C.c_int),
        ("max_particle_num_voxel", C.c_int),
        ("half_fov_h", C.c_int), ("half_fov_v", C.c_int),
        ("prediction_times", C.c_int),
        ("prediction_future_time", C.c_float * MAX_PRED),
        ("z_lo", C.c_int), ("z_hi", C.c_int),
        ("device", C.c_int),
        ("gaussian_table_size", C.c_int),
        ("seed", C.c_uint),
        ("pyramid_neighbor_n", C.c_int), ("safe_particle_factor", C.c_int), ("static_model", C.c_int),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "n_points_in", "n_valid", "n_obs", "n_live_in", "n_moved", "n_out_of_map", "n_voxel_full",
        "n_pyramid_full", "n_fov", "n_born", "n_born_dropped", "n_live_out", "n_exported_up",
        "n_exported_down", "n_reslotted", "n_overflow_inexact")] + [("newborn_weight", C.c_float), ("update_ms", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


VPOINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"),
                         ("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("intensity", "f4")])

# every symbol include/dspmap.h declares: name -> (restype, argtypes)
_P, _f, _i, _d = C.c_void_p, C.c_float, C.c_int, C.c_double
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)
SIGNATURES = {
    "dspmap_default_config": (None, [C.POINTER(Config)]),
    "dspmap_create": (_P, [C.POINTER(Config)]),
    "dspmap_destroy": (None, [_P]),
    "dspmap_init_device": (_i, [_P]),
    "dspmap_last_error": (C.c_char_p, [_P]),
    "dspmap_sync": (_i, [_P]),
    "dspmap_set_stream": (_i, [_P, _P]),
    "dspmap_set_param": (_i, [_P, _i, _d]),
    "dspmap_get_param": (_d, [_P, _i]),
    "dspmap_set_gaussian_tables": (_i, [_P, _P, _P, _i]),
    "dspmap_set_rand_table": (_i, [_P, _P, _i]),
    "dspmap_set_cursors": (_i, [_P, _i, _i, _i]),
    "dspmap_get_cursors": (_i, [_P, _ip, _ip, _ip]),
    "dspmap_update": (_i, [_P, _i, _i, _P, _f, _f, _f, _d, _f, _f, _f, _f]),
    "dspmap_update_device": (_i, [_P, _i, _P, _i, _P, _P, _d, _P]),
    "dspmap_set_birth_cloud": (_i, [_P, _P, _i]),
    "dspmap_get_birth_cloud": (_i, [_P, _P, _i, _ip]),
    "dspmap_get_occupancy": (_i, [_P, _f, _P, _i, _ip]),
    "dspmap_get_occupancy_with_future": (_i, [_P, _f, _P, _i, _ip, _P]),
    "dspmap_get_future": (_i, [_P, _P]),
    "dspmap_clear_future": (_i, [_P]),
    "dspmap_get_results": (_i, [_P, _P]),
    "dspmap_results_device": (_P, [_P]),
    "dspmap_future_device": (_P, [_P]),
    "dspmap_voxel_center": (None, [_P, _i, _fp, _fp, _fp]),
    "dspmap_point_voxel_index": (_i, [_P, _f, _f, _f, _ip]),
    "dspmap_voxel_num": (_i, [_P]),
    "dspmap_local_voxel_num": (_i, [_P]),
    "dspmap_local_voxel_base": (_i, [_P]),
    "dspmap_slots_per_voxel": (_i, [_P]),
    "dspmap_pyramid_num": (_i, [_P]),
    "dspmap_pyramid_capacity": (_i, [_P]),
    "dspmap_get_counters": (_i, [_P, C.POINTER(Counters)]),
    "dspmap_set_profiling": (_i, [_P, _i]),
    "dspmap_get_stage_ms": (_i, [_P, _fp, _ip]),
    "dspmap_get_event_overhead_ms": (_i, [_P, _fp]),
    "dspmap_debug_stream": (_i, [_P, _i, C.POINTER(C.c_longlong)]),
    "dspmap_debug_sweep_probe": (_i, [_P, _i, _i, _i, _i, _fp, C.POINTER(C.c_longlong)]),
    "dspmap_debug_tile_view": (_i, [_P, C.POINTER(C.c_int), _i]),
    "dspmap_debug_rollout_paths": (_i, [_P, C.POINTER(C.c_longlong)]),
    "dspmap_debug_estimator_queue": (_i, [_P, C.POINTER(C.c_longlong)]),
    "dspmap_debug_frame_branches": (_i, [_P, C.POINTER(C.c_longlong)]),
    "dspmap_debug_resample_split_frames": (C.c_longlong, [_P]),
    "dspmap_debug_estimator_path": (_i, [_P]),
    "dspmap_debug_tile_count": (_i, [_P]),
    "dspmap_debug_tile_of_voxels": (_i, [_P, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dspmap_debug_tile_moving": (_i, [_P, C.POINTER(C.c_int), _i]),
    "dspmap_debug_rdzv_publish": (_i, [C.c_char_p, C.c_char_p]),
    "dspmap_debug_rdzv_wait": (_i, [C.c_char_p, _i, C.c_char_p]),
    "dspmap_clear_state": (_i, [_P]),
    "dspmap_import_state": (_i, [_P, _i, _P, _P, _P]),
    "dspmap_export_state": (_i, [_P, _i, _P, _P, _P, _ip]),
    "dspmap_save_checkpoint": (_i, [_P, C.c_char_p]),
    "dspmap_load_checkpoint": (_i, [_P, C.c_char_p]),
    "dspmap_preprocess_cloud": (_i, [_P, _i, _P, _i, _f, _i, _i, _P, _ip, _ip]),
    "dspmap_add_random_particles": (_i, [_P, _i, _f]),
    "dspmap_seed_uniform_moving": (_i, [_P, _i, _f, C.c_uint, _f]),
    "dspmap_seed_uniform": (_i, [_P, _i, _f, C.c_uint]),
    "dspmap_stage_bin_points": (_i, [_P, _i, _i, _P, _f, _f, _f, _f]),
    "dspmap_set_current_position": (_i, [_P, _f, _f, _f]),
    "dspmap_stage_predict": (_i, [_P, _f, _f, _f, _f]),
    "dspmap_stage_update": (_i, [_P]),
    "dspmap_stage_birth": (_i, [_P]),
    "dspmap_stage_resample": (_i, [_P]),
    "dspmap_get_observations": (_i, [_P, _P, _P, _P, _fp]),
    "dspmap_set_expected_newborn": (_i, [_P, _f]),
    "dspmap_get_pyramid_counts": (_i, [_P, _P]),
    "dspmap_mgpu_bind": (_i, [_P, _P, _P, _i]),
    "dspmap_mgpu_place_interior": (_i, [_P]),
    "dspmap_mgpu_begin": (_i, [_P, _i, _P, _i, _P, _P, _d, _P]),
    "dspmap_mgpu_export": (_i, [_P, _i, _P, _i, _ip]),
    "dspmap_mgpu_export_both": (_i, [_P, _P, _P, _i, _P]),
    "dspmap_mgpu_set_export_counts": (_i, [_P, _i, _i]),
    "dspmap_mgpu_import": (_i, [_P, _i, _P]),
    "dspmap_mgpu_ck_partial": (_i, [_P]),
    "dspmap_mgpu_weights_and_split": (_i, [_P]),
    "dspmap_mgpu_finish": (_i, [_P]),
    "dspmap_mgpu_get_unique_id": (_i, [_P]),
    "dspmap_mgpu_comm_init": (_i, [_P, _i, _i, _P]),
    "dspmap_mgpu_comm_init_from_env": (_i, [_P]),
    "dspmap_mgpu_comm_destroy": (_i, [_P]),
    "dspmap_mgpu_update": (_i, [_P, _i, _P, _i, _P, _P, _d, _P]),
    "dspmap_mgpu_update_host": (_i, [_P, _i, _i, _P, _f, _f, _f, _d, _f, _f, _f, _f]),
    "dspmap_mgpu_message_records": (_i, [_P]),
    "dspmap_mgpu_group_create": (_i, [_P, _i]),
    "dspmap_mgpu_group_update": (_i, [_P, _i, _i, _P, _i, _P, _P, _d, _P]),
    "dspmap_mgpu_group_set_profiling": (_i, [_P, _i, _i]),
    "dspmap_mgpu_group_get_phase_ms": (_i, [_P, _i, _fp, _ip]),
}

_LIB = None


def load_library(path=None):
    """dlopen the HIP library and bind every declared symbol.  Raises if it is missing."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(
            "%s not found: build it with `python dsp-map_amd/build_ext.py` (hipcc, gfx950). "
            "There is no CPU fallback." % p)
    # Load order matters in a process that also uses PyTorch-ROCm: torch ships its own copy of the HIP / HSA
    # runtime, and the copy that is initialised first owns the device; a second one then reports "no device".
    # Import torch (when it is installed) BEFORE this library so that both share torch's runtime, whatever
    # order the caller imports things in.  A process without torch is unaffected.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001
        pass
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _LIB = lib
    return lib


def make_config(nx=66, ny=66, nz=40, res=0.15, ppv=9, angle=3, half_fov_h=42, half_fov_v=24,
                pred_times=(0.05, 0.2, 0.5, 1.0, 1.5, 2.0), z_lo=0, z_hi=0, device=-1,
                table_size=0, seed=0, neighbor_n=0, safe_factor=0, static_model=0):
    c = Config()
    c.pyramid_neighbor_n, c.safe_particle_factor, c.static_model = neighbor_n, safe_factor, static_model
    c.nx, c.ny, c.nz = nx, ny, nz
    c.voxel_resolution = res
    c.angle_resolution = angle
    c.max_particle_num_voxel = ppv
    c.half_fov_h, c.half_fov_v = half_fov_h, half_fov_v
    c.prediction_times = len(pred_times)
    for k, t in enumerate(pred_times):
        c.prediction_future_time[k] = t
    c.z_lo, c.z_hi = z_lo, z_hi
    c.device = device
    c.gaussian_table_size = table_size
    c.seed = seed
    return c


class DSPMapError(RuntimeError):
    pass


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DSPMap:
    """Mirror of the reference's DSPMap (dsp_dynamic.h:142) on top of the C ABI."""

    def __init__(self, cfg=None, example_params=True, init_particle_num=0, init_weight=0.01):
        self.L = load_library()
        self.cfg = cfg or make_config()
        self.h = self.L.dspmap_create(C.byref(self.cfg))
        if not self.h:
            raise DSPMapError("dspmap_create rejected the configuration")
        self.V = self.L.dspmap_voxel_num(self.h)
        self.V_local = self.L.dspmap_local_voxel_num(self.h)
        self.slots = self.L.dspmap_slots_per_voxel(self.h)
        self.NP = self.L.dspmap_pyramid_num(self.h)
        self.capp = self.L.dspmap_pyramid_capacity(self.h)
        self.T = self.cfg.prediction_times
        if example_params:  # src/map_sim_example.cpp:522-526
            self.setPredictionVariance(0.05, 0.05)
            self.setObservationStdDev(0.1)
            self.setNewBornParticleNumberofEachPoint(20)
            self.setNewBornParticleWeight(0.0001)
            self.setOriginalVoxelFilterResolution(0.1)
        if init_particle_num:
            self._chk(self.L.dspmap_add_random_particles(self.h, init_particle_num, init_weight))

    # -- plumbing
    def _chk(self, rc):
        if rc < 0:
            raise DSPMapError(self.L.dspmap_last_error(self.h).decode())
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.dspmap_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self.L.dspmap_sync(self.h))

    def set_param(self, key, value):
        self._chk(self.L.dspmap_set_param(self.h, key, float(value)))

    def get_param(self, key):
        return self.L.dspmap_get_param(self.h, key)

    def rollout_paths(self):
        """(variant, adds through k_rollout's LDS windows, single-atomic adds of k_rollout) of the last resampling stage;
        variant: bit 0 = k_resample_wg, bits 1-2 = rollout 0 inline / 1 k_rollout light / 2 k_rollout windows / 3 none"""
        out = (C.c_longlong * 3)()
        self._chk(self.L.dspmap_debug_rollout_paths(self.h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def estimator_queue(self):
        """(frames whose estimator ran on a queue of its own, hand-over word 0, hand-over word 1, give-ups, frames whose first birth kernel
        had to wait for the birth cloud, shares its workgroup 0 did for the others) -- DSPMAP_P_ESTIMATOR_QUEUE"""
        out = (C.c_longlong * 6)()
        self._chk(self.L.dspmap_debug_estimator_queue(self.h, out))
        return tuple(int(v) for v in out)

    def frame_branches(self):
        """(frames run as two branches, tiles of class Q, tiles of class P, tiles of the map, largest speed ever given in mm/s) -- DSPMAP_P_FRAME_BRANCHES"""
        out = (C.c_longlong * 5)()
        self._chk(self.L.dspmap_debug_frame_branches(self.h, out))
        return tuple(int(v) for v in out)

    def resample_split_frames(self):
        """frames whose resampling stage ran as two launches -- DSPMAP_P_RESAMPLE_SPLIT"""
        return int(self.L.dspmap_debug_resample_split_frames(self.h))

    def estimator_path(self):
        """where the last device-estimator frame ran the estimator: 'own_stream', 'forked_shared_queue' (the fallback), 'forked', or None"""
        return {0: None, 1: "own_stream", 2: "forked_shared_queue", 3: "forked"}.get(self.L.dspmap_debug_estimator_path(self.h))

    def tile_count(self):
        return self.L.dspmap_debug_tile_count(self.h)

    def tile_of(self, voxels):
        """the 64-voxel tile each of the given GLOBAL voxel indices lives in (runs of 64 indices or 4x4x4 cubes: DSPMAP_P_TILING)"""
        import numpy as np
        v = np.ascontiguousarray(voxels, np.int32)
        out = np.zeros(v.size, np.int32)
        self._chk(self.L.dspmap_debug_tile_of_voxels(self.h, v.size, v.ctypes.data_as(C.POINTER(C.c_int)), out.ctypes.data_as(C.POINTER(C.c_int))))
        return out

    def tile_moving(self):
        """per 64-voxel tile: 0 = all of its live particles are static (its velocity rows are not fetched by the sweeps)"""
        import numpy as np
        n = self.tile_count()
        out = np.zeros(n, np.int32)
        r = self.L.dspmap_debug_tile_moving(self.h, out.ctypes.data_as(C.POINTER(C.c_int)), n)
        if r < 0:
            self._chk(r)
        return out[:r]

    # -- reference setters (dsp_dynamic.h:355-382)
    def setPredictionVariance(self, p_stddev, v_stddev):
        self.set_param(P_POSITION_STDDEV, p_stddev)
        self.set_param(P_VELOCITY_STDDEV, v_stddev)
        self.set_param(P_REGENERATE_TABLES, 1)

    def setObservationStdDev(self, s):
        self.set_param(P_OBSERVATION_STDDEV, s)

    def setNewBornParticleWeight(self, w):
        self.set_param(P_NEWBORN_WEIGHT, w)

    def setNewBornParticleNumberofEachPoint(self, n):
        self.set_param(P_NEWBORN_NUMBER, n)

    def setOriginalVoxelFilterResolution(self, r):
        self.set_param(P_VOXEL_FILTER_RES, r)

    def useVelocityEstimator(self, on):
        self.set_param(P_VELOCITY_ESTIMATOR, 1 if on else 0)

    # -- randomness
    def set_tables(self, p_tab, v_tab, rand_ints=None):
        p_tab = np.ascontiguousarray(p_tab, np.float32)
        v_tab = np.ascontiguousarray(v_tab, np.float32)
        self._chk(self.L.dspmap_set_gaussian_tables(self.h, _ptr(p_tab), _ptr(v_tab), p_tab.size))
        if rand_ints is not None:
            r = np.ascontiguousarray(rand_ints, np.int32)
            self._chk(self.L.dspmap_set_rand_table(self.h, _ptr(r), r.size))

    def cursors(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.dspmap_get_cursors(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # -- the frame (dsp_dynamic.h:181)
    def update(self, pts, pos, stamp, quat):
        """pts: (n,3) float32 host array, sensor frame.  Returns 1 (ok) / 0 (rejected)."""
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        return self._chk(self.L.dspmap_update(self.h, pts.shape[0], 3, _ptr(pts), pos[0], pos[1], pos[2],
                                              float(stamp), quat[0], quat[1], quat[2], quat[3]))

    def update_device(self, pts_dev_ptr, n, pos, stamp, quat, birth_dev_ptr=None, n_birth=0):
        pos_a = (C.c_float * 3)(*pos)
        q_a = (C.c_float * 4)(*quat)
        return self._chk(self.L.dspmap_update_device(self.h, n, pts_dev_ptr, n_birth, birth_dev_ptr,
                                                     C.cast(pos_a, C.c_void_p), float(stamp),
                                                     C.cast(q_a, C.c_void_p)))

    def set_birth_cloud(self, vpts):
        vpts = np.ascontiguousarray(vpts, VPOINT_DTYPE)
        self._chk(self.L.dspmap_set_birth_cloud(self.h, _ptr(vpts), vpts.size))

    def get_birth_cloud(self):
        n = C.c_int()
        self._chk(self.L.dspmap_get_birth_cloud(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, VPOINT_DTYPE)
        if n.value:
            self._chk(self.L.dspmap_get_birth_cloud(self.h, _ptr(out), n.value, C.byref(n)))
        return out

    # -- readout (dsp_dynamic.h:385-438)
    def getOccupancyMap(self, threshold=0.7):
        xyz = np.zeros((self.V_local, 3), np.float32)
        n = C.c_int()
        self._chk(self.L.dspmap_get_occupancy(self.h, threshold, _ptr(xyz), self.V_local, C.byref(n)))
        return n.value, xyz[:n.value].copy()

    def getOccupancyMapWithFutureStatus(self, threshold=0.7):
        xyz = np.zeros((self.V_local, 3), np.float32)
        fut = np.zeros((self.V_local, self.T), np.float32)
        n = C.c_int()
        self._chk(self.L.dspmap_get_occupancy_with_future(self.h, threshold, _ptr(xyz), self.V_local,
                                                          C.byref(n), _ptr(fut)))
        return n.value, xyz[:n.value].copy(), fut

    def getFutureStatus(self):
        fut = np.zeros((self.V_local, self.T), np.float32)
        self._chk(self.L.dspmap_get_future(self.h, _ptr(fut)))
        return fut

    def clearOccupancyMapPrediction(self):
        self._chk(self.L.dspmap_clear_future(self.h))

    def results(self):
        """[V_local, 4]: occupancy mass, mean vx, vy, vz (voxels_objects_number[v][0..3])."""
        out = np.zeros((self.V_local, 4), np.float32)
        self._chk(self.L.dspmap_get_results(self.h, _ptr(out)))
        return out

    def getVoxelPositionFromIndexPublic(self, index):
        x, y, z = C.c_float(), C.c_float(), C.c_float()
        self.L.dspmap_voxel_center(self.h, index, C.byref(x), C.byref(y), C.byref(z))
        return x.value, y.value, z.value

    def getPointVoxelsIndexPublic(self, px, py, pz):
        idx = C.c_int()
        ok = self.L.dspmap_point_voxel_index(self.h, px, py, pz, C.byref(idx))
        return ok, idx.value

    def debug_tile_fov(self):
        n = self.tile_count()
        out = np.zeros(n, np.int32)
        got = self.L.dspmap_debug_tile_view(self.h, out.ctypes.data_as(C.POINTER(C.c_int)), n)
        if got < 0:
            self._chk(got)
        return out[:got]

    def counters(self):
        c = Counters()
        self._chk(self.L.dspmap_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    STAGES = ("setup+bin", "predict", "claim", "ck_partial", "weight", "ck_finalize", "birth", "resample")

    def set_profiling(self, on=True):
        self._chk(self.L.dspmap_set_profiling(self.h, 1 if on else 0))

    def stage_ms(self):
        """(per-stage summed device ms, frames) accumulated since set_profiling(True)"""
        out = (C.c_float * 8)()
        n = C.c_int()
        self._chk(self.L.dspmap_get_stage_ms(self.h, out, C.byref(n)))
        return dict(zip(self.STAGES, list(out))), n.value

    def event_overhead_ms(self):
        """what an event bracket adds to the one kernel inside it (calibrated by set_profiling(True))"""
        out = C.c_float()
        self._chk(self.L.dspmap_get_event_overhead_ms(self.h, C.byref(out)))
        return out.value

    # -- state
    def clear_state(self):
        self._chk(self.L.dspmap_clear_state(self.h))

    def import_state(self, voxel, rec8, slot=None):
        voxel = np.ascontiguousarray(voxel, np.int32)
        rec8 = np.ascontiguousarray(rec8, np.float32).reshape(-1, 8)
        s = np.ascontiguousarray(slot, np.int32) if slot is not None else None
        self._chk(self.L.dspmap_import_state(self.h, voxel.size, _ptr(voxel), _ptr(s), _ptr(rec8)))

    def export_state(self):
        cap = self.V_local * self.slots
        n = C.c_int()
        self._chk(self.L.dspmap_export_state(self.h, 0, None, None, None, C.byref(n)))
        cap = n.value
        voxel = np.zeros(cap, np.int32)
        slot = np.zeros(cap, np.int32)
        rec = np.zeros((cap, 8), np.float32)
        if cap:
            self._chk(self.L.dspmap_export_state(self.h, cap, _ptr(voxel), _ptr(slot), _ptr(rec), C.byref(n)))
        order = np.lexsort((slot, voxel))
        return voxel[order], slot[order], rec[order]

    def save_checkpoint(self, path):
        self._chk(self.L.dspmap_save_checkpoint(self.h, str(path).encode()))

    def load_checkpoint(self, path):
        self._chk(self.L.dspmap_load_checkpoint(self.h, str(path).encode()))

    def preprocess_cloud(self, points_ptr, n, out_ptr, max_points, leaf=0.1, swap_axes=True, stride=3):
        """voxel-grid filter + axis swap + crop + cap on the device (src/map_sim_example.cpp:309-336);
        returns (points written to out_ptr, occupied leaves touching the map box)"""
        n_out, n_leaves = C.c_int(), C.c_int()
        self._chk(self.L.dspmap_preprocess_cloud(self.h, n, points_ptr, stride, leaf, 1 if swap_axes else 0, max_points,
                                                 out_ptr, C.byref(n_out), C.byref(n_leaves)))
        return n_out.value, n_leaves.value

    def seed_uniform(self, per_voxel, weight=0.01, seed=99, vmax=0.0):
        self._chk(self.L.dspmap_seed_uniform_moving(self.h, per_voxel, weight, seed, vmax))

    # -- stages
    def bin_points(self, pts, quat=(1, 0, 0, 0)):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        self._chk(self.L.dspmap_stage_bin_points(self.h, pts.shape[0], 3, _ptr(pts), *[float(q) for q in quat]))

    def set_current_position(self, x, y, z):
        self._chk(self.L.dspmap_set_current_position(self.h, x, y, z))

    def predict(self, dx, dy, dz, dt):
        self._chk(self.L.dspmap_stage_predict(self.h, dx, dy, dz, dt))

    def map_update(self):
        self._chk(self.L.dspmap_stage_update(self.h))

    def add_newborn(self):
        self._chk(self.L.dspmap_stage_birth(self.h))

    def occupancy_resample(self):
        self._chk(self.L.dspmap_stage_resample(self.h))

    def pyramid_counts(self):
        out = np.zeros(self.NP, np.int32)
        self._chk(self.L.dspmap_get_pyramid_counts(self.h, _ptr(out)))
        return out

    def observations(self):
        obs = np.zeros((self.NP, 100, 5), np.float32)
        cnt = np.zeros(self.NP, np.int32)
        ml = np.zeros(self.NP, np.float32)
        e = C.c_float()
        self._chk(self.L.dspmap_get_observations(self.h, _ptr(obs), _ptr(cnt), _ptr(ml), C.byref(e)))
        return obs, cnt, ml, e.value
