"""ctypes binding of the CPU oracle (oracle/dsp_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (dsp-map_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

MAX_PRED = 16


class Config(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("voxel_resolution", C.c_float),
        ("angle_resolution", C.c_int),
        ("max_particle_num_voxel", C.c_int),
        ("half_fov_h", C.c_int), ("half_fov_v", C.c_int),
        ("prediction_times", C.c_int),
        ("prediction_future_time", C.c_float * MAX_PRED),
        ("pyramid_neighbor_n", C.c_int), ("safe_particle_factor", C.c_int), ("static_model", C.c_int),
    ]


VPOINT_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"),
                         ("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("intensity", "f4")])


def build(force=False):
    """make the oracle libraries (gcc only; a few seconds)."""
    strict = os.path.join(_BUILD, "libdsp_oracle.so")
    if force or not os.path.exists(strict) or \
            os.path.getmtime(strict) < os.path.getmtime(os.path.join(_HERE, "dsp_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return strict


def _load(fast=False):
    build()
    name = "libdsp_oracle_fast.so" if fast else "libdsp_oracle.so"
    lib = C.CDLL(os.path.join(_BUILD, name))
    P = C.c_void_p
    f = C.c_float
    i = C.c_int
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    sig = {
        "dspo_default_config": (None, [C.POINTER(Config)]),
        "dspo_create": (P, [C.POINTER(Config)]),
        "dspo_destroy": (None, [P]),
        "dspo_voxel_num": (i, [P]), "dspo_slots_per_voxel": (i, [P]), "dspo_pyramid_num": (i, [P]),
        "dspo_pyramid_capacity": (i, [P]), "dspo_result_dim": (i, [P]),
        "dspo_set_prediction_variance": (None, [P, f, f]),
        "dspo_set_observation_stddev": (None, [P, f]),
        "dspo_set_newborn_weight": (None, [P, f]),
        "dspo_set_newborn_number": (None, [P, i]),
        "dspo_set_voxel_filter_resolution": (None, [P, f]),
        "dspo_set_gaussian_tables": (None, [P, P, P, i]),
        "dspo_set_rand_table": (None, [P, P, i]),
        "dspo_set_cursors": (None, [P, i, i, i]),
        "dspo_get_cursors": (None, [P, ip, ip, ip]),
        "dspo_update": (i, [P, i, i, P, f, f, f, C.c_double, f, f, f, f]),
        "dspo_use_velocity_estimator": (None, [P, i]),
        "dspo_static_birth_cloud": (None, [P]),
        "dspo_set_birth_cloud": (None, [P, P, i]),
        "dspo_get_birth_cloud": (i, [P, P, i]),
        "dspo_bin_points": (i, [P, i, i, P, f, f, f, f]),
        "dspo_set_current_position": (None, [P, f, f, f]),
        "dspo_map_prediction": (None, [P, f, f, f, f]),
        "dspo_map_update": (None, [P]),
        "dspo_map_update_ck": (None, [P]),
        "dspo_map_update_weights": (None, [P]),
        "dspo_compute_nstatic": (None, [P, P]),
        "dspo_set_nstatic_override": (None, [P, P]),
        "dspo_add_newborn": (None, [P]),
        "dspo_occupancy_resample": (None, [P]),
        "dspo_velocity_estimation": (None, [P]),
        "dspo_get_occupancy_map": (i, [P, f, P, i]),
        "dspo_get_occupancy_map_with_future": (i, [P, f, P, i, P]),
        "dspo_clear_future": (None, [P]),
        "dspo_query_normal_pdf": (f, [P, f, f, f]),
        "dspo_pdf_lut": (fp, [P]),
        "dspo_rotate_vector": (None, [P, P, P]),
        "dspo_in_pyramids_area": (i, [P, f, f, f]),
        "dspo_pyramid_h": (i, [P, f, f, f]),
        "dspo_pyramid_v": (i, [P, f, f, f]),
        "dspo_voxel_index": (i, [P, f, f, f, ip]),
        "dspo_voxel_center": (None, [P, i, fp, fp, fp]),
        "dspo_neighbor_table": (ip, [P]),
        "dspo_generate_random_float": (f, [P, f, f]),
        "dspo_add_random_particles": (None, [P, i, f]),
        "dspo_inject": (i, [P, i, P, P, P, P, P, P, P, P, f]),
        "dspo_particles": (fp, [P]), "dspo_results": (fp, [P]), "dspo_pyramid_lists": (ip, [P]),
        "dspo_obs": (fp, [P]), "dspo_obs_count": (ip, [P]), "dspo_obs_max_length": (fp, [P]),
        "dspo_expected_newborn": (f, [P]), "dspo_set_expected_newborn": (None, [P, f]),
        "dspo_set_occlusion_margin": (None, [P, f]),
        "dspo_update_time": (f, [P]), "dspo_count_live": (i, [P]),
        "dspo_fill_gaussian_tables": (None, [P, P, i, f, f, C.c_uint]),
        "dspo_preprocess_cloud": (i, [i, P, i, f, i, f, f, f, i, P, ip]),
        "dspo_hungarian": (None, [P, i, i, P]),
        "dspo_euclidean_clusters": (i, [P, i, f, i, i, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_LIBS = {}


def lib(fast=False):
    if fast not in _LIBS:
        _LIBS[fast] = _load(fast)
    return _LIBS[fast]


def hungarian(cost):
    """the estimator's restated munkres-cpp (:1474-1481): assign[r] = column of row r or -1"""
    cost = np.ascontiguousarray(cost, np.float32)
    assign = np.empty(cost.shape[0], np.int32)
    lib().dspo_hungarian(cost.ctypes.data_as(C.c_void_p), cost.shape[0], cost.shape[1], assign.ctypes.data_as(C.c_void_p))
    return assign


def euclidean_clusters(pts, tol, min_size=5, max_size=10000):
    """the estimator's restated pcl::EuclideanClusterExtraction (:1406-1417): -> (labels (n,), cluster count); label = rank by size"""
    pts = np.ascontiguousarray(pts, np.float32)
    label = np.empty(pts.shape[0], np.int32)
    n = lib().dspo_euclidean_clusters(pts.ctypes.data_as(C.c_void_p), pts.shape[0], float(tol), min_size, max_size,
                                      label.ctypes.data_as(C.c_void_p))
    return label, n


def preprocess_cloud(pts, leaf, half, max_points=5000, swap_axes=True):
    """cloudCallback's pre-processing (src/map_sim_example.cpp:309-336): -> (points (n,3), occupied leaves)"""
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros((max_points, 3), np.float32)
    nl = C.c_int()
    n = lib().dspo_preprocess_cloud(pts.shape[0], pts.ctypes.data_as(C.c_void_p), pts.shape[1], float(leaf), 1 if swap_axes else 0,
                                    float(half[0]), float(half[1]), float(half[2]), max_points,
                                    out.ctypes.data_as(C.c_void_p), C.byref(nl))
    return out[:n].copy(), nl.value


def make_config(nx=66, ny=66, nz=40, res=0.15, ppv=9, angle=3, half_fov_h=42, half_fov_v=24,
                pred_times=(0.05, 0.2, 0.5, 1.0, 1.5, 2.0), neighbor_n=0, safe_factor=0, static_model=0):
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
    return c


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Thin object wrapper; numpy views alias the oracle's own arrays."""

    def __init__(self, cfg=None, fast=False, example_params=True):
        self.L = lib(fast)
        self.cfg = cfg or make_config()
        self.h = self.L.dspo_create(C.byref(self.cfg))
        if not self.h:
            raise MemoryError("dspo_create failed")
        self.V = self.L.dspo_voxel_num(self.h)
        self.slots = self.L.dspo_slots_per_voxel(self.h)
        self.NP = self.L.dspo_pyramid_num(self.h)
        self.capp = self.L.dspo_pyramid_capacity(self.h)
        self.rdim = self.L.dspo_result_dim(self.h)
        self.T = self.cfg.prediction_times
        self._keep = []
        if example_params:  # src/map_sim_example.cpp:522-526
            self.L.dspo_set_prediction_variance(self.h, 0.05, 0.05)
            self.L.dspo_set_observation_stddev(self.h, 0.1)
            self.L.dspo_set_newborn_number(self.h, 20)
            self.L.dspo_set_newborn_weight(self.h, 0.0001)
            self.L.dspo_set_voxel_filter_resolution(self.h, 0.1)

    def close(self):
        if self.h:
            self.L.dspo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- views on raw state
    def _view(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        ct = C.c_float if dtype == np.float32 else C.c_int
        buf = C.cast(ptr, C.POINTER(ct * n)).contents
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    @property
    def particles(self):
        return self._view(self.L.dspo_particles(self.h), (self.V, self.slots, 9), np.float32)

    @property
    def results(self):
        return self._view(self.L.dspo_results(self.h), (self.V, self.rdim), np.float32)

    @property
    def pyramid_lists(self):
        return self._view(self.L.dspo_pyramid_lists(self.h), (self.NP, self.capp, 3), np.int32)

    @property
    def obs(self):
        return self._view(self.L.dspo_obs(self.h), (self.NP, 100, 5), np.float32)

    @property
    def obs_count(self):
        return self._view(self.L.dspo_obs_count(self.h), (self.NP,), np.int32)

    @property
    def obs_max_length(self):
        return self._view(self.L.dspo_obs_max_length(self.h), (self.NP,), np.float32)

    @property
    def neighbors(self):
        return self._view(self.L.dspo_neighbor_table(self.h), (self.NP, 10), np.int32)

    @property
    def pdf_lut(self):
        return self._view(self.L.dspo_pdf_lut(self.h), (20000,), np.float32)

    # --- randomness
    def set_tables(self, p_tab, v_tab, rand_ints=None):
        p_tab = np.ascontiguousarray(p_tab, np.float32)
        v_tab = np.ascontiguousarray(v_tab, np.float32)
        assert p_tab.size == v_tab.size
        self._keep += [p_tab, v_tab]
        self.L.dspo_set_gaussian_tables(self.h, _ptr(p_tab), _ptr(v_tab), p_tab.size)
        if rand_ints is not None:
            r = np.ascontiguousarray(rand_ints, np.int32)
            self._keep.append(r)
            self.L.dspo_set_rand_table(self.h, _ptr(r), r.size)

    def cursors(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.dspo_get_cursors(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    # --- frame + stages
    def update(self, pts, pos, stamp, quat):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        return self.L.dspo_update(self.h, pts.shape[0], 3, _ptr(pts), pos[0], pos[1], pos[2],
                                  float(stamp), quat[0], quat[1], quat[2], quat[3])

    def bin_points(self, pts, quat=(1, 0, 0, 0)):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        return self.L.dspo_bin_points(self.h, pts.shape[0], 3, _ptr(pts), *[float(q) for q in quat])

    def set_birth_cloud(self, vpts):
        vpts = np.ascontiguousarray(vpts, VPOINT_DTYPE)
        self.L.dspo_set_birth_cloud(self.h, _ptr(vpts), vpts.size)

    def get_birth_cloud(self):
        n = self.L.dspo_get_birth_cloud(self.h, None, 0)
        out = np.zeros(n, VPOINT_DTYPE)
        if n:
            self.L.dspo_get_birth_cloud(self.h, _ptr(out), n)
        return out

    def predict(self, dx, dy, dz, dt):
        self.L.dspo_map_prediction(self.h, dx, dy, dz, dt)

    def map_update(self):
        self.L.dspo_map_update(self.h)

    def add_newborn(self):
        self.L.dspo_add_newborn(self.h)

    def occupancy_resample(self):
        self.L.dspo_occupancy_resample(self.h)

    def get_occupancy_with_future(self, thr):
        xyz = np.zeros((self.V, 3), np.float32)
        fut = np.zeros((self.V, self.T), np.float32)
        n = self.L.dspo_get_occupancy_map_with_future(self.h, thr, _ptr(xyz), self.V, _ptr(fut))
        return xyz[:n].copy(), fut

    # --- sparse state helpers: rows of (voxel, flag, vx, vy, vz, px, py, pz, w)
    def export_sparse(self):
        p = self.particles
        v, s = np.nonzero(p[:, :, 0] > 0.1)
        rec = p[v, s]
        return v.astype(np.int32), s.astype(np.int32), rec[:, :8].copy()

    def inject(self, px, py, pz, vx, vy, vz, w, flag=1.0):
        """place particles into first free slots of their voxels (like addAParticle but with a flag)."""
        arrs = [np.ascontiguousarray(a, np.float32) for a in (px, py, pz, vx, vy, vz, w)]
        fl = np.ascontiguousarray(flag, np.float32) if np.ndim(flag) else None
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        return self.L.dspo_inject(self.h, len(arrs[0]), *[ptr(a) for a in arrs], ptr(fl) if fl is not None else None,
                                  float(flag) if fl is None else 0.0)
