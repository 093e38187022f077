"""Shared builders for the parity tests: seeded scenes, tables, state injection on both sides."""
import numpy as np

EX_QUATS = [
    (1.0, 0.0, 0.0, 0.0),
    (0.9914449, 0.0, 0.0, 0.1305262),            # yaw 15 deg
    (0.9799247, 0.0436194, 0.0868241, 0.1736482),  # mixed roll/pitch/yaw (not exactly unit)
]


def tables(seed, n=200003, sigma_p=0.05, sigma_v=0.05, nrand=50021):
    rng = np.random.default_rng(seed)
    p = (rng.standard_normal(n) * sigma_p).astype(np.float32)
    v = (rng.standard_normal(n) * sigma_v).astype(np.float32)
    r = rng.integers(0, 2**31 - 1, nrand).astype(np.int32)
    return p, v, r


def wall_cloud(seed, n_side=60, dist=3.0, half_w=2.6, half_h=1.3, wav=0.2, sensor_frame=True):
    """wavy wall in front of the sensor + a ground strip; points in the SENSOR frame (x forward)."""
    rng = np.random.default_rng(seed)
    ys = np.linspace(-half_w, half_w, n_side)
    zs = np.linspace(-half_h, half_h, max(4, n_side // 2))
    Y, Z = np.meshgrid(ys, zs)
    X = dist + wav * np.sin(2.0 * Y) + 0.01 * rng.standard_normal(Y.shape)
    wall = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    gx = rng.uniform(0.8, dist, 300)
    gy = rng.uniform(-1.5, 1.5, 300)
    ground = np.stack([gx, gy, np.full(300, -1.0) + 0.005 * rng.standard_normal(300)], 1)
    return np.concatenate([wall, ground]).astype(np.float32)


def random_particles(seed, n, half, vmax=1.0, static_frac=0.5, wlo=0.002, whi=0.08):
    """particles spread over the map box: px,py,pz,vx,vy,w (vz = 0 as after any prediction)."""
    rng = np.random.default_rng(seed)
    hx, hy, hz = half
    px = rng.uniform(-hx * 0.98, hx * 0.98, n).astype(np.float32)
    py = rng.uniform(-hy * 0.98, hy * 0.98, n).astype(np.float32)
    pz = rng.uniform(-hz * 0.98, hz * 0.98, n).astype(np.float32)
    vx = (rng.uniform(-vmax, vmax, n) * (rng.random(n) > static_frac)).astype(np.float32)
    vy = (rng.uniform(-vmax, vmax, n) * (vx != 0)).astype(np.float32)
    w = rng.uniform(wlo, whi, n).astype(np.float32)
    return px, py, pz, vx, vy, w


def inject_both(o, m, px, py, pz, vx, vy, w, flag=1.0):
    """Put the same particles, in the same slots, into the oracle and the HIP map.
    Returns the number placed.  Slots are assigned by the oracle's first-free rule."""
    n = len(px)
    vz = np.zeros(n, np.float32)
    o.inject(px, py, pz, vx, vy, vz, w, flag)
    voxel, slot, rec = o.export_sparse()
    m.clear_state()
    if len(voxel):
        m.import_state(voxel, rec, slot)
    return len(voxel)


def half_extent(cfg):
    r = np.float32(cfg.voxel_resolution)
    return (float(r * np.float32(cfg.nx) * np.float32(0.5)), float(r * np.float32(cfg.ny) * np.float32(0.5)),
            float(r * np.float32(cfg.nz) * np.float32(0.5)))


def sorted_records(voxel, rec, cols=(4, 5, 6, 1, 2, 7)):
    """canonical per-voxel ordering of particle records for multiset comparison"""
    keys = [rec[:, c] for c in reversed(cols)] + [voxel]
    order = np.lexsort(keys)
    return voxel[order], rec[order]
