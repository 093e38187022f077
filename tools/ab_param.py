"""In-process A/B of a run-time parameter of the handle (DSPMAP_P_*): ONE map, ONE allocation, the parameter alternates between
two values every `--block` frames, each block timed with the wall clock around a stream sync; medians per value.

Why: the boxes of the pool differ by several per cent from call to call, and so do two processes on one box (clock / power
state drifts over seconds, every process gets other physical pages) -- more than most changes are worth.  Alternating
INSIDE one process removes both.  Only parameters that do not change the map's result can be compared this way
(scheduling knobs: SWEEP_ALTERNATE, PLACE_SPLIT_TILES, RESAMPLE_WG_TILES, ROLLOUT_INLINE, SPARSE_SWEEP, ...).

  python tools/ab_param.py --workload C_sat --param SWEEP_ALTERNATE --a 0 --b 1 --rounds 12 --block 20
"""
import argparse
import gc
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C_sat")
    ap.add_argument("--param", required=True, help="name without the P_ prefix, e.g. SWEEP_ALTERNATE")
    ap.add_argument("--a", type=float, default=None)
    ap.add_argument("--b", type=float, default=None)
    ap.add_argument("--values", default="", help="comma list of more than two values (instead of --a / --b): every round runs one block of each, rotated")
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--block", type=int, default=20)
    ap.add_argument("--prefill", type=int, default=3)
    ap.add_argument("--estimator", type=int, default=-1)
    args = ap.parse_args()
    import torch
    import bench
    import dsp_map_amd as D
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    w = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    kw = {"pred_times": w["pred_times"]} if "pred_times" in w else {}
    m = D.DSPMap(D.make_config(nx=w["nx"], ny=w["ny"], nz=w["nz"], res=w["res"], ppv=w["ppv"], device=0, seed=1234, **kw))
    m.L.dspmap_init_device(m.h)
    est = args.estimator if args.estimator >= 0 else (0 if w["sat"] else 2)
    if est:
        m.set_param(D.capi.P_VELOCITY_ESTIMATOR, est)
    key = getattr(D.capi, "P_" + args.param)
    vals = [float(x) for x in args.values.split(",") if x] or [args.a, args.b]
    n_frames = args.prefill + len(vals) * args.rounds * (args.block + 4) + 8
    sc = scene_mod.CorridorScene(w["nx"] * w["res"], w["ny"] * w["res"], w["nz"] * w["res"], seed=1234, device=dev,
                                 scale=1.0 if w["res"] >= 0.15 else 1.33)
    frames = []
    for f in range(n_frames):
        pts, pos, quat = sc.frame(f / 30.0)
        frames.append((pts, pos, quat, f / 30.0))
    torch.cuda.synchronize()
    if w["sat"]:
        m.seed_uniform(w["ppv"], 0.01, 99, w.get("vmax", 0.0))
    it = iter(frames)

    def run(n):
        for _ in range(n):
            pts, pos, quat, t = next(it)
            assert m.update_device(pts.data_ptr(), pts.shape[0], pos, t, quat) == 1
            m.clearOccupancyMapPrediction()
    run(args.prefill)
    ms = {v: [] for v in vals}
    gc.collect(); gc.disable()
    for r in range(args.rounds):
        for v in (vals[r % len(vals):] + vals[:r % len(vals)]):
            m.set_param(key, v)
            run(4)                      # (re-capture of the frame's graph + warm-up, untimed)
            m.sync()
            t0 = time.perf_counter()
            run(args.block)
            m.sync()
            ms[v].append((time.perf_counter() - t0) / args.block * 1e3)
    gc.enable()
    out = {}
    for v, xs in ms.items():
        out[v] = (statistics.median(xs), min(xs), max(xs))
        print("%s = %g : median %.4f ms  (min %.4f, max %.4f, %d blocks of %d frames)" % (args.param, v, *out[v], len(xs), args.block))
    a = out[vals[0]][0]
    for v in vals[1:]:
        print("%g / %g = %.4f  (%+.2f %%)" % (v, vals[0], out[v][0] / a, (out[v][0] / a - 1) * 100))
    m.close()


if __name__ == "__main__":
    main()
