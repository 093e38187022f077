"""A/B of a scheduling parameter (DSPMAP_P_*: same result whatever its value) on IDENTICAL maps: one map per value, all seeded alike and fed
the same frames, so that every arm sees the same state in every frame -- a saturated map thins out over a run (0.66 -> 0.2 ms per frame over
1 500 frames at 132x132x60), which an alternation of blocks inside ONE map (tools/ab_param.py) cannot separate from the parameter's effect
once more than two values are compared.  Every frame runs on every map, in rotated order, each update timed on its own (wall clock around
a stream sync); mean over the frames of a window per value.

  python tools/ab_maps.py --workload C_sat --param SIDE_PLACEMENT --values 3,7,19,35 --frames 40 --skip 8
"""
import argparse
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C_sat")
    ap.add_argument("--param", required=True)
    ap.add_argument("--values", required=True)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--skip", type=int, default=8, help="untimed frames at the start (first-use set-up, graph capture, bench.py's prefill + warmup)")
    ap.add_argument("--estimator", type=int, default=-1)
    args = ap.parse_args()
    import torch
    import bench
    import dsp_map_amd as D
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    w = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    kw = {"pred_times": w["pred_times"]} if "pred_times" in w else {}
    vals = [float(x) for x in args.values.split(",") if x]
    key = getattr(D.capi, "P_" + args.param)
    est = args.estimator if args.estimator >= 0 else (0 if w["sat"] else 2)
    maps = []
    for v in vals:
        m = D.DSPMap(D.make_config(nx=w["nx"], ny=w["ny"], nz=w["nz"], res=w["res"], ppv=w["ppv"], device=0, seed=1234, **kw))
        m.L.dspmap_init_device(m.h)
        if est:
            m.set_param(D.capi.P_VELOCITY_ESTIMATOR, est)
        m.set_param(key, v)
        if w["sat"]:
            m.seed_uniform(w["ppv"], 0.01, 99, w.get("vmax", 0.0))
        maps.append(m)
    sc = scene_mod.CorridorScene(w["nx"] * w["res"], w["ny"] * w["res"], w["nz"] * w["res"], seed=1234, device=dev,
                                 scale=1.0 if w["res"] >= 0.15 else 1.33)
    n = args.skip + args.frames
    frames = [sc.frame(f / 30.0) + (f / 30.0,) for f in range(n)]
    torch.cuda.synchronize()
    ms = [[] for _ in vals]
    gc.collect(); gc.disable()
    for f, (pts, pos, quat, t) in enumerate(frames):
        order = list(range(len(vals)))
        order = order[f % len(vals):] + order[:f % len(vals)]
        for i in order:
            m = maps[i]
            m.sync()
            t0 = time.perf_counter()
            assert m.update_device(pts.data_ptr(), pts.shape[0], pos, t, quat) == 1
            m.sync()
            dt = (time.perf_counter() - t0) * 1e3
            m.clearOccupancyMapPrediction()
            if f >= args.skip:
                ms[i].append(dt)
    gc.enable()
    live = [m.counters()["n_live_out"] for m in maps]
    base = sum(ms[0]) / len(ms[0])
    for i, v in enumerate(vals):
        a = sum(ms[i]) / len(ms[i])
        h = len(ms[i]) // 2
        print("%s = %g : mean %.4f ms (first half %.4f, second half %.4f; %d frames, n_live_out %d)  %+.2f %% vs %g"
              % (args.param, v, a, sum(ms[i][:h]) / h, sum(ms[i][h:]) / (len(ms[i]) - h), len(ms[i]), live[i], (a / base - 1) * 100, vals[0]))
    assert len(set(live)) == 1, "the maps diverged: the parameter changes the result"
    for m in maps:
        m.close()


if __name__ == "__main__":
    main()
