"""How much identical maps of ONE process differ in speed, stage by stage (where their arrays landed in memory): N maps, same configuration,
same seed, same frames, interleaved frame by frame; per-stage device time (HIP events) per map.
   python tools/map_variance.py --workload C_sat --maps 6 --frames 24"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C_sat")
    ap.add_argument("--maps", type=int, default=6)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--skip", type=int, default=6)
    ap.add_argument("--pre-gb", type=float, default=0.0, help="device memory allocated (and kept) BEFORE the first map")
    args = ap.parse_args()
    import torch
    import bench
    import dsp_map_amd as D
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])
    w = bench.WORKLOADS[args.workload]
    maps = []
    pre = torch.empty(int(args.pre_gb * (1 << 30)), dtype=torch.uint8, device="cuda") if args.pre_gb > 0 else None
    for i in range(args.maps):
        m = D.DSPMap(D.make_config(nx=w["nx"], ny=w["ny"], nz=w["nz"], res=w["res"], ppv=w["ppv"], device=0, seed=1234))
        m.L.dspmap_init_device(m.h)
        if w["sat"]:
            m.seed_uniform(w["ppv"], 0.01, 99, w.get("vmax", 0.0))
        else:
            m.set_param(D.capi.P_VELOCITY_ESTIMATOR, 2)
        maps.append(m)
    sc = scene_mod.CorridorScene(w["nx"] * w["res"], w["ny"] * w["res"], w["nz"] * w["res"], seed=1234, device=torch.device("cuda", 0),
                                 scale=1.0 if w["res"] >= 0.15 else 1.33)
    frames = [sc.frame(f / 30.0) + (f / 30.0,) for f in range(args.skip + args.frames)]
    torch.cuda.synchronize()
    for f, (pts, pos, quat, t) in enumerate(frames):
        if f == args.skip:
            for m in maps:
                m.sync(); m.set_profiling(True)
        for i in range(len(maps)):
            m = maps[(i + f) % len(maps)]
            assert m.update_device(pts.data_ptr(), pts.shape[0], pos, t, quat) == 1
            m.clearOccupancyMapPrediction()
            m.sync()
    keys = ("predict", "claim", "ck_partial", "weight", "birth", "resample")
    print("%-6s" % "map", " ".join("%10s" % k for k in keys), "%10s" % "sum")
    rows = []
    for i, m in enumerate(maps):
        st, n = m.stage_ms()
        r = [st[k] / n for k in keys]
        rows.append(r)
        print("%-6d" % i, " ".join("%10.4f" % v for v in r), "%10.4f" % sum(r))
    print("%-6s" % "max/min", " ".join("%10.3f" % (max(r[j] for r in rows) / min(r[j] for r in rows)) for j in range(len(keys))),
          "%10.3f" % (max(sum(r) for r in rows) / min(sum(r) for r in rows)))
    for m in maps:
        m.close()


if __name__ == "__main__":
    main()
