#!/usr/bin/env python
"""bench.py -- DSP map update() frames/s on MI355X.

Metric (BASELINE.json): map update() frames/sec @ 66x66x40, 24 particles/voxel; achieved HBM GB/s.
One "step" = one update() of the hot path (binning -> prediction -> weight update -> birth ->
occupancy/rollout/resampling, + the per-frame clearing of the future accumulators the reference's
protocol requires, include/dsp_dynamic.h:429-438) over one frame of a synthetic 640x480 depth
stream (dsp-map_amd/scene.py).  Clouds are generated and resident in HBM before the timed region;
the D2H readout of the result grid is not part of update() and is not timed.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload B|C|C_sat|E]

N > 1 (launched by torch.distributed.run, one rank per GPU): see DESIGN.md "multi-GPU".
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dsp-map_amd"))

WORKLOADS = {
    # name: grid, res, ppv, T times, saturated fill
    "A": dict(nx=66, ny=66, nz=40, res=0.15, ppv=9, sat=False),
    "B": dict(nx=66, ny=66, nz=40, res=0.15, ppv=24, sat=False),
    "B_sat": dict(nx=66, ny=66, nz=40, res=0.15, ppv=24, sat=True),
    # the metric's grid saturated with moving particles (every particle +-1 m/s): the worst case of the inline rollout of small maps
    "B_mov": dict(nx=66, ny=66, nz=40, res=0.15, ppv=24, sat=True, vmax=1.0),
    "C": dict(nx=132, ny=132, nz=60, res=0.15, ppv=24, sat=False),
    "C_sat": dict(nx=132, ny=132, nz=60, res=0.15, ppv=24, sat=True),
    "E": dict(nx=264, ny=264, nz=80, res=0.10, ppv=36, sat=False),
    "E_sat": dict(nx=264, ny=264, nz=80, res=0.10, ppv=36, sat=True),
    # E's grid with 24 particles per voxel (one occupancy word): the only way to run the four-wave resampler on 87 120 sparse tiles
    # (tools/ab_param.py --workload E24 --param RESAMPLE_WG_TILES --a 0 --b 1000000000: what a two-word k_resample_wg could bring E)
    "E24": dict(nx=264, ny=264, nz=80, res=0.10, ppv=24, sat=False),
    # one rank's share of E_sat on 8 GPUs (10 of the 80 layers) as a stand-alone map: driver-overhead studies
    "E8_sat": dict(nx=264, ny=264, nz=10, res=0.10, ppv=36, sat=True),
    # config D (the 10-horizon rollout on C's saturated grid), static / moving fill, as workloads of their own for profiling
    "D_sat": dict(nx=132, ny=132, nz=60, res=0.15, ppv=24, sat=True, pred_times=tuple(0.2 * (k + 1) for k in range(10))),
    "D_mov": dict(nx=132, ny=132, nz=60, res=0.15, ppv=24, sat=True, pred_times=tuple(0.2 * (k + 1) for k in range(10)), vmax=1.0),
}
REC = 32  # bytes of one live particle record in SURVEY 8(d)'s accounting
COUNTER_KEYS = ("n_live_in", "n_fov", "n_born", "n_obs", "n_moved", "n_out_of_map", "n_voxel_full", "n_pyramid_full", "n_live_out")


def csrc_fingerprint():
    """sha256 over the kernel sources (dsp-map_amd/csrc + include/dspmap.h): the PMC traffic figures committed under profiles/
    carry the fingerprint of the sources they were measured on; bench.py prints them only while it still matches"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dsp-map_amd", "csrc")
    for fn in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "dspmap.h")]:
        fp = os.path.join(d, fn)
        if os.path.isfile(fp):
            h.update(fn.encode()); h.update(open(fp, "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def b_alg(c, V, T):
    """algorithmic bytes of one frame."""
    return 4 * REC * c["n_live_in"] + 36 * c["n_fov"] + REC * c["n_born"] + 40 * c["n_obs"] + \
        4 * (4 + T) * V + 4 * T * V


def kernel_alg_bytes(stage, c, V, T, mw=1):
    """Bytes a kernel has to move per launch IN THIS DESIGN ("bytes per unit"), from the device
    counters of the same run.  They are what `roofline.achieved` prices the dominant kernel with and are never larger than
    the kernel's share of SURVEY 8(d)'s B_alg (a kernel is not credited for records it does not have to touch):
      predict    every live record in and out once                                   2 * 32 * N_live
      claim      k_place touches the particles that changed voxel only              2 * 32 * N_moved
      resample   weight + velocity of every particle in (12 B), weight of the kept ones out (4 B), the newborn records
                 (32 B), result grid (16 B) + static future mass (4 B) + occupancy words (2 x 8 B x words) per voxel;
                 positions are read for moving particles and copies only and are not counted (lower bound)
      ck_partial 16 B per particle in view + 20 B per observation
      weight     20 B per particle in view + 20 B per observation"""
    if stage == "predict":
        return 2 * REC * c["n_live_in"]
    if stage == "claim":
        return 2 * REC * c["n_moved"]
    if stage == "resample":
        n_in = c["n_live_in"] - c["n_out_of_map"] - c["n_voxel_full"] - c["n_pyramid_full"]
        return 12 * n_in + 4 * c["n_live_out"] + REC * c["n_born"] + (16 + 4 + 16 * mw) * V
    if stage == "ck_partial":
        return 16 * c["n_fov"] + 20 * c["n_obs"]
    if stage == "weight":
        return 20 * c["n_fov"] + 20 * c["n_obs"]
    return 0


FRAME_KERNELS = ("k_obs_points", "k_predict", "k_place", "k_pyr_prepare", "k_ck_partial", "k_weight", "k_birth_split_cksum", "k_birth_split_cksum_cvr",
                 "k_birth_cursors", "k_birth_insert", "k_resample", "k_resample_wg", "k_rollout", "k_ve_view", "k_ve_components", "k_ve_clusters",
                 "k_birth_children")
# the kernels behind each timed stage (the resampling stage runs the one-wave-per-tile kernel on large maps, the four-waves-per-tile
# one on small ones, and the rollout of the moving particles behind either)
STAGE_KERNELS = {"predict": ("k_predict",), "claim": ("k_place",), "ck_partial": ("k_ck_partial",), "weight": ("k_weight",),
                 "resample": ("k_resample", "k_resample_wg", "k_rollout")}


SQ_DB = {}       # profiles/pmc_traffic.json "sq": per workload and kernel the share of the chip's VALU issue slots used / of wave lifetime spent waiting
KERNEL_US = {}   # profiles/pmc_traffic.json "kernel_us" (rocprofv3 --kernel-trace --stats averages), while its sources are this run's


def roofline_block(stage_ms, cnt, V, T, mw, traffic_db, wl_name, peak=8000.0, traffic_meta=None, overhead_ms=0.0):
    """roofline of the dominant kernel + every kernel's fraction.  A kernel cannot beat the HBM peak on the bytes it has to
    move: a fraction above 1 means the accounting (or the timer) is wrong and is never printed.
    Kernel durations = the HIP-event bracket around the stage's launches AS RECORDED: `frac` / `GBps` / `ms` are computed from it
    and are therefore a little pessimistic (the bracket holds the record's cost on the queue and the launch gaps besides the
    kernel).  For stages that are ONE launch the same figures with the calibrated bracket overhead removed (`overhead_ms`, measured
    in this process with a kernel of known duration, dspmap_get_event_overhead_ms) are printed beside them as
    `ms_minus_bracket_overhead` / `frac_minus_bracket_overhead` -- labelled, never the headline; the committed rocprofv3 durations
    (`rocprof_kernel_us`, `frac_on_rocprof_duration`) are the independent check.  The dominant kernel is the one with the longest
    bracket."""
    single_launch = ("predict", "ck_partial", "weight")   # (claim = two k_place launches on large maps, resample = k_resample + k_rollout)
    timed = {k: v for k, v in stage_ms.items() if k not in ("setup+bin", "ck_finalize", "birth")}
    per = {}
    for k, ms in timed.items():
        b = kernel_alg_bytes(k, cnt, V, T, mw)
        fr = b / (ms * 1e-3) / 1e9 / peak if ms > 0 else 0.0
        per[k] = {"ms": round(ms, 5), "bytes": int(b), "GBps": round(b / (ms * 1e-3) / 1e9, 2) if ms > 0 else 0.0,
                  "frac": round(fr, 5)}
        if fr > 1.0:
            per[k] = {"ms": round(ms, 5), "bytes": int(b), "GBps": None, "frac": None,
                      "error": "bytes / time exceeds the HBM peak: accounting rejected"}
        elif k in single_launch and overhead_ms > 0 and ms > 2 * overhead_ms:
            msc = ms - overhead_ms
            per[k]["ms_minus_bracket_overhead"] = round(msc, 5)
            per[k]["frac_minus_bracket_overhead"] = round(min(b / (msc * 1e-3) / 1e9 / peak, 1.0), 5)
    dom = max((k for k in timed if per[k]["frac"] is not None), key=lambda k: timed[k])
    tdb = traffic_db.get(wl_name, {})
    name_of = {"claim": "k_place", "weight": "k_weight", "predict": "k_predict", "resample": "k_resample",
               "ck_partial": "k_ck_partial"}

    def pmc_of(stage):
        tot = sum(tdb.get(k, {}).get("hbm_bytes", 0) for k in STAGE_KERNELS.get(stage, ()))
        return int(tot) if tot else None
    roof = {"bound": "hbm", "kernel": name_of.get(dom, "k_" + dom), "achieved": per[dom]["GBps"], "peak": peak,
            "unit": "GB/s", "frac": per[dom]["frac"], "traffic": pmc_of(dom),
            "kernel_ms": per[dom]["ms"], "algorithmic_bytes": per[dom]["bytes"], "per_kernel": per,
            "timer": "HIP events on the library's stream around each stage's launches, as recorded (frame.stage_ms); the calibrated "
                     "bracket overhead of %.4f ms is removed only in the *_minus_bracket_overhead fields of single-launch stages" % overhead_ms}
    rk = KERNEL_US.get(wl_name, {})
    if rk:   # the committed rocprofv3 durations of the same command on the same sources: the figure the events must agree with
        for k, v in per.items():
            us = sum(rk.get(n, 0.0) for n in STAGE_KERNELS.get(k, ()))
            if us > 0 and v.get("bytes"):
                v["rocprof_kernel_us"] = round(us, 2)
                v["frac_on_rocprof_duration"] = round(v["bytes"] / (us * 1e-6) / 1e9 / peak, 5)
        if "rocprof_kernel_us" in per[dom]:
            roof["rocprof_kernel_us"] = per[dom]["rocprof_kernel_us"]
            roof["frac_on_rocprof_duration"] = per[dom]["frac_on_rocprof_duration"]
    if tdb:
        meta = traffic_meta or {}
        roof["traffic_source"] = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload "
                                  "(corrected 2*FETCH+WRITE per the MI355X guide), measured at commit %s on the sources with "
                                  "fingerprint %s = the sources of this run" % (meta.get("commit", "?"), meta.get("csrc_sha16", "?")))
        roof["traffic_frame"] = int(sum(v.get("hbm_bytes", 0) for k, v in tdb.items() if k in FRAME_KERNELS))
        # beside the contract fraction (algorithmic bytes / 8 TB/s): the bytes the kernel really moved against what a float4
        # copy sustains on this part (6.3 TB/s, MI355X guide)
        for k, v in per.items():
            pb = pmc_of(k)
            if pb and v.get("ms"):
                v["pmc_bytes"] = int(pb)
                v["frac_of_6.3TBps_on_pmc_bytes"] = round(pb / (v["ms"] * 1e-3) / 1e9 / 6300.0, 4)
    elif traffic_meta and traffic_meta.get("stale"):
        roof["traffic_source"] = traffic_meta["stale"]
    return roof


SCATTER_STORES_PER_S = 55e9   # tools/micro/atomic_bench.hip on MI355X: 55 - 90 G scattered 4-byte stores per second whatever the footprint
                              # (profiles/r03_micro.md); the lower figure is the ruler
COPY_TBPS = 6.3               # what a float4 copy sustains on this part (MI355X guide)


def ceiling_table(cnt, b_alg_bytes, frame_ms, kernel_us, pmc, sq, skeleton_TBps, peak=8000.0):
    """Every kernel of the saturated frame against ITS OWN ruler, so that "how far is the frame from what this
    design can reach under slot-exact parity" is arithmetic in the JSON line:
      sweeps (k_predict, k_resample)  the HBM bytes they really move (PMC) at the rate the sweep's bare memory skeleton sustains on
                                      this box (dspmap_debug_sweep_probe, measured live: same rows, same access pattern, no arithmetic)
      k_place                         scattered stores (2 per static arrival + 3 list words per arrival in view) at the part's
                                      scattered-store rate (tools/micro/atomic_bench.hip)
      pair kernels                    VALU-bound: their duration x the share of the chip's VALU issue slots they used (SQ counters):
                                      the time they would take at issue rate 1.0
      k_pyr_prepare, the rest         their PMC bytes at the sustained copy rate (they are latency chains: the ceiling says how much)
    kernel_us: average kernel durations (rocprofv3 of the committed profile when its sources are this run's, else this run's HIP-event
    brackets); returns per-kernel rows + the frame at its ceilings."""
    rows = {}
    n_live, n_mv, n_fov = max(cnt["n_live_in"], 1), cnt["n_moved"], cnt["n_fov"]
    for k, us in sorted(kernel_us.items(), key=lambda kv: -kv[1]):
        if us <= 0 or k in ("k_spin", "k_seed_uniform", "k_reduce_counters", "k_verify_div", "k_set_live_sample"):
            continue
        pb = pmc.get(k, {}).get("hbm_bytes")
        issue = sq.get(k, {}).get("valu_issue")
        r = {"us": round(us, 2)}
        if k in ("k_predict", "k_resample") and skeleton_TBps:
            bytes_ = pb or kernel_alg_bytes("predict" if k == "k_predict" else "resample", cnt, 0, 0)
            r.update(ruler="bytes moved (%s) at the sweep skeleton's %.2f TB/s" % ("PMC" if pb else "algorithmic", skeleton_TBps),
                     ceiling_us=round(bytes_ / (skeleton_TBps * 1e12) * 1e6, 2))
        elif k == "k_place":
            stores = 2 * n_mv + 3 * n_mv * n_fov / n_live
            r.update(ruler="%.2f M scattered stores at %.0f G/s" % (stores / 1e6, SCATTER_STORES_PER_S / 1e9),
                     ceiling_us=round(stores / SCATTER_STORES_PER_S * 1e6, 2))
        elif k in ("k_ck_partial", "k_weight") and issue:
            r.update(ruler="VALU issue %.2f -> 1.0" % issue, ceiling_us=round(us * issue, 2))
        elif pb:   # a latency chain: its bytes at the copy rate, but never below what one dependent kernel node costs (~5 us)
            r.update(ruler="PMC bytes at the %.1f TB/s copy rate, >= 5 us per dependent launch (latency chain)" % COPY_TBPS,
                     ceiling_us=round(max(pb / (COPY_TBPS * 1e12) * 1e6, 5.0), 2))
        else:
            r.update(ruler="none (taken at its own duration)", ceiling_us=round(us, 2))
        r["achieved_over_ceiling"] = round(us / r["ceiling_us"], 2) if r["ceiling_us"] > 0 else None
        rows[k] = r
    tot_us = sum(r["us"] for r in rows.values())
    tot_ceil = sum(r["ceiling_us"] for r in rows.values())
    out = {"per_kernel": rows, "sum_of_kernels_us": round(tot_us, 1), "sum_of_ceilings_us": round(tot_ceil, 1),
           "frame_ms": round(frame_ms, 4),
           "frac_of_8TBps_now": round(b_alg_bytes / (frame_ms * 1e-3) / 1e9 / peak, 4),
           "frac_of_8TBps_if_every_kernel_sat_on_its_ceiling": round(b_alg_bytes / (tot_ceil * 1e-6) / 1e9 / peak, 4) if tot_ceil > 0 else None,
           "reading": "sum_of_ceilings is a chain of kernels each at its own ruler with no launch gap and no overlap credit: the part of "
                      "the gap between frame_ms and it that belongs to a kernel is that kernel's achieved_over_ceiling"}
    return out


XGMI_COLLECTIVE_US = 30.0   # assumed latency of one small collective over xGMI (20 - 40 us); never measured here


def balanced_ranges(layer_cost, world):
    """contiguous z ranges with (nearly) equal summed cost: cut the prefix sum of the per-layer costs into `world` equal parts, every
    slab at least one layer"""
    nz = len(layer_cost)
    tot = float(sum(layer_cost))
    cuts, acc, z = [0], 0.0, 0
    for r in range(1, world):
        target = tot * r / world
        while z < nz - (world - r) and (acc + layer_cost[z] <= target or z < cuts[-1] + 1):
            acc += layer_cost[z]; z += 1
        # the cut closer to the target
        if z < nz - (world - r) and z > cuts[-1] + 0 and abs(acc + layer_cost[z] - target) < abs(acc - target):
            acc += layer_cost[z]; z += 1
        cuts.append(z)
    cuts.append(nz)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


SEGMENTS = ((0,), (1, 2, 4), (5,), (6,))   # phases between two collectives: begin | exchange .. Ck | weights + split | births + resampling


def critical_path(tab, n, select_ran):
    """tab[slab][phase] mean ms (slab n = the group's stand-ins for the collectives).  Every collective is a barrier (the neighbour
    exchange only between neighbours: treated as one): between two of them a rank runs its phases back to back, so the frame's critical
    path is the sum over those SEGMENTS of the slowest slab's segment -- + a rank's share of the list selection (measured for all slabs
    together) in the frames that run it."""
    per_seg = [max(sum(tab[i][ph] for ph in seg) for i in range(n)) for seg in SEGMENTS]
    sel = tab[n][3] / n if select_ran else 0.0
    return sum(per_seg) + sel, per_seg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--prefill", type=int, default=300, help="untimed frames before warmup (steady state: the live count of workload B levels off after ~8 s of stream)")
    ap.add_argument("--estimator", type=int, default=2, help="velocity estimator of the realistic workloads: 0 static tags, 1 host stage, 2 device (default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the saturated extra measurements")
    ap.add_argument("--only", default="", help="comma list: run only these extra blocks of the default line (saturated, realistic, rollout, variants, "
                                              "host, node, preprocess, origin, projection); the contract measurement always runs")
    args = ap.parse_args()

    import numpy as np
    import torch
    import build_ext
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0 or world == 1:
        build_ext.build()
    force_sharded = os.environ.get("DSPMAP_BENCH_FORCE_SHARDED") == "1"  # exercise the N>1 path on one GPU
    if world > 1 or force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()
    else:
        dist = None
        torch.cuda.set_device(0)
    import dsp_map_amd as D
    scene_mod = __import__("dsp-map_amd.scene", fromlist=["CorridorScene"])

    sharded_run = world > 1 or force_sharded
    wl_name = args.workload or ("B" if not sharded_run else "E_sat")
    wl = WORKLOADS[wl_name]
    dev = torch.device("cuda", local_rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
    globals_barrier = barrier

    def make_map(w):
        kw = {"pred_times": w["pred_times"]} if "pred_times" in w else {}
        cfg = D.make_config(nx=w["nx"], ny=w["ny"], nz=w["nz"], res=w["res"], ppv=w["ppv"], device=local_rank,
                            seed=1234, **kw)
        m = D.DSPMap(cfg)
        m.L.dspmap_init_device(m.h)
        return m

    def gen_frames(w, n, t0=0.0, seed=1234):
        sc = scene_mod.CorridorScene(w["nx"] * w["res"], w["ny"] * w["res"], w["nz"] * w["res"], seed=seed,
                                     device=dev, scale=1.0 if w["res"] >= 0.15 else 1.33)
        out = []
        for f in range(n):
            t = t0 + f / 30.0
            pts, pos, quat = sc.frame(t)
            out.append((pts, pos, quat, t))
        torch.cuda.synchronize()
        return out

    def quiet_host():
        """The harness is Python, the product's host side is C++ (include/dsp_dynamic.h): a generation-2 pass of the
        interpreter's collector over torch's object graph takes ~35 ms -- 180 frames of the headline workload -- and would
        land inside a 20-step timed region at random.  Collect now, keep the collector off until the region ends."""
        gc.collect()
        gc.disable()

    def run_frames(m, frames):
        for pts, pos, quat, t in frames:
            rc = m.update_device(pts.data_ptr(), pts.shape[0], pos, t, quat)
            assert rc == 1, rc
            m.clearOccupancyMapPrediction()  # the reference requires this once per frame (:429-438)

    def measure(w, steps, warmup, prefill, profile=True, solo=False, estimator=0):
        """solo: this rank measures alone (no collective barrier): the single-GPU origin of an N > 1 run
        estimator: DSPMAP_P_VELOCITY_ESTIMATOR (0 = every point in view is a static birth source, 1 = host stage,
        2 = on the device, inside the captured frame)"""
        barrier = (lambda: torch.cuda.synchronize()) if solo else globals_barrier
        m = make_map(w)
        if estimator:
            m.set_param(D.capi.P_VELOCITY_ESTIMATOR, estimator)
        n_total = prefill + warmup + steps + (steps if profile else 0)
        frames = gen_frames(w, n_total, seed=1234 + rank)
        if w["sat"]:
            m.seed_uniform(w["ppv"], 0.01, 99, w.get("vmax", 0.0))  # M zero-velocity particles in every voxel
        run_frames(m, frames[:prefill])
        run_frames(m, frames[prefill:prefill + warmup])
        quiet_host()
        barrier()
        t0 = time.perf_counter()
        run_frames(m, frames[prefill + warmup:prefill + warmup + steps])
        t_issue = time.perf_counter() - t0   # host time to enqueue the frames (launch-bound if ~= dt)
        m.sync()
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        measure.host_issue_ms = t_issue / steps * 1e3
        measure.est_path = m.estimator_path()   # (of the TIMED frames: the stage-profiled ones below keep the frame on one stream)
        cnt = m.counters()
        stage = None
        if profile:
            m.set_profiling(True)
            measure.event_overhead_ms = m.event_overhead_ms()
            run_frames(m, frames[prefill + warmup + steps:])
            sums, nfr = m.stage_ms()
            stage = {k: v / max(nfr, 1) for k, v in sums.items()}
            m.set_profiling(False)
        return m, frames, dt, cnt, stage

    def measure_sharded(w, steps, warmup, prefill):
        """N > 1: the map is split in Z-slabs, one per rank; every rank is fed the same cloud.  The frame is driven from C++
        (dspmap_mgpu_update, dspmap_dist.hip): boundary particles go to the neighbour ranks in fixed-size ncclSend / ncclRecv
        pairs, Ck and n_static are all-reduced, all on the library's stream.  torch.distributed only launches the ranks,
        hands rank 0's RCCL unique id around once and provides the barriers of the timing contract."""
        sharded = __import__("dsp-map_amd.sharded", fromlist=["CppShardedRank"])

        def bcast(t):
            t = t.to(dev)
            dist.broadcast(t, 0)
            return t
        rk = sharded.CppShardedRank(D, dict(nx=w["nx"], ny=w["ny"], nz=w["nz"], res=w["res"], ppv=w["ppv"], seed=1234),
                                    world, rank, local_rank, broadcast=bcast)
        # every rank must be fed the SAME cloud, bit for bit (the ranks bin the same observations and all-reduce arrays whose
        # length follows the point count): the scene's voxel filter sums with float atomics, so rank 0 generates the
        # frames and hands them to the others once, before anything is timed
        frames = gen_frames(w, prefill + warmup + steps, seed=1234)
        if world > 1:
            shared = []
            for pts, pos, quat, t in frames:
                n = torch.tensor([pts.shape[0]], device=dev, dtype=torch.int64)
                dist.broadcast(n, 0)
                buf = pts.contiguous() if rank == 0 else torch.empty((int(n.item()), 3), device=dev, dtype=torch.float32)
                dist.broadcast(buf, 0)
                shared.append((buf, pos, quat, t))
            frames = shared
            torch.cuda.synchronize()
        if w["sat"]:
            rk.map.seed_uniform(w["ppv"], 0.01, 99)

        def run(fr):
            for pts, pos, quat, t in fr:
                assert rk.update(pts, pos, t, quat) == 1
                rk.map.clearOccupancyMapPrediction()
        run(frames[:prefill + warmup])
        rk.sync()
        quiet_host()
        barrier()
        t0 = time.perf_counter()
        run(frames[prefill + warmup:])
        rk.sync()
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        cnt = rk.map.counters()
        keys = ("n_live_in", "n_fov", "n_born", "n_obs", "n_moved", "n_out_of_map", "n_voxel_full", "n_pyramid_full", "n_live_out")
        tot = torch.tensor([cnt[k] for k in keys], device=dev, dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        cnt_all = dict(cnt)
        for k, v in zip(keys, tot.tolist()):
            cnt_all[k] = int(v)
        cnt_all["n_obs"] = cnt["n_obs"]  # every rank bins the same observations
        cnt_all["message_records"] = rk.map.L.dspmap_mgpu_message_records(rk.map.h)
        return rk.map, frames, dt, cnt_all, None

    only = set(x for x in args.only.split(",") if x)

    def want(name):
        return rank == 0 and not sharded_run and not args.no_extra and wl_name == "B" and (not only or name in only)

    # ------------------------------------------------------------------ main measurement
    if not sharded_run:
        m, frames, dt, cnt, stage = measure(wl, args.steps, args.warmup, args.prefill, estimator=args.estimator if not wl["sat"] else 0)
    else:
        m, frames, dt, cnt, stage = measure_sharded(wl, args.steps, args.warmup, 5 if wl["sat"] else args.prefill)
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    fps = args.steps / dt
    V, T = (m.V_local, m.T) if not sharded_run else (m.V, m.T)
    balg = b_alg(cnt, V, T)
    ms = dt / args.steps * 1e3
    peak = 8000.0
    traffic_db, traffic_meta = {}, {}
    try:   # HBM bytes per launch from the committed PMC passes of the same commands -- only if measured on THESE sources
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        fp_now = csrc_fingerprint()
        if tj.get("csrc_sha16") == fp_now:
            traffic_db = tj["workloads"]
            KERNEL_US.update(tj.get("kernel_us", {}))
            SQ_DB.update(tj.get("sq", {}))
            traffic_meta = {"commit": tj.get("commit"), "csrc_sha16": fp_now}
        else:
            traffic_meta = {"stale": "profiles/pmc_traffic.json was measured on other kernel sources (fingerprint %s, commit %s; this "
                                     "run: %s): traffic not reported" % (tj.get("csrc_sha16"), tj.get("commit"), fp_now)}
    except Exception:
        pass
    mw = (2 * wl["ppv"] + 63) // 64
    if stage is not None:
        roof = roofline_block(stage, cnt, V, T, mw, traffic_db, wl_name, peak, traffic_meta, getattr(measure, "event_overhead_ms", 0.0))
        if V < 500_000:   # the metric's own size: the frame is a chain of dependent launches of 5-35 us, none of them bandwidth-bound
            roof["note"] = ("at this map size every kernel is a latency chain (one wave per SIMD, ~0.2 TB/s for the whole frame); "
                            "the HBM-bound case is saturated_132x132x60 in this same line")
    else:  # sharded run: whole-frame algorithmic bytes over all ranks against N x 8 TB/s
        roof = {"bound": "hbm", "kernel": "whole frame (all ranks)", "achieved": round(balg / (ms * 1e-3) / 1e9, 3),
                "peak": peak * world, "unit": "GB/s", "frac": round(balg / (ms * 1e-3) / 1e9 / (peak * world), 6),
                "traffic": None, "algorithmic_bytes": int(balg)}
    result = {
        "metric": ("map update() frames/sec @ 66x66x40, 24 particles/voxel; achieved HBM GB/s" if wl_name == "B" else
                   "map update() frames/sec @ %dx%dx%d, %d particles/voxel (workload %s, NOT the 66x66x40 headline); achieved HBM GB/s"
                   % (wl["nx"], wl["ny"], wl["nz"], wl["ppv"], wl_name)),
        "workload_id": wl_name,
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "weak" if not sharded_run else "strong",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%dx%d @ %.2f m, %d particles/voxel, synthetic 640x480 depth @ 30 Hz "
                               "(corridor scene, <=5000 points/frame after 0.1 m voxel filter), %s" %
                               (wl_name, wl["nx"], wl["ny"], wl["nz"], wl["res"], wl["ppv"],
                                "saturated fill" if wl["sat"] else "steady state after %d frames" % args.prefill),
                   "birth_tags": ("static (every in-FOV point is a zero-velocity birth source)" if wl["sat"] or args.estimator == 0 else
                                  "velocityEstimationThread (:1377-1544) %s" % ("on the device, inside the captured frame (dspmap_velest.hip)"
                                                                               if args.estimator == 2 else "as a host stage (velocity_estimator.cpp)")),
                   "parallelism": "1 GPU" if not sharded_run else "%d Z-slabs (one per GPU), C++ driver: ncclSend/ncclRecv neighbour "
                                                                "exchange + 2 small ncclAllReduce per frame on the library's stream" % world,
                   "n_points": int(frames[-1][0].shape[0]),
                   # where the device estimator of the timed frames ran: own_stream (DSPMAP_P_ESTIMATOR_QUEUE), forked_shared_queue (the
                   # fallback: no stream apart from the main stream's hardware queue in this process), forked
                   "estimator_path": getattr(measure, "est_path", None) if (not sharded_run and not wl["sat"] and args.estimator == 2) else None,
                   "storage": ("cubes of 4x4x4 voxels" if m.get_param(D.capi.P_TILING) == 1 else "runs of 64 voxel indices") if not sharded_run else "runs of 64 voxel indices"},
        "roofline": roof,
        "host_enqueue_ms_per_step": round(getattr(measure, "host_issue_ms", 0.0), 5),
        "frame": {"b_alg_bytes": int(balg), "b_alg_GBps": round(balg / (ms * 1e-3) / 1e9, 3),
                  "frac_of_8TBps": round(balg / (ms * 1e-3) / 1e9 / peak, 6),
                  "stage_ms": {k: round(v, 5) for k, v in stage.items()} if stage else None,
                  "counters": {k: cnt[k] for k in COUNTER_KEYS}},
    }

    # ------------------------------------------------------------------ saturated large map (C_sat): the roofline case
    if want("saturated"):
        try:
            del frames
            m.close()
            w2 = WORKLOADS["C_sat"]
            m2, fr2, dt2, c2, st2 = measure(w2, 40, 5, 3)
            V2, T2 = m2.V_local, m2.T
            skel = None
            try:   # the memory skeleton of the prediction sweep on THIS box: 24 B in + 12 B out per cell of 27 rows (a saturated tile's live rows)
                pms, pby = C.c_float(), C.c_longlong()
                m2._chk(m2.L.dspmap_debug_sweep_probe(m2.h, 15, 27, 2, 10, C.byref(pms), C.byref(pby)))
                skel = pby.value / (pms.value * 1e-3) / 1e12
            except Exception:
                pass
            ms2 = dt2 / 40 * 1e3
            b2 = b_alg(c2, V2, T2)
            result["saturated_132x132x60"] = {
                "workload": "C_sat: 132x132x60 @ 0.15 m, every voxel seeded with 24 zero-velocity particles, same depth stream",
                "frames_per_s": round(40 / dt2, 2), "ms_per_step": round(ms2, 4),
                "b_alg_bytes": int(b2), "b_alg_GBps": round(b2 / (ms2 * 1e-3) / 1e9, 2),
                "frac_of_8TBps": round(b2 / (ms2 * 1e-3) / 1e9 / peak, 5),
                # how saturated "saturated" is when the timed frames end: live particles over V x M (the fill erodes from frame to frame: 24 per
                # voxel are seeded, pyramid lists and the resampler's n_after = M cap take their share); every figure of this block depends on it
                "fill_n_live_over_VM": round(c2["n_live_in"] / float(V2 * w2["ppv"]), 4),
                "frames": {"prefill": 3, "warmup": 5, "steps": 40},
                "storage": "cubes of 4x4x4 voxels" if m2.get_param(D.capi.P_TILING) == 1 else "runs of 64 voxel indices",
                "frame": "two branches" if m2.frame_branches()[0] > 0 else "serial",
                "roofline": roofline_block(st2, c2, V2, T2, 1, traffic_db, "C_sat", peak, traffic_meta, getattr(measure, "event_overhead_ms", 0.0)),
                "stage_ms": {k: round(v, 5) for k, v in st2.items()},
                "counters": {k: c2[k] for k in COUNTER_KEYS}}
            kus = dict(KERNEL_US.get("C_sat", {}))
            src = "rocprofv3 averages of the committed profile (profiles/pmc_traffic.json, same kernel sources)"
            if not kus:   # no committed profile of these sources: this run's HIP-event brackets per stage
                kus = {"k_predict": st2["predict"] * 1e3, "k_place": st2["claim"] * 1e3, "k_ck_partial": st2["ck_partial"] * 1e3,
                       "k_weight": st2["weight"] * 1e3, "k_resample": st2["resample"] * 1e3, "k_birth_insert": st2["birth"] * 1e3,
                       "k_obs_points": st2["setup+bin"] * 1e3}
                src = "this run's HIP-event brackets per stage (no committed rocprofv3 profile of these kernel sources; k_pyr_prepare is inside k_ck_partial's bracket)"
            ct = ceiling_table(c2, b2, ms2, kus, traffic_db.get("C_sat", {}), SQ_DB.get("C_sat", {}), skel, peak)
            ct["durations_from"] = src
            ct["sweep_skeleton_TBps"] = round(skel, 3) if skel else None
            result["saturated_132x132x60"]["ceilings"] = ct
            m2.close()
            del fr2
            # the same map FRESHLY seeded: the first timed frame sees all 24 particles per voxel (SURVEY 8(d)'s 25.09 M), no untimed frame before it
            m3, fr3, dt3, c3, _ = measure(w2, 8, 0, 0, profile=False)
            b3 = b_alg(c3, m3.V_local, m3.T)
            result["saturated_132x132x60"]["freshly_seeded"] = {
                "what": "8 timed frames straight after the seeding (prefill 0, warmup 0: the first frame also pays the frame's first-use set-up)",
                "ms_per_step": round(dt3 / 8 * 1e3, 4), "n_live_in_last_frame": int(c3["n_live_in"]),
                "fill_n_live_over_VM": round(c3["n_live_in"] / float(m3.V_local * w2["ppv"]), 4),
                "frac_of_8TBps_on_last_frames_b_alg": round(b3 / (dt3 / 8) / 1e9 / peak, 5)}
            m3.close()
            del fr3
        except Exception as e:  # the extra line must never break the contract line
            result["saturated_132x132x60"] = {"error": repr(e)}

    # ------------------------------------------------------------------ C and E with a REALISTIC fill
    if want("realistic"):
        try:
            out = {"what": "the large grids filled by the depth stream itself (empty start, particles only near surfaces) instead "
                           "of the saturated fill: what a deployed map looks like; device velocity estimator in the frame"}
            for tag, wn, st in (("132x132x60", "C", 150), ("264x264x80", "E", 100)):
                mr, frr, dtr, cr, _ = measure(WORKLOADS[wn], st, 10, args.prefill, profile=False, estimator=args.estimator)
                br = b_alg(cr, mr.V_local, mr.T)
                out[tag] = {"frames_per_s": round(st / dtr, 2), "ms_per_step": round(dtr / st * 1e3, 4),
                            "n_live_in": cr["n_live_in"], "n_fov": cr["n_fov"], "n_born": cr["n_born"], "n_moved": cr["n_moved"],
                            "b_alg_bytes": int(br)}
                mr.close()
                del frr
            result["realistic_fill"] = out
        except Exception as e:
            result["realistic_fill"] = {"error": repr(e)}

    # ------------------------------------------------------------------ config D: the future-status rollout, T = 10
    if want("rollout"):
        try:
            out = {}
            for tag, vmax in (("static_fill", 0.0), ("moving_fill", 1.0)):
                wd = dict(WORKLOADS["C_sat"], pred_times=tuple(0.2 * (k + 1) for k in range(10)), vmax=vmax)
                md, frd, dtd, cd, sd = measure(wd, 20, 3, 2)
                n_old = cd["n_live_in"]  # particles that enter the rollout (newborns of the frame are excluded, :944)
                bd = 28 * n_old + 8 * md.T * md.V_local   # B_alg(D)
                out[tag] = {"k_resample_ms": round(sd["resample"], 5), "n_old": int(n_old), "T": md.T,
                            "b_alg_D_bytes": int(bd), "GBps": round(bd / (sd["resample"] * 1e-3) / 1e9, 2),
                            "frac_of_8TBps": round(bd / (sd["resample"] * 1e-3) / 1e9 / peak, 5),
                            "update_ms": round(dtd / 20 * 1e3, 4)}
                if vmax > 0:
                    # the byte roofline is the wrong ruler when every particle moves: the stage is T scattered fixed-point adds per
                    # particle into k_rollout's LDS windows (ds_add_u32) -- the ruler is the chip's LDS integer-atomic rate
                    adds = float(n_old) * md.T
                    out[tag].update({"scattered_adds": int(adds), "adds_per_s": round(adds / (sd["resample"] * 1e-3), 0),
                                     "frac_of_lds_u32_atomic_rate": round(adds / (sd["resample"] * 1e-3) / 1.7e12, 4),
                                     "ruler": "ds_add_u32 to random cells of a 30 000-cell window sustains 1.7 T adds/s device-wide, ds_add_f32 0.20 T/s "
                                              "(tools/micro/lds_atomic_bench.hip, profiles/r03_micro.md); the stage also reads every particle once "
                                              "(28 B) and flushes its windows with one global 64-bit atomic per touched cell and horizon"})
                md.close()
                del frd
            result["rollout_D_132x132x60_T10"] = {
                "workload": "D: C's grid, PREDICTION_TIMES = 10, horizons 0.2*(k+1) s; k_resample_ms = k_resample (cull + mass + "
                            "mean velocity + resampling; static particles add their mass once) + k_rollout (the moving "
                            "particles' future mass: LDS windows + coalesced atomics), timed with HIP events; "
                            "static_fill = SURVEY's saturated zero-velocity state, "
                            "moving_fill = same fill with velocities uniform in +-1 m/s",
                **out}
        except Exception as e:
            result["rollout_D_132x132x60_T10"] = {"error": repr(e)}

    # ------------------------------------------------------------------ the metric's workload with the other birth-tag sources
    if want("variants"):
        try:
            var = {}
            for tag, est, nst in (("static_tags", 0, 150), ("host_estimator", 1, 150), ("device_estimator_300_steps", 2, 300)):
                mv, frv, dtv, cv, _ = measure(wl, nst, 15, args.prefill, profile=False, estimator=est)
                var[tag] = {"frames_per_s": round(nst / dtv, 2), "ms_per_step": round(dtv / nst * 1e3, 4),
                            "n_born": cv["n_born"], "n_live_in": cv["n_live_in"]}
                mv.close()
                del frv
            result["birth_tag_variants"] = {
                "what": "workload B with the two other sources of the birth tags: static_tags = every point in view is a "
                        "zero-velocity source (round 1's headline); host_estimator = the reference's helper thread (:297,311) as "
                        "the host stage of velocity_estimator.cpp (one D2H + H2D round trip of the cloud per frame).  The "
                        "headline value runs the estimator on the device (dspmap_velest.hip); device_estimator_300_steps is "
                        "that same configuration timed over 300 steps (the contract line times --steps, 20 in the driver's "
                        "run: 4 ms, where the first frames after the barrier weigh in).", **var}
        except Exception as e:
            result["birth_tag_variants"] = {"error": repr(e)}

    # ------------------------------------------------------------------ the same frame as plain launches (DSPMAP_P_USE_GRAPH = 2)
    if want("plain"):
        try:
            mp = make_map(wl)
            if args.estimator:
                mp.set_param(D.capi.P_VELOCITY_ESTIMATOR, args.estimator)
            mp.set_param(D.capi.P_USE_GRAPH, 2)
            frp = gen_frames(wl, args.prefill + 330, seed=1234)
            run_frames(mp, frp[:args.prefill + 30])
            mp.sync()
            quiet_host()
            t0 = time.perf_counter()
            run_frames(mp, frp[args.prefill + 30:])
            t_issue = time.perf_counter() - t0
            mp.sync()
            dtp = (time.perf_counter() - t0) / 300
            gc.enable()
            result["plain_launches_66x66x40"] = {
                "what": "NOT the contract line's configuration: the metric's workload with the frame queued as plain launches instead of a graph "
                        "replay (DSPMAP_P_USE_GRAPH = 2: same kernels, parameter block through the same pinned ring, estimator on its own "
                        "stream) -- no graph boundary between two frames (8.7 us on this runtime), but twelve launches of host work per frame: "
                        "the device goes faster, the host-pointer update() becomes host-bound, hence not the default (include/dspmap.h)",
                "frames_per_s": round(1.0 / dtp, 1), "ms_per_frame": round(dtp * 1e3, 4), "host_enqueue_ms_per_frame": round(t_issue / 300 * 1e3, 4)}
            mp.close()
        except Exception as e:
            result["plain_launches_66x66x40"] = {"error": repr(e)}

    # ------------------------------------------------------------------ the boundary's own call: update(float* host, ...)
    if want("host"):
        try:
            mh = make_map(wl)
            if args.estimator:
                mh.set_param(D.capi.P_VELOCITY_ESTIMATOR, args.estimator)
            frh = gen_frames(wl, args.prefill + 330, seed=1234)
            run_frames(mh, frh[:args.prefill])
            host = [(np.ascontiguousarray(p.cpu().numpy(), np.float32), pos, quat, t) for p, pos, quat, t in frh[args.prefill:]]
            torch.cuda.synchronize()

            def host_frames(fr):
                for pts, pos, quat, t in fr:
                    # dspmap_update: what DSPMap::update(int, int, float*, ...) forwards to (src/map_sim_example.cpp:345-347):
                    # the caller's HOST cloud is copied into a slot of a pinned, device-mapped ring during the call and the captured
                    # frame's first kernel reads it over the bus (no copy node, no event: one graph launch per frame, round 5);
                    # returns after enqueue like the reference returns after its own work
                    assert mh.L.dspmap_update(mh.h, pts.shape[0], 3, pts.ctypes.data_as(C.c_void_p), pos[0], pos[1], pos[2], t,
                                              quat[0], quat[1], quat[2], quat[3]) == 1
                    mh.clearOccupancyMapPrediction()
            host_frames(host[:30])
            mh.sync()
            quiet_host()
            t0 = time.perf_counter()
            host_frames(host[30:])
            mh.sync()
            dth = (time.perf_counter() - t0) / len(host[30:])
            gc.enable()
            result["host_update_66x66x40"] = {
                "what": "the drop-in boundary's own call, PCIe-inclusive: dspmap_update(float* HOST cloud, pose) -- what DSPMap::update "
                        "(include/dsp_dynamic.h, reference :181) forwards to and src/map_sim_example.cpp:345-347 calls -- %d frames of "
                        "workload B, cloud copied into a pinned device-mapped ring and read over the bus by the frame's first kernel, "
                        "estimator on the device; never the contract line's value (the contract defines that one with the cloud "
                        "resident in HBM)" % len(host[30:]),
                "ratio_to_device_resident": round((1.0 / dth) / fps, 4),
                "frames_per_s": round(1.0 / dth, 1), "ms_per_frame": round(dth * 1e3, 4),
                "h2d_bytes_per_frame": int(host[-1][0].nbytes), "estimator_path": mh.estimator_path()}
            # the boundary's own call beside `value` in the line's first level (the contract keeps `value` for inputs resident in HBM)
            result["value_host_pointer_update"] = round(1.0 / dth, 1)
            mh.close()
            del frh, host
        except Exception as e:
            result["host_update_66x66x40"] = {"error": repr(e)}

    # ------------------------------------------------------------------ the reference node's loop: update() + the getter
    if want("node"):
        try:
            mn = make_map(wl)
            if args.estimator:
                mn.set_param(D.capi.P_VELOCITY_ESTIMATOR, args.estimator)
            frn = gen_frames(wl, args.prefill + 110, seed=1234)
            run_frames(mn, frn[:args.prefill])
            xyz = np.zeros((mn.V_local, 3), np.float32)
            futb = np.zeros((mn.V_local, mn.T), np.float32)
            nocc = C.c_int()
            pxyz, pfut = xyz.ctypes.data_as(C.c_void_p), futb.ctypes.data_as(C.c_void_p)

            def node_frame(fr):
                pts, pos, quat, t = fr
                assert mn.update_device(pts.data_ptr(), pts.shape[0], pos, t, quat) == 1
                # src/map_sim_example.cpp:378: occupied voxels + the [V][T] future status into host arrays (also clears the accumulators)
                mn._chk(mn.L.dspmap_get_occupancy_with_future(mn.h, 0.2, pxyz, mn.V_local, C.byref(nocc), pfut))
            for fr in frn[args.prefill:args.prefill + 10]:
                node_frame(fr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for fr in frn[args.prefill + 10:]:
                node_frame(fr)
            dtn = (time.perf_counter() - t0) / 100
            result["node_loop_66x66x40"] = {
                "what": "what the reference's node does per cloud (src/map_sim_example.cpp:345-384): update() and then "
                        "getOccupancyMapWithFutureStatus into host arrays -- a synchronous call that compacts the occupied voxels on "
                        "the device and copies them and the [V][T] future status (4.2 MB) over PCIe into pageable memory; never the "
                        "contract line's value",
                "frames_per_s": round(1.0 / dtn, 1), "ms_per_frame": round(dtn * 1e3, 4), "occupied_voxels": int(nocc.value)}
            mn.close()
            del frn
        except Exception as e:
            result["node_loop_66x66x40"] = {"error": repr(e)}

    # ------------------------------------------------------------------ next row: the caller's pre-processing on the device
    if want("preprocess"):
        try:
            mp = make_map(wl)
            sc = scene_mod.CorridorScene(wl["nx"] * wl["res"], wl["ny"] * wl["res"], wl["nz"] * wl["res"], seed=1234, device=dev)
            raws = [sc.raw(f / 30.0)[0] for f in range(8)]
            outb = torch.zeros((5000, 3), dtype=torch.float32, device=dev)
            for r in raws[:2]:
                mp.preprocess_cloud(r.data_ptr(), r.shape[0], outb.data_ptr(), 5000, leaf=0.1, swap_axes=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps, kept = 0, 0
            for _ in range(10):
                for r in raws:
                    kept, _ = mp.preprocess_cloud(r.data_ptr(), r.shape[0], outb.data_ptr(), 5000, leaf=0.1, swap_axes=False)
                    reps += 1
            dtp = (time.perf_counter() - t0) / reps
            result["preprocess_640x480"] = {
                "what": "dspmap_preprocess_cloud: 0.1 m voxel-grid centroid filter + crop + cap on the device "
                        "(src/map_sim_example.cpp:309-336), synchronous call incl. two small D2H reads",
                "points_in": int(raws[0].shape[0]), "points_out": int(kept), "ms_per_cloud": round(dtp * 1e3, 4)}
            mp.close()
        except Exception as e:
            result["preprocess_640x480"] = {"error": repr(e)}

    # ------------------------------------------------------------------ strong-scaling origin: config E on ONE GPU, unsharded
    if rank == 0 and not args.no_extra and (wl_name == "B" or sharded_run) and os.environ.get("DSPMAP_BENCH_ORIGIN", "1") == "1" and (sharded_run or want("origin")):
        try:
            if sharded_run:
                m.close()
            we = WORKLOADS["E_sat"]
            # the origin of an N-rank run times THE SAME FRAMES FROM THE SAME STATE as the ranks did (same prefill, warmup and steps as
            # measure_sharded: the saturated fill erodes from frame to frame, and round 5's origin -- 12 frames after 5 against `steps`
            # frames after 5 + warmup -- showed a "speedup" of 1.305 at world 1); the stand-alone block of the N = 1 line keeps
            # its short run
            o_steps, o_warm, o_pre = (args.steps, args.warmup, 5) if sharded_run else (12, 2, 3)
            me, fre, dte, ce, ste = measure(we, o_steps, o_warm, o_pre, profile=False, solo=True)
            be = b_alg(ce, me.V_local, me.T)
            mse = dte / o_steps * 1e3
            result["single_gpu_264x264x80"] = {
                "workload": "E_sat unsharded on one GPU (the N = 1 point of the Z-slab strong-scaling series that "
                            "bench.py --gpus N > 1 reports)",
                "frames_per_s": round(o_steps / dte, 2), "ms_per_step": round(mse, 4), "b_alg_bytes": int(be),
                "frac_of_8TBps": round(be / (mse * 1e-3) / 1e9 / peak, 5),
                "frames": {"prefill": o_pre, "warmup": o_warm, "steps": o_steps}, "n_live_in": int(ce["n_live_in"])}
            me.close()
            del fre
            if sharded_run and wl_name == "E_sat":
                # the series the north star asks for: frames/s on the 264x264x80 grid at 1 / 2 / 4 / 8 GPUs.  Its N = 1
                # point is the unsharded map measured above in this same run -- NOT the default N = 1 bench line, which
                # runs the metric's own 66x66x40 workload.
                result["strong_scaling_264x264x80"] = {
                    "n_gpus": world, "frames_per_s": round(fps, 2), "one_gpu_frames_per_s": round(o_steps / dte, 2),
                    "speedup_vs_one_gpu": round(fps / (o_steps / dte), 3),
                    "n_live_in": {"sharded_all_ranks": int(cnt["n_live_in"]), "one_gpu": int(ce["n_live_in"])},
                    "same_frames": "both runs: 5 prefill + %d warmup frames untimed, then %d timed frames of the same stream from the same saturated fill" % (args.warmup, args.steps),
                    "note": "compare value with one_gpu_frames_per_s (same workload, same box), not with the N = 1 bench "
                            "line (workload B, 66x66x40)"}
        except Exception as e:
            result["single_gpu_264x264x80"] = {"error": repr(e)}

    # ------------------------------------------------------------------ projected multi-GPU critical path (N > 1 is unmeasured)
    if want("projection") and os.environ.get("DSPMAP_BENCH_PROJECTION", "1") == "1":
        try:
            sharded = __import__("dsp-map_amd.sharded", fromlist=["CppGroup"])
            we = WORKLOADS["E_sat"]
            NW = 8
            ckw = dict(nx=we["nx"], ny=we["ny"], nz=we["nz"], res=we["res"], ppv=we["ppv"], seed=1234)
            fre = gen_frames(we, 3 + 6, seed=1234)

            def group_run(ranges):
                g = sharded.CppGroup(D, ckw, len(ranges), device_index=local_rank, ranges=ranges)
                for mm in g.maps:
                    mm.seed_uniform(we["ppv"], 0.01, 99)
                g.create()

                def run(fr, tolerant=False):
                    for pts, pos, quat, t in fr:
                        try:
                            assert g.update(pts, pos, t, quat) == 1
                        except RuntimeError as e:   # the first frames size the exchange messages (an undersized one is reported once, then raised)
                            if not (tolerant and "exchange message" in str(e)):
                                raise
                        for mm in g.maps:
                            mm.clearOccupancyMapPrediction()
                run(fre[:3], tolerant=True)
                g.sync()
                g.set_profiling(True)
                run(fre[3:])
                g.sync()
                tab, nf = g.phase_ms()
                sel = tab[len(ranges)][3] > 0
                cnts = [mm.counters() for mm in g.maps]
                try:   # the longest pyramid list of the whole map (the slabs' lists added up) against the reference's capacity (:66): what decides
                    # whether a frame has to select the lists' cut over all ranks
                    pc = np.sum([np.asarray(mm.pyramid_counts(), np.int64) for mm in g.maps], axis=0)
                    cnts[0]["longest_list_over_capp"] = round(float(pc.max()) / float(g.maps[0].capp), 4)
                except Exception:
                    pass
                g.close()
                return tab, sel, cnts

            eq = sharded.slab_ranges(we["nz"], NW)
            tab_eq, sel_eq, cnt_eq = group_run(eq)
            crit_eq, per_eq = critical_path(tab_eq, NW, sel_eq)
            # slab boundaries by measured work: every layer of a slab is charged the slab's time per layer, the prefix sum is cut in equal parts
            layer_cost = []
            for (z0, z1), row in zip(eq, tab_eq[:NW]):
                layer_cost += [sum(row) / (z1 - z0)] * (z1 - z0)
            bal = balanced_ranges(layer_cost, NW)
            tab_b, sel_b, cnt_b = (tab_eq, sel_eq, cnt_eq) if bal == eq else group_run(bal)
            crit_b, per_b = critical_path(tab_b, NW, sel_b)
            ranges, tab, crit, per, selr, cnts = (bal, tab_b, crit_b, per_b, sel_b, cnt_b) if crit_b < crit_eq else (eq, tab_eq, crit_eq, per_eq, sel_eq, cnt_eq)
            # What a BLOCK-CYCLIC assignment would do to the critical path: 16 blocks of nz / 16 layers, rank r owning blocks r and r + 8 (a
            # lower and an upper one -- the field of view looks at the lower half, so every rank gets a share of the in-view work).  Not
            # built as a driver (a rank would hold two slabs and exchange with both neighbours for each); the 16 blocks run through the
            # group driver and a rank is charged the SUM of its two blocks in every segment.
            cyc = None
            try:
                if we["nz"] % 16 == 0:
                    r16 = sharded.slab_ranges(we["nz"], 16)
                    tab16, sel16, _ = group_run(r16)
                    per_rank = [[tab16[r][ph] + tab16[r + 8][ph] for ph in range(len(tab16[0]))] for r in range(8)] + [tab16[16]]
                    crit16, per16 = critical_path(per_rank, 8, sel16)
                    cyc = {"critical_path_ms": round(crit16, 4), "slowest_rank_per_segment_ms": [round(x, 4) for x in per16],
                           "mean_rank_ms": round(sum(sum(r) for r in per_rank[:8]) / 8, 4),
                           "what": "16 blocks of %d layers through the group driver, rank r = blocks r and r + 8, a rank's segment = the sum of its two "
                                   "blocks' (an emulation: no such driver exists)" % (we["nz"] // 16)}
            except Exception as e:
                cyc = {"error": repr(e)}
            # what the RCCL calls themselves cost a rank (launch + protocol, no wire): the C++ driver with a world-1 communicator against the
            # same driver inside a one-slab group (device copies / reduction kernels), on one rank's share of the map
            w8 = WORKLOADS["E8_sat"]
            ckw8 = dict(nx=w8["nx"], ny=w8["ny"], nz=w8["nz"], res=w8["res"], ppv=w8["ppv"], seed=1234)
            fr8 = gen_frames(w8, 4 + 12, seed=1234)

            def timed(update, sync, seed_maps):
                for mm in seed_maps:
                    mm.seed_uniform(w8["ppv"], 0.01, 99)
                for pts, pos, quat, t in fr8[:4]:
                    try:
                        assert update(pts, pos, t, quat) == 1
                    except Exception as e:   # (message sizing, see above)
                        if "exchange message" not in str(e):
                            raise
                    for mm in seed_maps:
                        mm.clearOccupancyMapPrediction()
                sync()
                t0 = time.perf_counter()
                for pts, pos, quat, t in fr8[4:]:
                    assert update(pts, pos, t, quat) == 1
                    for mm in seed_maps:
                        mm.clearOccupancyMapPrediction()
                sync()
                return (time.perf_counter() - t0) / 12 * 1e3
            g1 = sharded.CppGroup(D, ckw8, 1, device_index=local_rank)
            ms_group1 = timed(g1.update, g1.sync, g1.maps)
            g1.close()
            r1 = sharded.CppShardedRank(D, ckw8, 1, 0, local_rank)
            ms_rccl1 = timed(r1.update, r1.sync, [r1.map])
            r1.map.close()
            n_coll = 3 + (4 if selr else 0)   # the send/recv group, the Ck all-reduce, the n_static all-reduce (+ 4 digit all-reduces in frames that select)
            rccl_over = max(ms_rccl1 - ms_group1, 0.0)
            proj = crit + rccl_over + n_coll * XGMI_COLLECTIVE_US * 1e-3
            one = result.get("single_gpu_264x264x80", {}).get("ms_per_step")
            result["projected_8gpu_264x264x80"] = {
                "what": "a PROJECTION, not a measurement (no multi-GPU box in this environment; N > 1 has never run on hardware): the 8 "
                        "slabs of E_sat through the C++ frame driver inside ONE process (dspmap_mgpu_group_update), HIP events around every "
                        "phase of every slab; inside the group a slab has the GPU to itself for the length of its phase, as its own GPU "
                        "would.  projected_ms = sum over phases of the slowest slab + the RCCL calls' own cost at world 1 + %d collectives "
                        "x %.0f us ASSUMED xGMI latency" % (n_coll, XGMI_COLLECTIVE_US),
                "phases": list(sharded.CppGroup.GROUP_PHASES),
                "slab_ranges": [list(r) for r in ranges], "equal_height_ranges": [list(r) for r in eq],
                "phase_ms_per_slab": [[round(x, 4) for x in row] for row in tab[:NW]],
                "group_stand_ins_ms": [round(x, 4) for x in tab[NW]],
                "segments": [list(x) for x in SEGMENTS], "slowest_slab_per_segment_ms": [round(x, 4) for x in per],
                "critical_path_ms": round(crit, 4), "critical_path_equal_height_ms": round(crit_eq, 4),
                "mean_slab_ms": round(sum(sum(r) for r in tab[:NW]) / NW, 4),
                "rccl_calls_world1_ms": round(rccl_over, 4), "world1_driver_ms": round(ms_rccl1, 4), "one_slab_group_ms": round(ms_group1, 4),
                "assumed_xgmi_ms": round(n_coll * XGMI_COLLECTIVE_US * 1e-3, 4), "list_selection_ran": bool(selr),
                "projected_ms": round(proj, 4), "projected_frames_per_s": round(1e3 / proj, 1),
                "one_gpu_ms": one, "projected_speedup": round(one / proj, 2) if one else None,
                "projected_efficiency_8": round(one / proj / NW, 3) if one else None,
                "in_view_particles_per_slab": [int(c["n_fov"]) for c in cnts],
                "longest_list_over_capp": cnts[0].get("longest_list_over_capp"),
                "block_cyclic_emulation": cyc}
            del fre, fr8
        except Exception as e:
            result["projected_8gpu_264x264x80"] = {"error": repr(e)}

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1 only)
    if rank == 0 and not sharded_run and not args.no_cpu:
        try:
            from oracle import oracle_py as O
            import subprocess
            subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-B"])  # -march=native on THIS host
            fr = gen_frames(wl, 90, seed=1234)
            host = [(p.cpu().numpy(), pos, q, t) for p, pos, q, t in fr]
            o = O.Oracle(O.make_config(nx=wl["nx"], ny=wl["ny"], nz=wl["nz"], res=wl["res"], ppv=wl["ppv"]), fast=True)
            p_tab = np.zeros(10_000_000, np.float32)
            v_tab = np.zeros(10_000_000, np.float32)
            o.L.dspo_fill_gaussian_tables(p_tab.ctypes.data_as(C.c_void_p), v_tab.ctypes.data_as(C.c_void_p),
                                          p_tab.size, 0.05, 0.05, 1234)
            o.set_tables(p_tab, v_tab)
            o.L.dspo_use_velocity_estimator(o.h, 2)
            t_acc, n_acc = 0.0, 0
            for i, (pts, pos, q, t) in enumerate(host):
                t0 = time.perf_counter()
                o.update(pts, pos, t, q)
                o.L.dspo_clear_future(o.h)
                e = time.perf_counter() - t0
                if i >= 30:  # steady state
                    t_acc += e
                    n_acc += 1
                if t_acc > 25.0:
                    break
            result["cpu_baseline"] = {
                "value": round(n_acc / t_acc, 3), "unit": "frames/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                "host_cores_available": os.cpu_count(),
                "sample": "oracle/dsp_oracle.c (reference-faithful dense AoS restatement, reference flags "
                          "-O3 -ffast-math -march=native, 1 thread) on frames 30..%d of the same stream" % (29 + n_acc),
                "ms_per_frame": round(t_acc / n_acc * 1e3, 2)}
            o.close()
        except Exception as e:
            result["cpu_baseline"] = {"error": repr(e)}

    if rank == 0:
        try:  # RCCL's banner and other C stdio output must not trail the JSON line
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
