"""Merge the outputs of profiles/collect.sh (gpurun_out/<tag>_*) into the tracked summaries:
   profiles/pmc_traffic.json  (read by bench.py for roofline.traffic; pmc_traffic_r01.json is round 1's)
   profiles/<tag>_pmc_traffic.md, profiles/<tag>_kernel_stats.md, profiles/<tag>_<W>_kernel_stats.csv"""
import io
import json
import os
import shutil
import sys
from contextlib import redirect_stdout

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import summarize  # noqa: E402

CMD = {
    "B": "python bench.py --steps 300 --warmup 30 --no-cpu --no-extra",
    "C_sat": "python bench.py --workload C_sat --steps 40 --warmup 5 --prefill 3 --no-cpu --no-extra",
    "E_sat": "python bench.py --workload E_sat --steps 12 --warmup 2 --prefill 3 --no-cpu --no-extra",
}
PMC_CMD = {
    "B": "python bench.py --steps 100 --warmup 10 --no-cpu --no-extra",
    "C_sat": "python bench.py --workload C_sat --steps 20 --warmup 3 --prefill 3 --no-cpu --no-extra",
}


def base(name):
    return name.split("<")[0]


def main(tag):
    g = os.path.join(ROOT, "gpurun_out")
    cal_f = json.load(open(os.path.join(g, tag + "_cal_FETCH_SIZE.json")))
    cal_w = json.load(open(os.path.join(g, tag + "_cal_WRITE_SIZE.json")))
    cal_bytes = dict(l.split()[1::2] for l in open(os.path.join(g, tag + "_cal_bytes.txt")) if l.startswith("mode"))
    rd_bytes, wr_bytes = int(cal_bytes["0"]), int(cal_bytes["1"])
    f_ratio = cal_f["k_calib_read"]["FETCH_SIZE"]["avg"] * 1024 / rd_bytes
    w_ratio = cal_w["k_calib_write"]["WRITE_SIZE"]["avg"] * 1024 / wr_bytes
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate counter-only passes, KB per dispatch, averaged over the "
            "dispatches of the run). Calibration on the same box with streams of known size (dspmap_debug_stream: the "
            "sweeps' access pattern): FETCH_SIZE*1024 = %.3f x bytes read (%d B read), WRITE_SIZE*1024 = %.3f x bytes "
            "written (%d B). hbm_bytes = FETCH_SIZE*1024/%.3f + WRITE_SIZE*1024/%.3f." %
            (f_ratio, rd_bytes, w_ratio, wr_bytes, f_ratio, w_ratio))
    out = {"note": note, "commands": {w: "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- " + c for w, c in PMC_CMD.items()},
           "workloads": {}}
    md = ["# %s -- HBM traffic per kernel launch (PMC)\n" % tag, note, ""]
    for w in ("B", "C_sat"):
        jf = json.load(open(os.path.join(g, "%s_pmc_%s_FETCH_SIZE.json" % (tag, w))))
        jw = json.load(open(os.path.join(g, "%s_pmc_%s_WRITE_SIZE.json" % (tag, w))))
        rows = {}
        for k in jf:
            f = jf[k]["FETCH_SIZE"]["avg"]
            wv = jw.get(k, {}).get("WRITE_SIZE", {}).get("avg", 0.0)
            rows[base(k)] = {"fetch_kb": round(f, 1), "write_kb": round(wv, 1),
                             "hbm_bytes": int(f * 1024 / f_ratio + wv * 1024 / w_ratio)}
        out["workloads"][w] = dict(sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes"]))
        md += ["\n## workload %s\n" % w, "`rocprofv3 --pmc <C> --kernel-trace --output-format csv -- %s`\n" % PMC_CMD[w],
               "| kernel | FETCH_SIZE KB | WRITE_SIZE KB | corrected HBM bytes / launch |", "|---|---|---|---|"]
        for k, v in out["workloads"][w].items():
            md.append("| %s | %.1f | %.1f | %s |" % (k, v["fetch_kb"], v["write_kb"], format(v["hbm_bytes"], ",")))
    # rocprofv3's own kernel durations of the bench commands (--kernel-trace --stats): what bench.py's event-derived kernel
    # durations must agree with; bench.py prints the dominant kernel's fraction on both
    import csv
    out["kernel_us"] = {}
    for w in ("B", "C_sat", "E_sat"):
        try:
            rows = list(csv.DictReader(open(os.path.join(g, "%s_%s_kernel_stats.csv" % (tag, w)))))
        except OSError:
            continue
        d = {}
        for r in rows:
            n = r["Name"].replace("void ", "").split("(")[0]
            if n.startswith("k_"):
                k = base(n)
                a = d.setdefault(k, [0.0, 0])
                a[0] += float(r["TotalDurationNs"]); a[1] += int(r["Calls"])
        out["kernel_us"][w] = {k: round(a[0] / max(a[1], 1) / 1e3, 2) for k, a in d.items()}
    # the sources the passes ran on, and the commit that holds them
    try:
        out["csrc_sha16"] = open(os.path.join(g, tag + "_csrc_sha16.txt")).read().strip()
    except OSError:
        out["csrc_sha16"] = None
    try:
        import subprocess
        out["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
        sys.path.insert(0, ROOT)
        import bench
        if out["csrc_sha16"] != bench.csrc_fingerprint():
            print("WARNING: the kernel sources changed since the PMC passes (%s vs %s): bench.py will not print this traffic"
                  % (out["csrc_sha16"], bench.csrc_fingerprint()))
    except Exception as e:  # noqa: BLE001
        out["commit"] = None
        print("commit not recorded:", e)
    out["tag"] = tag
    # issue-slot shares (SQ counters) per workload and kernel: bench.py's per-kernel ceiling table prices the VALU-bound kernels with them
    out["sq"] = {}
    for w in ("B", "C_sat"):
        try:
            js = json.load(open(os.path.join(g, "%s_sq_%s.json" % (tag, w))))
        except OSError:
            continue
        d = {}
        for k, v in js.items():
            a = {c: v.get(c, {}).get("avg", 0.0) for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "duration_ns")}
            if a["SQ_WAVES"] <= 0 or a["duration_ns"] <= 0:
                continue
            d[base(k)] = {"valu_issue": round(4.0 * a["SQ_ACTIVE_INST_VALU"] / (a["duration_ns"] * 2.4 * 1024.0), 4),
                          "waiting": round(a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0), 4)}
        out["sq"][w] = d
    json.dump(out, open(os.path.join(HERE, "pmc_traffic.json"), "w"), indent=1)
    # issue-slot picture (SQ counters)
    sq = ["# %s -- issue slots of the frame's kernels (SQ counters)\n" % tag,
          "`rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU "
          "SQ_INSTS_SALU --kernel-trace` on the bench commands of the PMC passes; averages per dispatch, summed over the device. "
          "SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_ACTIVE_INST_* count in units of 4 cycles. `VALU issue` = 4 x SQ_ACTIVE_INST_VALU / "
          "(kernel duration from the same run's kernel trace x 2.4 GHz x 1024 SIMDs): the share of all VALU issue slots of the chip "
          "the kernel used while it ran -- at most 1 by construction (the earlier tables divided by SQ_BUSY_CYCLES / 32, which sums the "
          "busy cycles of the shader engines that had work and under-counts the elapsed time of a kernel that leaves some of them idle: "
          "shares above 1); `waiting` = SQ_WAIT_ANY / SQ_WAVE_CYCLES: the share of its waves' lifetime spent in s_waitcnt / barriers.\n"]
    for w in ("B", "C_sat"):
        try:
            js = json.load(open(os.path.join(g, "%s_sq_%s.json" % (tag, w))))
        except OSError:
            continue
        sq += ["\n## workload %s\n" % w, "| kernel | waves | VALU inst / wave | SALU inst / wave | VALU issue | waiting |", "|---|---|---|---|---|---|"]
        for k, v in sorted(js.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", {}).get("avg", 0)):
            a = {c: v.get(c, {}).get("avg", 0.0) for c in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "duration_ns")}
            if a["SQ_WAVES"] <= 0 or a["SQ_BUSY_CYCLES"] <= 0:
                continue
            slots = a["duration_ns"] * 2.4 * 1024.0 if a["duration_ns"] > 0 else a["SQ_BUSY_CYCLES"] / 32.0 * 1024.0
            sq.append("| %s | %d | %.0f | %.0f | %.2f | %.2f |" % (base(k), a["SQ_WAVES"], a["SQ_INSTS_VALU"] / a["SQ_WAVES"], a["SQ_INSTS_SALU"] / a["SQ_WAVES"],
                                                                4.0 * a["SQ_ACTIVE_INST_VALU"] / slots,
                                                                a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0)))
    open(os.path.join(HERE, tag + "_sq_issue.md"), "w").write("\n".join(sq) + "\n")
    open(os.path.join(HERE, tag + "_pmc_traffic.md"), "w").write("\n".join(md) + "\n")
    buf = io.StringIO()
    with redirect_stdout(buf):
        for w in ("B", "C_sat", "E_sat"):
            src = os.path.join(g, "%s_%s_kernel_stats.csv" % (tag, w))
            shutil.copy(src, os.path.join(HERE, "%s_%s_kernel_stats.csv" % (tag, w)))
            summarize.main(src, "%s -- workload %s: rocprofv3 --kernel-trace --stats -- %s" % (tag, w, CMD[w]))
            try:
                j = json.loads(open(os.path.join(g, "%s_%s_bench.json" % (tag, w))).read())
                print("bench line of this run (under the profiler): %.1f frames/s, %.4f ms/step; stage_ms (HIP events) %s\n"
                      % (j["value"], j["ms_per_step"], json.dumps(j["frame"]["stage_ms"])))
            except Exception as e:  # noqa: BLE001
                print("(bench line not captured: %r)\n" % (e,))
    open(os.path.join(HERE, tag + "_kernel_stats.md"), "w").write(buf.getvalue())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01_e")
