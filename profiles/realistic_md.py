"""Turn the outputs of tools/prof/pmc_realistic_E.sh and tools/prof/timeline.sh (gpurun_out/<tag>_*) into the tracked summaries
profiles/<tag>_E_realistic_pmc.md, profiles/<tag>_B_timeline.md and profiles/<tag>_C_sat_timeline.md.
  python profiles/realistic_md.py r04_c"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(os.path.dirname(HERE), "gpurun_out")


def main(tag):
    sq = json.load(open(os.path.join(G, tag + "_sq_E.json")))
    fe = json.load(open(os.path.join(G, tag + "_pmc_E_FETCH_SIZE.json")))
    wr = json.load(open(os.path.join(G, tag + "_pmc_E_WRITE_SIZE.json")))
    cal = json.load(open(os.path.join(HERE, "pmc_traffic.json")))
    fr, wrr = cal.get("fetch_ratio", 0.5), cal.get("write_ratio", 1.0)
    out = ["# %s -- the realistic fill of 264x264x80 (depth stream from an empty map, ~510 k live particles in ~12 k of 87 120 tiles): "
           "HBM traffic, durations and issue slots per kernel" % tag, "",
           "`rocprofv3 --pmc <C> --kernel-trace -- python bench.py --workload E --steps 60 --warmup 10 --no-cpu --no-extra` (separate "
           "counter-only passes; `TAG=%s tools/prof/pmc_realistic_E.sh`); corrected bytes = FETCH_SIZE*1024/%.3f + WRITE_SIZE*1024/%.3f "
           "(the calibration of `profiles/%s_pmc_traffic.md`); duration = the SQ pass's kernel trace; `VALU issue` = 4 x "
           "SQ_ACTIVE_INST_VALU / (duration x 2.4 GHz x 1024 SIMDs), at most 1 by construction." % (tag, fr, wrr, tag), "",
           "| kernel | dispatches | avg us | FETCH_SIZE KB | WRITE_SIZE KB | corrected HBM bytes / launch | waves | VALU issue | waiting |",
           "|---|---|---|---|---|---|---|---|---|"]
    rows = []
    for k, v in sq.items():
        if "duration_ns" not in v:
            continue
        dur = v["duration_ns"]["avg"]
        f = fe.get(k, {}).get("FETCH_SIZE", {}).get("avg", 0.0)
        w = wr.get(k, {}).get("WRITE_SIZE", {}).get("avg", 0.0)
        issue = 4.0 * v["SQ_ACTIVE_INST_VALU"]["avg"] / (dur * 2.4 * 1024)
        wait = v["SQ_WAIT_ANY"]["avg"] / max(v["SQ_WAVE_CYCLES"]["avg"], 1.0)
        rows.append((dur, "| %s | %d | %.1f | %.1f | %.1f | %s | %d | %.2f | %.2f |" % (
            k, v["duration_ns"]["dispatches"], dur / 1e3, f, w, format(int(f * 1024 / fr + w * 1024 / wrr), ","), v["SQ_WAVES"]["avg"], issue, wait)))
    out += [r for _, r in sorted(rows, key=lambda x: -x[0])]
    open(os.path.join(HERE, tag + "_E_realistic_pmc.md"), "w").write("\n".join(out) + "\n")
    for wl, note in (("B", "Under the profiler every dependent launch costs ~6 us instead of ~2; queue 2 = the estimator's branch."),
                     ("C_sat", "Frames from the middle of a long run (the saturated map thins out as the sensor moves on: these frames are shorter "
                               "than the benchmark's); the second k_place (another queue) is the side-stream placement of the tiles without a view.")):
        src = os.path.join(G, "%s_%s_timeline.txt" % (tag, wl))
        if not os.path.exists(src):
            continue
        body = open(src).read().rstrip()
        args = "" if wl == "B" else " C_sat 40 5"
        open(os.path.join(HERE, "%s_%s_timeline.md" % (tag, wl)), "w").write(
            "# %s -- timeline of three replayed frames of workload %s (rocprofv3 --kernel-trace, start / end relative to the frame's first "
            "kernel)\n\n`tools/prof/timeline.sh%s`.  %s\n\n```\n%s\n```\n" % (tag, wl, args, note, body))


if __name__ == "__main__":
    main(sys.argv[1])
