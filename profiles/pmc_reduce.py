"""Reduce a rocprofv3 --pmc counter_collection CSV to per-kernel averages for this repo's kernels (k_*).
With the kernel trace of the SAME run as third argument, every kernel also gets its average duration (`duration_ns`): the
denominator of the issue-slot shares in pmc_merge.py (SQ_BUSY_CYCLES sums the busy cycles of the shader engines that had work and
under-counts the elapsed time of a kernel that leaves some of them idle -- shares above 1 came from there)."""
import csv
import json
import sys
from collections import defaultdict


def base(name):
    return name.replace("void ", "").split("(")[0]


def main(path, out, trace=None):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = base(r.get("Kernel_Name", ""))
            if not name.startswith("k_"):
                continue
            c = r.get("Counter_Name")
            v = float(r.get("Counter_Value", 0))
            a = acc[name][c]
            a[0] += v
            a[1] += 1
    if trace:
        with open(trace) as f:
            for r in csv.DictReader(f):
                name = base(r.get("Kernel_Name", ""))
                if not name.startswith("k_"):
                    continue
                a = acc[name]["duration_ns"]
                a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a[1] += 1
    res = {k: {c: {"avg": a[0] / max(a[1], 1), "dispatches": a[1]} for c, a in v.items()} for k, v in acc.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res)[:2000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
