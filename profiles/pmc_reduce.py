"""Reduce a rocprofv3 --pmc counter_collection CSV to per-kernel averages for this repo's kernels (k_*)."""
import csv
import json
import sys
from collections import defaultdict


def main(path, out):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "").replace("void ", "").split("(")[0]
            if not name.startswith("k_"):
                continue
            c = r.get("Counter_Name")
            v = float(r.get("Counter_Value", 0))
            a = acc[name][c]
            a[0] += v
            a[1] += 1
    res = {k: {c: {"avg": a[0] / max(a[1], 1), "dispatches": a[1]} for c, a in v.items()} for k, v in acc.items()}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res)[:2000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
