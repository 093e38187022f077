#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/tl
# usage: timeline.sh [workload [steps [warmup]]]   (frames from the TIMED region: the stage-profiled frames at the end of a
# bench run keep large maps on one stream)
W=${1:-B}; S=${2:-60}; WU=${3:-10}
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --workload $W --steps $S --warmup $WU --no-cpu --no-extra > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kernel_Name"].replace("void ", "").startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete frames: find k_obs_points starts
idx = [i for i, r in enumerate(rows) if "k_obs_points" in r["Kernel_Name"]]
mid = len(idx) // 2 - 3   # (the run's second half is the stage-profiled repeat of the timed frames)
for fi in (mid, mid + 1, mid + 2):
    a, b = idx[fi], idx[fi + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    print("frame", fi)
    for r in rows[a:b + 1]:
        n = r["Kernel_Name"].replace("void ", "").split("(")[0][:34]
        print("  %-34s queue %-3s start %7.1f us  end %7.1f us  (%5.1f)" % (n, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
