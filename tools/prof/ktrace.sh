#!/bin/bash
# usage: ktrace.sh "<bench args>" [ENV=val ...]   -> per-kernel average durations
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
args="$1"; shift
cd /tmp; rm -rf /tmp/kt
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py $args > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Name"].replace("void ", "").startswith("k_")]
for r in rows[:18]:
    print("%-34s calls %5s avg %8.1f us  min %8.1f max %8.1f" % (r["Name"].split("(")[0].replace("void ", "")[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep '^{"metric' /tmp/kt.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
