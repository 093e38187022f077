#!/bin/bash
# usage: ab_env.sh "<bench args>" ENVVAR val1 val2 ...
args="$1"; var="$2"; shift; shift
for v in "$@"; do
  env $var=$v python bench.py $args > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
    print("$var=$v", d["ms_per_step"], json.dumps(d["frame"]["stage_ms"]))
except Exception as e:
    print("$v", "FAILED", e)
PY
done
