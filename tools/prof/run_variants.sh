#!/bin/bash
# usage: run_variants.sh "<bench args>" v0 v1 ...
args="$1"; shift
for v in "$@"; do
  cp scratch/so/$v.so dsp-map_amd/lib/libdspmap_hip.so
  python bench.py $args > gpurun_out/exp_$v.json 2> gpurun_out/exp_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/exp_$v.json").read().strip().splitlines()[-1])
    print("$v", d["ms_per_step"], json.dumps(d["frame"]["stage_ms"]))
except Exception as e:
    print("$v", "FAILED", e)
PY
done
