python bench.py --no-cpu > gpurun_out/b22.json 2> gpurun_out/b22.err; python - <<PY
import json
j=json.loads(open("gpurun_out/b22.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["frame"]["stage_ms"])
s=j["saturated_132x132x60"]; print(s["frames_per_s"], s["ms_per_step"], s["frac_of_8TBps"], s["stage_ms"])
PY
