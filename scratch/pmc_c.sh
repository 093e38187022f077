export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -- python $R/bench.py --workload C_sat --steps 20 --warmup 3 --prefill 3 --no-cpu --no-extra > /tmp/pmc_$C.log 2>&1)
  f=$(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1)
  python $R/profiles/pmc_reduce.py $f /tmp/pmc_$C.json > /dev/null
  python - <<PY
import json
j=json.load(open("/tmp/pmc_$C.json"))
for k in ("k_predict<1, 4>","k_place<1>","k_resample<1>","k_birth_insert"):
    for kk,v in j.items():
        if kk.startswith(k.split("<")[0]): print("$C", kk, round(v["$C"]["avg"]/1024,1), "MB(KB-units)")
PY
done
