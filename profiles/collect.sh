#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to (run on the MI355X box from the repo root):
#   kernel stats of the three bench commands, PMC FETCH_SIZE / WRITE_SIZE in separate counter-only passes,
#   and the calibration of those counters against streams of known size (dspmap_debug_stream).
# Outputs go to gpurun_out/ (scratch); profiles/pmc_merge.py turns them into the tracked summaries.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01_e}
cd /tmp
declare -A ARGS
ARGS[B]="--steps 300 --warmup 30 --no-cpu --no-extra"
ARGS[C_sat]="--workload C_sat --steps 40 --warmup 5 --prefill 3 --no-cpu --no-extra"
ARGS[E_sat]="--workload E_sat --steps 12 --warmup 2 --prefill 3 --no-cpu --no-extra"
for W in B C_sat E_sat; do
  rm -rf /tmp/ks_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$W -- python $R/bench.py ${ARGS[$W]} > /tmp/ks_$W.log 2>&1
  cp $(find /tmp/ks_$W -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_${W}_kernel_stats.csv
  grep '^{"metric' /tmp/ks_$W.log | tail -1 > $R/gpurun_out/${TAG}_${W}_bench.json
done
ARGS[B]="--steps 100 --warmup 10 --no-cpu --no-extra"
ARGS[C_sat]="--workload C_sat --steps 20 --warmup 3 --prefill 3 --no-cpu --no-extra"
for W in B C_sat; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${W}_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${W}_$C -- python $R/bench.py ${ARGS[$W]} > /tmp/pmc_${W}_$C.log 2>&1
    python $R/profiles/pmc_reduce.py $(find /tmp/pmc_${W}_$C -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_pmc_${W}_$C.json > /dev/null
  done
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  (cd $R && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/cal_$C -- python tools/calib.py > /tmp/cal_$C.log 2>&1)
  python $R/profiles/pmc_reduce.py $(find /tmp/cal_$C -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_cal_$C.json > /dev/null
  grep bytes /tmp/cal_$C.log > $R/gpurun_out/${TAG}_cal_bytes.txt
done
# issue-slot picture of the sweeps and the pair kernels (their "VALU-bound" / "latency-bound" labels in DESIGN.md): SQ counters
for W in B C_sat; do
  rm -rf /tmp/sq_$W
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/sq_$W -- python $R/bench.py ${ARGS[$W]} > /tmp/sq_$W.log 2>&1
  python $R/profiles/pmc_reduce.py $(find /tmp/sq_$W -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_sq_$W.json $(find /tmp/sq_$W -name "*kernel_trace.csv" | head -1) > /dev/null
done
# the sources these figures belong to (bench.py prints the traffic only while the fingerprint still matches)
(cd $R && python -c "import bench; print(bench.csrc_fingerprint())" > $R/gpurun_out/${TAG}_csrc_sha16.txt)
ls $R/gpurun_out | grep $TAG
