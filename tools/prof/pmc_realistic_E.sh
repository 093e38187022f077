#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
A="--workload E --steps 60 --warmup 10 --no-cpu --no-extra"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pe_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pe_$C -- python $R/bench.py $A > /tmp/pe_$C.log 2>&1
  python $R/profiles/pmc_reduce.py $(find /tmp/pe_$C -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG:-r04_a}_pmc_E_$C.json > /dev/null
done
rm -rf /tmp/pe_sq
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pe_sq -- python $R/bench.py $A > /tmp/pe_sq.log 2>&1
python $R/profiles/pmc_reduce.py $(find /tmp/pe_sq -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG:-r04_a}_sq_E.json $(find /tmp/pe_sq -name "*kernel_trace.csv" | head -1) > /dev/null
ls $R/gpurun_out | grep "${TAG:-r04_a}.*_E"
