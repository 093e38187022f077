export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pb /tmp/pc /tmp/pe
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python $R/bench.py --steps 300 --warmup 30 --no-cpu --no-extra > /tmp/pb.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python $R/bench.py --workload C_sat --steps 40 --warmup 5 --prefill 3 --no-cpu --no-extra > /tmp/pc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -- python $R/bench.py --workload E_sat --steps 12 --warmup 2 --prefill 3 --no-cpu --no-extra > /tmp/pe.log 2>&1
cd $R
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) gpurun_out/r01_e_B_kernel_stats.csv
cp $(find /tmp/pc -name "*kernel_stats.csv" | head -1) gpurun_out/r01_e_Csat_kernel_stats.csv
cp $(find /tmp/pe -name "*kernel_stats.csv" | head -1) gpurun_out/r01_e_Esat_kernel_stats.csv
tail -1 /tmp/pb.log > gpurun_out/r01_e_B_bench.json; tail -1 /tmp/pc.log > gpurun_out/r01_e_Csat_bench.json; tail -1 /tmp/pe.log > gpurun_out/r01_e_Esat_bench.json
# PMC passes (separate runs, counters only)
for W in B C_sat; do
  if [ $W = B ]; then ARGS="--steps 100 --warmup 10 --no-cpu --no-extra"; else ARGS="--workload C_sat --steps 20 --warmup 3 --prefill 3 --no-cpu --no-extra"; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$W_$C
    (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${W}_$C -- python $R/bench.py $ARGS > /tmp/pmc_${W}_$C.log 2>&1)
    f=$(find /tmp/pmc_${W}_$C -name "*counter_collection.csv" | head -1)
    python profiles/pmc_reduce.py $f gpurun_out/pmc_${W}_$C.json > /dev/null
  done
done
ls -la gpurun_out | tail -12
