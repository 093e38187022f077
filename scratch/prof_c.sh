export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -- python $R/bench.py --workload C_sat --steps 40 --warmup 5 --prefill 3 --no-cpu --no-extra > /tmp/pc.log 2>&1
cd $R; cp $(find /tmp/pc -name "*kernel_stats.csv" | head -1) gpurun_out/prof_Csat_kernel_stats.csv
python profiles/summarize.py gpurun_out/prof_Csat_kernel_stats.csv Csat | head -12
