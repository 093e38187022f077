export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu --no-extra > /tmp/prof_b.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/prof_B_kernel_stats.csv
python profiles/summarize.py gpurun_out/prof_B_kernel_stats.csv "B"
