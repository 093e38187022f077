#!/bin/bash
# A/B of the headline workload with the DRIVER's command line (bench.py --gpus 1 --steps 20 --warmup 5), interleaved on one box:
# DSPMAP_ESTIMATOR_QUEUE 0 / 1 (and, for reference, plain launches).  Output: gpurun_out/ab_headline.txt, one line per run.
out=gpurun_out/ab_headline.txt
mkdir -p gpurun_out
: > $out
N=${1:-5}
for i in $(seq 1 $N); do
  for q in 0 1; do
    v=$(DSPMAP_ESTIMATOR_QUEUE=$q python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('estimator_path'))")
    echo "run $i queue $q : $v" >> $out
  done
done
cat $out
