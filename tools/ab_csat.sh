#!/bin/bash
# A/B of the saturated 132x132x60 map (bench.py's C_sat block) with the frame as two branches (-1: the handle's choice) / serial (0),
# interleaved on one box.  Output: gpurun_out/ab_csat.txt
out=gpurun_out/ab_csat.txt
mkdir -p gpurun_out
: > $out
N=${1:-3}
for i in $(seq 1 $N); do
  for q in 0 -1; do
    v=$(DSPMAP_FRAME_BRANCHES=$q python bench.py --gpus 1 --steps 20 --warmup 5 --prefill 20 --only saturated --no-cpu 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['saturated_132x132x60']; print(s.get('ms_per_step'), s.get('frac_of_8TBps'), s.get('counters',{}).get('n_live_in'), s.get('error'))")
    echo "run $i branches $q : $v" >> $out
  done
done
cat $out
