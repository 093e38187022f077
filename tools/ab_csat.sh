#!/bin/bash
# A/B of the saturated 132x132x60 map (bench.py's C_sat block) under environment switches, interleaved on one box.
# usage: tools/ab_csat.sh N "VAR=val VAR2=val" "VAR=val" ...   (each quoted argument = one arm; "" = defaults)   -> gpurun_out/ab_csat.txt
out=gpurun_out/ab_csat.txt
mkdir -p gpurun_out
: > $out
N=${1:-3}; shift
for i in $(seq 1 $N); do
  for arm in "$@"; do
    v=$(env $arm python bench.py --gpus 1 --steps 20 --warmup 5 --prefill 20 --only saturated --no-cpu 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['saturated_132x132x60']; print(s.get('ms_per_step'), s.get('frac_of_8TBps'), s.get('counters',{}).get('n_live_in'), s.get('error'))")
    echo "run $i [$arm] : $v" >> $out
  done
done
cat $out
