#!/bin/bash
# A/B of the headline workload with the DRIVER's command line (bench.py --gpus 1 --steps 20 --warmup 5) under environment switches,
# interleaved on one box.  usage: tools/ab_headline.sh N "VAR=val ..." "VAR=val" ...   -> gpurun_out/ab_headline.txt, one line per run
out=gpurun_out/ab_headline.txt
mkdir -p gpurun_out
: > $out
N=${1:-5}; shift
for i in $(seq 1 $N); do
  for arm in "$@"; do
    v=$(env $arm python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('config',{}).get('estimator_path'))")
    echo "run $i [$arm] : $v" >> $out
  done
done
cat $out
python - <<'PY'
import re,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/ab_headline.txt'):
    m=re.match(r'run \d+ \[(.*)\] : ([\d.]+)',l)
    if m: d[m.group(1)].append(float(m.group(2)))
for k,v in d.items(): print('mean [%s] = %.1f frames/s over %d runs (min %.1f max %.1f)'%(k,sum(v)/len(v),len(v),min(v),max(v)))
PY
