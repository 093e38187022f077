#!/bin/bash
# A/B of bench.py's large-map blocks (saturated C, config D rollout, E on one GPU, realistic fills) under environment switches, one box.
# usage: tools/ab_extras.sh "VAR=val ..." "VAR=val" ...   -> gpurun_out/ab_extras.txt
out=gpurun_out/ab_extras.txt
mkdir -p gpurun_out
: > $out
for arm in "$@"; do
  env $arm python bench.py --gpus 1 --steps 20 --warmup 5 --prefill 20 --only saturated,rollout,origin,realistic --no-cpu 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read())
s=d.get('saturated_132x132x60',{}); r=d.get('rollout_D_132x132x60_T10',{}); e=d.get('single_gpu_264x264x80',{}); f=d.get('realistic_fill',{})
print('C_sat ms', s.get('ms_per_step'), 'stages', {k: round(v,4) for k,v in (s.get('stage_ms') or {}).items()})
print('D static', r.get('static_fill',{}).get('update_ms'), r.get('static_fill',{}).get('k_resample_ms'), 'D moving', r.get('moving_fill',{}).get('update_ms'), r.get('moving_fill',{}).get('k_resample_ms'), r.get('error'))
print('E_sat 1gpu ms', e.get('ms_per_step'), e.get('error'))
print('realistic C', f.get('132x132x60',{}).get('ms_per_step'), 'E', f.get('264x264x80',{}).get('ms_per_step'), f.get('error'))
" > /tmp/arm.txt
  echo "[$arm]" >> $out; cat /tmp/arm.txt >> $out
done
cat $out
