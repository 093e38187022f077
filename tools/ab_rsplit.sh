#!/bin/bash
# A/B of DSPMAP_P_RESAMPLE_SPLIT on identical saturated maps (maps of one process differ by where their arrays landed: compare NEIGHBOURING arms)
out=gpurun_out/ab_rsplit.txt
mkdir -p gpurun_out
export DSPMAP_SIDE_PLACEMENT=${SIDE:-19}
{
echo "== E_sat (DSPMAP_SIDE_PLACEMENT=$DSPMAP_SIDE_PLACEMENT)"; python tools/ab_maps.py --workload E_sat --param RESAMPLE_SPLIT --values 0,1,0,1,0,1 --frames 24 --skip 5 2>&1 | grep -v amdgpu.ids | grep -v "^ \|Traceback\|Assert"
} > $out
cat $out
