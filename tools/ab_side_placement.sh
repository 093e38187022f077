#!/bin/bash
# A/B of DSPMAP_P_SIDE_PLACEMENT (16 * fork + workgroups per CU) on identical saturated maps -> gpurun_out/ab_side_placement.txt
out=gpurun_out/ab_side_placement.txt
mkdir -p gpurun_out
{
echo "== C_sat"; python tools/ab_maps.py --workload C_sat --param SIDE_PLACEMENT --values 3,5,7,19,21,35,37,39 --frames 48 --skip 8 2>&1 | grep -v amdgpu.ids
echo "== E_sat"; python tools/ab_maps.py --workload E_sat --param SIDE_PLACEMENT --values 3,5,19,35,37 --frames 20 --skip 5 2>&1 | grep -v amdgpu.ids
} > $out
cat $out
