#!/bin/bash
# A/B of DSPMAP_P_TILE_BITMAPS on identical maps filled by the depth stream -> gpurun_out/ab_bits.txt
out=gpurun_out/ab_bits.txt
mkdir -p gpurun_out
{
echo "== E (264x264x80 filled by the depth stream)"; python tools/ab_maps.py --workload E --param TILE_BITMAPS --values 0,1,0,1 --frames 200 --skip 100 2>&1 | grep -v amdgpu.ids
echo "== C (132x132x60 filled by the depth stream)"; python tools/ab_maps.py --workload C --param TILE_BITMAPS --values 0,1,0,1 --frames 200 --skip 100 2>&1 | grep -v amdgpu.ids
} > $out
cat $out
