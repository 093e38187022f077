#!/bin/bash
# VGPR / SGPR / LDS / occupancy of the kernels of one source file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
# usage: tools/kernel_resources.sh dsp-map_amd/csrc/dspmap_sweep.hip [name-filter]
f=${1:-dsp-map_amd/csrc/dspmap_sweep.hip}; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -x hip "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage --cuda-device-only 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); rows[cur]={}; continue
    for key in ('VGPRs','AGPRs','SGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(key+r': (\d+)',line)
        if m and cur: rows[cur][key.split(' ')[0]]=int(m.group(1))
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().split('(')[0]
    if re.search(r'$pat',name): print('%-60s'%name[:60],' '.join('%s=%s'%(a,b) for a,b in v.items()))
"
