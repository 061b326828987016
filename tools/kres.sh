#!/bin/bash
# kernel resource usage of a HIP source: tools/kres.sh file.hip [extra hipcc flags]  ->  name, VGPRs, spills, scratch, occupancy, LDS
src=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -o /dev/null "$src" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None
rows=[]
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for key,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('spill',r'VGPRs Spill: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)'),('sgpr',r' SGPRs: (\d+)')):
        m=re.search(pat,line)
        if m and cur is not None: cur[key]=m.group(1)
for r in rows:
    import subprocess
    name=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    print(f\"{name[:110]:110s} vgpr {r.get('vgpr')} agpr {r.get('agpr')} spill {r.get('spill')} scratch {r.get('scratch')} occ {r.get('occ')} lds {r.get('lds')} sgpr {r.get('sgpr')}\")
"
