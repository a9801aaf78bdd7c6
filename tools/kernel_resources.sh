#!/bin/bash
# usage: res.sh file.hip [regex]  -> compact kernel resource table
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNDEBUG -c $1 -o /tmp/res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('VGPRs','AGPRs','VGPRs Spill','SGPRs Spill','Occupancy \[waves/SIMD\]','ScratchSize \[bytes/lane\]'):
        m=re.search(r' '+k+r': (\d+)',l)
        if m and cur: rows[cur][k]=m.group(1)
for k,v in rows.items():
    if '$2' and not re.search('$2',k): continue
    print(k[:70], v)
"
