#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [extra hipcc flags...]   -> one line per function: VGPRs, spills, scratch, occupancy
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Iwitch_amd/csrc "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | \
python3 -c '
import sys,re,subprocess
cur=None; d={}
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur=m.group(1); d[cur]={}; continue
    m=re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)",line)
    if m and cur: d[cur][m.group(1)]=int(m.group(2))
names=list(d)
dem=subprocess.run(["c++filt"]+names,capture_output=True,text=True).stdout.split("\n")
for n,dn in zip(names,dem):
    v=d[n]
    print("%-110s vgpr %3d spill %3d scratch %4d occ %d" % (dn[:110], v.get("VGPRs",-1), v.get("VGPRs Spill",-1), v.get("ScratchSize [bytes/lane]",-1), v.get("Occupancy [waves/SIMD]",-1)))
'
