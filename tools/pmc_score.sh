#!/bin/bash
# Collect SQ stall/issue counters for the score kernel on a reduced workload (separate --pmc passes).
# usage: tools/pmc_score.sh <tag> [nq]
set -e
tag=${1:-pmc}; nq=${2:-2048}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set -d $out/p$i -o p$i --output-format csv -- python3 bench.py --nq $nq --steps 1 --warmup 0 --no-cpu-baseline --no-level1 > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/fail.log
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
with open("$out/summary.txt", "w") as o:
    for k in sorted(acc): o.write(f"{k} {acc[k]:.6g}\n")
print(open("$out/summary.txt").read())
PY
