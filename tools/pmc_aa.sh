#!/bin/bash
# SQ counters of the kernels on the protein shape (config 5): tools/ab_score.py 2000 workload=aa_50k_x500 with 100 HMMs.
# Run through gpurun from the repo root; two counter passes, kernel-trace only.
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export AB_NH=100 AB_REPS=1
out=gpurun_out/pmc_aa; mkdir -p $out
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $out/p1 -o pmc --output-format csv -- python3 tools/ab_score.py 2000 workload=aa_50k_x500 > $out/p1.log 2>&1
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_ANY -d $out/p2 -o pmc --output-format csv -- python3 tools/ab_score.py 2000 workload=aa_50k_x500 > $out/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_aa/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
with open("gpurun_out/pmc_aa/summary.txt", "w") as o:
    for k in sorted(acc):
        o.write(k + "\n")
        for c in sorted(acc[k]):
            o.write("  %-24s %.4g  (%d dispatches)\n" % (c, acc[k][c], cnt[(k, c)]))
print(open("gpurun_out/pmc_aa/summary.txt").read())
PY
