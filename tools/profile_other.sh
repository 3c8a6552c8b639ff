#!/bin/bash
# rocprofv3 kernel-trace summaries of the two non-headline shapes the round changed: the several-waves-per-pair kernels
# (dna_6k_nodes) and the resolver on a slice of config 5.  usage (through gpurun): tools/profile_other.sh <tag>
set -e
tag=${1:-r04}; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace6k -o trace --output-format csv -- python3 bench.py --workload dna_6k_nodes --steps 2 --warmup 1 --no-cpu-baseline --no-level1 --no-also > $out/trace6k.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/tracec5 -o trace --output-format csv -- python3 bench.py --workload aa_50k_x500 --nq 3000 --steps 1 --warmup 1 --no-cpu-baseline --no-level1 --no-also > $out/tracec5.log 2>&1
for t in trace6k tracec5; do f=$(find $out/$t -name "*kernel_stats.csv" | head -1); cp "$f" $out/${t}_kernel_stats.csv; head -6 $out/${t}_kernel_stats.csv | cut -c1-160; done
