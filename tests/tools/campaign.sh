#!/bin/bash
# Randomised campaign on the GPU box (run through gpurun from the repo root): the fuzz tools of this directory against the
# float64 oracle, on the default scoring kernel and on the alternative schedules (WH_SCORE_KERNEL is read at wh_ehmm_load).
# usage: tests/tools/campaign.sh <tag> [seeds per run]      -> gpurun_out/<tag>/*.log, a one-line summary per run on stdout
tag=${1:-campaign}; n=${2:-40}
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/$tag; mkdir -p $out
run() {  # name, kernel, tool, args...
  local name=$1 kern=$2; shift 2
  WH_SCORE_KERNEL=$kern timeout -k 10 400 python3 tests/tools/"$@" > $out/$name.log 2>&1
  echo "$name (WH_SCORE_KERNEL=$kern): exit $? : $(tail -1 $out/$name.log | cut -c1-200)"
}
run align_default 7 fuzz_align.py 7000 $n
run align_hybrid 10 fuzz_align.py 7100 $n
run align_split 11 fuzz_align.py 7200 $n
run align_quad 12 fuzz_align.py 7300 $n 900 1024
run resolver_default 7 fuzz_resolver.py 500 $((n * 2))
run resolver_split 11 fuzz_resolver.py 700 $n
run level1_default 7 fuzz_level1.py 900 $((n / 4 + 1)) 4
run built_default 7 fuzz_built_models.py 300 $n
run long_list_default 7 fuzz_long_list.py 100 $n
run long_list_hybrid 10 fuzz_long_list.py 300 $((n / 2))
