#!/bin/bash
# Last GPU call of round 5 (run through gpurun from the repo root, on a budget of seconds): the tests that exercise the
# long-query ("SG") instantiations of the scoring kernels, the example-data stage times, then the profile passes of
# tools/profile_round.sh in the order that matters for the stamp (FETCH, WRITE first), each only if time is left.
# usage: tools/round5_final.sh <tag> <seconds the whole script may take>
tag=${1:-r05_v8}; limit=${2:-125}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/$tag; mkdir -p $out
timeout -k 5 95 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_queries or long_protein_queries or score_against_oracle or every_kernel_instantiation or end_to_end or underflow_to_zero" > $out/tests.log 2>&1
rc=$?; tail -3 $out/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; tail -30 $out/tests.log; exit 1; }
echo "tests done at ${SECONDS}s"
timeout -k 5 30 python3 tools/bench_example.py 4 2>&1 | grep -E "iter 1|align 1" | cut -c1-200 | tee $out/bench_example.log
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-level1 --no-also"
stamp() { python3 tools/prof_summary.py $out > /dev/null 2>&1; python3 tools/stamp_profile.py $out dna_100k_x200 2709930000000.0 > $out/stamp.log 2>&1 && cp profiles/traffic.json $out/stamped_traffic.json; }
for c in FETCH_SIZE WRITE_SIZE; do
  [ $((limit - SECONDS)) -lt 16 ] && { echo "no time for $c at ${SECONDS}s"; exit 0; }
  timeout -k 5 40 rocprofv3 --pmc $c -d $out/pmc_$c -o pmc --output-format csv -- $B > $out/pmc_$c.log 2>&1
done
stamp; echo "stamped (FETCH/WRITE) at ${SECONDS}s"
if [ $((limit - SECONDS)) -ge 16 ]; then
  timeout -k 5 40 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $out/pmc_SQ -o pmc --output-format csv -- $B > $out/pmc_SQ.log 2>&1
  stamp; echo "stamped (SQ) at ${SECONDS}s"
fi
if [ $((limit - SECONDS)) -ge 26 ]; then
  timeout -k 5 50 rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-level1 --no-also > $out/trace.log 2>&1
  stamp; echo "trace done at ${SECONDS}s"
fi
rm -rf $out/pmc_*/ $out/trace/ 2>/dev/null
cat $out/kernel_stats.csv 2>/dev/null | cut -c1-150
