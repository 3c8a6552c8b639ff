cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_waves_per_pair" > gpurun_out/wide_tests.log 2>&1 || { tail -30 gpurun_out/wide_tests.log; exit 1; }
tail -n 1 gpurun_out/wide_tests.log
for q in 16 24; do
WH_FORCE_WIDE=$q WH_STATS=1 timeout -k 10 300 python bench.py --workload dna_6k_nodes --steps 2 --warmup 1 --no-cpu-baseline --no-level1 --no-also > gpurun_out/wide$q.json 2> gpurun_out/wide$q.err || exit 1
python3 -c "
import json;d=json.load(open('gpurun_out/wide$q.json'));print($q, d['value'],d['stage_ms_per_step'])"; grep "cycles of the first wave" gpurun_out/wide$q.err | tail -1
done
( timeout -k 10 500 python tests/tools/fuzz_align.py 6600 3 6200 8100 1 | tail -n 2 ) > gpurun_out/fuzz_w16.log 2>&1 || { echo FAIL; tail gpurun_out/fuzz_w16.log; exit 1; }
tail -n 1 gpurun_out/fuzz_w16.log
