cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_waves_per_pair or models_beyond_3072 or backbone_of_more_than_3072 or multihit_queries_on_a_long_model" > gpurun_out/wide_tests.log 2>&1 || { tail -30 gpurun_out/wide_tests.log; exit 1; }
tail -3 gpurun_out/wide_tests.log
WH_TRACE=1 timeout -k 10 300 python bench.py --workload dna_6k_nodes --steps 2 --warmup 1 --no-cpu-baseline --no-level1 --no-also > gpurun_out/wide_bench.json 2> gpurun_out/wide_bench.err || { tail -20 gpurun_out/wide_bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/wide_bench.json'));print(d['value'],d['stage_ms_per_step'],d['config'].get('topk_crc32'))"
grep "wide" gpurun_out/wide_bench.err | tail -4
