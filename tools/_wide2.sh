cd $GRAFT_REPO_ROOT
WH_STATS=1 WH_TRACE=1 timeout -k 10 300 python bench.py --workload dna_6k_nodes --steps 1 --warmup 1 --no-cpu-baseline --no-level1 --no-also > gpurun_out/wide_bench.json 2> gpurun_out/wide_bench.err || { tail -20 gpurun_out/wide_bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/wide_bench.json'));print(d['value'],d['stage_ms_per_step'],d['config'].get('topk_crc32'))"
grep "wide" gpurun_out/wide_bench.err | tail -6
