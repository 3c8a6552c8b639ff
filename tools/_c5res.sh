cd $GRAFT_REPO_ROOT
WH_STATS=1 timeout -k 10 500 python bench.py --workload aa_50k_x500 --nq 3000 --steps 1 --warmup 1 --no-cpu-baseline --no-level1 --no-also > gpurun_out/c5_res.json 2> gpurun_out/c5_res.err
grep -i "resolv\|trace\|fetch" gpurun_out/c5_res.err | tail -12
python3 -c "
import json;d=json.load(open('gpurun_out/c5_res.json'));print(d['value'],d['stage_ms_per_step'],d['config']['pairs_multidomain_rank0'])"
