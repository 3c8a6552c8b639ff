cd $GRAFT_REPO_ROOT
for pf in 0 1; do
if [ $pf = 0 ]; then export WH_RES_NO_PREFETCH=1; else unset WH_RES_NO_PREFETCH; fi
WH_STATS=1 timeout -k 10 300 python bench.py --workload aa_50k_x500 --nq 3000 --steps 1 --warmup 1 --no-cpu-baseline --no-level1 --no-also > gpurun_out/c5_res_$pf.json 2> gpurun_out/c5_res_$pf.err || exit 1
echo "prefetch $pf"; grep "cycles per fetch\|fetch order\|inside the traces\|load alone" gpurun_out/c5_res_$pf.err | tail -4
python3 -c "
import json;d=json.load(open('gpurun_out/c5_res_$pf.json'));print(d['stage_ms_per_step']['score_parts'], d['config']['topk_crc32'])"
done
