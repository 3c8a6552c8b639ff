cd $GRAFT_REPO_ROOT
export WH_RES_NO_PREFETCH=1
for wv in 1 2 4 8; do
WH_RES_WAVES=$wv WH_STATS=1 timeout -k 10 400 python bench.py --workload aa_50k_x500 --nq 1000 --steps 1 --warmup 0 --no-cpu-baseline --no-level1 --no-also > gpurun_out/c5_w$wv.json 2> gpurun_out/c5_w$wv.err || exit 1
echo "waves $wv"; grep "cycles per fetch\|load alone\|resolver wave cycles" gpurun_out/c5_w$wv.err | tail -3
python3 -c "
import json;d=json.load(open('gpurun_out/c5_w$wv.json'));print(d['stage_ms_per_step']['score_parts'])"
done
