cd $GRAFT_REPO_ROOT
for wl in dna_rrna_like dna_m1900_long; do
for q in 0 12; do
if [ $q = 0 ]; then unset WH_FORCE_WIDE; else export WH_FORCE_WIDE=$q; fi
timeout -k 10 300 python bench.py --workload $wl --nq 2000 --steps 2 --warmup 1 --no-cpu-baseline --no-level1 --no-also > gpurun_out/mid_$q.json 2> gpurun_out/mid_$q.err || { tail -5 gpurun_out/mid_$q.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/mid_$q.json'));print('$wl', $q, d['value'],d['stage_ms_per_step'], d['config'].get('model_len_min'), d['config'].get('model_len_max'), d['config']['topk_crc32'])"
done; done
