cd $GRAFT_REPO_ROOT
WH_STATS=1 WH_NO_RESOLVE=1 timeout -k 10 500 python bench.py --workload aa_50k_x500 --nq 4000 --steps 1 --warmup 1 --no-cpu-baseline --no-level1 --no-also > gpurun_out/c5_stats.json 2> gpurun_out/c5_stats.err
