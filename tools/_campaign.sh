cd $GRAFT_REPO_ROOT
set -o pipefail
( timeout -k 10 900 python tests/tools/fuzz_resolver.py 1 24 | tail -n 6 ) > gpurun_out/camp_res.log 2>&1; echo "resolver rc=$?"; tail -n 2 gpurun_out/camp_res.log
( timeout -k 10 600 python tests/tools/fuzz_align.py 1000 60 | tail -n 2 ) > gpurun_out/camp_a.log 2>&1; echo "align rc=$?"; tail -n 1 gpurun_out/camp_a.log
( timeout -k 10 600 python tests/tools/fuzz_built_models.py 100 30 | tail -n 2 ) > gpurun_out/camp_c.log 2>&1; echo "built rc=$?"; tail -n 1 gpurun_out/camp_c.log
( timeout -k 10 600 python tests/tools/fuzz_topk.py 50 40 | tail -n 2 ) > gpurun_out/camp_t.log 2>&1; echo "topk rc=$?"; tail -n 1 gpurun_out/camp_t.log
