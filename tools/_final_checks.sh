cd $GRAFT_REPO_ROOT
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "null2_by_trace or models_beyond_3072" > gpurun_out/new_tests.log 2>&1 || { tail -20 gpurun_out/new_tests.log; exit 1; }
tail -n 1 gpurun_out/new_tests.log
( timeout -k 10 600 python tests/tools/fuzz_align.py 400 24 | tail -n 2 ) > gpurun_out/fuzz_a.log 2>&1 || { echo FAIL a; tail gpurun_out/fuzz_a.log; exit 1; }
( timeout -k 10 600 python tests/tools/fuzz_align.py 3400 4 1600 3072 | tail -n 2 ) > gpurun_out/fuzz_b.log 2>&1 || { echo FAIL b; tail gpurun_out/fuzz_b.log; exit 1; }
( timeout -k 10 600 python tests/tools/fuzz_built_models.py 20 12 | tail -n 2 ) > gpurun_out/fuzz_c.log 2>&1 || { echo FAIL c; tail gpurun_out/fuzz_c.log; exit 1; }
( timeout -k 10 600 python tests/tools/fuzz_level1.py 300 6 4 | tail -n 2 ) > gpurun_out/fuzz_d.log 2>&1 || { echo FAIL d; tail gpurun_out/fuzz_d.log; exit 1; }
for f in a b c d; do tail -n 1 gpurun_out/fuzz_$f.log; done
timeout -k 10 300 python tools/bench_example.py > gpurun_out/bench_example.log 2>&1; tail -n 6 gpurun_out/bench_example.log
