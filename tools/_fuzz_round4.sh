cd $GRAFT_REPO_ROOT
set -o pipefail
( timeout -k 10 500 python tests/tools/fuzz_align.py 6100 9 3100 6100 1 | tail -3 ) > gpurun_out/fuzz_wide_tr.log 2>&1 || { echo "FAIL wide_tr"; tail -5 gpurun_out/fuzz_wide_tr.log; exit 1; }
( timeout -k 10 500 python tests/tools/fuzz_align.py 6300 4 3100 6100 3 | tail -3 ) > gpurun_out/fuzz_wide_tr_long.log 2>&1 || { echo "FAIL wide_tr_long"; tail -5 gpurun_out/fuzz_wide_tr_long.log; exit 1; }
( timeout -k 10 500 python tests/tools/fuzz_align.py 6500 3 6200 9000 1 | tail -3 ) > gpurun_out/fuzz_wide_24.log 2>&1 || { echo "FAIL wide_24"; tail -5 gpurun_out/fuzz_wide_24.log; exit 1; }
( timeout -k 10 500 python tests/tools/fuzz_align.py 700 9 300 1600 6 | tail -3 ) > gpurun_out/fuzz_sg.log 2>&1 || { echo "FAIL sg"; tail -5 gpurun_out/fuzz_sg.log; exit 1; }
for f in wide_tr wide_tr_long wide_24 sg; do tail -n 1 gpurun_out/fuzz_$f.log; done
