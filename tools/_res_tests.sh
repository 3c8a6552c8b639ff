cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resolver or multidomain or score_against_oracle or long_protein or config5_shape or more_than_eight or end_to_end or randomised_small" > gpurun_out/res_tests.log 2>&1 || { tail -30 gpurun_out/res_tests.log; exit 1; }
tail -3 gpurun_out/res_tests.log
