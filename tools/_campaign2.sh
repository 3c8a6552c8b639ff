cd $GRAFT_REPO_ROOT
set -o pipefail
run() { name=$1; shift; ( timeout -k 10 1100 python "$@" | tail -n 3 ) > gpurun_out/c2_$name.log 2>&1; echo "$name rc=$?: $(tail -n 1 gpurun_out/c2_$name.log | cut -c1-220)"; }
run res tests/tools/fuzz_resolver.py 1000 800
run align tests/tools/fuzz_align.py 2000 150
run alignlong tests/tools/fuzz_align.py 2500 30 300 1600 4
run big tests/tools/fuzz_align.py 3000 12 1600 3072
run wide12 tests/tools/fuzz_align.py 6700 10 3100 6100 2
run wide16 tests/tools/fuzz_align.py 6800 6 6200 8100 2
run wide24 tests/tools/fuzz_align.py 6900 4 8200 11000 1
run level1 tests/tools/fuzz_level1.py 500 20 6
