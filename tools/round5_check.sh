#!/bin/bash
# Round-5 GPU check of the long-list pass (run through gpurun from the repo root): the tests that exercise the region
# lists of every scoring kernel and the resolver, then the round profile and a default bench line.
# usage: tools/round5_check.sh <tag>
set -e -o pipefail
tag=${1:-r05_v6}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${WH_CHECK_K:-more_regions or more_than_eight or score_against_oracle or beyond_3072 or resolver_on_every or resolver_queue or long_protein or end_to_end or multihit_queries_on_a_long or several_waves or level0}" > gpurun_out/${tag}_tests.log 2>&1 || { tail -40 gpurun_out/${tag}_tests.log; exit 1; }
tail -3 gpurun_out/${tag}_tests.log
tools/profile_round.sh $tag
# stamp the traffic figures with this build (bench.py refuses a stamp of another build), keep a copy where gpurun merges it back
python3 tools/stamp_profile.py gpurun_out/$tag dna_100k_x200 2709930000000.0 > gpurun_out/${tag}_stamp.log
cp profiles/traffic.json gpurun_out/${tag}_stamped_traffic.json
timeout -k 10 600 python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
tail -c 600 gpurun_out/${tag}_bench_default.json
