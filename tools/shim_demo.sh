#!/bin/bash
# Level-0 demo on the GPU box: auto-started server, WITCH's two command lines, call rates.
set -e
cd "$GRAFT_REPO_ROOT"
make -C witch_amd/shim >/dev/null
export WITCH_HIP_SOCKET=/tmp/witch_demo.sock
W=$(mktemp -d)
python3 - "$W" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench
from witch_amd import synth
wd = sys.argv[1]
fam, se, names, seqs, k = bench.make_workload("dna_1k_x10", os.path.join(wd, "hmm"), 2000, 10)
synth.write_fasta(os.path.join(wd, "chunk.fa"), names, seqs, fam.alphabet)
for i in range(256):
    synth.write_fasta(os.path.join(wd, "q%d.fa" % i), [names[i]], [seqs[i]], fam.alphabet)
open(os.path.join(wd, "hmms.txt"), "w").write("\n".join(se.paths))
PY
H0=$(head -1 $W/hmms.txt)
t0=$(date +%s.%N)
witch_amd/shim/bin/hmmsearch --cpu 1 --noali -E 99999999 -o $W/r0.txt --max $H0 $W/chunk.fa
t1=$(date +%s.%N)
echo "first hmmsearch call (starts the server): $(python3 -c "print(round($t1 - $t0, 3))") s; rows: $(grep -c '^ *[0-9].* q0' $W/r0.txt)"
t0=$(date +%s.%N)
for h in $(cat $W/hmms.txt); do witch_amd/shim/bin/hmmsearch --cpu 1 --noali -E 99999999 -o $W/r.$(basename $(dirname $h)).txt --max $h $W/chunk.fa; done
t1=$(date +%s.%N)
echo "10 hmmsearch calls x 2000 queries: $(python3 -c "print(round($t1 - $t0, 3))") s"
t0=$(date +%s.%N)
for i in $(seq 0 63); do witch_amd/shim/bin/hmmalign -o $W/a$i.sto $H0 $W/q$i.fa; done
t1=$(date +%s.%N)
echo "64 sequential hmmalign calls: $(python3 -c "print(round($t1 - $t0, 3))") s"
t0=$(date +%s.%N)
for i in $(seq 0 255); do witch_amd/shim/bin/hmmalign -o $W/b$i.sto $H0 $W/q$i.fa & done; wait
t1=$(date +%s.%N)
echo "256 concurrent hmmalign calls: $(python3 -c "print(round($t1 - $t0, 3))") s; outputs: $(ls $W/b*.sto | wc -l)"
python3 -c "
import sys; sys.path.insert(0,'.')
from witch_amd.shim.server import request
print(request('$WITCH_HIP_SOCKET','ping',[])); print(request('$WITCH_HIP_SOCKET','shutdown',[]))"
tail -3 $WITCH_HIP_SOCKET.log 2>/dev/null || true
