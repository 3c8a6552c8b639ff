#!/usr/bin/env python3
"""Per-stage breakdown of the LEVEL-1 path (bench.py reports the end-to-end number of the same chain as
`level1_e2e`, bench.level1_stage).  Throughput of the LEVEL-1 path (witch_amd.gcmm, what a WITCH maintainer calls): synthetic headline
family, <nq> queries x <nh> HMMs, every stage from text queries to the two merged FASTA files.
usage: tools/bench_level1.py [nq] [nh]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from witch_amd import gcmm, synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wd = tempfile.mkdtemp(prefix="witch_l1_")
fam, se, names, seqs, k = bench.make_workload("dna_100k_x200", wd, nq, nh)


class _Sub:
    def __init__(self, path, n):
        self.hmm_model_path, self.num_taxa = path, n


index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(se.index, se.paths, se.nseq)}
retained = {i: (h.map_cols[1:] - 1).tolist() for i, h in zip(se.index, se.hmms)}
nongaps = {i: h.nongaps.tolist() for i, h in zip(se.index, se.hmms)}
B = fam.msa.shape[1]
texts = [synth.to_text(s, "dna") for s in seqs]
bpath = os.path.join(wd, "backbone.fasta")
synth.write_msa_fasta(bpath, fam, 0, 64)
T = {}
t0 = time.time()
gcmm.warm_up(0)            # library + HIP context: process start-up, like the imports above; reported, not in the total
t_warm = time.time() - t0
t0 = time.time()
eng = gcmm.install(gcmm.QueryAlignmentEngine.run(index_to_hmm, list(zip(names, texts)), k, subset_to_retained_columns=retained,
                                                 subset_to_nongaps_per_column=nongaps, backbone_length=B))
T["engine.run (digitize + GPU + tables)"] = time.time() - t0
for kk, v in eng.timings.items():
    T["  gpu stage %s" % kk] = v
t0 = time.time()
ranked = gcmm.rankBitscores(index_to_hmm, {})
T["rankBitscores"] = time.time() - t0
t0 = time.time()
weights = gcmm.writeWeights(index_to_hmm, ranked)
T["writeWeights"] = time.time() - t0
t0 = time.time()
queries = [gcmm.alignSubQueriesNew(bpath, B, index_to_hmm, None, 0, t, s, weights[t], i)[0]
           for i, (t, s) in enumerate(zip(names, texts)) if t in weights]
T["alignSubQueriesNew x %d" % len(queries)] = time.time() - t0
t0 = time.time()
gcmm.mergeAlignmentsCollapsed(bpath, queries, {}, None, output_path=os.path.join(wd, "out.fasta"))
T["mergeAlignmentsCollapsed"] = time.time() - t0
t0 = time.time()
gcmm.mergeAlignmentsDevice(bpath, {}, output_path=os.path.join(wd, "out_dev.fasta"), taxa=[t for t in names if t in weights])
t_dev = time.time() - t0
same = open(os.path.join(wd, "out.fasta"), "rb").read() == open(os.path.join(wd, "out_dev.fasta"), "rb").read()
tot = sum(v for kk, v in T.items() if not kk.startswith("  "))
print("%-44s %8.2f s  (process start-up: library load + HIP context; not in the totals)" % ("warm_up", t_warm))
for kk, v in T.items():
    print("%-44s %8.2f s" % (kk, v))
print("%-44s %8.2f s  -> %.0f queries/s end to end (level 1, one GPU)" % ("total", tot, nq / tot))
host = T["alignSubQueriesNew x %d" % len(queries)] + T["mergeAlignmentsCollapsed"]
print("%-44s %8.2f s  (wh_merge from the consensus codes, instead of the %.2f s of per-query strings + host merge; same bytes: %s)"
      % ("mergeAlignmentsDevice", t_dev, host, same))
print("%-44s %8.2f s  -> %.0f queries/s end to end with the device merge" % ("total", tot - host + t_dev, nq / (tot - host + t_dev)))
