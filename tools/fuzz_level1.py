#!/usr/bin/env python3
"""Randomised check of the level-1 chain's consensus step on the GPU box: the cases of tools/fuzz_align.py (families
of 300-1 600 nodes, flanked / degenerate / unrelated queries) through witch_amd.gcmm - engine run, ranking, weights,
alignSubQueriesNew - and every query's merged row against the numpy restatement of the reference's consensus DP
(oracle/consensus.py) on the same top-k and weights.  usage: tools/fuzz_level1.py [first_seed] [n_seeds] [k]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_align  # noqa: E402
from oracle import consensus as ocons  # noqa: E402
from witch_amd import gcmm, synth  # noqa: E402


class _Sub:
    def __init__(self, path, n):
        self.hmm_model_path, self.num_taxa = path, n


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nbad = nq_tot = 0
for seed in range(first, first + n):
    alph, root, eh, seqs = fuzz_align.make_case(seed, tempfile.mkdtemp(prefix="fuzz_l1_"))
    # text form, as the reference hands queries over: Easel's symbols, degenerate codes included
    sym = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~" if alph == "amino" else "ACGT-RYMKSWHBVDN*~"
    texts = ["".join(sym[int(c)] for c in s) for s in seqs]
    names = ["q%02d" % i for i in range(len(seqs))]
    index_to_hmm = {i: _Sub(p, m) for i, p, m in zip(eh.index, eh.paths, eh.nseq)}
    retained = {i: (h.map_cols[1:] - 1).tolist() for i, h in zip(eh.index, eh.hmms)}
    nongaps = {i: h.nongaps.tolist() for i, h in zip(eh.index, eh.hmms)}
    B = int(max(max(v) for v in retained.values())) + 1
    gcmm.install(gcmm.QueryAlignmentEngine.run(index_to_hmm, list(zip(names, texts)), k, subset_to_retained_columns=retained,
                                               subset_to_nongaps_per_column=nongaps, backbone_length=B))
    weights = gcmm.writeWeights(index_to_hmm, gcmm.rankBitscores(index_to_hmm, {}))
    for q, (qn, qs) in enumerate(zip(names, texts)):
        if qn not in weights:
            continue
        nq_tot += 1
        query, _, _ = gcmm.alignSubQueriesNew("bb", B, index_to_hmm, None, 120, qn, qs, weights[qn], q)
        _, wmap, cols = gcmm.getBackbones(index_to_hmm, qn, q, qs, "p", weights[qn], ".", ".", use_gcm=False)
        codes, _ = ocons.consensus_trace(len(qs), list(cols.items()), wmap, retained, nongaps, B)
        if query[qn] != ocons.trace_to_string(qs, codes, B):
            nbad += 1
            print("MISMATCH seed", seed, "query", qn, "L", len(qs), flush=True)
    print("seed", seed, alph, "root", root, "backbone", B, "mismatches so far:", nbad, "of", nq_tot, flush=True)
print("mismatches", nbad, "of", nq_tot)
sys.exit(1 if nbad else 0)
