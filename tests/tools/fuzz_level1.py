#!/usr/bin/env python3
"""Randomised check of the level-1 chain's consensus step on the GPU box: the cases of tests/tools/fuzz_align.py (families
of 300-1 600 nodes, flanked / degenerate / unrelated queries) through witch_amd.gcmm - engine run, ranking, weights,
alignSubQueriesNew - and every query's merged row against the numpy restatement of the reference's consensus DP
(oracle/consensus.py) on the same top-k and weights.  usage: tests/tools/fuzz_level1.py [first_seed] [n_seeds] [k]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
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
    fam = eh.family
    # text form, as the reference hands queries over: Easel's symbols, degenerate codes included
    sym = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~" if alph == "amino" else "ACGT-RYMKSWHBVDN*~"
    K = 20 if alph == "amino" else 4
    texts = ["".join(sym[int(c) if int(c) != K else K + 1] for c in s) for s in seqs]     # (no gap symbol inside a query)
    names = ["q%02d" % i for i in range(len(seqs))]
    index_to_hmm = {i: _Sub(p, m) for i, p, m in zip(eh.index, eh.paths, eh.nseq)}
    retained = {i: (h.map_cols[1:] - 1).tolist() for i, h in zip(eh.index, eh.hmms)}
    nongaps = {i: h.nongaps.tolist() for i, h in zip(eh.index, eh.hmms)}
    B = int(fam.msa.shape[1])
    gcmm.install(gcmm.QueryAlignmentEngine.run(index_to_hmm, list(zip(names, texts)), k, subset_to_retained_columns=retained,
                                               subset_to_nongaps_per_column=nongaps, backbone_length=B))
    weights = gcmm.writeWeights(index_to_hmm, gcmm.rankBitscores(index_to_hmm, {}))
    queries = []
    for q, (qn, qs) in enumerate(zip(names, texts)):
        if qn not in weights:
            continue
        nq_tot += 1
        query, _, _ = gcmm.alignSubQueriesNew("bb", B, index_to_hmm, None, 120, qn, qs, weights[qn], q)
        queries.append(query)
        _, wmap, cols = gcmm.getBackbones(index_to_hmm, qn, q, qs, "p", weights[qn], ".", ".", use_gcm=False)
        codes, _ = ocons.consensus_trace(len(qs), list(cols.items()), wmap, retained, nongaps, B)
        if query[qn] != ocons.trace_to_string(qs, codes, B):
            nbad += 1
            print("MISMATCH seed", seed, "query", qn, "L", len(qs), flush=True)
    # final merge: the device merge (wh_merge) and the host merger write the same two files
    wd = tempfile.mkdtemp(prefix="fuzz_l1_out_")
    bpath = os.path.join(wd, "backbone.fasta")
    synth.write_msa_fasta(bpath, fam, 0, 8)
    host = gcmm.mergeAlignmentsCollapsed(bpath, queries, {}, None, output_path=os.path.join(wd, "host.fasta"))
    dev = gcmm.mergeAlignmentsDevice(bpath, {}, output_path=os.path.join(wd, "dev.fasta"), taxa=[t for t in names if t in weights])
    for what, a_, b_ in zip(("full", "masked"), host, dev):
        if open(a_, "rb").read() != open(b_, "rb").read():
            nbad += 1
            print("MERGE MISMATCH seed", seed, what, flush=True)
    print("seed", seed, alph, "root", root, "backbone", B, "mismatches so far:", nbad, "of", nq_tot, flush=True)
print("mismatches", nbad, "of", nq_tot)
sys.exit(1 if nbad else 0)
