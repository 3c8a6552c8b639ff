#!/usr/bin/env python3
"""Randomised check of the weights / top-k / adaptive-cut kernel on the GPU box: random deci-bit matrices (few distinct
values: many exact ties), random reported masks, model sizes with shared odd parts and powers of two (the integer tie
key of wh_topk.hip), H from 1 to 200 and k from 1 to 40, against the oracle's restatement of the reference's
rankBitscores / calculateWeights / 0.999 cut (the comparison of tests/test_gpu_parity.py::test_topk_*).
usage: tests/tools/fuzz_topk.py [first_seed] [n_seeds]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from witch_amd import synth  # noqa: E402
from witch_amd.ehmm import EHMM  # noqa: E402
from oracle import oracle as orc  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fam = synth.make_family(77, 40, 8, "dna", 0.05, 2e-3)
eh = synth.make_ehmm(fam, 3, tempfile.mkdtemp(prefix="fuzz_topk_"), witch_layout=False)
nbad = nrows = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    H = int(rng.choice([1, 2, 3, 5, 17, 64, 200]))
    k = int(rng.choice([1, 2, 3, 4, 10, 16, 20, 40]))
    nq = 200
    index = rng.permutation(1000)[:H].astype(np.int32)
    kind = seed % 3
    if kind == 0:
        nseq = rng.integers(1, 5000, size=H)
    elif kind == 1:
        nseq = rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24, 48, 96, 100, 200, 400], size=H)      # shared odd parts, powers of two
    else:
        nseq = np.full(H, int(rng.integers(1, 300)))
    e = EHMM([eh.paths[i % 3] for i in range(H)], hmm_index=index, nseq=nseq.astype(np.int32))
    spread = int(rng.choice([3, 30, 400, 3000]))
    deci = (rng.integers(0, spread, size=(nq, H)) * int(rng.choice([1, 1, 10])) - int(rng.integers(0, 200))).astype(np.int32)
    flags = (rng.random((nq, H)) < rng.choice([0.05, 0.5, 0.9, 1.0])).astype(np.uint8)
    idx, w, nk, nu = e.topk(deci, flags, k)
    size_of = dict(zip(index.tolist(), nseq.tolist()))
    for qi in range(nq):
        nrows += 1
        ranked = orc.rank_bitscores(index.tolist(), deci[qi], flags[qi] & 1)
        ok = True
        if not ranked:
            ok = nk[qi] == 0 and nu[qi] == 0
        else:
            idxs = [r[0] for r in ranked]
            ref = orc.calculate_weights(idxs, [r[1] for r in ranked], [size_of[i] for i in idxs], k)
            got_w = w[qi, :nk[qi]]
            ref_w = np.array([x[1] for x in ref])
            ok = nk[qi] == len(ref) and np.allclose(got_w, ref_w, rtol=1e-12, atol=0)
            if ok:
                got_i = idx[qi, :nk[qi]].tolist()
                for a_, b_, wa, wb in zip(got_i, [x[0] for x in ref], got_w, ref_w):
                    if a_ != b_ and not abs(wa - wb) <= 1e-12 * max(wa, wb):
                        ok = False
                ok = ok and nu[qi] == orc.adaptive_cut(list(zip(got_i, got_w)))
        if not ok:
            nbad += 1
            if nbad <= 10:
                print("MISMATCH seed", seed, "H", H, "k", k, "query", qi, flush=True)
    e.close()
    print("seed", seed, "H", H, "k", k, "sizes", kind, "mismatches so far:", nbad, "of", nrows, flush=True)
print("mismatches", nbad, "of", nrows)
sys.exit(1 if nbad else 0)
