#!/usr/bin/env python3
"""Randomised check on the GPU box with models written by wh_hmmbuild from random alignments (tools/fuzz_hmmbuild.py's
generator: 1-40 sequences, 8-400 columns, gaps, fragments, degenerate residues - Dirichlet-prior transitions, tiny and
single-sequence models, entropy weighting), instead of the synthetic families' directly written model files: scores
(exact boundary rule), flags and aligned columns of the alignment's own rows, their fragments and unrelated sequences
against the float64 oracle.  usage: tests/tools/fuzz_built_models.py [first_seed] [n_seeds]"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from witch_amd.ehmm import EHMM, pack_queries  # noqa: E402
from witch_amd.gcmm.hmmbuild import hmmbuild_text  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from test_gpu_parity import _near_boundary_eps as near_boundary, BOUNDARY_EPS  # noqa: E402


def load_generator():
    src = open(os.path.join(ROOT, "tools", "fuzz_hmmbuild.py")).read().split("first = int(sys.argv[1])")[0]
    ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_hmmbuild.py")}
    exec(src, ns)
    return ns["random_alignment"], ns["ALPH"]


random_alignment, ALPH = load_generator()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
nbad = npairs = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    mol = ["dna", "rna", "amino"][seed % 3]
    wd = tempfile.mkdtemp(prefix="fuzz_built_")
    paths, nseq, all_rows = [], [], []
    for h in range(3):
        rows = random_alignment(rng, mol)
        text, M, _ = hmmbuild_text(rows, mol, "m%d" % h, stats=bool(h % 2))
        p = os.path.join(wd, "m%d.hmm" % h)
        open(p, "w").write(text)
        paths.append(p); nseq.append(len(rows)); all_rows += rows
    e = EHMM(paths, hmm_index=list(range(3)), nseq=nseq)
    can = ALPH[mol][0]
    texts = []
    for r in all_rows[:10]:
        t = "".join(c for c in r.upper() if c not in "-.")
        if t:
            texts.append(t)
            if len(t) > 6:
                lo = int(rng.integers(0, len(t) // 2))
                texts.append(t[lo:lo + max(3, len(t) // 2)])
    texts += ["".join(can[int(x)] for x in rng.integers(0, len(can), size=int(rng.integers(1, 120)))) for _ in range(3)]
    seqs = [e.digitize(t) for t in texts]
    res, offs = pack_queries(seqs)
    ohm = [orc.OracleHMM(p) for p in paths]
    deci, flags = e.score(res, offs)
    od, of, _, osc = orc.score_batch(ohm, res, offs)
    bad = int(np.sum((flags & 7) != (of & 7)))
    for qi, hj in np.argwhere((deci != od) & ((of & 1) == 1)):
        if abs(int(deci[qi, hj]) - int(od[qi, hj])) != 1 or not near_boundary(osc[qi, hj], BOUNDARY_EPS):
            bad += 1
    pq = [q for q in range(len(seqs)) for _ in range(3)]
    ph = [h for q in range(len(seqs)) for h in range(3)]
    for order in (np.arange(len(pq)), np.arange(len(pq))[::-1]):
        cols, co = e.align(res, offs, [pq[p] for p in order], [ph[p] for p in order])
        for t, p in enumerate(order):
            npairs += 1
            if not np.array_equal(cols[co[t]:co[t + 1]], ohm[ph[p]].align(seqs[pq[p]])):
                bad += 1
                print("ALIGN MISMATCH seed", seed, "query", pq[p], "model", ph[p], "L", len(seqs[pq[p]]), "M", int(e.M[ph[p]]), flush=True)
    e.close()
    nbad += bad
    print("seed", seed, mol, "M", [int(x) for x in e.M], "queries", len(seqs), "mismatches so far:", nbad, flush=True)
print("mismatches", nbad, "alignments checked", npairs)
sys.exit(1 if nbad else 0)
