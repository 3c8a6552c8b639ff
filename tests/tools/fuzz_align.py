#!/usr/bin/env python3
"""Randomised alignment check on the GPU box: families of 300-1 600 nodes (DNA and protein), queries of 5-500
residues - family windows with and without random flanks, unrelated sequences, degenerate residue codes - every
(query, model) pair aligned in one batch and compared, column by column, with the float64 oracle.  Found the
round-3 bug of the full-width passes (Forward rows that had underflowed to zero everywhere were not written, the
Backward sweep then read the previous pair's cells: seed 103).  usage: tests/tools/fuzz_align.py [first_seed] [n_seeds] [min_nodes] [max_nodes] [length_scale]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from witch_amd import synth  # noqa: E402
from witch_amd.ehmm import EHMM, pack_queries  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def make_case(seed, workdir, lo=300, hi=1600, lscale=1):
    rng = np.random.default_rng(seed)
    alph = "amino" if seed % 3 == 0 else "dna"
    root = int(rng.integers(lo, hi))
    fam = synth.make_family(2000 + seed, root, 8, alph, 0.05, 2e-3)
    eh = synth.make_ehmm(fam, 3, workdir, witch_layout=False)
    K = 20 if alph == "amino" else 4
    Kp = 29 if alph == "amino" else 18
    bg = synth.background(alph)
    seqs = []
    for t in range(12):
        L = int(rng.choice([5, 25, 60, 150, 233, 333, 401, 500])) * lscale
        if t % 4 == 0:
            s_ = rng.choice(K, size=L, p=bg).astype(np.uint8)
        else:
            _, w = synth.make_queries(fam, seed * 100 + t, 1, min(L, root - 1), 0.05 + 0.1 * (t % 3), flank_frac=0.3 if t % 2 else 0.0)
            s_ = w[0].astype(np.uint8)
        if t % 5 == 1 and len(s_) > 4:
            pos = rng.integers(0, len(s_), size=max(1, len(s_) // 10))
            s_ = s_.copy()
            s_[pos] = rng.integers(K, Kp - 3, size=len(pos)).astype(np.uint8)
        seqs.append(s_)
    return alph, root, eh, seqs


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1600
    lscale = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    nbad = npairs = nboundary = 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _near_boundary_eps as near_boundary, BOUNDARY_EPS as boundary_eps   # the tests' rule
    paths = {}
    for seed in range(first, first + n):
        alph, root, eh, seqs = make_case(seed, tempfile.mkdtemp(prefix="fuzz_align_"), lo, hi, lscale)
        e = EHMM(eh.paths, hmm_index=eh.index, nseq=eh.nseq)
        res, offs = pack_queries(seqs)
        pq = [q for q in range(len(seqs)) for _ in range(e.H)]
        ph = [h for q in range(len(seqs)) for h in range(e.H)]
        ohm = [orc.OracleHMM(p) for p in eh.paths]
        want = [ohm[ph[p]].align(seqs[pq[p]]) for p in range(len(pq))]
        # scores too: reported mask and multidomain flag equal, deci-bits within 1 of the oracle's (a difference of 1
        # is the rounding boundary class of SURVEY 8.0; the parity tests apply the exact boundary rule)
        deci, flags = e.score(res, offs)
        od, of, _, osc = orc.score_batch(ohm, res, offs)
        rep_ = (of & 1) == 1
        sbad = int(np.sum((flags & 7) != (of & 7)))
        # the parity tests' rule: equal, or one unit apart with the oracle's float score at a "%6.1f" boundary
        d1 = np.argwhere((deci != od) & rep_)
        for qi_, hj_ in d1:
            if abs(int(deci[qi_, hj_]) - int(od[qi_, hj_])) != 1 or not near_boundary(osc[qi_, hj_], boundary_eps):
                sbad += 1
        nboundary += len(d1)
        if sbad:
            nbad += sbad
            print("SCORE MISMATCH seed", seed, "pairs", sbad, flush=True)
        # three calls in different pair orders: every call finds OTHER pairs' rows in the slabs it gets
        prng = np.random.default_rng(seed + 7)
        for rep in range(3):
            order = np.arange(len(pq)) if rep == 0 else np.arange(len(pq))[::-1] if rep == 1 else prng.permutation(len(pq))
            cols, co = e.align(res, offs, [pq[p] for p in order], [ph[p] for p in order])
            for k_, v in e.last_align_paths().items():
                paths[k_] = paths.get(k_, 0) + v
            paths["log_space_redo"] = paths.get("log_space_redo", 0) + e.last_align_status()[0]
            for t, p in enumerate(order):
                npairs += 1
                if not np.array_equal(cols[co[t]:co[t + 1]], want[p]):
                    nbad += 1
                    print("MISMATCH seed", seed, "call", rep, "query", pq[p], "model", ph[p], "L", len(seqs[pq[p]]), "M", int(e.M[ph[p]]), flush=True)
        # scores of the queries in reversed order: the same numbers
        rres, roffs = pack_queries(seqs[::-1])
        rdeci, rflags = e.score(rres, roffs)
        if not (np.array_equal(rdeci[::-1], deci) and np.array_equal(rflags[::-1], flags)):
            nbad += 1
            print("SCORE ORDER MISMATCH seed", seed, int(np.sum(rdeci[::-1] != deci)), "deci-bit values differ", flush=True)
        e.close()
        print("seed", seed, alph, "root", root, "mismatches so far:", nbad, "of", npairs, flush=True)
    print("pairs by path", paths, "deci-bit values one unit off at a rounding boundary:", nboundary, "mismatches", nbad, "of", npairs)
    sys.exit(1 if nbad else 0)
