#!/usr/bin/env python3
"""Randomised check of the multidomain resolver on the GPU box: random DNA and protein families of 150-1 500 nodes,
queries that hold two or three copies of the family (with random flanks and a few degenerate codes), every (query,
model) pair scored and compared with the float64 oracle, which runs the same 200 seeded tracebacks: reported mask,
multidomain flag, Forward log-odds, deci-bit scores under the tests' boundary rule (0.02 bit for this class).
usage: tests/tools/fuzz_resolver.py [first_seed] [n_seeds]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from witch_amd import synth  # noqa: E402
from witch_amd.ehmm import EHMM, pack_queries  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    from test_gpu_parity import _near_boundary_eps as near_boundary, LONG_EPS
    npairs = nmulti = nbad = nbound = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(90000 + seed)
        alph = "amino" if seed % 2 == 0 else "dna"
        root = int(rng.integers(150, 1500 if alph == "dna" else 900))
        fam = synth.make_family(91000 + seed, root, 16, alph, 0.04, 1e-3)
        eh = synth.make_ehmm(fam, 2, tempfile.mkdtemp(prefix="fuzz_res_"), witch_layout=False)
        _, seqs = synth.make_queries(fam, 92000 + seed, 6, (2 * root, 3 * root), flank_frac=0.3)
        _, one = synth.make_queries(fam, 93000 + seed, 2, (root // 2, root), flank_frac=0.2)
        seqs = [s_.astype(np.uint8) for s_ in list(seqs) + list(one)]
        K = 20 if alph == "amino" else 4
        Kp = 29 if alph == "amino" else 18
        for t in (1, 4):
            pos = rng.integers(0, len(seqs[t]), size=max(1, len(seqs[t]) // 40))
            seqs[t] = seqs[t].copy()
            seqs[t][pos] = rng.integers(K, Kp - 3, size=len(pos)).astype(np.uint8)
        e = EHMM(eh.paths, hmm_index=eh.index, nseq=eh.nseq)
        res, offs = pack_queries(seqs)
        deci, flags, fwd = e.score(res, offs, want_fwd=True)
        e.close()
        ohm = [orc.OracleHMM(p) for p in eh.paths]
        od, of, ofwd, osc = orc.score_batch(ohm, res, offs)
        bad = 0
        if not np.array_equal(flags & 3, of & 3):
            bad += int(((flags & 3) != (of & 3)).sum())
        fin = np.isfinite(ofwd)
        if fin.any() and np.max(np.abs(fwd[fin] - ofwd[fin])) > 3e-4 * max(1.0, root / 500.0):
            bad += 1
        rep = (of & 1) == 1
        diff = rep & (deci != od)
        for q, h in zip(*np.nonzero(diff)):
            if abs(int(deci[q, h]) - int(od[q, h])) == 1 and near_boundary(float(osc[q, h]), LONG_EPS):
                nbound += 1
            else:
                bad += 1
                print("   seed %d pair (%d, %d): gpu %d oracle %d (score %.6f) flags %d/%d" % (seed, q, h, deci[q, h], od[q, h], osc[q, h], flags[q, h], of[q, h]))
        npairs += deci.size
        nmulti += int(((of & 2) != 0).sum())
        nbad += bad
        print("seed %d %s root %d: %d pairs, %d multidomain, mismatches so far %d" % (seed, alph, root, deci.size, int(((of & 2) != 0).sum()), nbad), flush=True)
    print("pairs %d, multidomain %d, deci-bit values one unit off at a rounding boundary: %d, mismatches %d" % (npairs, nmulti, nbound, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
