#!/usr/bin/env python3
"""Randomised check of the long-list pass on the GPU box: random DNA and protein families of
150-1 200 nodes, queries that hold 18-45 fragments of the family with random spacers (some fragments back to back:
multidomain candidates) - more regions than a scoring kernel's list holds - and one tandem repeat of 8-26 fragments (one
region of as many domains).  Every (query, model) pair is scored and
compared with the float64 oracle (whose own list holds 256 envelopes): no pair flagged WH_FLAG_TRUNC, reported mask and
multidomain flag, deci-bit scores under the tests' boundary rule (0.02 bit for this class), the pass's own count of
pairs against the regions the detail records report.
usage: tests/tools/fuzz_long_list.py [first_seed] [n_seeds]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from witch_amd import synth  # noqa: E402
from witch_amd._lib import WH_MAX_ENVELOPES  # noqa: E402
from witch_amd.ehmm import EHMM, pack_queries  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    from test_gpu_parity import _near_boundary_eps as near_boundary, LONG_EPS
    npairs = nlong = nbad = nbound = most = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(70000 + seed)
        alph = "amino" if seed % 2 == 0 else "dna"
        K = 20 if alph == "amino" else 4
        root = int(rng.integers(150, 1200 if alph == "dna" else 700))
        fam = synth.make_family(71000 + seed, root, 16, alph, 0.04, 1e-3)
        eh = synth.make_ehmm(fam, 2, tempfile.mkdtemp(prefix="fuzz_ll_"), witch_layout=False)
        flen = int(rng.integers(40, 80))
        _, frags = synth.make_queries(fam, 72000 + seed, 12, flen)
        seqs = []
        for _q in range(4):
            parts = []
            for c in range(int(rng.integers(18, 46))):
                parts.append(frags[int(rng.integers(0, len(frags)))].astype(np.uint8))
                if rng.random() > 0.15:
                    parts.append(rng.integers(0, K, size=int(rng.integers(15, 60))).astype(np.uint8))
            seqs.append(np.concatenate(parts))
        # tandem repeats: ONE region that the resolver splits into as many envelopes (vertex stacks in HBM beyond 2 048 segments)
        seqs.append(np.concatenate([frags[int(rng.integers(0, len(frags)))].astype(np.uint8) for _c in range(int(rng.integers(8, 27)))]))
        seqs.append(frags[0].astype(np.uint8))
        e = EHMM(eh.paths, hmm_index=eh.index, nseq=eh.nseq)
        res, offs = pack_queries(seqs)
        deci, flags, det = e.score(res, offs, want_detail=True)
        n_ll = e.last_long_list_pairs()
        e.close()
        nreg = np.array([d.nregions for d in det]).reshape(len(seqs), -1)
        ohm = [orc.OracleHMM(p) for p in eh.paths]
        od, of, ofwd, osc = orc.score_batch(ohm, res, offs)
        bad = int(((flags & 8) != 0).sum())
        bad += int(((flags & 3) != (of & 3)).sum())
        bad += int(n_ll != int((nreg > WH_MAX_ENVELOPES).sum()))
        for q in range(len(seqs)):
            for h in range(len(ohm)):
                if nreg[q, h] != ohm[h].score(seqs[q]).nregions:
                    bad += 1
                    print("   seed %d pair (%d, %d): %d regions, oracle %d" % (seed, q, h, nreg[q, h], ohm[h].score(seqs[q]).nregions))
        rep = (of & 1) == 1
        for q, h in zip(*np.nonzero(rep & (deci != od))):
            if abs(int(deci[q, h]) - int(od[q, h])) == 1 and near_boundary(float(osc[q, h]), LONG_EPS):
                nbound += 1
            else:
                bad += 1
                print("   seed %d pair (%d, %d): gpu %d oracle %d (score %.6f) flags %d/%d regions %d" % (seed, q, h, deci[q, h], od[q, h], osc[q, h], flags[q, h], of[q, h], nreg[q, h]))
        npairs += deci.size
        nlong += n_ll
        most = max(most, int(nreg.max()))
        nbad += bad
        print("seed %d %s root %d fragment %d: %d pairs, %d through the long-list pass (up to %d regions), mismatches so far %d"
              % (seed, alph, root, flen, deci.size, n_ll, int(nreg.max()), nbad), flush=True)
    print("pairs %d, through the long-list pass %d (most regions on a pair: %d), deci-bit values one unit off at a rounding boundary: %d, mismatches %d"
          % (npairs, nlong, most, nbound, nbad))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
