#!/usr/bin/env python3
"""null2 by trace from prefix sums (round 4) against the one-row-per-residue sums it replaced, on a slice of a bench
workload: deci-bit scores of every pair, and how the differing ones sit relative to a %6.1f rounding boundary.
usage: tools/ab_null2.py [NQ] [workload]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    wl = sys.argv[2] if len(sys.argv) > 2 else "aa_50k_x500"
    from witch_amd.ehmm import EHMM, pack_queries
    wd = tempfile.mkdtemp(prefix="witch_ab_")
    fam, se, names, seqs, k = bench.make_workload(wl, wd, nq, None)
    e = EHMM(se.paths, hmm_index=se.index, nseq=se.nseq, device=0)
    res, offs = pack_queries([s.astype(np.uint8) for s in seqs])
    out = {}
    for mode in ("runs", "gather"):
        if mode == "gather":
            os.environ["WH_RES_NULL2_GATHER"] = "1"
        else:
            os.environ.pop("WH_RES_NULL2_GATHER", None)
        deci, flags, det = e.score(res, offs, want_detail=True) if "want_detail" in e.score.__code__.co_varnames else (*e.score(res, offs), None)
        out[mode] = (deci.copy(), flags.copy())
    os.environ.pop("WH_RES_NULL2_GATHER", None)
    d0, f0 = out["runs"]
    d1, f1 = out["gather"]
    multi = (f0 & 2) != 0
    diff = d0 != d1
    print("pairs %d, multidomain %d, flags differ %d, deci-bits differ %d (all multidomain: %s), max |diff| %d"
          % (d0.size, int(multi.sum()), int(((f0 & 7) != (f1 & 7)).sum()), int(diff.sum()), bool((diff & ~multi).sum() == 0),
             int(np.abs(d0.astype(np.int64) - d1).max())))
    e.close()


if __name__ == "__main__":
    main()
