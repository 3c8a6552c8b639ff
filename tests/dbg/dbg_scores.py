#!/usr/bin/env python3
"""Compare GPU scoring with the CPU oracle on a slice of a bench workload; print mismatches."""
import os, sys, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from oracle import oracle as orc

def main():
    wl, nq, nh = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from witch_amd.ehmm import EHMM, pack_queries
    wd = tempfile.mkdtemp(prefix="witch_dbg_")
    try:
        fam, se, names, seqs, k = bench.make_workload(wl, wd, nq, nh)
        e = EHMM(se.paths, hmm_index=se.index, nseq=se.nseq, device=0)
        res, offs = pack_queries([s.astype(np.uint8) for s in seqs])
        deci, flags, fwd, det = e.score(res, offs, want_fwd=True, want_detail=True)
        ohm = [orc.OracleHMM(p) for p in se.paths]
        od, of, ofwd, osc = orc.score_batch(ohm, res, offs)
        H = e.H
        rep = (of & 1) == 1
        bad = np.argwhere(((deci != od) & rep) | ((flags & 7) != (of & 7)))
        print("pairs", deci.size, "reported", int(rep.sum()), "multi", int(((of & 2) != 0).sum()), "mismatches", len(bad),
              "max |deci diff|", int(np.abs(deci.astype(np.int64) - od)[rep].max()), "dense redo", int(((flags & 16) != 0).sum()))
        for qi, hj in bad[:12]:
            d = det[qi * H + hj]
            r = ohm[hj].score(seqs[qi].astype(np.uint8))
            print("q", qi, "L", len(seqs[qi]), "h", hj, "M", e.M[hj], "gpu", deci[qi, hj], flags[qi, hj], "orc", od[qi, hj], of[qi, hj],
                  "nenv", d.nenv, r.nenv, "env", [(d.env_i[t], d.env_j[t], round(d.envsc[t], 3), round(d.domcorr[t], 4)) for t in range(d.nenv)],
                  "orc env", [(r.env_i[t], r.env_j[t], round(r.envsc[t], 3), round(r.domcorr[t], 4)) for t in range(min(r.nenv, 8))])
    finally:
        shutil.rmtree(wd, ignore_errors=True)

main()
