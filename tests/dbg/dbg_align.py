#!/usr/bin/env python3
"""Compare GPU MEA alignment columns with the CPU oracle on a slice of a bench workload."""
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
        seqs = [s.astype(np.uint8) for s in seqs]
        e = EHMM(se.paths, hmm_index=se.index, nseq=se.nseq, device=0)
        res, offs = pack_queries(seqs)
        ohm = [orc.OracleHMM(p) for p in se.paths]
        pq = [q for q in range(len(seqs)) for _ in range(e.H)]
        ph = [h for q in range(len(seqs)) for h in range(e.H)]
        cols, co = e.align(res, offs, pq, ph)
        bad = 0
        for p in range(len(pq)):
            want = ohm[ph[p]].align(seqs[pq[p]])
            got = cols[co[p]:co[p + 1]]
            if not np.array_equal(got, want):
                bad += 1
                nd = int((got != want).sum())
                if bad <= 8:
                    w = np.argwhere(got != want)[:6, 0]
                    print("pair q%d L%d h%d M%d: %d of %d columns differ; first at %s got %s want %s" % (pq[p], len(seqs[pq[p]]), ph[p], e.M[ph[p]], nd, len(want), w.tolist(), got[w].tolist(), want[w].tolist()))
        print("pairs", len(pq), "mismatching", bad)
    finally:
        shutil.rmtree(wd, ignore_errors=True)

main()
