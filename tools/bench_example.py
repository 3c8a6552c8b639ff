#!/usr/bin/env python3
"""Stage times of the hot path on the reference's example data (tests/golden/example_e2e: 500 fragments
x 15 HMMs of 1278-2574 nodes; 28 % of the pairs have a multidomain region), replicated <rep> times to
give the GPU a batch worth timing.  usage: tools/bench_example.py [rep]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.conftest import load_case  # noqa: E402
from witch_amd.ehmm import EHMM, pack_queries  # noqa: E402

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
case = load_case("example_e2e")
e = EHMM(case.hmm_paths, hmm_index=case.hmm_index, nseq=case.nseq)
seqs = [e.digitize(s) for s in case.qseqs] * rep
res, offs = pack_queries(seqs)
e.set_timing(True)
for it in range(2):
    t0 = time.time()
    deci, flags = e.score(res, offs)
    t1 = time.time()
    ms0, _ = e.last_kernel_ms(0)
    ms4, _ = e.last_kernel_ms(4)
    cells = float(np.diff(offs).sum()) * float(e.M.sum())
    nm = int(((flags & 2) != 0).sum())
    print("iter %d: %d queries x %d HMMs (M %d..%d): scoring kernels %.1f ms (%.3g cells/s), resolver %.1f ms for %d multidomain pairs (%.0f pairs/s), wall %.2f s"
          % (it, len(seqs), e.H, e.M.min(), e.M.max(), ms0, cells / (ms0 * 1e-3), ms4, nm, nm / (ms4 * 1e-3) if ms4 > 0 else 0, t1 - t0), flush=True)
idx, w, nk, nu = e.topk(deci, flags, case.k)
pq = [q for q in range(len(seqs)) for _ in range(int(nu[q]))]
ph = [e.pos_of_index[int(idx[q, j])] for q in range(len(seqs)) for j in range(int(nu[q]))]
for it in range(2):      # (the first call allocates the workspace)
    t0 = time.time()
    cols, co = e.align(res, offs, pq, ph)
    ms2, _ = e.last_kernel_ms(2)
    print("align %d: %d pairs, %.1f ms kernel, wall %.2f s" % (it, len(pq), ms2, time.time() - t0))
e.close()
