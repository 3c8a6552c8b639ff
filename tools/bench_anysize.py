#!/usr/bin/env python3
"""Throughput of the any-size kernels (wh_generic.hip): <nq> fragments against <nh> models of ~<root> nodes.
usage: tools/bench_anysize.py [root=6000] [nq=2000] [qlen=150] [nh=2]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from witch_amd import synth  # noqa: E402
from witch_amd.ehmm import EHMM, pack_queries  # noqa: E402

root = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
qlen = int(sys.argv[3]) if len(sys.argv) > 3 else 150
nh = int(sys.argv[4]) if len(sys.argv) > 4 else 2
wd = tempfile.mkdtemp(prefix="witch_any_")
fam = synth.make_family(1234 + root, root, 16, "dna", 0.03, 1e-4)
eh = synth.make_ehmm(fam, nh, wd, witch_layout=False)
names, seqs = synth.make_queries(fam, 77, nq, qlen)
res, offs = pack_queries([s.astype(np.uint8) for s in seqs])
e = EHMM(eh.paths, hmm_index=eh.index, nseq=eh.nseq)
e.set_timing(True)
cells = float(np.diff(offs).sum()) * float(e.M.sum())
for it in range(2):
    t0 = time.time()
    deci, flags = e.score(res, offs)[:2]
    t1 = time.time()
    ms0, _ = e.last_kernel_ms(0)
    ms4, _ = e.last_kernel_ms(4)
    print("iter %d: %d queries of %d x %d models (M %d..%d): front end %.1f ms, resolver queue %.1f ms, %.3g cells/s, wall %.2f s"
          % (it, nq, qlen, e.H, e.M.min(), e.M.max(), ms0, ms4, cells / ((ms0 + ms4) * 1e-3), t1 - t0), flush=True)
pq = list(range(nq))
ph = [0] * nq
for it in range(2):
    t0 = time.time()
    cols, co = e.align(res, offs, pq, ph)
    ms2, _ = e.last_kernel_ms(2)
    c2 = float(np.diff(offs).sum()) * float(e.M[0])
    print("align %d: %d pairs, %.1f ms (%.3g cells/s), wall %.2f s; aligned residues %.1f%%"
          % (it, nq, ms2, c2 / (ms2 * 1e-3), time.time() - t0, 100.0 * float((cols >= 0).mean())), flush=True)
e.close()
