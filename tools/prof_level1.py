#!/usr/bin/env python3
"""Where the level-1 engine's wall time goes (development aid): wh_init / wh_ehmm_load / each stage of
QueryAlignmentEngine.run with WH_TRACE timings.  usage: tools/prof_level1.py [nq] [nh]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WH_TRACE", "1")
import bench  # noqa: E402
from witch_amd import synth  # noqa: E402
from witch_amd._lib import lib  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wd = tempfile.mkdtemp(prefix="witch_l1p_")
fam, se, names, seqs, k = bench.make_workload("dna_100k_x200", wd, nq, nh)
texts = [synth.to_text(s, "dna") for s in seqs]
t0 = time.time()
L = lib()
t1 = time.time()
L.wh_init(0)
t2 = time.time()
print("dlopen %.3f s, wh_init %.3f s" % (t1 - t0, t2 - t1), flush=True)
from witch_amd.ehmm import EHMM  # noqa: E402
t0 = time.time()
e = EHMM(se.paths, hmm_index=se.index, nseq=se.nseq)
print("EHMM() %.3f s" % (time.time() - t0), flush=True)
t0 = time.time()
res, offs = e.digitize_many(texts)
print("digitize %.3f s" % (time.time() - t0), flush=True)
for it in range(2):
    t0 = time.time()
    deci, flags = e.score(res, offs)
    t1 = time.time()
    idx, w, nk, nu = e.topk(deci, flags, k)
    t2 = time.time()
    import numpy as np
    keep = np.arange(k)[None, :] < nu[:, None]
    pq = np.nonzero(keep)[0].astype(np.int64)
    ph = np.array([e.pos_of_index[int(x)] for x in idx[keep]], dtype=np.int32)
    t3 = time.time()
    cols, co = e.align(res, offs, pq, ph)
    t4 = time.time()
    print("iter %d: score %.3f s, topk %.3f s, pairs %.3f s, align %.3f s (%d pairs)" % (it, t1 - t0, t2 - t1, t3 - t2, t4 - t3, len(pq)), flush=True)
e.close()
