#!/usr/bin/env python3
"""Design aid for the banded envelope sweeps: on a sample of pairs of a bench workload, how many model
nodes hold posterior mass >= 2^e on some envelope row (lane-block granularity), as known from the
multihit sweeps and as needed by the unihit envelope sweeps.  Uses the CPU oracle (tool, not product).
usage: tests/tools/band_stats.py [workload] [n_pairs] [log2 eps]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "dna_100k_x200"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
l2e = float(sys.argv[3]) if len(sys.argv) > 3 else -50.0
wd = tempfile.mkdtemp()
fam, se, names, seqs, k = bench.make_workload(wl, wd, 512, 40)
L = orc.lib()
L.orc_band_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
rng = np.random.default_rng(0)
hm = [orc.OracleHMM(p) for p in se.paths]
wm, wu, cover = [], [], 0
for t in range(n):
    q, h = int(rng.integers(len(seqs))), int(rng.integers(len(hm)))
    s = np.ascontiguousarray(seqs[q], dtype=np.uint8)
    out = np.zeros(6, dtype=np.int32)
    if not L.orc_band_stats(hm[h]._h, s.ctypes.data, len(s), 16, l2e, out.ctypes.data):
        continue
    wm.append(out[1] - out[0] + 1)
    wu.append(out[3] - out[2] + 1)
    cover += out[0] <= out[2] and out[1] >= out[3]
wm, wu = np.array(wm), np.array(wu)
print("%s, eps 2^%g, %d pairs: band from multihit sweeps: mean %.0f, 90%% %d, 99%% %d, max %d nodes; needed by the envelope: mean %.0f, max %d; multihit band covers the need in %d of %d"
      % (wl, l2e, len(wm), wm.mean(), np.percentile(wm, 90), np.percentile(wm, 99), wm.max(), wu.mean(), wu.max(), cover, len(wm)))
print("band <= 256: %.1f%%   <= 512: %.1f%%" % (100 * (wm <= 256).mean(), 100 * (wm <= 512).mean()))
