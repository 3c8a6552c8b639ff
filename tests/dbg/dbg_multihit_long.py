"""Multi-hit long DNA queries against 1900-node models: alignment vs the oracle (log-space pass on the pass-synchronous kernel)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from witch_amd import synth
from witch_amd.ehmm import EHMM, pack_queries
from oracle import oracle as orc
tmp = tempfile.mkdtemp()
root = int(sys.argv[1]) if len(sys.argv) > 1 else 1900
fam = synth.make_family(4242 + root, root, 16, "dna", 0.03, 1e-4)
eh = synth.make_ehmm(fam, 2, tmp, witch_layout=False)
names, seqs = synth.make_queries(fam, 17, 6, (root + 200, 2 * root + 400), flank_frac=0.3)
seqs = [s.astype(np.uint8) for s in seqs]
e = EHMM(eh.paths, hmm_index=eh.index, nseq=eh.nseq)
print("M", e.M.tolist(), "L", [len(s) for s in seqs])
res, offs = pack_queries(seqs)
ohm = [orc.OracleHMM(p) for p in eh.paths]
pq = [q for q in range(len(seqs)) for _ in range(e.H)]
ph = [h for q in range(len(seqs)) for h in range(e.H)]
cols, co = e.align(res, offs, pq, ph)
bad = 0
for p in range(len(pq)):
    want = ohm[ph[p]].align(seqs[pq[p]])
    got = cols[co[p]:co[p + 1]]
    if not np.array_equal(got, want):
        bad += 1
        g = np.where(got >= 0)[0]; w = np.where(want >= 0)[0]
        print("pair", pq[p], ph[p], "gpu rows", (g.min(), g.max(), len(g)) if len(g) else None, "oracle", (w.min(), w.max(), len(w)) if len(w) else None)
print("pairs", len(pq), "mismatching", bad)
