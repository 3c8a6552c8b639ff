import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from witch_amd import synth
from witch_amd.ehmm import EHMM, pack_queries
from oracle import oracle as orc
tmp = tempfile.mkdtemp()
fam = synth.make_family(20251206, 600, 64, "amino", 0.03, 1e-4)
eh = synth.make_ehmm(fam, 6, tmp, witch_layout=False)
names, seqs = synth.make_queries(fam, 20251207, 16, (500, 2000), flank_frac=0.3)
seqs = [s.astype(np.uint8) for s in seqs]
e = EHMM(eh.paths, hmm_index=eh.index, nseq=eh.nseq)
res, offs = pack_queries(seqs)
deci, flags, fwd, det = e.score(res, offs, want_fwd=True, want_detail=True)
ohm = [orc.OracleHMM(p) for p in eh.paths]
for (q, h) in [(0, 0), (0, 1), (1, 0)]:
    d = det[q * e.H + h]
    print("q", q, "L", len(seqs[q]), "h", h, "M", e.M[h], "env", [(d.env_i[t], d.env_j[t], round(d.envsc[t] / 0.693, 1)) for t in range(d.nenv)])
    cols, co = e.align(res, offs, [q], [h])
    want = ohm[h].align(seqs[q])
    g = np.where(cols >= 0)[0]; w = np.where(want >= 0)[0]
    print("   gpu aligned rows", (g.min(), g.max(), len(g)) if len(g) else None, "oracle", (w.min(), w.max(), len(w)) if len(w) else None, "equal", np.array_equal(cols, want))
