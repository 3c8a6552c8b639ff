"""Any-size kernels (wh_generic.hip) on boundary model sizes, 3 075 ... 16 309 nodes: scores, flags and aligned columns
against the oracle.  Run through gpurun: python tests/dbg/any_shapes.py"""
import sys, os, tempfile
import numpy as np
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from witch_amd import synth
from witch_amd.ehmm import EHMM, pack_queries
from oracle import oracle as orc
for root in (3075, 4090, 4100, 8200, 16300):
    wd = tempfile.mkdtemp()
    fam = synth.make_family(500 + root, root, 8, "dna", 0.03, 1e-4)
    eh = synth.make_ehmm(fam, 2, wd, witch_layout=False)
    _, seqs = synth.make_queries(fam, 3, 6, (120, 400))
    seqs = [s.astype(np.uint8) for s in seqs]
    e = EHMM(eh.paths, hmm_index=eh.index, nseq=eh.nseq)
    res, offs = pack_queries(seqs)
    deci, flags, fwd = e.score(res, offs, want_fwd=True)
    ohm = [orc.OracleHMM(p) for p in eh.paths]
    od, of, ofwd, osc = orc.score_batch(ohm, res, offs)
    nd = int((deci != od).sum()); nf = int(((flags & 3) != (of & 3)).sum())
    pq = [q for q in range(len(seqs)) for _ in range(e.H)]; ph = [h for q in range(len(seqs)) for h in range(e.H)]
    cols, co = e.align(res, offs, pq, ph)
    na = sum(0 if np.array_equal(cols[co[p]:co[p+1]], ohm[ph[p]].align(seqs[pq[p]])) else 1 for p in range(len(pq)))
    print("root %d: M %s cells per lane %s  decibit diffs %d  flag diffs %d  max|fwd diff| %.2g  alignments differing %d of %d" % (root, [int(m) for m in e.M], [((int(m) + 63) // 64 + 3) // 4 * 4 for m in e.M], nd, nf, float(np.max(np.abs(fwd - ofwd))), na, len(pq)), flush=True)
    e.close()
