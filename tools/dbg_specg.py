import sys, time, tempfile, numpy as np
sys.path.insert(0, '/root/repo')
from witch_amd import synth
from witch_amd.ehmm import EHMM, pack_queries
root = int(sys.argv[1]); nq = int(sys.argv[2])
fam = synth.make_family(5, root, 8, "dna", 0.03, 0.001)
d = tempfile.mkdtemp()
eh = synth.make_ehmm(fam, 1, d, witch_layout=False)
names, seqs = synth.make_queries(fam, 1, nq, 150)
e = EHMM(eh.paths)
res, offs = pack_queries([s.astype(np.uint8) for s in seqs])
print("M", e.M, flush=True)
d_, f_ = e.score(res, offs)
print("score ok", d_[:, 0].tolist(), flush=True)
if len(sys.argv) > 3:
    pq = list(range(nq)); ph = [0] * nq
    cols, co = e.align(res, offs, pq, ph)
    print("align ok", int((cols >= 0).sum()), flush=True)
