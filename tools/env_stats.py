import sys, tempfile, numpy as np, collections
sys.path.insert(0, '/root/repo')
from witch_amd import synth
from witch_amd.ehmm import EHMM, pack_queries
fam = synth.make_family(20251205, 900, 1024, "dna", 0.03, 1e-4)
d = tempfile.mkdtemp()
eh = synth.make_ehmm(fam, 200, d)
names, seqs = synth.make_queries(fam, 20251206, 256, 150)
e = EHMM(eh.paths, hmm_index=eh.index, nseq=eh.nseq)
res, offs = pack_queries([s.astype(np.uint8) for s in seqs])
deci, flags, det = e.score(res, offs, want_detail=True)
c = collections.Counter()
H = e.H
for q in range(len(seqs)):
    for h in range(H):
        dd = det[q * H + h]
        if dd.nenv == 1:
            c[(dd.env_i[0], dd.env_j[0])] += 1
        else:
            c[("nenv", dd.nenv)] += 1
tot = sum(c.values())
for k, v in c.most_common(12):
    print(k, v, "%.1f%%" % (100.0 * v / tot))
