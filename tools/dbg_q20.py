import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from tests.conftest import load_case
from witch_amd.ehmm import EHMM, pack_queries
case = load_case(sys.argv[1])
n = int(sys.argv[2])
e = EHMM(case.hmm_paths, hmm_index=case.hmm_index, nseq=case.nseq)
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
seqs = [e.digitize(s)[:cap] for s in case.qseqs[:n]]
print("M", e.M, "lens", [len(s) for s in seqs], flush=True)
res, offs = pack_queries(seqs)
t = time.time()
d, f = e.score(res, offs)
print("score ok", time.time() - t, d[:, 0].tolist(), f[:, 0].tolist(), flush=True)
