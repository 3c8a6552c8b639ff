"""Debug: one (query, model) pair of a golden case: stage-by-stage detail, GPU (with and without
the multidomain resolver) next to the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
case_name, qi, hj = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
from oracle import oracle as orc
import witch_amd._lib as _L
if os.environ.get('WITCH_LIB'):
    _L.LIB_PATH = os.environ['WITCH_LIB']
    _L.SYMBOLS.pop('wh_set_option', None) if 'old_lib' in _L.LIB_PATH else None
from tests.conftest import load_case
from witch_amd.ehmm import EHMM, pack_queries
case = load_case(case_name)
e = EHMM([case.hmm_paths[hj]])
s = e.digitize(case.qseqs[qi])
res, offs = pack_queries([s])
for opt in ("", "1"):
    if not os.environ.get('WITCH_LIB'):
        e.set_option("WH_NO_RESOLVE", opt)
    deci, flags, fwd, det = e.score(res, offs, want_fwd=True, want_detail=True)
    d = det[0]
    print("GPU no_resolve=%r" % opt, int(deci[0, 0]), int(flags[0, 0]), "fwd_bits %.5f seq %.5f pre %.5f seqbias %.6f nreg %d" %
          (d.fwd_bits, d.seq_score, d.pre_score, d.seqbias_nats, d.nregions),
          [(d.env_i[t], d.env_j[t], round(d.envsc[t], 4), round(d.domcorr[t], 5)) for t in range(d.nenv)], flush=True)
r = orc.OracleHMM(case.hmm_paths[hj]).score(s)
print("oracle", r.decibits, r.flags, "fwd_bits %.5f seq %.5f pre %.5f seqbias %.6f nreg %d" % (r.fwd_bits, r.seq_score, r.pre_score, r.seqbias_nats, r.nregions),
      [(r.env_i[t], r.env_j[t], round(r.envsc[t], 4), round(r.domcorr[t], 5)) for t in range(r.nenv)])
