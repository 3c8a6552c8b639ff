"""Debug: one (query, model) pair of a golden case through the device resolver and the oracle,
both printing their sampled segments and cluster statistics (WH_RDBG / ORC_DBG = segments shown)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
case_name, qi, hj = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nshow = sys.argv[4] if len(sys.argv) > 4 else "40"
os.environ["ORC_DBG"] = nshow
from oracle import oracle as orc
from tests.conftest import load_case
from witch_amd.ehmm import EHMM, pack_queries
case = load_case(case_name)
e = EHMM([case.hmm_paths[hj]])
e.set_option("WH_RDBG", nshow)
s = e.digitize(case.qseqs[qi])
res, offs = pack_queries([s])
deci, flags, det = e.score(res, offs, want_detail=True)
d = det[0]
print("GPU   ", int(deci[0, 0]), int(flags[0, 0]), [(d.env_i[t], d.env_j[t]) for t in range(d.nenv)], flush=True)
r = orc.OracleHMM(case.hmm_paths[hj]).score(s)
print("oracle", r.decibits, r.flags, [(r.env_i[t], r.env_j[t]) for t in range(r.nenv)])
