"""writeWeights and friends with the reference's signatures (witch_msa/gcmm/weighting.py)."""
import re
from collections.abc import Mapping

import numpy as np

from .engine import current_engine


def calculateWeights(packed_data, num_hmms=None):
    """Reference formula (weighting.py:58-74) for ONE query given as
    (taxon, indexes, bitscores, sizes); used for queries the engine has not seen.
    w_i = 1 / sum_j 2^((s_j - s_i) + log2(n_j / n_i)); keep the top num_hmms, ties by
    (-weight, -score, +index)."""
    taxon, indexes, bitscores, sizes = packed_data
    assert len(indexes) == len(bitscores) == len(sizes)
    bits = np.array(bitscores, dtype=np.float64)
    sz = np.array(sizes, dtype=np.float64)
    weights = {}
    for i in range(len(bitscores)):
        exponents = bits - bits[i] + np.log2(sz / sz[i])
        weights[indexes[i]] = 1. / np.sum(np.power(2, exponents))
    k = current_engine().num_hmms if num_hmms is None else num_hmms
    score_of = dict(zip(indexes, bitscores))
    kept = sorted(weights.items(), key=lambda t: (-t[1], -score_of[t[0]], t[0]))[:min(k, len(weights))]
    return {taxon: tuple(kept)}


class WeightTable(Mapping):
    """{taxon: ((idx, np.float64 weight), ...)} answered from the engine's (gathered) top-k table on access.  The
    reference materialises every tuple (weighting.py:140-163); at 10^5 queries x 10 kept models that is 10^6 Python
    objects per rank - and with several ranks EVERY rank answers for EVERY query - that the callers then read once or
    never (getBackbones reads one row per query, writeWeightsToLocal walks them once).  Same values, built when asked."""

    def __init__(self, eng, rows, extra):
        self._eng, self._rows, self._extra = eng, rows, extra      # taxon -> row of the top-k table; foreign taxa -> their tuples

    def __getitem__(self, taxon):
        row = self._rows.get(taxon)
        return self._eng.weights(row) if row is not None else self._extra[taxon]

    def __iter__(self):
        yield from self._rows
        yield from self._extra

    def __len__(self):
        return len(self._rows) + len(self._extra)

    def __contains__(self, taxon):
        return taxon in self._rows or taxon in self._extra


def writeWeights(index_to_hmm, ranked_bitscores, pool=None):
    """{taxon: ((idx, np.float64 weight), ...)} for every taxon of ranked_bitscores
    (weighting.py:121-169).  Weights come from the device top-k kernel; the mapping is read-only and builds a query's
    tuple on access (WeightTable)."""
    eng = current_engine()
    taxon_row = eng.taxon_row
    rows, extra = {}, {}
    for taxon in ranked_bitscores.keys():
        row = taxon_row.get(taxon)
        if row is None:
            # renamed or foreign taxon: fall back to the formula on the scores handed in
            scores = ranked_bitscores[taxon]
            idxs = [x[0] for x in scores]
            extra.update(calculateWeights((taxon, idxs, [x[1] for x in scores],
                                           [index_to_hmm[i].num_taxa for i in idxs])))
        else:
            rows[taxon] = row
    return WeightTable(eng, rows, extra)


def writeWeightsToLocal(taxon_to_weights, path):
    """weights.txt: one line per query 'taxon:((idx, w), ...)' (weighting.py:174-179)."""
    with open(path, 'w') as f:
        for taxon, weights in taxon_to_weights.items():
            f.write('{}:{}\n'.format(taxon, tuple((int(i), float(w)) for i, w in weights)))


_NP_SCALAR = re.compile(r'(?:np|numpy)\.(?:float64|int64|int32)\(([^()]*)\)')


def readWeightsFromLocal(path):
    """Inverse of writeWeightsToLocal (weighting.py:185-194), without eval().  Accepts both texts the
    reference's writer produces: plain numbers (numpy 1 prints np.float64 as '0.5') and numpy 2's
    'np.float64(0.5)' (tests/golden/wire/ref_weights.txt is such a file, written by the reference)."""
    import ast
    out = {}
    with open(path, 'r') as f:
        for line in f:
            if not line.strip():
                continue
            taxon, raw = line.rsplit(':', 1)
            raw = _NP_SCALAR.sub(r'\1', raw.strip())
            out[taxon] = tuple((int(i), np.float64(w)) for i, w in ast.literal_eval(raw))
    return out
