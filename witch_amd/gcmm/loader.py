"""rankBitscores / readAndRankBitscoreMP with the reference's signatures
(witch_msa/gcmm/loader.py:299-376), answered from the batched GPU run."""
from collections.abc import Mapping

import numpy as np

from .engine import current_engine


class RankedBitscores(Mapping):
    """{taxon: [(hmm index, bit-score), ...]} answered from the engine's score table on access.  The reference
    materialises every list (loader.py:310-332); at 100 000 queries x 200 HMMs that is 2*10^7 Python tuples
    nobody reads (writeWeights only walks the keys), so the lists are built when asked for."""

    def __init__(self, eng, wanted, renamed_taxa):
        self._eng, self._wanted = eng, wanted
        all_wanted = set(int(i) for i in eng.hmm_index.tolist()) <= wanted
        self._filter = not all_wanted
        # a taxon with no reported HMM never appears (loader.py:291-293): the rows with a hit, from one pass over the flags
        rows = (np.nonzero(eng.reported_counts() > 0)[0] + eng.row_lo).tolist()
        taxa = eng.taxa
        if renamed_taxa:
            self._row = {renamed_taxa.get(taxa[r], taxa[r]): r for r in rows}
        else:
            self._row = dict(zip([taxa[r] for r in rows], rows))
        if self._filter:
            self._row = {n: r for n, r in self._row.items() if self._get(r)}

    def _get(self, row):
        scores = self._eng.ranked(row)
        return [(i, s) for (i, s) in scores if i in self._wanted] if self._filter else scores

    def __getitem__(self, name):
        return self._get(self._row[name])

    def __iter__(self):
        return iter(self._row)

    def __len__(self):
        return len(self._row)

    def __contains__(self, name):
        return name in self._row


def readAndRankBitscoreMP(index_to_hmm, renamed_taxa, lock=None, pool=None):
    """{taxon: [(hmm index, bit-score), ...]} sorted by score, descending (a read-only mapping).

    The reference sorts with Python's stable sort over the arrival order of pool futures
    (loader.py:310-330), so its order among equal scores is not reproducible; here equal
    scores are ordered by ascending HMM index.  With several ranks (one per
    GPU) a rank returns the queries of its own block; writeWeights answers for every query."""
    eng = current_engine()
    return RankedBitscores(eng, set(int(i) for i in index_to_hmm.keys()), renamed_taxa)


def rankBitscores(index_to_hmm, renamed_taxa, lock=None, pool=None):
    return readAndRankBitscoreMP(index_to_hmm, renamed_taxa, lock, pool)
