"""rankBitscores / readAndRankBitscoreMP with the reference's signatures
(witch_msa/gcmm/loader.py:299-376), answered from the batched GPU run."""
from collections import defaultdict

from .engine import current_engine


def readAndRankBitscoreMP(index_to_hmm, renamed_taxa, lock=None, pool=None):
    """{taxon: [(hmm index, bit-score), ...]} sorted by score, descending.

    The reference sorts with Python's stable sort over the arrival order of pool futures
    (loader.py:310-330), so its order among equal scores is not reproducible; here equal
    scores are ordered by ascending HMM index.  With several ranks (one per
    GPU) a rank returns the queries of its own block; writeWeights answers for every query."""
    eng = current_engine()
    wanted = set(int(i) for i in index_to_hmm.keys())
    ranked = defaultdict(list)
    for row in range(eng.row_lo, eng.row_hi):
        taxon = eng.taxa[row]
        scores = [(i, s) for (i, s) in eng.ranked(row) if i in wanted]
        if not scores:
            continue          # a taxon with no reported HMM never appears (loader.py:291-293)
        name = renamed_taxa[taxon] if renamed_taxa and taxon in renamed_taxa else taxon
        ranked[name] = scores
    return ranked


def rankBitscores(index_to_hmm, renamed_taxa, lock=None, pool=None):
    return readAndRankBitscoreMP(index_to_hmm, renamed_taxa, lock, pool)
