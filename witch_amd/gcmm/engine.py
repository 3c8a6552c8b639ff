"""QueryAlignmentEngine: the batched GPU run behind the reference-shaped functions.

Input contract = what the reference has at gcmm.py:205-222: ``index_to_hmm`` (objects
with ``hmm_model_path`` and ``num_taxa``, loader.py:17-65) and the query sequences.
Output contract = SURVEY.md section 8.0: deci-bit scores + reported mask, top-k
(idx, np.float64 weight) tuples, per-residue aligned match columns.
"""
from __future__ import annotations

import numpy as np

_ENGINE = None


def install(engine):
    """Make <engine> the table the reference-shaped functions answer from."""
    global _ENGINE
    _ENGINE = engine
    return engine


def current_engine():
    if _ENGINE is None:
        raise RuntimeError("witch_amd.gcmm: no QueryAlignmentEngine installed "
                           "(call witch_amd.gcmm.install(QueryAlignmentEngine.run(...)) in the parent process)")
    return _ENGINE


class QueryAlignmentEngine:
    def __init__(self):
        self.taxa = []                # query names in batch order
        self.taxon_row = {}           # name -> row
        self.hmm_index = None         # int32 [H] labels (A_0_<idx>)
        self.num_taxa = None          # int32 [H]
        self.decibits = None          # int32 [nq, H]
        self.flags = None             # uint8 [nq, H]
        self.topk_idx = None          # int32 [nq, k]
        self.topk_w = None            # float64 [nq, k]
        self.n_kept = None
        self.n_used = None
        self.cols = None              # int32 CSR
        self.col_offsets = None       # int64 [npairs+1]
        self.pair_of = {}             # (row, hmm label) -> pair number
        self.num_hmms = 0
        self.timings = {}
        self.merged = None            # int32 CSR by query: consensus codes (gcmm/merge.py)
        self.query_offsets = None     # int64 [nq+1]
        self.merged_minmax = None

    # ------------------------------------------------------------------ construction
    @classmethod
    def run(cls, index_to_hmm, unaligned, num_hmms: int, device: int = 0, multidomain_policy: str = "envelope",
            subset_to_retained_columns=None, subset_to_nongaps_per_column=None, backbone_length=None):
        """Score, weight and align every query of ``unaligned`` ({taxon: sequence text} or
        a list of (taxon, text)) against every HMM of ``index_to_hmm`` on one MI355X."""
        import time
        from ..ehmm import EHMM, pack_queries
        items = list(unaligned.items()) if hasattr(unaligned, "items") else list(unaligned)
        labels = sorted(index_to_hmm.keys())
        paths = [index_to_hmm[i].hmm_model_path for i in labels]
        nseq = [int(index_to_hmm[i].num_taxa) for i in labels]
        e = EHMM(paths, hmm_index=labels, nseq=nseq, device=device)
        self = cls()
        self.num_hmms = int(num_hmms)
        self.hmm_index = np.asarray(labels, dtype=np.int32)
        self.num_taxa = np.asarray(nseq, dtype=np.int32)
        self.taxa = [t for t, _ in items]
        self.taxon_row = {t: r for r, t in enumerate(self.taxa)}
        # the reference upper-cases sequences on read (helpers/alignment_tools.py:730-731)
        seqs = [e.digitize(s.upper()) for _, s in items]
        res, offs = pack_queries(seqs)
        t0 = time.time()
        self.decibits, self.flags = e.score(res, offs)
        if multidomain_policy == "drop":
            drop = (self.flags & 2) != 0
            self.flags = np.where(drop, self.flags & ~np.uint8(1), self.flags).astype(np.uint8)
        t1 = time.time()
        self.topk_idx, self.topk_w, self.n_kept, self.n_used = e.topk(self.decibits, self.flags, self.num_hmms)
        t2 = time.time()
        pq, ph, key = [], [], []
        for r in range(len(self.taxa)):
            for j in range(int(self.n_used[r])):
                lab = int(self.topk_idx[r, j])
                pq.append(r)
                ph.append(e.pos_of_index[lab])
                key.append((r, lab))
        self.cols, self.col_offsets = e.align(res, offs, pq, ph)
        self.pair_of = {k: p for p, k in enumerate(key)}
        t3 = time.time()
        # same three stage names the reference logs (algorithm.py:333-335, weighting.py:165-168, aligner.py:520-525)
        self.timings = {"search": t1 - t0, "weights": t2 - t1, "align": t3 - t2}
        self.query_offsets = offs
        if subset_to_retained_columns is not None:
            # the weighted consensus DP of alignSubQueriesNew (aligner.py:376-473), all queries at once
            qpo = np.zeros(len(self.taxa) + 1, dtype=np.int64)
            qpo[1:] = np.cumsum(self.n_used)
            pw = np.array([self.topk_w[r, j] for r in range(len(self.taxa)) for j in range(int(self.n_used[r]))], dtype=np.float64)
            ret = [np.asarray(subset_to_retained_columns[i], dtype=np.int32) for i in labels]
            ng = [np.asarray(subset_to_nongaps_per_column[i], dtype=np.int32) for i in labels]
            self.merged, self.merged_minmax = e.consensus(offs, qpo, ph, pw, self.col_offsets, self.cols, ret, ng,
                                                          int(backbone_length))
            self.timings["merge"] = time.time() - t3
        e.close()
        return self

    @classmethod
    def from_results(cls, taxa, hmm_index, num_taxa, decibits, flags, num_hmms, topk=None, aligned=None):
        """Assemble an engine from precomputed arrays (host-logic tests, checkpoints)."""
        self = cls()
        self.taxa = list(taxa)
        self.taxon_row = {t: r for r, t in enumerate(self.taxa)}
        self.hmm_index = np.asarray(hmm_index, dtype=np.int32)
        self.num_taxa = np.asarray(num_taxa, dtype=np.int32)
        self.decibits = np.asarray(decibits, dtype=np.int32)
        self.flags = np.asarray(flags, dtype=np.uint8)
        self.num_hmms = int(num_hmms)
        if topk is not None:
            self.topk_idx, self.topk_w, self.n_kept, self.n_used = topk
        if aligned is not None:
            self.cols, self.col_offsets, self.pair_of = aligned
        return self

    # ------------------------------------------------------------------ lookups
    def ranked(self, row: int):
        """[(idx, score)] sorted by score descending (loader.py:325-330), ties by idx."""
        rep = (self.flags[row] & 1) != 0
        order = np.lexsort((self.hmm_index, -self.decibits[row]))
        return [(int(self.hmm_index[j]), float(self.decibits[row, j]) / 10.0) for j in order if rep[j]]

    def weights(self, row: int):
        """((idx, np.float64 w), ...) - the value type calculateWeights returns (weighting.py:71-74)."""
        n = int(self.n_kept[row])
        return tuple((int(self.topk_idx[row, j]), np.float64(self.topk_w[row, j])) for j in range(n))

    def aligned_columns(self, row: int, label: int):
        p = self.pair_of[(row, int(label))]
        return self.cols[self.col_offsets[p]:self.col_offsets[p + 1]].tolist()
