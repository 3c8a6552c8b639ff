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


def warm_up(device: int = 0):
    """Process start-up work, once: load libwitch_hip.so (and with it the HIP runtime PyTorch bundles) and create the
    HIP context of ``device``.  Optional - QueryAlignmentEngine.run does it on first use - but a caller that times its
    first batch (tools/bench_level1.py) or wants the ~0.7 s out of its first call does it beside its own imports."""
    from .._lib import check, lib
    L = lib()
    check(L.wh_init(int(device)), "wh_init")
    name = (__import__("ctypes").c_char * 64)()
    check(L.wh_device_info(name, 64, None, None), "wh_device_info")
    return name.value.decode()


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
        self.pair_of = None           # (row, hmm label) -> pair number: only for engines built from precomputed arrays (from_results)
        self.qpair_off = None         # int64 [nloc+1]: first aligned pair of every local query (pairs of a query in top-k order)
        self.num_hmms = 0
        self.timings = {}
        self.merged = None            # int32 CSR by query: consensus codes (gcmm/merge.py)
        self.query_offsets = None     # int64 [nq+1]
        self.merged_minmax = None
        self.truncated_pairs = []     # (taxon, hmm label) with WH_FLAG_TRUNC
        self.long_list_pairs = 0      # pairs with more regions than a scoring kernel lists, scored in full by the long-list pass
        self.unaligned_pairs = []     # (taxon, hmm label) the alignment stage returned unaligned
        self.query_text = None        # uint8: the local queries' characters (upper-cased on read), concatenated like query_offsets
        self.device = 0
        # multi-GPU (one process per GPU): this rank scored, aligned and merged the
        # contiguous block [row_lo, row_hi) of the batch; the top-k tables cover EVERY query once gathered
        self.world, self.rank = 1, 0
        self.row_lo, self.row_hi = 0, 0
        self.topk_rows = (0, 0)       # rows the top-k tables cover

    # ------------------------------------------------------------------ construction
    @classmethod
    def run(cls, index_to_hmm, unaligned, num_hmms: int, device: int = 0, multidomain_policy: str = "resolve",
            subset_to_retained_columns=None, subset_to_nongaps_per_column=None, backbone_length=None,
            world: int = 1, rank: int = 0, group=None, chunk: int = 20000, keep_scores: bool = True):
        """Score, weight and align every query of ``unaligned`` ({taxon: sequence text} or a list of
        (taxon, text)) against every HMM of ``index_to_hmm``.

        One process per GPU: with ``world`` > 1 (torch.distributed initialised, backend "nccl" = RCCL on
        the GPU box, "gloo" on CPU) this rank works on its contiguous block of queries
        (distributed.shard_range; the eHMM is replicated), the per-query top-k records of all ranks are
        all-gathered (the path's only exchange step), aligned columns and consensus results stay
        rank-local.  multidomain_policy: "resolve" (HMMER's stochastic resolver, default) or "drop"
        (pairs with a multidomain region are not reported).

        ``chunk``: queries per pass (default 20 000, the reference's own hmmsearch chunk, algorithm.py:209; 0 = all at
        once).  ``keep_scores=False`` drops the nq x H deci-bit / flag tables once a chunk's top-k is formed - the host
        then holds top-k records, aligned columns and consensus codes only (rankBitscores / ranked() need the tables)."""
        import time
        import warnings
        if multidomain_policy not in ("resolve", "drop", "envelope"):
            raise ValueError("multidomain_policy must be 'resolve', 'drop' or 'envelope', not %r" % (multidomain_policy,))
        from ..distributed import shard_range
        from ..ehmm import EHMM, pack_queries
        from .algorithm import check_query_names
        items = list(unaligned.items()) if hasattr(unaligned, "items") else list(unaligned)
        check_query_names([t for t, _ in items])                       # algorithm.py:351-359
        labels = sorted(index_to_hmm.keys())
        paths = [index_to_hmm[i].hmm_model_path for i in labels]
        nseq = [int(index_to_hmm[i].num_taxa) for i in labels]
        t_load = time.time()
        e = EHMM(paths, hmm_index=labels, nseq=nseq, device=device)
        if multidomain_policy == "envelope":                  # round-1 behaviour: a multidomain region stays ONE envelope
            e.set_option("WH_NO_RESOLVE", "1")
        t_load = time.time() - t_load
        self = cls()
        self.num_hmms = int(num_hmms)
        self.hmm_index = np.asarray(labels, dtype=np.int32)
        self.num_taxa = np.asarray(nseq, dtype=np.int32)
        self.taxa = [t for t, _ in items]
        self.taxon_row = {t: r for r, t in enumerate(self.taxa)}
        self.world, self.rank = int(world), int(rank)
        self.row_lo, self.row_hi = shard_range(len(items), self.rank, self.world)
        local = items[self.row_lo:self.row_hi]
        # the reference upper-cases sequences on read (helpers/alignment_tools.py:730-731)
        t0 = time.time()
        upper = [s.upper() for _, s in local]
        res, offs = e.digitize_many(upper)
        self.query_text = np.frombuffer("".join(upper).encode("ascii"), dtype=np.uint8)
        self.device = int(device)
        t_digit = time.time() - t0
        # ---- the batch in CHUNKS of at most <chunk> queries, as the reference feeds hmmsearch (algorithm.py:209,280-284:
        # chunks of <= 20 000 sequences): score -> top-k -> align -> consensus per chunk, so the device-side tables and
        # workspaces are bounded by the chunk whatever the query count, and a chunk is far below the 2^31 pairs one
        # scoring call serves.  Chunks are independent (every stage is per query), so the concatenated results ARE the
        # one-call results (tests/test_gpu_parity.py: chunk = 257 on the example data).
        nloc = len(local)
        step = nloc if not chunk or int(chunk) <= 0 else int(chunk)
        lut = np.full(int(self.hmm_index.max()) + 1, -1, dtype=np.int32)
        lut[self.hmm_index] = np.arange(len(labels), dtype=np.int32)
        do_merge = subset_to_retained_columns is not None
        if do_merge:
            ret = [np.asarray(subset_to_retained_columns[i], dtype=np.int32) for i in labels]
            ng = [np.asarray(subset_to_nongaps_per_column[i], dtype=np.int32) for i in labels]
        parts = {k_: [] for k_ in ("deci", "flags", "idx", "w", "nk", "nu", "cols", "colen", "merged", "mm")}
        tim = {"search": 0.0, "weights": 0.0, "align": 0.0, "merge": 0.0}
        n_pairs_before = 0
        for c0 in range(0, max(nloc, 1), max(step, 1)):
            c1 = min(c0 + step, nloc)
            coffs = offs[c0:c1 + 1] - offs[c0]
            cres = res[offs[c0]:offs[c1]]
            t0 = time.time()
            deci, flags = e.score(cres, coffs)
            self.long_list_pairs += e.last_long_list_pairs()
            if multidomain_policy == "drop":
                drop = (flags & 2) != 0
                flags = np.where(drop, flags & ~np.uint8(1), flags).astype(np.uint8)
            trunc = np.argwhere((flags & 8) != 0)
            if len(trunc):
                # WH_FLAG_TRUNC: a list of the pair overflowed (include/witch_hip.h: since round 5 NOT the region list - pairs
                # with more than WH_MAX_ENVELOPES regions are scored in full by the long-list pass - but more than 32 domains
                # or 64 clusters inside ONE multidomain region); its score may differ from hmmsearch's.  Never silent.
                self.truncated_pairs += [(self.taxa[int(q) + c0 + self.row_lo], int(labels[int(h)])) for q, h in trunc]
            t1 = time.time()
            idx, w, nk, nu = e.topk(deci, flags, self.num_hmms)
            t2 = time.time()
            # pairs (local query, kept model) of the 0.999 prefix (aligner.py:58-63), built with array ops
            keep = np.arange(self.num_hmms)[None, :] < nu[:, None]
            pq = np.nonzero(keep)[0].astype(np.int64)
            plab = idx[keep].astype(np.int64)
            ph = lut[plab]
            cols, co = e.align(cres, coffs, pq, ph)
            _, unal_pairs = e.last_align_status()
            pw_mask = np.ones(len(pq), dtype=bool)
            if len(unal_pairs):
                # pairs the any-size kernel could not align come back all -1: they are NOT all-insertion alignments.
                # They are dropped from the consensus (weight 0: the max-weight trace is invariant to the common
                # scale of the remaining weights) and reported.
                pw_mask[unal_pairs] = False
                self.unaligned_pairs += [(self.taxa[int(pq[p_]) + c0 + self.row_lo], int(plab[p_])) for p_ in unal_pairs]
            t3 = time.time()
            if do_merge:
                # the weighted consensus DP of alignSubQueriesNew (aligner.py:376-473), all queries of the chunk at once
                qpo = np.zeros(c1 - c0 + 1, dtype=np.int64)
                qpo[1:] = np.cumsum(nu)
                pw = np.where(pw_mask, w[keep].astype(np.float64), 0.0)
                merged, mm = e.consensus(coffs, qpo, ph, pw, co, cols, ret, ng, int(backbone_length))
                parts["merged"].append(merged)
                parts["mm"].append(mm)
                tim["merge"] += time.time() - t3
            if keep_scores:
                parts["deci"].append(deci)
                parts["flags"].append(flags)
            for k_, v_ in (("idx", idx), ("w", w), ("nk", nk), ("nu", nu), ("cols", cols), ("colen", np.diff(co))):
                parts[k_].append(v_)
            n_pairs_before += len(pq)
            tim["search"] += t1 - t0
            tim["weights"] += t2 - t1
            tim["align"] += t3 - t2
        cat = lambda k_, empty: np.concatenate(parts[k_]) if parts[k_] else empty
        H_ = len(labels)
        self.decibits = cat("deci", np.zeros((0, H_), np.int32)) if keep_scores else None
        self.flags = cat("flags", np.zeros((0, H_), np.uint8)) if keep_scores else None
        self.topk_idx, self.topk_w = cat("idx", np.zeros((0, self.num_hmms), np.int32)), cat("w", np.zeros((0, self.num_hmms), np.float64))
        self.n_kept, self.n_used = cat("nk", np.zeros(0, np.int32)), cat("nu", np.zeros(0, np.int32))
        self.topk_rows = (self.row_lo, self.row_hi)
        self.cols = cat("cols", np.zeros(0, np.int32))
        self.col_offsets = np.zeros(n_pairs_before + 1, dtype=np.int64)
        self.col_offsets[1:] = np.cumsum(cat("colen", np.zeros(0, np.int64)))
        # pair number of (local row r, its j-th kept model) = qpair_off[r] + j: no dictionary over the pairs
        self.qpair_off = np.zeros(nloc + 1, dtype=np.int64)
        self.qpair_off[1:] = np.cumsum(self.n_used)
        self.pair_of = None
        if self.truncated_pairs:
            warnings.warn("witch_amd: %d (query, HMM) pair(s) overflowed a list of the scoring kernels (WH_FLAG_TRUNC: more than 32 domains or 64 "
                          "clusters in one multidomain region); their scores may differ from hmmsearch's (first: %s vs A_0_%d)"
                          % ((len(self.truncated_pairs),) + self.truncated_pairs[0]), RuntimeWarning)
        if self.unaligned_pairs:
            warnings.warn("witch_amd: %d pair(s) on models of more than 3072 nodes could not be aligned and are left out "
                          "of the consensus (first: %s vs A_0_%d)" % ((len(self.unaligned_pairs),) + self.unaligned_pairs[0]), RuntimeWarning)
        # same three stage names the reference logs (algorithm.py:333-335, weighting.py:165-168, aligner.py:520-525)
        self.timings = {"load_ehmm": t_load, "digitize": t_digit, "search": tim["search"], "weights": tim["weights"], "align": tim["align"],
                        "chunks": (nloc + max(step, 1) - 1) // max(step, 1)}
        self.query_offsets = offs
        if do_merge:
            self.merged = cat("merged", np.zeros(0, np.int32))
            self.merged_minmax = cat("mm", np.zeros((0, 2), np.int32))
            self.timings["merge"] = tim["merge"]
        e.close()
        from ..distributed import collectives_forced
        if self.world > 1 or collectives_forced():
            self.gather(group)
        return self

    def gather(self, group=None):
        """The path's one exchange step: all-gather of the per-query top-k records
        (int32 idx[k], float64 w[k], n_kept, n_used); afterwards writeWeights answers for every query on
        every rank.  Scores, aligned columns and consensus results stay with the rank that owns the query."""
        import torch
        import torch.distributed as dist
        from ..distributed import collectives_forced, gather_topk
        if self.world <= 1 and not collectives_forced():
            return self
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t0 = __import__("time").time()
        parts = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (self.topk_idx, self.topk_w, self.n_kept, self.n_used)]
        idx, w, nk, nu = gather_topk(*parts, group=group, n_total=len(self.taxa))
        self.topk_idx, self.topk_w, self.n_kept, self.n_used = [t.cpu().numpy() for t in (idx, w, nk, nu)]
        assert self.topk_idx.shape[0] == len(self.taxa), "ranks were given different query lists"
        self.topk_rows = (0, len(self.taxa))
        self.timings["gather"] = __import__("time").time() - t0
        return self

    def owns(self, row: int) -> bool:
        return self.row_lo <= row < self.row_hi

    @classmethod
    def from_results(cls, taxa, hmm_index, num_taxa, decibits, flags, num_hmms, topk=None, aligned=None,
                     rows=None, world: int = 1, rank: int = 0):
        """Assemble an engine from precomputed arrays (host-logic tests, checkpoints).  ``rows`` = (lo, hi):
        the arrays cover only that block of ``taxa`` (a rank's shard)."""
        self = cls()
        self.taxa = list(taxa)
        self.taxon_row = {t: r for r, t in enumerate(self.taxa)}
        self.hmm_index = np.asarray(hmm_index, dtype=np.int32)
        self.num_taxa = np.asarray(num_taxa, dtype=np.int32)
        self.decibits = np.asarray(decibits, dtype=np.int32)
        self.flags = np.asarray(flags, dtype=np.uint8)
        self.num_hmms = int(num_hmms)
        self.world, self.rank = int(world), int(rank)
        self.row_lo, self.row_hi = rows if rows is not None else (0, len(self.taxa))
        self.topk_rows = (self.row_lo, self.row_hi)
        if topk is not None:
            self.topk_idx, self.topk_w, self.n_kept, self.n_used = topk
        if aligned is not None:
            self.cols, self.col_offsets, self.pair_of = aligned
        return self

    # ------------------------------------------------------------------ lookups
    def _local(self, row: int, what: str) -> int:
        if not self.owns(row):
            raise KeyError("%s of query row %d live on the rank that owns rows [%d, %d); this is rank %d of %d with rows [%d, %d)"
                           % (what, row, *self._owner_rows(row), self.rank, self.world, self.row_lo, self.row_hi))
        return row - self.row_lo

    def _owner_rows(self, row: int):
        from ..distributed import shard_range
        for r in range(self.world):
            lo, hi = shard_range(len(self.taxa), r, self.world)
            if lo <= row < hi:
                return lo, hi
        return 0, 0

    def _need_scores(self):
        if self.decibits is None:
            raise RuntimeError("this engine ran with keep_scores=False: the nq x H score table was not kept (top-k records, aligned "
                               "columns and consensus codes are; rankBitscores / ranked() need keep_scores=True)")

    def reported_counts(self):
        """Per local row the number of models that reported the query (one pass over the flag table, cached)."""
        if getattr(self, "_nrep", None) is None:
            self._need_scores()
            self._nrep = ((self.flags & 1) != 0).sum(axis=1)
        return self._nrep

    def ranked(self, row: int):
        """[(idx, score)] sorted by score descending (loader.py:325-330), ties by idx.  ONE row is sorted on request
        (a composite integer key over its reported models): nobody walks all 10^5 lists, writeWeights reads the keys only."""
        r = self._local(row, "the scores")
        self._need_scores()
        if getattr(self, "_label_rank", None) is None:
            self._label_rank = np.argsort(np.argsort(self.hmm_index, kind="stable"), kind="stable").astype(np.int64)
        js = np.nonzero(self.flags[r] & 1)[0]
        key = (-self.decibits[r, js].astype(np.int64)) * (len(self.hmm_index) + 1) + self._label_rank[js]
        js = js[np.argsort(key, kind="stable")]
        return list(zip(self.hmm_index[js].tolist(), (self.decibits[r, js] / 10.0).tolist()))

    def has_hit(self, row: int) -> bool:
        return bool(self.reported_counts()[self._local(row, "the scores")])

    def weights(self, row: int):
        """((idx, np.float64 w), ...) - the value type calculateWeights returns (weighting.py:71-74)."""
        lo, hi = self.topk_rows
        if not lo <= row < hi:
            raise KeyError("top-k record of query row %d is not on this rank: call gather() first" % row)
        r = row - lo
        n = int(self.n_kept[r])
        return tuple((int(self.topk_idx[r, j]), np.float64(self.topk_w[r, j])) for j in range(n))

    def pair_number(self, row: int, label: int) -> int:
        """Number of the aligned pair (query row, model label) in cols / col_offsets: the pairs are ordered by query and,
        inside a query, by rank in its top-k table - qpair_off[r] + j (KeyError for a model outside the query's 0.999 prefix)."""
        r = self._local(row, "the aligned columns")
        if self.pair_of is not None:                      # engines assembled from precomputed arrays bring a dictionary
            return self.pair_of[(row, int(label))]
        t = row - self.topk_rows[0]
        js = np.nonzero(self.topk_idx[t, :int(self.n_used[t])] == int(label))[0]
        if not len(js):
            raise KeyError((row, int(label)))
        return int(self.qpair_off[r] + js[0])

    def aligned_columns(self, row: int, label: int):
        p = self.pair_number(row, label)
        return self.cols[self.col_offsets[p]:self.col_offsets[p + 1]].tolist()
