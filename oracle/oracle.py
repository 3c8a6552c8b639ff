"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY - see p7_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The HMMER-side arithmetic lives in p7_oracle.c (float64); the
reference's own Python arithmetic on the path is restated here in numpy:

* rank_bitscores      <- witch_msa/gcmm/loader.py:310-330      (stable sort, desc)
* calculate_weights   <- witch_msa/gcmm/weighting.py:58-74     (w_i = 1/sum_j 2^(...))
* adaptive_cut        <- witch_msa/gcmm/aligner.py:58-63       (prefix until sum >= 0.999)
* canonical tie-break <- SURVEY.md section 8.0 (the reference's own tie order is the
  arrival order of futures, i.e. nondeterministic; both sides are canonicalised
  to (-weight, -decibit, +hmm index) before comparison)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_REPORTED, FLAG_MULTI, FLAG_OVERRIDE = 1, 2, 4
MAXENV = 256


class OrcResult(C.Structure):
    _fields_ = [
        ("flags", C.c_int), ("nregions", C.c_int), ("nenv", C.c_int),
        ("env_i", C.c_int * MAXENV), ("env_j", C.c_int * MAXENV), ("env_multi", C.c_int * MAXENV),
        ("envsc", C.c_double * MAXENV), ("domcorr", C.c_double * MAXENV),
        ("fwd_nats", C.c_double), ("null_nats", C.c_double), ("seqbias_nats", C.c_double),
        ("fwd_bits", C.c_double),
        ("pre_score", C.c_float), ("seq_score", C.c_float), ("sum_score", C.c_float),
        ("decibits", C.c_int),
    ]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libp7oracle.so")
    src = os.path.join(_HERE, "p7_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libp7oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_hmm_read.restype = C.c_void_p
        L.orc_hmm_read.argtypes = [C.c_char_p]
        L.orc_hmm_free.argtypes = [C.c_void_p]
        for f in ("M", "K", "Kp", "nseq", "alphabet"):
            getattr(L, "orc_hmm_" + f).argtypes = [C.c_void_p]
            getattr(L, "orc_hmm_" + f).restype = C.c_int
        L.orc_hmm_name.argtypes = [C.c_void_p]
        L.orc_hmm_name.restype = C.c_char_p
        L.orc_hmm_map.argtypes = [C.c_void_p]
        L.orc_hmm_map.restype = C.POINTER(C.c_int)
        for f in ("entry", "odds", "pt"):
            getattr(L, "orc_hmm_" + f).argtypes = [C.c_void_p]
            getattr(L, "orc_hmm_" + f).restype = C.POINTER(C.c_double)
        L.orc_digitize.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_void_p]
        L.orc_score_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(OrcResult)]
        L.orc_align_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_score_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_score_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_result_size.restype = C.c_int
        assert L.orc_result_size() == C.sizeof(OrcResult)
        _LIB = L
    return _LIB


class OracleHMM:
    def __init__(self, path: str):
        self._h = lib().orc_hmm_read(path.encode())
        if not self._h:
            raise ValueError("oracle: cannot parse HMM file %s" % path)
        L = lib()
        self.path = path
        self.M = L.orc_hmm_M(self._h)
        self.K = L.orc_hmm_K(self._h)
        self.Kp = L.orc_hmm_Kp(self._h)
        self.nseq = L.orc_hmm_nseq(self._h)
        self.alphabet = L.orc_hmm_alphabet(self._h)
        self.name = L.orc_hmm_name(self._h).decode()

    def __del__(self):
        try:
            if self._h:
                lib().orc_hmm_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def map(self):
        return np.ctypeslib.as_array(lib().orc_hmm_map(self._h), shape=(self.M + 1,)).copy()

    @property
    def entry(self):
        return np.ctypeslib.as_array(lib().orc_hmm_entry(self._h), shape=(self.M + 2,)).copy()

    @property
    def odds(self):
        return np.ctypeslib.as_array(lib().orc_hmm_odds(self._h), shape=(self.Kp, self.M + 1)).copy()

    @property
    def pt(self):
        return np.ctypeslib.as_array(lib().orc_hmm_pt(self._h), shape=(self.M + 1, 7)).copy()

    def digitize(self, text: str) -> np.ndarray:
        out = np.empty(len(text), dtype=np.uint8)
        lib().orc_digitize(self.alphabet, text.encode(), len(text), out.ctypes.data)
        return out

    def score(self, dsq: np.ndarray) -> OrcResult:
        dsq = np.ascontiguousarray(dsq, dtype=np.uint8)
        r = OrcResult()
        lib().orc_score_pair(self._h, dsq.ctypes.data, len(dsq), C.byref(r))
        return r

    def align(self, dsq: np.ndarray) -> np.ndarray:
        dsq = np.ascontiguousarray(dsq, dtype=np.uint8)
        cols = np.empty(len(dsq), dtype=np.int32)
        lib().orc_align_pair(self._h, dsq.ctypes.data, len(dsq), cols.ctypes.data)
        return cols


def pack(seqs):
    """list of uint8 arrays -> (residues, offsets[int64])"""
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    if len(seqs):
        offs[1:] = np.cumsum([len(s) for s in seqs])
    res = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if len(seqs) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(res), offs


def score_batch(hmms, residues, offsets, nthreads: int = 0):
    """All-vs-all scoring: returns (decibits[nq,H] int32, flags[nq,H] uint8,
    fwd_bits[nq,H] float64, seq_score[nq,H] float32)."""
    nq, nh = len(offsets) - 1, len(hmms)
    arr = (C.c_void_p * nh)(*[h._h for h in hmms])
    deci = np.zeros((nq, nh), dtype=np.int32)
    flags = np.zeros((nq, nh), dtype=np.uint8)
    fwd = np.zeros((nq, nh), dtype=np.float64)
    sc = np.zeros((nq, nh), dtype=np.float32)
    residues = np.ascontiguousarray(residues, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    lib().orc_score_batch(arr, nh, residues.ctypes.data, offsets.ctypes.data, nq, deci.ctypes.data,
                          flags.ctypes.data, fwd.ctypes.data, sc.ctypes.data, nthreads)
    return deci, flags, fwd, sc


def score_pairs(hmms, residues, offsets, pair_q, pair_h, nthreads: int = 0):
    """Scores a LIST of (query, model position) pairs (OpenMP over the pairs): returns (decibits int32, flags uint8,
    fwd_bits float64, seq_score float32), one entry per pair.  <hmms> may hold None for models no pair uses."""
    nh = len(hmms)
    arr = (C.c_void_p * nh)(*[h._h if h is not None else None for h in hmms])
    pair_q = np.ascontiguousarray(pair_q, dtype=np.int64)
    pair_h = np.ascontiguousarray(pair_h, dtype=np.int32)
    n = len(pair_q)
    deci = np.zeros(n, dtype=np.int32)
    flags = np.zeros(n, dtype=np.uint8)
    fwd = np.zeros(n, dtype=np.float64)
    sc = np.zeros(n, dtype=np.float32)
    residues = np.ascontiguousarray(residues, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    if n:
        lib().orc_score_pairs(arr, residues.ctypes.data, offsets.ctypes.data, pair_q.ctypes.data, pair_h.ctypes.data, n,
                              deci.ctypes.data, flags.ctypes.data, fwd.ctypes.data, sc.ctypes.data, nthreads)
    return deci, flags, fwd, sc


def align_batch(hmms, residues, offsets, pair_q, pair_h, nthreads: int = 0):
    """cols CSR over the residues of each pair; returns (cols, col_offsets)."""
    nh = len(hmms)
    arr = (C.c_void_p * nh)(*[h._h for h in hmms])
    pair_q = np.ascontiguousarray(pair_q, dtype=np.int64)
    pair_h = np.ascontiguousarray(pair_h, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    lens = offsets[pair_q + 1] - offsets[pair_q]
    co = np.zeros(len(pair_q) + 1, dtype=np.int64)
    co[1:] = np.cumsum(lens)
    cols = np.full(int(co[-1]), -1, dtype=np.int32)
    residues = np.ascontiguousarray(residues, dtype=np.uint8)
    lib().orc_align_batch(arr, residues.ctypes.data, offsets.ctypes.data, pair_q.ctypes.data,
                          pair_h.ctypes.data, len(pair_q), co.ctypes.data, cols.ctypes.data, nthreads)
    return cols, co


# ---------------------------------------------------------------------------
# The reference's own Python arithmetic on the path, restated
# ---------------------------------------------------------------------------
def rank_bitscores(hmm_index, decibits_row, reported_row):
    """loader.py:310-330: [(idx, score)] sorted by score descending.  The reference's
    tie order is arrival order; canonical order here is (-decibit, +idx)."""
    items = [(int(hmm_index[j]), int(decibits_row[j])) for j in range(len(hmm_index)) if reported_row[j]]
    items.sort(key=lambda t: (-t[1], t[0]))
    return [(i, d / 10.0) for i, d in items]


def calculate_weights(indexes, bitscores, sizes, num_hmms):
    """weighting.py:58-74 verbatim arithmetic (numpy float64), canonical tie order."""
    weights = {}
    bits = np.array(bitscores, dtype=np.float64)
    sz = np.array(sizes, dtype=np.float64)
    for i in range(len(bitscores)):
        exponents = bits - bits[i] + np.log2(sz / sz[i])
        weights[indexes[i]] = 1.0 / np.sum(np.power(2, exponents))
    score_of = {indexes[i]: bitscores[i] for i in range(len(indexes))}
    k = min(num_hmms, len(weights))
    out = sorted(weights.items(), key=lambda t: (-t[1], -score_of[t[0]], t[0]))[:k]
    return tuple(out)


def adaptive_cut(sorted_weights, target=0.999):
    """aligner.py:58-63: number of HMMs used = shortest prefix with sum >= target."""
    cur, idx = 0.0, 0
    while idx < len(sorted_weights) and cur < target:
        cur += sorted_weights[idx][1]
        idx += 1
    return idx


def set_resolve_multidomain(on: bool) -> None:
    """True (default): multidomain regions go through the A.4b trace ensemble + clustering;
    False: the round-1 behaviour (the whole region is one envelope) - kept for comparison only."""
    lib().orc_set_resolve_multidomain(1 if on else 0)
