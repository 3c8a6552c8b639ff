"""EHMM: an ensemble of profile HMMs resident on one MI355X, and the three operators of
the hot path (score -> top-k weights -> align) over the C ABI of libwitch_hip.so.

Host entry points take/return numpy arrays (the library stages them over PCIe);
the ``*_t`` entry points take torch CUDA tensors already resident in HBM and enqueue
on torch's current stream - PyTorch is only the allocator/stream plumbing here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PairDetail, WitchHipError, check, lib


def pack_queries(seqs):
    """list of uint8 arrays -> (residues uint8 [total], offsets int64 [n+1])"""
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    if len(seqs):
        offs[1:] = np.cumsum([len(s) for s in seqs])
        res = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs])
    else:
        res = np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(res), offs


class EHMM:
    def __init__(self, hmm_paths, hmm_index=None, nseq=None, device: int = 0):
        L = lib()
        check(L.wh_init(int(device)), "wh_init")
        n = len(hmm_paths)
        arr = (C.c_char_p * n)(*[str(p).encode() for p in hmm_paths])
        idx = np.ascontiguousarray(hmm_index if hmm_index is not None else np.arange(n), dtype=np.int32)
        ns = None if nseq is None else np.ascontiguousarray(nseq, dtype=np.int32)
        self._h = L.wh_ehmm_load(arr, idx.ctypes.data, None if ns is None else ns.ctypes.data, n)
        if not self._h:
            raise WitchHipError("wh_ehmm_load failed: %s" % L.wh_last_error().decode())
        self.device = int(device)
        self.paths = list(hmm_paths)
        self.H = L.wh_ehmm_count(self._h)
        self.alphabet = L.wh_ehmm_alphabet(self._h)
        self.M = np.zeros(self.H, dtype=np.int32)
        self.nseq = np.zeros(self.H, dtype=np.int32)
        self.index = np.zeros(self.H, dtype=np.int32)
        check(L.wh_ehmm_info(self._h, self.M.ctypes.data, self.nseq.ctypes.data, self.index.ctypes.data),
              "wh_ehmm_info")
        self.pos_of_index = {int(v): i for i, v in enumerate(self.index)}

    def close(self):
        if getattr(self, "_h", None):
            lib().wh_ehmm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def digitize(self, text: str) -> np.ndarray:
        out = np.empty(len(text), dtype=np.uint8)
        bad = lib().wh_digitize(self.alphabet, text.encode(), len(text), out.ctypes.data)
        if bad != 0:
            raise ValueError("query contains %d characters outside the %s alphabet" %
                             (bad, "amino" if self.alphabet == 2 else "nucleic"))
        return out

    def digitize_many(self, texts):
        """Digitise a list of sequence texts in ONE library call: (residues uint8 [total], offsets int64 [n+1])."""
        lens = np.fromiter((len(t) for t in texts), dtype=np.int64, count=len(texts))
        offs = np.zeros(len(texts) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        blob = "".join(texts).encode()
        if len(blob) != int(offs[-1]):
            raise ValueError("query sequences must be ASCII")
        out = np.empty(len(blob), dtype=np.uint8)
        bad = lib().wh_digitize(self.alphabet, blob, len(blob), out.ctypes.data) if len(blob) else 0
        if bad != 0:
            raise ValueError("queries contain %d characters outside the %s alphabet" %
                             (bad, "amino" if self.alphabet == 2 else "nucleic"))
        return out, offs

    def map_columns(self, h: int) -> np.ndarray:
        out = np.zeros(int(self.M[h]), dtype=np.int32)
        check(lib().wh_ehmm_map(self._h, int(h), out.ctypes.data), "wh_ehmm_map")
        return out

    def set_timing(self, on: bool):
        check(lib().wh_set_timing(self._h, 1 if on else 0), "wh_set_timing")

    def set_option(self, name: str, value: str = ""):
        """Development knob on a live handle (include/witch_hip.h: wh_set_option)."""
        check(lib().wh_set_option(self._h, name.encode(), str(value).encode()), "wh_set_option")

    def last_kernel_ms(self, which: int):
        ms, n = C.c_double(0), C.c_int(0)
        check(lib().wh_last_kernel_ms(self._h, which, C.byref(ms), C.byref(n)), "wh_last_kernel_ms")
        return ms.value, n.value

    def last_score_launches(self):
        """[(cells per lane, kernel family, ms)] of the scoring launches of the last score call (timing mode)."""
        q = np.zeros(64, dtype=np.int32)
        kind = np.zeros(64, dtype=np.int32)
        ms = np.zeros(64, dtype=np.float64)
        n = lib().wh_last_score_launches(self._h, q.ctypes.data, kind.ctypes.data, ms.ctypes.data, 64)
        check(min(n, 0), "wh_last_score_launches")
        return [(int(q[t]), int(kind[t]), float(ms[t])) for t in range(min(n, 64))]

    # ------------------------------------------------------------------ host (numpy) operators
    def last_score_paths(self):
        """Backward sweeps of the last score call by path: envelope sweeps {"window256", "window512", "window_rejected",
        "full_width"}, multihit sweeps {"p2_window", "p2_window_in_doubt"} (include/witch_hip.h: wh_last_score_paths)."""
        p6 = np.zeros(6, dtype=np.int64)
        check(lib().wh_last_score_paths(self._h, p6.ctypes.data), "wh_last_score_paths")
        return {"window256": int(p6[0]), "window512": int(p6[1]), "window_rejected": int(p6[2]), "full_width": int(p6[3]),
                "p2_window": int(p6[4]), "p2_window_in_doubt": int(p6[5])}

    def last_score_spill_bytes(self) -> int:
        """Bytes of Forward rows the envelope sweeps of the last score call stored (include/witch_hip.h: wh_last_score_counters)."""
        c8 = np.zeros(8, dtype=np.int64)
        check(lib().wh_last_score_counters(self._h, c8.ctypes.data), "wh_last_score_counters")
        return int(c8[6])

    def last_long_list_pairs(self) -> int:
        """Pairs of the last score call that had more regions than a scoring kernel's list holds (WH_MAX_ENVELOPES) and were
        scored again by the long-list pass (include/witch_hip.h: wh_last_score_counters, out8[7])."""
        c8 = np.zeros(8, dtype=np.int64)
        check(lib().wh_last_score_counters(self._h, c8.ctypes.data), "wh_last_score_counters")
        return int(c8[7])

    def set_path_buffer(self, paths_t):
        """Registers a CUDA uint8 tensor of nq x H bytes that later score calls fill with WH_PATH_* bits per pair
        (staged launches only; None switches it off).  The caller keeps the tensor alive (include/witch_hip.h)."""
        check(lib().wh_set_path_buffer(self._h, paths_t.data_ptr() if paths_t is not None else None), "wh_set_path_buffer")
        self._path_t = paths_t

    def last_queue_reruns(self) -> int:
        """Scoring passes the last score call repeated because the resolver's queue overflowed its estimate (0 or 1)."""
        n = lib().wh_last_queue_reruns(self._h)
        check(min(n, 0), "wh_last_queue_reruns")
        return int(n)

    def score(self, residues, offsets, want_fwd=False, want_detail=False):
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        nq = len(offsets) - 1
        deci = np.zeros((nq, self.H), dtype=np.int32)
        flags = np.zeros((nq, self.H), dtype=np.uint8)
        fwd = np.zeros((nq, self.H), dtype=np.float32) if want_fwd else None
        det = (PairDetail * (nq * self.H))() if want_detail else None
        check(lib().wh_score(self._h, residues.ctypes.data, offsets.ctypes.data, nq, deci.ctypes.data,
                             flags.ctypes.data, None if fwd is None else fwd.ctypes.data,
                             None if det is None else C.addressof(det)), "wh_score")
        out = [deci, flags]
        if want_fwd:
            out.append(fwd)
        if want_detail:
            out.append(det)
        return tuple(out)

    def topk(self, decibits, flags, k: int):
        decibits = np.ascontiguousarray(decibits, dtype=np.int32)
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        nq = decibits.shape[0]
        idx = np.full((nq, k), -1, dtype=np.int32)
        w = np.zeros((nq, k), dtype=np.float64)
        nk = np.zeros(nq, dtype=np.int32)
        nu = np.zeros(nq, dtype=np.int32)
        check(lib().wh_topk(self._h, decibits.ctypes.data, flags.ctypes.data, nq, int(k), idx.ctypes.data,
                            w.ctypes.data, nk.ctypes.data, nu.ctypes.data), "wh_topk")
        return idx, w, nk, nu

    def align(self, residues, offsets, pair_q, pair_h):
        """cols (CSR over the residues of each pair) and col_offsets."""
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        pair_q = np.ascontiguousarray(pair_q, dtype=np.int64)
        pair_h = np.ascontiguousarray(pair_h, dtype=np.int32)
        lens = offsets[pair_q + 1] - offsets[pair_q] if len(pair_q) else np.zeros(0, np.int64)
        co = np.zeros(len(pair_q) + 1, dtype=np.int64)
        co[1:] = np.cumsum(lens)
        cols = np.full(int(co[-1]), -1, dtype=np.int32)
        if len(pair_q):
            check(lib().wh_align(self._h, residues.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                 pair_q.ctypes.data, pair_h.ctypes.data, len(pair_q), co.ctypes.data,
                                 cols.ctypes.data), "wh_align")
        return cols, co

    def last_align_status(self):
        """(pairs redone in log space, pair numbers the last align call returned UNALIGNED - all columns -1 -
        because the any-size kernel could not align them; include/witch_hip.h: wh_last_align_status)."""
        nlog, nun = C.c_int64(0), C.c_int64(0)
        check(lib().wh_last_align_status(self._h, C.byref(nlog), C.byref(nun), None, 0), "wh_last_align_status")
        pairs = np.zeros(nun.value, dtype=np.int64)
        if nun.value:
            check(lib().wh_last_align_status(self._h, None, None, pairs.ctypes.data, nun.value), "wh_last_align_status")
        return int(nlog.value), pairs

    def last_align_paths(self):
        """Pairs of the last align call by sweep path: {"window256", "window512", "window_rejected", "full_width"}
        (include/witch_hip.h: wh_last_align_paths)."""
        p4 = np.zeros(4, dtype=np.int64)
        check(lib().wh_last_align_paths(self._h, p4.ctypes.data), "wh_last_align_paths")
        return {"window256": int(p4[0]), "window512": int(p4[3]), "window_rejected": int(p4[1]), "full_width": int(p4[2])}

    def consensus(self, offsets, qpair_off, pair_h, pair_w, col_offsets, cols, retained, nongaps, backbone_length):
        """Weighted consensus DP (wh_consensus).  retained / nongaps: one int array per model."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        qpair_off = np.ascontiguousarray(qpair_off, dtype=np.int64)
        pair_h = np.ascontiguousarray(pair_h, dtype=np.int32)
        pair_w = np.ascontiguousarray(pair_w, dtype=np.float64)
        col_offsets = np.ascontiguousarray(col_offsets, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        ro = np.zeros(self.H + 1, dtype=np.int64)
        ro[1:] = np.cumsum([len(r) for r in retained])
        ret = np.ascontiguousarray(np.concatenate(retained), dtype=np.int32)
        ng = np.ascontiguousarray(np.concatenate(nongaps), dtype=np.int32)
        nq = len(offsets) - 1
        out = np.zeros(int(offsets[-1]), dtype=np.int32)
        mm = np.zeros(2 * nq, dtype=np.int32)
        check(lib().wh_consensus(self._h, offsets.ctypes.data, nq, qpair_off.ctypes.data, pair_h.ctypes.data,
                                 pair_w.ctypes.data, col_offsets.ctypes.data, cols.ctypes.data, ro.ctypes.data,
                                 ret.ctypes.data, ng.ctypes.data, int(backbone_length), out.ctypes.data,
                                 mm.ctypes.data), "wh_consensus")
        return out, mm.reshape(nq, 2)

    # ------------------------------------------------------------------ device (torch) operators
    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def label_to_pos_t(self, labels_t):
        """hmm_index labels (as returned by topk_t) -> model positions 0..H-1, on the device."""
        import torch
        lut = getattr(self, "_lut_t", None)
        if lut is None or lut.device != labels_t.device:
            lut = torch.full((int(self.index.max()) + 1,), -1, dtype=torch.int32, device=labels_t.device)
            lut[torch.from_numpy(self.index.astype(np.int64)).to(labels_t.device)] = torch.arange(
                self.H, dtype=torch.int32, device=labels_t.device)
            self._lut_t = lut
        return lut[labels_t.long()].contiguous()

    def score_t(self, residues_t, offsets_t, max_len: int, want_fwd=False):
        import torch
        nq = offsets_t.numel() - 1
        dev = residues_t.device
        deci = torch.empty((nq, self.H), dtype=torch.int32, device=dev)
        flags = torch.empty((nq, self.H), dtype=torch.uint8, device=dev)
        fwd = torch.empty((nq, self.H), dtype=torch.float32, device=dev) if want_fwd else None
        check(lib().wh_score_dev(self._h, residues_t.data_ptr(), offsets_t.data_ptr(), nq, residues_t.numel(),
                                 int(max_len), deci.data_ptr(), flags.data_ptr(),
                                 None if fwd is None else fwd.data_ptr(), None, self._stream()), "wh_score_dev")
        return (deci, flags, fwd) if want_fwd else (deci, flags)

    def topk_t(self, deci_t, flags_t, k: int):
        import torch
        nq = deci_t.shape[0]
        dev = deci_t.device
        idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
        w = torch.empty((nq, k), dtype=torch.float64, device=dev)
        nk = torch.empty(nq, dtype=torch.int32, device=dev)
        nu = torch.empty(nq, dtype=torch.int32, device=dev)
        check(lib().wh_topk_dev(self._h, deci_t.data_ptr(), flags_t.data_ptr(), nq, int(k), idx.data_ptr(),
                                w.data_ptr(), nk.data_ptr(), nu.data_ptr(), self._stream()), "wh_topk_dev")
        return idx, w, nk, nu

    def align_t(self, residues_t, offsets_t, max_len: int, pair_q_t, pair_h_t, col_offsets_t, total_cols: int):
        import torch
        cols = torch.full((int(total_cols),), -1, dtype=torch.int32, device=residues_t.device)
        npairs = pair_q_t.numel()
        if npairs:
            check(lib().wh_align_dev(self._h, residues_t.data_ptr(), offsets_t.data_ptr(), offsets_t.numel() - 1,
                                     residues_t.numel(), int(max_len), pair_q_t.data_ptr(), pair_h_t.data_ptr(),
                                     npairs, col_offsets_t.data_ptr(), cols.data_ptr(), self._stream()),
                  "wh_align_dev")
        return cols

    def consensus_t(self, offsets_t, max_len: int, qpair_off_t, pair_h_t, pair_w_t, col_offsets_t, cols_t,
                    ret_off_t, retained_t, nongaps_t, backbone_length: int, max_pairs_per_query: int):
        """Weighted consensus DP on device-resident inputs (wh_consensus_dev): per residue the backbone
        column or -1 - (column before which the insertion sits), and (min, max) touched column per query."""
        import torch
        nq = offsets_t.numel() - 1
        dev = offsets_t.device
        out = torch.empty(int(cols_t.numel() and offsets_t[-1].item()), dtype=torch.int32, device=dev)
        mm = torch.empty((nq, 2), dtype=torch.int32, device=dev)
        check(lib().wh_consensus_dev(self._h, offsets_t.data_ptr(), nq, int(max_len), qpair_off_t.data_ptr(),
                                     pair_h_t.data_ptr(), pair_w_t.data_ptr(), col_offsets_t.data_ptr(),
                                     cols_t.data_ptr(), ret_off_t.data_ptr(), retained_t.data_ptr(),
                                     nongaps_t.data_ptr(), int(backbone_length), int(max_pairs_per_query),
                                     out.data_ptr(), mm.data_ptr(), self._stream()), "wh_consensus_dev")
        return out, mm
