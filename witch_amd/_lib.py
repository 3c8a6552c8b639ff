"""ctypes binding of libwitch_hip.so (the C ABI in include/witch_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails,
an exception is raised.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwitch_hip.so")
CSRC = os.path.join(_HERE, "csrc")

WH_MAX_ENVELOPES = 16
FLAG_REPORTED, FLAG_MULTI, FLAG_OVERRIDE, FLAG_TRUNC, FLAG_EXACT = 1, 2, 4, 8, 16
ALPH_DNA, ALPH_RNA, ALPH_AMINO = 0, 1, 2
WH_OK, WH_EINVAL, WH_EIO, WH_ENODEV, WH_EHIP, WH_ERANGE, WH_ENOMEM = 0, -1, -2, -3, -4, -5, -6      # include/witch_hip.h
PATH_P2_WIN, PATH_P2_FULL, PATH_P4_W256, PATH_P4_W512, PATH_P4_WFAIL, PATH_P4_FULL, PATH_DENSE, PATH_MULTI = 1, 2, 4, 8, 16, 32, 64, 128


class WitchHipError(RuntimeError):
    pass


class PairDetail(C.Structure):
    _fields_ = [
        ("fwd_bits", C.c_float), ("seq_score", C.c_float), ("pre_score", C.c_float),
        ("seqbias_nats", C.c_float), ("nregions", C.c_int32), ("nenv", C.c_int32),
        ("env_i", C.c_int32 * WH_MAX_ENVELOPES), ("env_j", C.c_int32 * WH_MAX_ENVELOPES),
        ("envsc", C.c_float * WH_MAX_ENVELOPES), ("domcorr", C.c_float * WH_MAX_ENVELOPES),
    ]


# every symbol include/witch_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "wh_version": (C.c_char_p, []),
    "wh_last_error": (C.c_char_p, []),
    "wh_init": (C.c_int, [C.c_int]),
    "wh_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "wh_digitize": (C.c_int, [C.c_int, C.c_char_p, C.c_int64, _P]),
    "wh_ehmm_load": (_P, [C.POINTER(C.c_char_p), _P, _P, C.c_int]),
    "wh_hmmbuild": (C.c_int, [C.c_char_p, C.c_int32, C.c_int64, C.POINTER(C.c_char_p), C.c_char_p, C.c_double, C.c_double,
                            C.c_double, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "wh_hmmbuild2": (C.c_int, [C.c_char_p, C.c_int32, C.c_int64, C.POINTER(C.c_char_p), C.c_char_p, C.c_double, C.c_double,
                             C.c_double, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "wh_free_text": (None, [C.c_void_p]),
    "wh_merge_sharded": (C.c_int, [C.c_int, _P, _P, C.c_int64, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "wh_merge": (C.c_int, [C.c_int, _P, _P, C.c_int64, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                         C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "wh_ehmm_free": (None, [_P]),
    "wh_ehmm_count": (C.c_int, [_P]),
    "wh_ehmm_alphabet": (C.c_int, [_P]),
    "wh_ehmm_info": (C.c_int, [_P, _P, _P, _P]),
    "wh_ehmm_map": (C.c_int, [_P, C.c_int, _P]),
    "wh_ehmm_max_query_len": (C.c_int, [_P]),
    "wh_score": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, _P]),
    "wh_score_dev": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int32, _P, _P, _P, _P, _P]),
    "wh_topk": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P]),
    "wh_topk_dev": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, _P]),
    "wh_align": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, C.c_int64, _P, _P]),
    "wh_align_dev": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int32, _P, _P, C.c_int64, _P, _P, _P]),
    "wh_consensus": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P]),
    "wh_consensus_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "wh_last_align_status": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P, C.c_int64]),
    "wh_last_align_paths": (C.c_int, [_P, _P]),
    "wh_last_score_paths": (C.c_int, [_P, _P]),
    "wh_last_score_counters": (C.c_int, [_P, _P]),
    "wh_set_path_buffer": (C.c_int, [_P, _P]),
    "wh_last_queue_reruns": (C.c_int, [_P]),
    "wh_last_score_launches": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "wh_last_kernel_ms": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "wh_set_timing": (C.c_int, [_P, C.c_int]),
    "wh_set_option": (C.c_int, [_P, C.c_char_p, C.c_char_p]),
}

_LIB = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise WitchHipError("building libwitch_hip.so failed:\n" + r.stdout[-4000:])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise WitchHipError(
                "libwitch_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C witch_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime; two HIP runtimes in one process cannot
        # both own the GPU.  Importing torch first makes the loader bind this library to the
        # runtime torch already loaded (same SONAME) whenever torch is used in the process.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)      # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().wh_last_error()
        raise WitchHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
