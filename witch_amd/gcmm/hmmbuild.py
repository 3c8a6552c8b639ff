"""eHMM construction: the reference's `subset_alignment_and_hmmbuild`
(witch_msa/gcmm/algorithm.py:394-477) without the hmmbuild process - the model comes from wh_hmmbuild in
libwitch_hip.so (witch_amd/csrc/wh_build.cpp, host code; HMMER 3.1b2's algorithm for the reference's exact
command line `hmmbuild --cpu 1 --<molecule> --ere 0.59 --symfrac 0.0 --informat afa`).

The returned tuples are the reference's: the retained (not all-gap) backbone columns of the subset and the
non-gap count of every backbone column (algorithm.py:423-429, 476-477).
"""
import ctypes as C
import os

import numpy as np

from .._lib import lib, check

_MOLECULES = {"dna": b"dna", "rna": b"rna", "amino": b"amino"}


def hmmbuild_text(rows, molecule="dna", name="sub", ere=0.59, symfrac=0.0, fragthresh=0.5, stats=False):
    """rows: aligned sequences (str or bytes, equal length).  Returns (HMMER3/f text, M, Neff).
    stats=True adds hmmbuild's three STATS LOCAL lines (E-value calibration; include/witch_hip.h: WH_BUILD_STATS):
    only stock HMMER needs them, this path never reads them."""
    if molecule not in _MOLECULES:
        raise ValueError("molecule must be dna, rna or amino")
    rows = [r.encode("ascii") if isinstance(r, str) else bytes(r) for r in rows]
    if not rows:
        raise ValueError("empty alignment")
    alen = len(rows[0])
    if any(len(r) != alen for r in rows):
        raise ValueError("rows of an alignment must have equal length")
    arr = (C.c_char_p * len(rows))(*rows)
    text, n, M, neff = C.c_void_p(), C.c_int64(0), C.c_int32(0), C.c_double(0.0)
    check(lib().wh_hmmbuild2(_MOLECULES[molecule], len(rows), alen, arr, name.encode(), ere, symfrac, fragthresh,
                             1 if stats else 0, C.byref(text), C.byref(n), C.byref(M), C.byref(neff)), "wh_hmmbuild")
    try:
        out = C.string_at(text.value, n.value).decode("ascii")
    finally:
        lib().wh_free_text(text)
    return out, int(M.value), float(neff.value)


def subset_alignment_and_hmmbuild(names, rows, molecule, outdirprefix, label, ere=0.59, symfrac=0.0):
    """The reference's per-subset step (algorithm.py:394-477) for the backbone rows of ONE subset (names, rows:
    the subset's taxa and their backbone rows; the reference upper-cases sequences when it reads the backbone):
    all-gap columns are deleted first, the reduced alignment is written to
    <outdirprefix>/<label>/hmmbuild.input.<label>.fasta, the model built from it to
    <outdirprefix>/<label>/hmmbuild.model.<label>.  Returns the reference's tuple
    (model_path, label, retained_columns, nongaps_per_column): the backbone columns that survive, and the
    non-gap count of every surviving column."""
    d = os.path.join(outdirprefix, label)
    os.makedirs(d, exist_ok=True)
    ax = np.stack([np.frombuffer((r.upper().encode("ascii") if isinstance(r, str) else bytes(r).upper()), dtype=np.uint8)
                   for r in rows])
    nongap = ax != ord("-")
    keep = np.nonzero(nongap.any(axis=0))[0]
    retained_columns = tuple(int(x) for x in keep)
    nongaps_per_column = tuple(int(x) for x in nongap[:, keep].sum(axis=0))
    reduced = [r.tobytes() for r in ax[:, keep]]
    with open(os.path.join(d, "hmmbuild.input.%s.fasta" % label), "w") as f:
        for n, r in zip(names, reduced):
            f.write(">%s\n%s\n" % (n, r.decode("ascii")))
    text, _, _ = hmmbuild_text(reduced, molecule, "hmmbuild.input.%s" % label, ere=ere, symfrac=symfrac)
    path = os.path.join(d, "hmmbuild.model.%s" % label)
    with open(path, "w") as f:
        f.write(text)
    return path, label, retained_columns, nongaps_per_column


def build_ehmm(names, rows, subsets, molecule, outdirprefix, threads=8, ere=0.59, symfrac=0.0):
    """All models of an eHMM: `subsets` is a list of (label, row indices) over the backbone rows.  Returns the
    list of the reference's tuples in subset order.  wh_hmmbuild runs outside the GIL, so a thread pool scales
    with the host cores (the reference starts one hmmbuild process per subset, algorithm.py:152-154)."""
    from concurrent.futures import ThreadPoolExecutor

    def one(item):
        label, idx = item
        return subset_alignment_and_hmmbuild([names[i] for i in idx], [rows[i] for i in idx], molecule, outdirprefix,
                                             label, ere=ere, symfrac=symfrac)
    if threads <= 1 or len(subsets) <= 1:
        return [one(s) for s in subsets]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(one, subsets))
