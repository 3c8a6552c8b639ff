#!/usr/bin/env python3
"""Randomised check of wh_hmmbuild against the reference's bundled hmmbuild 3.1b2 - BUILD CONTAINER ONLY (needs
/root/reference; CPU, no GPU): random alignments (DNA / RNA / protein, 1-40 sequences, 8-400 columns, gaps, fragments,
lower case, degenerate residues, '.' gaps, duplicated rows) through both with the reference's command line
(witch_msa/gcmm/algorithm.py:463-470); every line of the model file except NAME / DATE must be identical - MAXL and
the three STATS LOCAL lines included.  usage: tools/fuzz_hmmbuild.py [first_seed] [n_seeds]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from witch_amd.gcmm.hmmbuild import hmmbuild_text  # noqa: E402

HMMER = "/root/reference/witch_msa/tools/magus/tools/hmmer"
ALPH = {"dna": ("ACGT", "RYMKSWHBVDN"), "rna": ("ACGU", "RYMKSWHBVDN"), "amino": ("ACDEFGHIKLMNPQRSTVWY", "BJZOUX")}


def random_alignment(rng, mol):
    can, deg = ALPH[mol]
    n = int(rng.choice([1, 2, 3, 5, 8, 16, 40]))
    L = int(rng.choice([8, 20, 45, 90, 200, 400]))
    sub = float(rng.choice([0.02, 0.1, 0.3]))
    root = rng.integers(0, len(can), size=L)
    rows = []
    for i in range(n):
        s = root.copy()
        m = rng.random(L) < sub
        s[m] = rng.integers(0, len(can), size=int(m.sum()))
        r = [can[int(x)] for x in s]
        for p in np.flatnonzero(rng.random(L) < rng.choice([0.0, 0.03, 0.15])):     # interior gaps
            r[p] = "-" if rng.random() < 0.8 else "."
        if rng.random() < 0.3 and L > 10:                                             # fragment
            lo = int(rng.integers(0, L // 2)); hi = int(rng.integers(lo + 2, L + 1))
            r = ["-"] * lo + r[lo:hi] + ["-"] * (L - hi)
        for p in np.flatnonzero(rng.random(L) < 0.02):                               # degenerate residues
            if r[p] not in "-.":
                r[p] = deg[int(rng.integers(len(deg)))]
        if rng.random() < 0.2:
            r = [c.lower() for c in r]
        rows.append("".join(r))
    if n > 2 and rng.random() < 0.3:
        rows[1] = rows[0]                                                             # duplicated sequence
    if rng.random() < 0.3:
        c = int(rng.integers(L))
        rows = [r[:c] + "-" + r[c + 1:] for r in rows]                                # an all-gap column
    if all(set(r) <= set("-.") for r in rows):
        rows[0] = can[0] * L
    return rows


def body(text):
    return [l.rstrip() for l in text.splitlines() if not l.startswith(("NAME", "DATE"))]


first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
wd = tempfile.mkdtemp(prefix="fuzz_hmmbuild_")
nbad = nrun = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    mol = ["dna", "rna", "amino"][seed % 3]
    rows = random_alignment(rng, mol)
    afa, hmm = os.path.join(wd, "a.afa"), os.path.join(wd, "a.hmm")
    with open(afa, "w") as f:
        for i, r in enumerate(rows):
            f.write(">s%d\n%s\n" % (i, r))
    r = subprocess.run([HMMER + "/hmmbuild", "--cpu", "1", "--" + mol, "--ere", "0.59", "--symfrac", "0.0", "--informat", "afa", "-o", "/dev/null", hmm, afa],
                       capture_output=True, text=True)
    try:
        mine, M, _ = hmmbuild_text(rows, mol, "a", stats=True)
        mine_err = None
    except Exception as ex:
        mine, mine_err = None, str(ex)
    if r.returncode != 0:
        ok = mine is None          # both refuse
        what = "hmmbuild refused (%s); here: %s" % (r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?", mine_err or "accepted")
    elif mine is None:
        ok, what = False, "refused here (%s) but accepted by hmmbuild" % mine_err
    else:
        a, b = body(mine), body(open(hmm).read())
        ok = a == b
        what = "" if ok else "first differing line: %r vs %r" % next(((x, y) for x, y in zip(a, b) if x != y), (len(a), len(b)))
    nrun += 1
    if not ok:
        nbad += 1
        print("MISMATCH seed", seed, mol, "nseq", len(rows), "alen", len(rows[0]), what, flush=True)
print("%d alignments, %d differ" % (nrun, nbad))
sys.exit(1 if nbad else 0)
