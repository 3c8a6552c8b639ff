#!/usr/bin/env python3
"""STATS LOCAL lines of wh_hmmbuild2(..., WH_BUILD_STATS) against every golden model file written by hmmbuild 3.1b2
(20 edge cases, 12 family models, the 15 models of the reference's example backbone: 1 278 .. 2 574 nodes).
The CPU tests check 37 of the 47 (the ten largest are left out there for time); this prints all of them.
No GPU needed.  usage: tools/check_stats_lines.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hmmbuild_host as T  # noqa: E402
from witch_amd import synth  # noqa: E402
from witch_amd.gcmm.hmmbuild import hmmbuild_text  # noqa: E402

bad = 0


def cmp(text, gold, what, dt):
    global bad
    g, m = T.stats_lines(gold), T.stats_lines(text)
    same = g == m and len(g) == 3
    bad += 0 if same else 1
    print("%-32s %6.2f s  %s" % (what, dt, "identical" if same else "DIFFERENT\n   hmmbuild %s\n   here     %s" % (g, m)), flush=True)


d = os.path.join(T.GOLD, "hmmbuild_cases")
for case in T.CASES:
    rows = [l.strip() for l in open(os.path.join(d, case + ".afa")) if not l.startswith(">")]
    gold = T.read(os.path.join(d, case + ".hmm"))
    mol = {"DNA": "dna", "RNA": "rna", "amino": "amino"}[[l.split()[1] for l in gold.splitlines() if l.startswith("ALPH")][0]]
    t0 = time.time()
    text, _, _ = hmmbuild_text(rows, mol, case, stats=True)
    cmp(text, gold, case, time.time() - t0)
for case, args, mol in (("dna_hmmbuild", ("dna", 11, 120, 32, 8, 0.04, 0.004), "dna"), ("amino_hmmbuild", ("amino", 13, 90, 16, 4, 0.08, 0.004), "amino")):
    rows, subs = T.family_rows(*args)
    for idx, (lo, hi) in enumerate(subs):
        t0 = time.time()
        text, _, _ = hmmbuild_text(rows[lo:hi], mol, "sub", stats=True)
        cmp(text, T.read(os.path.join(T.GOLD, case, "hmms", "A_0_%d.hmm" % idx)), "%s/%d" % (case, idx), time.time() - t0)
names, rows = T.backbone_rows()
for idx, (lo, hi) in enumerate(synth.bfs_subsets(len(rows), 15)):
    t0 = time.time()
    text, M, _ = hmmbuild_text([r.upper() for r in rows[lo:hi]], "dna", "sub", stats=True)
    cmp(text, T.read(os.path.join(T.GOLD, "example_e2e", "hmms", "A_0_%d.hmm.gz" % idx)), "example/%d (%d nodes)" % (idx, M), time.time() - t0)
print("%d of 47 files differ" % bad)
sys.exit(1 if bad else 0)
