#!/usr/bin/env python3
"""Randomised pinning of the ORACLE against HMMER itself - BUILD CONTAINER ONLY (needs /root/reference; CPU, no GPU):
random alignments -> the bundled hmmbuild (the reference's command line, algorithm.py:463-470) -> the alignment's rows,
fragments of them and unrelated sequences through the bundled `hmmsearch --cpu 1 --noali -E 99999999 --max`
(algorithm.py:526-532) and `hmmalign` (aligner.py:98-100), compared with oracle/p7_oracle.c: the printed "%6.1f" score
as deci-bits, the reported set, and the aligned columns decoded from the Stockholm row the way the reference does
(aligner.py:126-142: upper case or '-' = match column, lower case = insert).  The golden vectors under tests/golden pin
the oracle on fixed cases; this counts agreement on random ones.  usage: tests/tools/fuzz_oracle_vs_hmmer.py [first_seed] [n]"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

HMMER = "/root/reference/witch_msa/tools/magus/tools/hmmer"
src = open(os.path.join(ROOT, "tools", "fuzz_hmmbuild.py")).read().split("first = int(sys.argv[1])")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_hmmbuild.py")}
exec(src, ns)
random_alignment, ALPH = ns["random_alignment"], ns["ALPH"]


def hmmsearch_scores(hmm, fa, wd):
    out = os.path.join(wd, "s.out")
    subprocess.run([HMMER + "/hmmsearch", "--cpu", "1", "--noali", "-E", "99999999", "-o", out, "--max", hmm, fa], check=True)
    res, started = {}, False
    for line in open(out):
        s = line.strip()
        if not started and s.startswith("E-value"):
            started = True
        elif started and s == "":
            break
        elif started and "--" not in s and len(s.split()) >= 9:
            f = s.split()
            res[f[8]] = int(round(float(f[1]) * 10))
    return res


def hmmalign_columns(hmm, name, text, wd):
    fa, out = os.path.join(wd, "one.fa"), os.path.join(wd, "one.sto")
    open(fa, "w").write(">%s\n%s\n" % (name, text))
    r = subprocess.run([HMMER + "/hmmalign", "-o", out, hmm, fa], capture_output=True)
    if r.returncode != 0:
        return None
    row = "".join(l.split()[1] for l in open(out) if l.split() and l.split()[0] == name)
    cols, c = [], 0
    for ch in row:
        if ch == ".":
            continue
        if ch == "-":
            c += 1
        elif ch.isupper():
            cols.append(c); c += 1
        else:
            cols.append(-1)
    return np.array(cols, dtype=np.int64)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = dict(pairs=0, reported_equal=0, score_equal=0, score_off1=0, score_other=0, multidomain=0, aligned=0, align_equal=0)
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    mol = ["dna", "rna", "amino"][seed % 3]
    wd = tempfile.mkdtemp(prefix="fuzz_orc_")
    rows = random_alignment(rng, mol)
    afa, hmm = os.path.join(wd, "a.afa"), os.path.join(wd, "a.hmm")
    open(afa, "w").write("".join(">s%d\n%s\n" % (i, r) for i, r in enumerate(rows)))
    if subprocess.run([HMMER + "/hmmbuild", "--cpu", "1", "--" + mol, "--ere", "0.59", "--symfrac", "0.0", "--informat", "afa", "-o", "/dev/null", hmm, afa],
                      capture_output=True).returncode != 0:
        continue
    can = ALPH[mol][0]
    texts = []
    for r in rows[:8]:
        t = "".join(c for c in r.upper() if c in can)
        if t:
            texts.append(t)
            if len(t) > 6:
                lo = int(rng.integers(0, len(t) // 2))
                texts.append(t[lo:lo + max(3, len(t) // 2)])
    if texts and len(texts[0]) > 10:
        texts.append(texts[0] + "".join(can[int(x)] for x in rng.integers(0, len(can), size=30)) + texts[0])     # two copies
    texts += ["".join(can[int(x)] for x in rng.integers(0, len(can), size=int(rng.integers(5, 150)))) for _ in range(3)]
    names = ["q%d" % i for i in range(len(texts))]
    fa = os.path.join(wd, "q.fa")
    open(fa, "w").write("".join(">%s\n%s\n" % (a, b) for a, b in zip(names, texts)))
    got = hmmsearch_scores(hmm, fa, wd)
    oh = orc.OracleHMM(hmm)
    for qn, t in zip(names, texts):
        d = oh.digitize(t)
        r = oh.score(d)
        rep = bool(r.flags & 1)
        tot["pairs"] += 1
        tot["multidomain"] += int(bool(r.flags & 2))
        if rep == (qn in got):
            tot["reported_equal"] += 1
            if rep:
                dd = abs(int(r.decibits) - got[qn])
                tot["score_equal" if dd == 0 else "score_off1" if dd == 1 else "score_other"] += 1
                if dd > 1:
                    print("SCORE seed", seed, qn, "oracle", r.decibits, "hmmsearch", got[qn], "flags", r.flags, "L", len(t), flush=True)
        else:
            print("REPORTED seed", seed, qn, "oracle", rep, "hmmsearch", qn in got, "flags", r.flags, "L", len(t), flush=True)
        want = hmmalign_columns(hmm, qn, t, wd)
        if want is not None:
            tot["aligned"] += 1
            mine = oh.align(d)
            if np.array_equal(mine, want):
                tot["align_equal"] += 1
            else:
                print("ALIGN seed", seed, qn, "L", len(t), "differing residues", int(np.sum(mine != want)), flush=True)
    print("seed", seed, mol, "M", oh.M, tot, flush=True)
print(tot)
