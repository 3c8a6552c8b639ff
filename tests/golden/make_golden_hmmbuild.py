#!/usr/bin/env python3
"""Edge-case fixtures for the hmmbuild equivalent.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference): small hand-shaped alignments through the reference's
bundled hmmbuild 3.1b2 with the reference's command line (witch_msa/gcmm/algorithm.py:463-470).  Stores ONLY
data under tests/golden/hmmbuild_cases/: <case>.afa (the input alignment) and <case>.hmm (hmmbuild's output,
DATE line dropped).  The cases probe what the big golden models do not: fragments (first..last residue span
below / at / above half the alignment), degenerate residues, lower case, '.' gaps, U in DNA, RNA, amino with
B/Z/X, one sequence, duplicated sequences, all-gap columns, and entropy weighting on a conserved family.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HMMER = "/root/reference/witch_msa/tools/magus/tools/hmmer"
OUT = os.path.join(HERE, "hmmbuild_cases")


def hmmbuild(mol, hmm, afa):
    subprocess.run([HMMER + "/hmmbuild", "--cpu", "1", "--" + mol, "--ere", "0.59", "--symfrac", "0.0", "--informat", "afa",
                    "-o", "/dev/null", hmm, afa], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = [l for l in open(hmm) if not l.startswith("DATE")]
    open(hmm, "w").writelines(lines)


def family(rng, alphabet, n, L, sub):
    root = rng.integers(0, len(alphabet), size=L)
    rows = []
    for _ in range(n):
        s = root.copy()
        m = rng.random(L) < sub
        s[m] = rng.integers(0, len(alphabet), size=int(m.sum()))
        rows.append([alphabet[int(x)] for x in s])
    return rows


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20251301)
    cases = {}
    # 1. fragments: alen 40; spans 19 (fragment), 20 (exactly half: not < 0.5 alen), 21, full
    rows = family(rng, "ACGT", 6, 40, 0.1)
    def span(row, lo, hi):
        return ["-"] * lo + row[lo:hi] + ["-"] * (40 - hi)
    rows[1] = span(rows[1], 5, 24)       # span 19
    rows[2] = span(rows[2], 10, 30)      # span 20
    rows[3] = span(rows[3], 0, 21)       # span 21
    rows[4] = span(rows[4], 30, 40)      # span 10, at the right end
    rows[5][7] = "-"; rows[5][8] = "-"   # interior gaps of a full-length sequence
    cases["dna_fragments"] = ("dna", rows)
    # 2. degenerate residues, lower case, '.' gaps, U, an all-gap column, N runs
    rows = family(rng, "ACGT", 5, 30, 0.15)
    rows[0][3] = "N"; rows[1][3] = "R"; rows[2][4] = "y"; rows[3][5] = "U"; rows[4][6] = "n"
    for r in rows:
        r[12] = "-"
    rows[0][13] = "."; rows[1][13] = "."
    rows[2] = [c.lower() for c in rows[2]]
    rows[4][20:25] = list("NNNNN")
    cases["dna_degenerate"] = ("dna", rows)
    # 3. RNA
    rows = family(rng, "ACGU", 7, 35, 0.2)
    rows[1][0:4] = list("----"); rows[6][30:35] = list("-----")
    cases["rna_small"] = ("rna", rows)
    # 4. amino with B/Z/X, a fragment and gappy columns
    aa = "ACDEFGHIKLMNPQRSTVWY"
    rows = family(rng, aa, 8, 45, 0.25)
    rows[0][2] = "X"; rows[1][2] = "B"; rows[2][9] = "Z"; rows[3][9] = "x"
    rows[4] = ["-"] * 30 + rows[4][30:45]
    for i in (5, 6, 7):
        rows[i][17] = "-"; rows[i][18] = "-"
    cases["amino_mixed"] = ("amino", rows)
    # 5. one sequence; identical sequences
    cases["dna_single"] = ("dna", family(rng, "ACGT", 1, 25, 0.0))
    r = family(rng, "ACGT", 1, 25, 0.0)[0]
    cases["dna_identical"] = ("dna", [list(r) for _ in range(4)])
    # 6. a conserved family (entropy weighting pulls Neff far below nseq) with indel structure
    rows = family(rng, "ACGT", 40, 80, 0.03)
    for i in range(0, 40, 3):
        a = int(rng.integers(5, 60)); rows[i][a:a + 4] = list("----")
    for i in range(1, 40, 7):
        rows[i][0:6] = list("------")
    cases["dna_conserved"] = ("dna", rows)
    # 7. only fragments cover the first columns (their leading gaps are missing data; no sequence has a gap there)
    rows = family(rng, "ACGT", 5, 50, 0.1)
    for i in (0, 1, 2):
        rows[i][0:8] = list("--------")
    rows[3] = ["-"] * 2 + rows[3][2:20] + ["-"] * 30      # fragment starting at column 3
    rows[4] = ["-"] * 35 + rows[4][35:50]                 # fragment at the right end
    cases["dna_fragment_edges"] = ("dna", rows)
    # 8. randomised alignments: random gap runs, fragments, degenerate codes and case, all three alphabets
    deg = {"dna": "RYMKSWHBVDN", "rna": "RYMKSWHBVDN", "amino": "BJZOUX"}
    alpha = {"dna": "ACGT", "rna": "ACGU", "amino": aa}
    for t in range(12):
        mol = ("dna", "rna", "amino")[t % 3]
        n, L = int(rng.integers(2, 24)), int(rng.integers(12, 70))
        rows = family(rng, alpha[mol], n, L, float(rng.uniform(0.02, 0.4)))
        for r in rows:
            for _ in range(int(rng.integers(0, 4))):                    # interior gap runs
                a = int(rng.integers(0, L)); b = min(L, a + int(rng.integers(1, 6)))
                r[a:b] = ["-"] * (b - a)
            if rng.random() < 0.35:                                      # fragment: keep a random window
                a = int(rng.integers(0, L - 3)); b = min(L, a + int(rng.integers(3, L)))
                r[:a] = ["-"] * a; r[b:] = ["-"] * (L - b)
            for _ in range(int(rng.integers(0, 3))):
                r[int(rng.integers(0, L))] = deg[mol][int(rng.integers(len(deg[mol])))]
            if rng.random() < 0.3:
                r[:] = [c.lower() for c in r]
        if all(c == "-" for r in rows for c in r[:1]):
            rows[0][0] = alpha[mol][0]
        cases["random_%02d_%s" % (t, mol)] = (mol, rows)
    for name, (mol, rows) in cases.items():
        afa = os.path.join(OUT, name + ".afa")
        with open(afa, "w") as f:
            for i, r in enumerate(rows):
                f.write(">s%d\n%s\n" % (i, "".join(r)))
        hmmbuild(mol, os.path.join(OUT, name + ".hmm"), afa)
        head = {l.split()[0]: l.split()[1] for l in open(os.path.join(OUT, name + ".hmm")) if l.split() and l.split()[0] in ("LENG", "NSEQ", "EFFN", "ALPH")}
        print(name, mol, head)


if __name__ == "__main__":
    main()
