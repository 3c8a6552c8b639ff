#!/usr/bin/env python3
"""Host time of the eHMM construction (wh_hmmbuild): the 15 subsets of the reference's example
backbone and the 200 subsets of the headline family, with 1 and 8 threads.  No GPU needed.
usage: tools/bench_hmmbuild.py [--reference]   (--reference also times HMMER's hmmbuild; build container only)"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from witch_amd import synth  # noqa: E402
from witch_amd.gcmm.hmmbuild import build_ehmm, hmmbuild_text  # noqa: E402


def example():
    names, rows = [], []
    with gzip.open(os.path.join(ROOT, "tests/golden/example_e2e/backbone.fasta.gz"), "rt") as fh:
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:].split()[0]); rows.append("")
            elif line:
                rows[-1] += line
    return names, rows, synth.bfs_subsets(len(rows), 15)


def headline():
    import bench
    alph, seed, root_len, leaves, sub, indel, n_hmms, nq, qlen, k = bench.WORKLOADS["dna_100k_x200"]
    fam = synth.make_family(seed, root_len, leaves, alph, sub, indel)
    sym = synth.symbols(alph) + "-"
    rows = []
    for i in range(leaves):
        r = fam.msa[i].astype(np.int64).copy(); r[r < 0] = len(sym) - 1
        rows.append("".join(sym[int(x)] for x in r))
    return list(fam.names), rows, synth.bfs_subsets(leaves, n_hmms)


hmmbuild_text(["ACGT", "ACGA"], "dna")          # load the library once
for tag, (names, rows, subs) in (("example backbone (500 x 2574)", example()), ("headline family", headline())):
    subsets = [("A_0_%d" % i, list(range(lo, hi))) for i, (lo, hi) in enumerate(subs)]
    for th in (1, 8):
        d = tempfile.mkdtemp(prefix="wh_hb_")
        t0 = time.time()
        out = build_ehmm(names, rows, subsets, "dna", d, threads=th)
        dt = time.time() - t0
        print("%-32s %3d subsets (%d..%d sequences, %d columns): %6.3f s with %d thread(s) = %.1f ms per model"
              % (tag, len(subsets), min(hi - lo for lo, hi in subs), max(hi - lo for lo, hi in subs), len(rows[0]), dt, th, 1e3 * dt / len(subsets)))
    if "--reference" in sys.argv:
        hb = "/root/reference/witch_msa/tools/magus/tools/hmmer/hmmbuild"
        t0 = time.time()
        n = 0
        for label, idx in subsets[:15]:
            fa = os.path.join(d, label, "hmmbuild.input.%s.fasta" % label)
            subprocess.run([hb, "--cpu", "1", "--dna", "--ere", "0.59", "--symfrac", "0.0", "--informat", "afa", "-o", "/dev/null",
                            os.path.join(d, "ref.hmm"), fa], check=True, stdout=subprocess.DEVNULL)
            n += 1
        dt = time.time() - t0
        print("%-32s HMMER hmmbuild, first %d subsets, one process each, serial: %6.3f s = %.1f ms per model" % (tag, n, dt, 1e3 * dt / n))
