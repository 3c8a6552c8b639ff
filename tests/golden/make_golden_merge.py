#!/usr/bin/env python3
"""Golden vectors for the final transitive merge.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).  Feeds query alignments (data: the
per-query consensus strings already stored in golden.json.gz, and seeded random ones) to the
reference's own mergeAlignmentsCollapsed (witch_msa/gcmm/merger.py:40-131, which drives
ExtendedAlignment.merge_in, helpers/alignment_tools.py:1183-1316) and stores the two FASTA files
it writes (<name>.fasta and <name>.masked.fasta) as data.  No reference source is copied.
"""
import gzip
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def read_fasta_text(path):
    out, name = [], None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            name = line[1:]
            out.append([name, ""])
        elif name is not None:
            out[-1][1] += line
    return out


def reference_merge(Configs, ExtendedAlignment, mergeAlignmentsCollapsed, backbone, queries, renamed):
    """queries: list of (name, string) or 'skipped'.  Labels as aligner.py:486-493 sets them."""
    tmp = tempfile.mkdtemp(prefix="gmerge_")
    bpath = os.path.join(tmp, "backbone.fasta")
    with open(bpath, "w") as f:
        for n, s in backbone:
            f.write(">%s\n%s\n" % (n, s))
    objs = []
    for q in queries:
        if q == 'skipped':
            objs.append('skipped')
            continue
        name, text = q
        ea = ExtendedAlignment([])
        ea[name] = text
        ea._reset_col_names()
        ins, reg = -1, 0
        for i, ch in enumerate(text):
            if ch.islower():
                ea._col_labels[i] = ins
                ins -= 1
            else:
                ea._col_labels[i] = reg
                reg += 1
        objs.append(ea)
    Configs.output_path = os.path.join(tmp, "out.fasta")
    Configs.log_path = None
    Configs.runtime_path = os.path.join(tmp, "runtime.txt")
    mergeAlignmentsCollapsed(bpath, objs, renamed, None)
    full = read_fasta_text(os.path.join(tmp, "out.fasta"))
    masked = read_fasta_text(os.path.join(tmp, "out.masked.fasta"))
    return full, masked


def random_case(seed, B, n_backbone, n_queries, alphabet="ACGT"):
    rng = np.random.default_rng(seed)
    backbone = []
    for i in range(n_backbone):
        s = "".join(rng.choice(list(alphabet + "-"), size=B, p=[0.2] * 4 + [0.2]))
        backbone.append(["bb%03d" % i, s])
    queries = []
    for q in range(n_queries):
        if rng.random() < 0.05:
            queries.append('skipped')
            continue
        lo = int(rng.integers(0, B))
        hi = int(rng.integers(lo, B)) + 1
        parts = []
        def ins(maxn):
            n = int(rng.integers(0, maxn + 1)) if rng.random() < 0.4 else 0
            return "".join(rng.choice(list(alphabet.lower()), size=n)) if n else ""
        parts.append(ins(6) if lo == 0 or rng.random() < 0.5 else "")
        for c in range(B):
            if c < lo or c >= hi:
                parts.append("-")
            else:
                parts.append(str(rng.choice(list(alphabet))) if rng.random() < 0.85 else "-")
                if c + 1 < hi:
                    parts.append(ins(4) if rng.random() < 0.15 else "")
        parts.append(ins(7))
        queries.append(["q%04d" % q, "".join(parts)])
    return backbone, queries


def main():
    scratch, Configs, *_ = mg.import_reference()
    from witch_msa.helpers.alignment_tools import ExtendedAlignment
    from witch_msa.gcmm.merger import mergeAlignmentsCollapsed
    Configs.log = staticmethod(lambda *a, **k: None)
    Configs.runtime = staticmethod(lambda *a, **k: None)
    out = {}
    # (1) the reference's example backbone (first 12 rows) + the 40 consensus strings of the golden case
    g = json.load(gzip.open(os.path.join(HERE, "example_ehmm", "golden.json.gz"), "rt"))
    rows = []
    with gzip.open(os.path.join(mg.REF, "examples/data/backbone.aln.fasta.gz"), "rt") as f:
        name = None
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if len(rows) == 12:
                    break
                rows.append([line[1:], ""])
            else:
                rows[-1][1] += line
    rows = [[n, s.upper()] for n, s in rows[:12]]
    queries = [[qn, g["merged"][qn]] for qn in g["queries"] if g["merged"].get(qn)]
    full, masked = reference_merge(Configs, ExtendedAlignment, mergeAlignmentsCollapsed, rows, queries, {})
    out["example_ehmm"] = {"backbone": rows, "queries": queries, "renamed": {}, "full": full, "masked": masked}
    # (2) seeded random alignments: insertions at both ends, empty spans, skipped queries, renamed taxa
    for seed, B, nb, nq in [(1, 40, 5, 60), (2, 7, 2, 30), (3, 120, 3, 200)]:
        backbone, queries = random_case(seed, B, nb, nq)
        renamed = {}
        if seed == 3:
            # gcmm.py renames taxa with illegal characters: {original: renamed}; the merger maps back
            for q in queries[:5]:
                if q != 'skipped':
                    renamed["orig|" + q[0]] = q[0]
        full, masked = reference_merge(Configs, ExtendedAlignment, mergeAlignmentsCollapsed, backbone, queries, renamed)
        out["random_%d" % seed] = {"backbone": backbone, "queries": queries, "renamed": renamed,
                                   "full": full, "masked": masked}
    with gzip.open(os.path.join(HERE, "final_merge.json.gz"), "wt") as f:
        json.dump(out, f, separators=(",", ":"))
    for k, v in out.items():
        print(k, len(v["backbone"]), "backbone rows,", len(v["queries"]), "queries ->", len(v["full"]), "rows x",
              len(v["full"][0][1]), "columns; masked", len(v["masked"][0][1]))


if __name__ == "__main__":
    main()
