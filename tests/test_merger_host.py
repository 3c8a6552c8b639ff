"""Final transitive merge: the closed-form merger must write the same
two FASTA files as the reference's sequential merge (witch_msa/gcmm/merger.py:40-131 over
ExtendedAlignment.merge_in).  Golden vectors: tests/golden/make_golden_merge.py ran the reference's
own function on the example backbone with the golden consensus strings and on seeded random
alignments (insertions at both ends, skipped queries, renamed taxa)."""
import gzip
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    with gzip.open(os.path.join(HERE, "golden", "final_merge.json.gz"), "rt") as f:
        return json.load(f)


def _read(path):
    rows, name = [], None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            name = line[1:]
            rows.append([name, ""])
        else:
            rows[-1][1] += line
    return rows


@pytest.mark.parametrize("case", ["example_ehmm", "random_1", "random_2", "random_3"])
def test_merge_equals_reference_files(case, tmp_path):
    from witch_amd.gcmm.merger import mergeAlignmentsCollapsed, masked_path
    from witch_amd.gcmm.merge import QueryAlignment
    g = _cases()[case]
    bpath = tmp_path / "backbone.fasta"
    with open(bpath, "w") as f:
        for n, s in g["backbone"]:
            f.write(">%s\n%s\n" % (n, s.lower() if case == "random_2" else s))   # the reader upper-cases
    queries = []
    for q in g["queries"]:
        if q == "skipped":
            queries.append("skipped")
            continue
        qa = QueryAlignment()
        qa[q[0]] = q[1]
        queries.append(qa)
    queries.insert(1, QueryAlignment())          # a failed query: empty alignment, ignored by merge_in
    out = str(tmp_path / "result.fasta")
    o, m = mergeAlignmentsCollapsed(str(bpath), queries, g["renamed"], None, output_path=out)
    assert m == masked_path(out) == str(tmp_path / "result.masked.fasta")
    assert _read(o) == g["full"]
    assert _read(m) == g["masked"]


def test_masked_path_rule():
    from witch_amd.gcmm.merger import masked_path
    assert masked_path("a/b.fasta") == "a/b.masked.fasta"
    assert masked_path("x.fa") == "x.masked.fa"
    assert masked_path("x.aln") == "x.aln.masked.fasta"


def test_merger_errors():
    from witch_amd.gcmm.merger import mergeAlignmentsCollapsed, merge_collapsed
    with pytest.raises(SystemExit):
        mergeAlignmentsCollapsed("unused", [], {}, None, output_path="unused")
    with pytest.raises(ValueError):
        merge_collapsed({"a": "AC-T"}, [{"q": "ACT"}])          # a query that lacks a backbone column


def test_merge_scales_linearly(tmp_path):
    """20k queries x 300 columns merge in seconds (the sequential splice is quadratic)."""
    import time
    import numpy as np
    from witch_amd.gcmm.merger import merge_collapsed
    rng = np.random.default_rng(5)
    B = 300
    backbone = {"b%d" % i: "".join(rng.choice(list("ACGT-"), size=B)) for i in range(20)}
    qs = []
    for q in range(20000):
        s = list(rng.choice(list("ACGT-"), size=B))
        g = int(rng.integers(0, B))
        s.insert(g, "acg"[: int(rng.integers(0, 4))])
        qs.append({"q%d" % q: "".join(s)})
    t0 = time.time()
    names, mat, col_pos = merge_collapsed(backbone, qs)
    dt = time.time() - t0
    assert mat.shape[0] == 20020 and len(col_pos) == B and dt < 30, dt
