"""The hmmbuild equivalent (witch_amd/csrc/wh_build.cpp) against HMMER 3.1b2.

The expected files were written by the reference's bundled hmmbuild with the reference's command line
(witch_msa/gcmm/algorithm.py:463-470): the golden models of the scoring tests (tests/golden/make_golden*.py)
and hand-shaped edge cases (tests/golden/make_golden_hmmbuild.py).  The bar is TEXT identity of every line
except NAME and DATE - MAXL (nucleotide models; hmmbuild's bound on the emitted length) included.  The three STATS LOCAL
lines (E-value calibration by simulation, witch_amd/csrc/wh_calibrate.h) are optional in the C ABI
(WH_BUILD_STATS) and, when asked for, text-identical too: their own tests below.  No GPU is needed: the builder
is host code behind the C ABI.
"""
import gzip
import json
import os

import numpy as np
import pytest

from witch_amd import synth
from witch_amd.gcmm.hmmbuild import hmmbuild_text, subset_alignment_and_hmmbuild, build_ehmm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SKIP = ("NAME", "DATE", "STATS")


def body(text):
    return [l for l in text.splitlines() if not l.startswith(SKIP)]


def read(path):
    return (gzip.open(path, "rt") if path.endswith(".gz") else open(path)).read()


def backbone_rows():
    names, rows = [], []
    with gzip.open(os.path.join(GOLD, "example_e2e", "backbone.fasta.gz"), "rt") as fh:
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                names.append(line[1:].split()[0])
                rows.append("")
            elif line:
                rows[-1] += line
    return names, rows


def family_rows(alphabet, seed, root_len, n_leaves, n_sub, sub_rate, indel_rate):
    """The alignments tests/golden/make_golden.py::family_case handed to hmmbuild (seeded, reproducible)."""
    fam = synth.make_family(seed, root_len, n_leaves, alphabet, sub_rate, indel_rate)
    sym = synth.symbols(alphabet) + "-"
    rows = []
    for i in range(n_leaves):
        r = fam.msa[i].astype(np.int64).copy()
        r[r < 0] = len(sym) - 1
        rows.append("".join(sym[int(x)] for x in r))
    return rows, synth.bfs_subsets(n_leaves, n_sub)


def stats_lines(text):
    return [l.rstrip() for l in text.splitlines() if l.startswith("STATS")]


def assert_same(text, gold_text, what):
    a, b = body(text), body(gold_text)
    assert len(a) == len(b), (what, len(a), len(b))
    bad = [(x, y) for x, y in zip(a, b) if x != y]
    assert not bad, (what, len(bad), bad[:3])


CASES = sorted(f[:-4] for f in os.listdir(os.path.join(GOLD, "hmmbuild_cases")) if f.endswith(".afa"))


@pytest.mark.parametrize("case", CASES)
def test_edge_cases_against_hmmbuild(case):
    """Fragments (span below / at / above half the alignment, at both ends), degenerate residues, lower case,
    '.' gaps, U in DNA, RNA, amino with B/Z/X, one sequence, identical sequences, all-gap columns, entropy
    weighting far below nseq."""
    d = os.path.join(GOLD, "hmmbuild_cases")
    rows = [l.strip() for l in open(os.path.join(d, case + ".afa")) if not l.startswith(">")]
    gold = read(os.path.join(d, case + ".hmm"))
    mol = {"DNA": "dna", "RNA": "rna", "amino": "amino"}[[l.split()[1] for l in gold.splitlines() if l.startswith("ALPH")][0]]
    text, M, neff = hmmbuild_text(rows, mol, case)
    assert_same(text, gold, case)
    assert "EFFN  %f" % neff in text and "LENG  %d" % M in text
    assert not stats_lines(text)                       # not asked for: no calibration, no lines
    # E-value calibration (MSV / Viterbi filter scores and Forward scores of 3 x 200 seeded random sequences):
    # hmmbuild's printed digits, and nothing else in the file moves
    text2, _, _ = hmmbuild_text(rows, mol, case, stats=True)
    assert stats_lines(text2) == stats_lines(gold) and len(stats_lines(gold)) == 3, (case, stats_lines(text2), stats_lines(gold))
    assert body(text2) == body(text)
    assert text2.splitlines().index(stats_lines(text2)[0]) == [l.split()[0] for l in text2.splitlines()].index("CKSUM") + 1


@pytest.mark.parametrize("case,args,mol", [
    ("dna_hmmbuild", ("dna", 11, 120, 32, 8, 0.04, 0.004), "dna"),          # entropy weighting active: Neff 1.9 .. 3.8
    ("amino_hmmbuild", ("amino", 13, 90, 16, 4, 0.08, 0.004), "amino"),      # nine-component mixture prior, Neff ~1
])
def test_golden_family_models(case, args, mol):
    rows, subs = family_rows(*args)
    for idx, (lo, hi) in enumerate(subs):
        text, _, _ = hmmbuild_text(rows[lo:hi], mol, "sub", stats=True)
        gold = read(os.path.join(GOLD, case, "hmms", "A_0_%d.hmm" % idx))
        assert_same(text, gold, (case, idx))
        assert stats_lines(text) == stats_lines(gold), (case, idx, stats_lines(text), stats_lines(gold))


def test_example_backbone_models_and_column_tuples(tmp_path):
    """The 15 models of the end-to-end golden (the reference's example backbone, 62..500 sequences, 1278..2574
    nodes): text identity, and the reference's retained-column / non-gap tuples (algorithm.py:423-429)."""
    names, rows = backbone_rows()
    subs = synth.bfs_subsets(len(rows), 15)
    gold = json.load(gzip.open(os.path.join(GOLD, "example_e2e", "golden.json.gz"), "rt"))
    for idx, (lo, hi) in enumerate(subs):
        # (calibration on the five smallest of these 1 278 .. 2 574-node models: ~0.6 s each; all 15 were checked once,
        # tools/README.md)
        want_stats = idx >= 10
        text, M, _ = hmmbuild_text([r.upper() for r in rows[lo:hi]], "dna", "sub", stats=want_stats)
        gold_text = read(os.path.join(GOLD, "example_e2e", "hmms", "A_0_%d.hmm.gz" % idx))
        assert_same(text, gold_text, idx)
        if want_stats:
            assert stats_lines(text) == stats_lines(gold_text), (idx, stats_lines(text), stats_lines(gold_text))
    out = build_ehmm(names, rows, [("A_0_%d" % i, list(range(lo, hi))) for i, (lo, hi) in enumerate(subs)], "dna",
                     str(tmp_path), threads=4)
    for idx, (path, label, retained, nongaps) in enumerate(out):
        assert label == "A_0_%d" % idx and os.path.exists(path)
        assert list(retained) == gold["retained"][str(idx)]
        assert list(nongaps) == gold["nongaps"][str(idx)]
        # built from the reduced alignment (as the reference does), the model has the same probabilities; only the
        # MAP column numbers are those of the reduced alignment
        mine = [l for l in body(open(path).read()) if not l.startswith("CKSUM")]
        ref = [l for l in body(read(os.path.join(GOLD, "example_e2e", "hmms", "A_0_%d.hmm.gz" % idx))) if not l.startswith("CKSUM")]
        assert len(mine) == len(ref)
        k = 0
        for x, y in zip(mine, ref):
            fx, fy = x.split(), y.split()
            if len(fx) == 4 + 6 and fx[0].isdigit():          # match line: node, 4 emissions, MAP, cons, rf, mm, cs
                k += 1
                assert int(fx[0]) == k and int(fx[5]) == k and int(fy[5]) == retained[k - 1] + 1
                assert fx[:5] + fx[6:] == fy[:5] + fy[6:]
            else:
                assert x == y
        assert k == len(retained)


def test_numpy_restatement_agrees():
    """oracle/hmmbuild_np.py (written first, independently) and the C++ agree on probabilities and Neff."""
    from oracle import hmmbuild_np as hb
    d = os.path.join(GOLD, "hmmbuild_cases")
    for case in ("dna_fragments", "dna_degenerate", "dna_conserved", "dna_fragment_edges"):
        rows = [l.strip() for l in open(os.path.join(d, case + ".afa")) if not l.startswith(">")]
        m = hb.build(rows, "dna")
        text, M, neff = hmmbuild_text(rows, "dna", case)
        assert M == m["M"] and abs(neff - m["neff"]) < 1e-9
        lines = text.splitlines()
        start = next(i for i, l in enumerate(lines) if l.startswith("  COMPO")) + 3
        for k in range(1, M + 1):
            f = lines[start + 3 * (k - 1)].split()
            got = np.array([float(x) for x in f[1:5]])
            with np.errstate(divide="ignore"):
                want = -np.log(m["mat"][k].astype(np.float64))
            assert np.max(np.abs(got - want)) < 2e-5, (case, k)


def test_backbone_of_more_than_3072_columns():
    """Large backbones (every populated column becomes a node under --symfrac 0.0): the builder has no size limit of
    its own; the model it writes parses back with as many nodes as the alignment has populated columns, and the
    numpy restatement agrees on its probabilities."""
    from oracle import hmmbuild_np as hb
    from oracle import oracle as orc
    fam = synth.make_family(77, 3400, 8, "dna", 0.03, 1e-3)
    sym = "ACGT"
    rows = ["".join(sym[c] if c >= 0 else "-" for c in fam.msa[i]) for i in range(fam.msa.shape[0])]
    populated = int((fam.msa >= 0).any(axis=0).sum())
    text, M, neff = hmmbuild_text(rows, "dna", "big")
    assert M == populated and M > 3072
    m = hb.build(rows, "dna")
    assert M == m["M"] and abs(neff - m["neff"]) < 1e-9
    lines = text.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("  COMPO")) + 3
    for k in (1, 2, M // 2, M - 1, M):
        got = np.array([float(x) for x in lines[start + 3 * (k - 1)].split()[1:5]])
        with np.errstate(divide="ignore"):
            want = -np.log(m["mat"][k].astype(np.float64))
        assert np.max(np.abs(got - want)) < 2e-5, k
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".hmm", delete=False) as fh:
        fh.write(text)
    try:
        assert orc.OracleHMM(fh.name).M == M
    finally:
        os.unlink(fh.name)


def test_bad_input_is_refused():
    from witch_amd._lib import WitchHipError
    with pytest.raises(ValueError):
        hmmbuild_text(["ACGT", "ACG"], "dna")
    with pytest.raises(WitchHipError):
        hmmbuild_text(["AC#T", "ACGT"], "dna")
    with pytest.raises(WitchHipError):
        hmmbuild_text(["----", "----"], "dna")          # no consensus column
    with pytest.raises(ValueError):
        hmmbuild_text(["ACGT"], "protein")


def test_level0_hmmbuild_executable(tmp_path):
    """WITCH's `hmmbuildpath` key pointed at witch_amd/shim/bin/hmmbuild: the reference's command line
    (algorithm.py:463-470) produces hmmbuild's model; the name is the alignment file's basename (HMMER's rule);
    options that would change the model are refused."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "witch_amd", "shim", "bin", "hmmbuild")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "witch_amd", "shim"), "bin/hmmbuild"], check=True, stdout=subprocess.DEVNULL)
    for case, mol in (("dna_fragments", "dna"), ("amino_mixed", "amino"), ("rna_small", "rna")):
        out = tmp_path / ("hmmbuild.model.%s" % case)
        afa = os.path.join(GOLD, "hmmbuild_cases", case + ".afa")
        cmd = [exe, "--cpu", "1", "--" + mol, "--ere", "0.59", "--symfrac", "0.0", "--informat", "afa", "-o", "/dev/null", str(out), afa]
        subprocess.run(cmd, check=True)
        text = out.read_text()
        gold = read(os.path.join(GOLD, "hmmbuild_cases", case + ".hmm"))
        assert_same(text, gold, case)
        assert "NAME  %s\n" % case in text
        assert stats_lines(text) == stats_lines(gold)        # like hmmbuild, the executable calibrates by default
        subprocess.run(cmd[:1] + ["--nostats"] + cmd[1:], check=True)
        assert not stats_lines(out.read_text()) and body(out.read_text()) == body(text)
    r = subprocess.run([exe, "--dna", "--wblosum", "x", "y"], capture_output=True, text=True)
    assert r.returncode != 0 and "not supported" in r.stderr
    r = subprocess.run([exe, "--ere", "0.59", str(tmp_path / "o"), os.path.join(GOLD, "hmmbuild_cases", "dna_single.afa")], capture_output=True, text=True)
    assert r.returncode != 0 and "required" in r.stderr


HMMER_BIN = "/root/reference/witch_msa/tools/magus/tools/hmmer"


@pytest.mark.skipif(not os.path.exists(os.path.join(HMMER_BIN, "hmmsearch")), reason="the reference's bundled HMMER is not on this machine")
def test_stock_hmmsearch_reads_the_calibrated_file(tmp_path):
    """Interoperation with stock HMMER (a `-p <hmmdir>` rerun of the reference on an eHMM this build wrote,
    witch_msa/gcmm/gcmm.py:163-171): the reference's bundled hmmsearch, with the reference's command line
    (algorithm.py:526-532), prints the same report - scores AND E-values - for a model written by the level-0
    hmmbuild executable as for hmmbuild's own file; without the STATS lines it reports no hit at all."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "witch_amd", "shim", "bin", "hmmbuild")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "witch_amd", "shim"), "bin/hmmbuild"], check=True, stdout=subprocess.DEVNULL)
    for case, mol in (("dna_conserved", "dna"), ("amino_mixed", "amino"), ("random_07_rna", "rna")):
        afa = os.path.join(GOLD, "hmmbuild_cases", case + ".afa")
        rows = [l.strip() for l in open(afa) if not l.startswith(">")]
        fa = tmp_path / (case + ".fa")
        fa.write_text("".join(">q%d\n%s\n" % (i, r.replace("-", "").replace(".", "")) for i, r in enumerate(rows[:6]) if len(r.replace("-", "").replace(".", "")) > 0))
        reports = {}
        for tag, extra in (("mine", []), ("nostats", ["--nostats"])):
            model = tmp_path / ("%s.%s.hmm" % (case, tag))
            subprocess.run([exe] + extra + ["--cpu", "1", "--" + mol, "--ere", "0.59", "--symfrac", "0.0", "--informat", "afa", "-o", "/dev/null", "-n", case,
                                            str(model), afa], check=True)
            reports[tag] = model
        reports["gold"] = os.path.join(GOLD, "hmmbuild_cases", case + ".hmm")
        out = {}
        for tag, model in reports.items():
            o = tmp_path / ("%s.%s.out" % (case, tag))
            subprocess.run([os.path.join(HMMER_BIN, "hmmsearch"), "--cpu", "1", "--noali", "-E", "99999999", "-o", str(o), "--max", str(model), str(fa)], check=True)
            out[tag] = [l for l in o.read_text().splitlines() if not l.startswith("#")]
        assert out["mine"] == out["gold"], case
        assert any("No hits detected" in l for l in out["nostats"]) and not any("No hits detected" in l for l in out["gold"]), case
