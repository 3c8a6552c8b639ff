"""Host logic of the reference-shaped functions (witch_amd.gcmm) on CPU: the engine table is
filled from the golden vectors (no GPU), then every function must reproduce what the
reference's own functions returned when make_golden.py ran them."""
import ast
import os

import numpy as np
import pytest

from oracle import oracle as orc
from witch_amd import gcmm


class _Sub:
    def __init__(self, path, n):
        self.hmm_model_path = path
        self.num_taxa = n


def _engine_from_golden(case):
    H = len(case.hmm_index)
    nq = len(case.qnames)
    deci = np.zeros((nq, H), np.int32)
    flags = np.zeros((nq, H), np.uint8)
    for j, hf in enumerate(case.hmm_files):
        for q, qn in enumerate(case.qnames):
            if qn in case.g["search"][hf]:
                deci[q, j] = int(round(case.g["search"][hf][qn]["score"] * 10))
                flags[q, j] = 1
    k = case.k
    idx = np.full((nq, k), -1, np.int32)
    w = np.zeros((nq, k))
    nk = np.zeros(nq, np.int32)
    nu = np.zeros(nq, np.int32)
    size_of = dict(zip(case.hmm_index, case.nseq))
    cols, co, pair_of = [], [0], {}
    for q, qn in enumerate(case.qnames):
        ranked = orc.rank_bitscores(case.hmm_index, deci[q], flags[q])
        if not ranked:
            continue
        ids = [r[0] for r in ranked]
        ww = orc.calculate_weights(ids, [r[1] for r in ranked], [size_of[i] for i in ids], k)
        nk[q] = len(ww)
        for t, (i, x) in enumerate(ww):
            idx[q, t] = i
            w[q, t] = x
        nu[q] = orc.adaptive_cut(ww)
        for i, _ in ww[:nu[q]]:
            g = case.g["align"][qn]["cols"].get(str(i))
            if g is None:      # tie at the cut: the reference aligned a different, equally weighted HMM
                g = [-1] * len(case.qseqs[q])
            pair_of[(q, i)] = len(co) - 1
            cols.extend(g)
            co.append(len(cols))
    eng = gcmm.QueryAlignmentEngine.from_results(
        case.qnames, case.hmm_index, case.nseq, deci, flags, k, topk=(idx, w, nk, nu),
        aligned=(np.array(cols, np.int32), np.array(co, np.int64), pair_of))
    return gcmm.install(eng)


def test_reference_shaped_functions(golden_case, tmp_path):
    case = golden_case
    eng = _engine_from_golden(case)
    index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(case.hmm_index, case.hmm_paths, case.nseq)}
    ranked = gcmm.rankBitscores(index_to_hmm, {}, None, None)
    weights = gcmm.writeWeights(index_to_hmm, ranked, None)
    n_checked = 0
    for q, qn in enumerate(case.qnames):
        if qn not in case.g["weights"]:
            assert qn not in ranked
            continue
        # ranking: descending score
        sc = [s for _, s in ranked[qn]]
        assert sc == sorted(sc, reverse=True)
        gold_w = case.g["weights"][qn]
        got = weights[qn]
        assert all(isinstance(x[1], np.float64) for x in got)      # the reference's value type
        assert sorted(x[1] for x in got) == pytest.approx(sorted(x[1] for x in gold_w), rel=1e-12)
        ret = gcmm.getBackbones(index_to_hmm, qn, q, case.qseqs[q], "unused.fasta", got,
                                str(tmp_path), str(tmp_path), use_gcm=False)
        ret_str, weights_map, cols = ret
        gold = case.g["align"][qn]
        assert ret_str.split("\t")[0] == qn
        assert ret_str.startswith("%s\tpassed to main pipeline with top %d weights: [" % (qn, len(cols)))
        assert len(cols) == len(gold["cols"])
        assert weights_map == {i: x for i, x in got}
        tied = len(set(x[1] for x in gold_w)) < len(gold_w)
        if not tied:
            assert ret_str == gold["ret_str"], (ret_str, gold["ret_str"])
            for i, c in cols.items():
                assert c == gold["cols"][str(i)]
            n_checked += 1
    assert n_checked > 0
    # the empty-weights quirk of the reference (aligner.py:46-47)
    assert gcmm.getBackbones(index_to_hmm, "x", 0, "ACGT", "p", (), ".", ".", use_gcm=False) == ("N/A", None)


def test_result_files_and_weights_txt(golden_case, tmp_path):
    case = golden_case
    eng = _engine_from_golden(case)
    dirs = {i: str(tmp_path / ("A_0_%d" % i)) for i in case.hmm_index}
    files, _ = gcmm.search(dirs)
    assert len(files) == len(case.hmm_index)
    for i, hf in zip(case.hmm_index, case.hmm_files):
        path = os.path.join(dirs[i], "hmmsearch.results.A_0_%d.fragment_chunk_0" % i)
        got = ast.literal_eval(open(path).read())        # the reference eval()s these files (loader.py:289-293)
        want = {q: v["score"] for q, v in case.g["search"][hf].items()}
        assert {q: v[1] for q, v in got.items()} == want
    index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(case.hmm_index, case.hmm_paths, case.nseq)}
    weights = gcmm.writeWeights(index_to_hmm, gcmm.rankBitscores(index_to_hmm, {}), None)
    wpath = str(tmp_path / "weights.txt")
    gcmm.writeWeightsToLocal(weights, wpath)
    back = gcmm.readWeightsFromLocal(wpath)
    assert back.keys() == weights.keys()
    for t in weights:
        assert [i for i, _ in back[t]] == [i for i, _ in weights[t]]
        assert [float(x) for _, x in back[t]] == [float(x) for _, x in weights[t]]
    # one line per query 'taxon:((idx, w), ...)' exactly as weighting.py:174-179 writes it
    first = open(wpath).readline()
    assert first.split(":")[0] in weights and first.split(":", 1)[1].startswith("((")


def test_search_uses_the_reference_chunk_layout(tmp_path):
    """a1: lcm(#HMMs, #cpus) // #HMMs chunks, queries dealt round-robin in input order, empty
    chunks skipped, at most 20 000 per chunk, names with blanks/tabs rejected
    (witch_msa/gcmm/algorithm.py:280-284, 351-359, 376-383; helpers/alignment_tools.py:674-686)."""
    import numpy as np
    assert gcmm.num_chunks_for(200, 8) == 1 and gcmm.num_chunks_for(10, 8) == 4 and gcmm.num_chunks_for(3, 16) == 16
    names = ["q%d" % i for i in range(7)]
    assert gcmm.divide_to_equal_chunks(names, 3) == [["q0", "q3", "q6"], ["q1", "q4"], ["q2", "q5"]]
    assert gcmm.divide_to_equal_chunks(["a", "b"], 4) == [["a"], ["b"], None, None]
    big = gcmm.divide_to_equal_chunks(range(50001), 2)          # 25 000.5 per chunk > 20 000 -> 3 chunks
    assert len(big) == 3 and max(len(c) for c in big) <= 20000
    hmm_index = [0, 1, 2]
    deci = np.arange(21, dtype=np.int32).reshape(7, 3) * 7 - 30
    flags = (deci % 5 != 0).astype(np.uint8)
    gcmm.install(gcmm.QueryAlignmentEngine.from_results(names, hmm_index, [5, 6, 7], deci, flags, 2))
    dirs = {i: str(tmp_path / "root" / ("A_0_%d" % i)) for i in hmm_index}
    seqs = {n: "ACGT" * (i + 1) for i, n in enumerate(names)}
    files, frags = gcmm.search(dirs, num_cpus=4, fragment_chunk_dir=str(tmp_path / "fragment_chunks"), sequences=seqs)
    assert gcmm.num_chunks_for(3, 4) == 4 and len(frags) == 4 and len(files) == 12
    assert open(frags[1]).read() == ">q1\nACGTACGT\n>q5\n" + "ACGT" * 6 + "\n"
    for col, i in enumerate(hmm_index):
        seen = {}
        for c in range(4):
            d = ast.literal_eval(open(os.path.join(dirs[i], "hmmsearch.results.A_0_%d.fragment_chunk_%d" % (i, c))).read())
            assert set(d) <= set(names[c::4])
            seen.update(d)
        assert seen == {n: (0.0, deci[r, col] / 10.0) for r, n in enumerate(names) if flags[r, col]}
    gcmm.install(gcmm.QueryAlignmentEngine.from_results(["ok", "has blank"], hmm_index, [5, 6, 7], deci[:2], flags[:2], 2))
    with pytest.raises(ValueError, match="whitespaces or tabs"):
        gcmm.search(dirs)


def test_no_engine_installed_fails_loudly():
    from witch_amd.gcmm import engine
    old = engine._ENGINE
    engine._ENGINE = None
    try:
        with pytest.raises(RuntimeError):
            gcmm.rankBitscores({}, {})
    finally:
        engine._ENGINE = old


def test_checkpoint_file_format(tmp_path):
    """f4: <outdir>/checkpoint_alignments.txt.gz as callback.py:19-26 appends it (one gzip member per
    query, 'taxon\\tsequence\\n') and loader.py:95-150 reads it back (taxon = text before the LAST tab,
    labels from the case of the characters, later lines win)."""
    import gzip
    from witch_amd.gcmm.merge import QueryAlignment
    path = str(tmp_path / "checkpoint_alignments.txt.gz")

    def qa(name, text):
        q = QueryAlignment()
        q[name] = text
        return q
    success, ignored, retry = [], [], []
    gcmm.callback_queryAlignment(success, ignored, retry, 0, qa("q1", "--ACgtT-"), 0, "q1", path)
    gcmm.callback_queryAlignment(success, ignored, retry, 0, QueryAlignment(), 1, "empty", path)      # failed -> ignored
    gcmm.callback_queryAlignment(success, ignored, retry, 1, None, 2, "again", path)                 # retry once
    assert (len(success), ignored, retry) == (1, ["empty"], [2])
    n = gcmm.writeCheckpointAlignments([qa("name\twith tab", "acGT"), "skipped", QueryAlignment(), qa("q1", "AC--")], path)
    assert n == 2
    # the file is a concatenation of single-line gzip members: what the reference's 'ab' appends produce
    raw = open(path, "rb").read()
    assert raw.count(b"\x1f\x8b\x08") == 3
    assert gzip.decompress(raw) == b"q1\t--ACgtT-\nname\twith tab\tacGT\nq1\tAC--\n"
    back = gcmm.readCheckpointAlignments(path)
    assert set(back) == {"q1", "name\twith tab"}
    assert back["q1"]["q1"] == "AC--" and back["q1"]._col_labels == [0, 1, 2, 3]                  # the later line wins
    assert back["name\twith tab"]._col_labels == [-1, -2, 0, 1]


def test_result_files_read_back_by_the_reference_reader():
    """tests/golden/readback.json.gz holds what the REFERENCE's readHMMSearch + ranking read from the
    result files gcmm.search wrote (3 chunks per HMM); it must be what the engine answers."""
    import gzip
    import json
    from tests.conftest import GOLDEN, load_case
    g = json.load(gzip.open(os.path.join(GOLDEN, "readback.json.gz"), "rt"))
    case = load_case(g["case"])
    _engine_from_golden(case)
    assert g["n_files"] == 3 * len(case.hmm_index)
    index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(case.hmm_index, case.hmm_paths, case.nseq)}
    ranked = gcmm.rankBitscores(index_to_hmm, {})
    assert set(ranked) == set(g["ranked"])
    for t, want in g["ranked"].items():
        assert [[i, s] for i, s in ranked[t]] == want, t


def _wire():
    import gzip
    import json
    from tests.conftest import GOLDEN
    with gzip.open(os.path.join(GOLDEN, "wire", "wire.json.gz"), "rt") as f:
        return json.load(f), os.path.join(GOLDEN, "wire")


def test_weights_txt_against_the_reference_writer_and_reader(tmp_path):
    """f4, both directions, on data made by tests/golden/make_golden_wire.py with the REFERENCE's own
    writeWeightsToLocal / readWeightsFromLocal (witch_msa/gcmm/weighting.py:174-194):
    (i) the file the reference wrote (numpy-2 text: 'np.float64(...)' inside the tuples) is read by this
    package's reader and holds the engine's weights; (ii) what the reference read from the file this
    package wrote is, bit for bit, what the engine answers."""
    from tests.conftest import load_case
    meta, d = _wire()
    case = load_case(meta["weights_case"])
    _engine_from_golden(case)
    index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(case.hmm_index, case.hmm_paths, case.nseq)}
    weights = gcmm.writeWeights(index_to_hmm, gcmm.rankBitscores(index_to_hmm, {}), None)
    ref = gcmm.readWeightsFromLocal(os.path.join(d, "ref_weights.txt"))
    assert "np.float64(" in open(os.path.join(d, "ref_weights.txt")).readline()
    assert list(ref.keys()) == list(weights.keys())                     # same taxa, same line order
    n_exact = 0
    for t, got in weights.items():
        want = ref[t]
        assert all(isinstance(x[1], np.float64) for x in want)
        assert len(want) == len(got)
        assert [x[1] for x in got] == pytest.approx([x[1] for x in want], rel=1e-12)
        if len(set(x[1] for x in want)) == len(want):                   # no exact weight ties: same order of models
            assert [x[0] for x in got] == [x[0] for x in want]
            n_exact += 1
    assert n_exact > 10
    back = meta["ref_read_of_repo_weights"]
    assert sorted(back.keys()) == sorted(weights.keys())                # (the JSON fixture is stored with sorted keys)
    for t, got in weights.items():
        assert [[int(i), float(w).hex()] for i, w in got] == back[t]
    # and this package's own round trip of the reference's file reproduces it value for value
    p = str(tmp_path / "w.txt")
    gcmm.writeWeightsToLocal(ref, p)
    again = gcmm.readWeightsFromLocal(p)
    assert {t: [(i, float(w)) for i, w in v] for t, v in again.items()} == {t: [(i, float(w)) for i, w in v] for t, v in ref.items()}


def test_checkpoint_file_against_the_reference_writer_and_reader(tmp_path):
    """f4, both directions (callback.py:9-29, loader.py:95-150): (i) the gzip file the reference's
    callback_queryAlignment appended (one member per finished query, a failed and a retried one, one taxon
    twice) is read by this package; this package's writer reproduces its BYTES from the same strings;
    (ii) what the reference's readCheckpointAlignments read from this package's file = the strings and labels
    this package holds."""
    import gzip
    from tests.conftest import load_case
    from witch_amd.gcmm.merge import QueryAlignment
    meta, d = _wire()
    case = load_case(meta["checkpoint_case"])
    merged = case.g["merged"]
    ref_path = os.path.join(d, "ref_checkpoint_alignments.txt.gz")
    order = meta["ref_callback"]["order"]
    assert meta["ref_callback"]["ignored"] == ["empty_query"] and meta["ref_callback"]["retry"] == [901]
    assert meta["ref_callback"]["n_success"] == len(order) == len(merged) + 1
    back = gcmm.readCheckpointAlignments(ref_path)
    first = order[0]
    assert set(back) == set(merged)
    for qn, text in merged.items():
        want = text if qn != first else text.replace("-", "", 1) + "-"      # the later line wins
        assert back[qn][qn] == want
        low = [c.islower() for c in want]
        assert [x < 0 for x in back[qn]._col_labels] == low
    # same strings through this package's writer: the decompressed stream and the member structure are the
    # reference's (the gzip header carries a timestamp, so the raw bytes differ only there)
    qas = []
    for qn in order[:-1]:
        a = QueryAlignment()
        a[qn] = merged[qn]
        qas.append(a)
    a = QueryAlignment()
    a[first] = merged[first].replace("-", "", 1) + "-"
    qas.append(a)
    mine = str(tmp_path / "checkpoint_alignments.txt.gz")
    success, ignored, retry = [], [], []
    for n, q in enumerate(qas):
        gcmm.callback_queryAlignment(success, ignored, retry, 0, q, n, next(iter(q)), mine)
    raw_ref, raw_mine = open(ref_path, "rb").read(), open(mine, "rb").read()
    assert gzip.decompress(raw_mine) == gzip.decompress(raw_ref)
    assert raw_mine.count(b"\x1f\x8b\x08") == raw_ref.count(b"\x1f\x8b\x08") == len(order)
    batch = str(tmp_path / "batch.txt.gz")
    assert gcmm.writeCheckpointAlignments(qas, batch) == len(order)
    assert gzip.decompress(open(batch, "rb").read()) == gzip.decompress(raw_ref)
    # (ii) the reference read this package's file
    got = meta["ref_read_of_repo_checkpoint"]
    qas = []
    for qn, text in merged.items():
        a = QueryAlignment()
        a[qn] = text
        qas.append(a)
    tabbed = QueryAlignment()
    tabbed["name\twith tab"] = "acGT-x"
    qas.insert(3, tabbed)
    assert set(got) == {next(iter(q)) for q in qas}
    for q in qas:
        t = next(iter(q))
        assert got[t][0] == q[t]
        assert got[t][1] == q._col_labels


def test_engine_rejects_an_unknown_multidomain_policy():
    """QueryAlignmentEngine.run validates multidomain_policy before any GPU work (a typo used to select the resolver)."""
    with pytest.raises(ValueError, match="multidomain_policy"):
        gcmm.QueryAlignmentEngine.run({}, [], 3, multidomain_policy="envelop")
    with pytest.raises(ValueError, match="multidomain_policy"):
        gcmm.QueryAlignmentEngine.run({}, [], 3, multidomain_policy="")


def test_weights_reader_accepts_both_number_forms(tmp_path):
    """weights.txt as numpy 1 wrote it (plain numbers) and as numpy 2 writes it (np.float64(...)), incl. a one-model tuple,
    an empty tuple and a taxon with a colon in its name."""
    p = tmp_path / "w.txt"
    p.write_text("a:((3, 0.75), (1, 0.25))\n"
                 "b:((7, np.float64(1.0)),)\n"
                 "c:()\n"
                 "ns:x:((2, np.float64(0.5)), (0, numpy.float64(0.5)))\n")
    w = gcmm.readWeightsFromLocal(str(p))
    assert w["a"] == ((3, 0.75), (1, 0.25)) and w["b"] == ((7, 1.0),) and w["c"] == () and w["ns:x"] == ((2, 0.5), (0, 0.5))
    assert all(isinstance(x[1], np.float64) for v in w.values() for x in v)
