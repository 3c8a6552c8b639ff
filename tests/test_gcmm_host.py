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
    files = gcmm.search(dirs)
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


def test_no_engine_installed_fails_loudly():
    from witch_amd.gcmm import engine
    old = engine._ENGINE
    engine._ENGINE = None
    try:
        with pytest.raises(RuntimeError):
            gcmm.rankBitscores({}, {})
    finally:
        engine._ENGINE = old
