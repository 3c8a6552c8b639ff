"""Pins the consensus-DP oracle (oracle/consensus.py) to the strings the reference's own
alignSubQueriesNew (witch_msa/gcmm/aligner.py:350-538) produced (golden 'merged')."""
import pytest

from oracle import consensus


def test_consensus_matches_reference(golden_case):
    case = golden_case
    g = case.g
    if not g.get("merged"):
        pytest.skip("case has no backbone alignment")
    retained = {int(k): v for k, v in g["retained"].items()}
    nongaps = {int(k): v for k, v in g["nongaps"].items()}
    n = 0
    for qn, qs in zip(case.qnames, case.qseqs):
        want = g["merged"].get(qn)
        if want is None:
            continue
        w = {int(i): x for i, x in g["weights"][qn]}
        order = g["align"][qn]["order"]
        aligned = [(i, g["align"][qn]["cols"][str(i)]) for i in order]
        codes, _ = consensus.consensus_trace(len(qs), aligned, w, retained, nongaps, g["backbone_length"])
        got = consensus.trace_to_string(qs, codes, g["backbone_length"])
        assert got == want, (case.name, qn)
        n += 1
    assert n > 0
