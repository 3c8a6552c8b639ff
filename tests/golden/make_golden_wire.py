#!/usr/bin/env python3
"""f4 fidelity of the two remaining wire formats, with the reference's OWN writers and readers
(VERDICT r2 item 1a).

RUNS ONLY IN THE BUILD CONTAINER (imports the reference from a /tmp copy, make_golden.import_reference).
Both directions are stored as DATA under tests/golden/wire/:

  weights.txt   (witch_msa/gcmm/weighting.py:174-194)
    ref_weights.txt              written by the reference's writeWeightsToLocal from the values its own
                                 calculateWeights returned for the golden case (np.float64 in the tuples:
                                 under numpy 2 the line text is "((3, np.float64(0.5)), ...)", which the
                                 reference eval()s back with numpy in scope)
    ref_read_of_repo_weights     what the reference's readWeightsFromLocal read from the file
                                 witch_amd.gcmm.writeWeightsToLocal wrote (float64 as hex strings)

  checkpoint_alignments.txt.gz   (witch_msa/gcmm/callback.py:9-29, loader.py:95-150)
    ref_checkpoint_alignments.txt.gz   appended by the reference's callback_queryAlignment, one call per
                                 query, from ExtendedAlignment objects holding the reference's own
                                 alignSubQueriesNew strings of the golden case (+ a failed, an empty and a
                                 repeated query)
    ref_read_of_repo_checkpoint  what the reference's readCheckpointAlignments (thread pool of 2) read
                                 from the file witch_amd.gcmm.writeCheckpointAlignments wrote:
                                 {taxon: [sequence, col_labels]}
"""
import gzip
import json
import os
import shutil
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

OUT = os.path.join(HERE, "wire")


def main():
    ref = mg.import_reference()
    Configs, calculateWeights = ref[1], ref[3]
    from witch_msa.gcmm.weighting import writeWeightsToLocal, readWeightsFromLocal
    from witch_msa.gcmm.callback import callback_queryAlignment
    from witch_msa.gcmm.loader import readCheckpointAlignments
    from witch_msa.helpers.alignment_tools import ExtendedAlignment
    for name in ("log", "warning", "runtime", "debug", "error"):
        setattr(Configs, name, staticmethod(lambda *a, **k: None))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.conftest import load_case
    from tests.test_gcmm_host import _engine_from_golden, _Sub
    from witch_amd import gcmm
    from witch_amd.gcmm.merge import QueryAlignment
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    tmp = tempfile.mkdtemp(prefix="wire_")
    meta = {}

    # ------------------------------------------------------------------ weights.txt
    case = load_case("dna_hmmbuild")
    _engine_from_golden(case)
    index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(case.hmm_index, case.hmm_paths, case.nseq)}
    # (i) the reference writes: its own calculateWeights on the golden ranking (weighting.py:58-74, 146-160)
    Configs.num_hmms = case.k
    size_of = dict(zip(case.hmm_index, case.nseq))
    ref_weights = {}
    for qn in case.qnames:
        pairs = []
        for i, hf in zip(case.hmm_index, case.hmm_files):
            if qn in case.g["search"][hf]:
                pairs.append((i, case.g["search"][hf][qn]["score"]))
        if not pairs:
            continue
        pairs.sort(key=lambda x: x[1], reverse=True)
        ids = [p[0] for p in pairs]
        ref_weights.update(calculateWeights((qn, ids, [p[1] for p in pairs], [size_of[i] for i in ids])))
    writeWeightsToLocal(ref_weights, os.path.join(OUT, "ref_weights.txt"))
    # (ii) the reference reads what the repo wrote
    repo_weights = gcmm.writeWeights(index_to_hmm, gcmm.rankBitscores(index_to_hmm, {}), None)
    rp = os.path.join(tmp, "weights.txt")
    gcmm.writeWeightsToLocal(repo_weights, rp)
    back = readWeightsFromLocal(rp)
    meta["weights_case"] = "dna_hmmbuild"
    meta["ref_read_of_repo_weights"] = {t: [[int(i), float(w).hex()] for i, w in v] for t, v in back.items()}
    import numpy
    meta["numpy_of_the_generator"] = numpy.__version__

    # ------------------------------------------------------------------ checkpoint_alignments.txt.gz
    case = load_case("example_ehmm")
    merged = case.g["merged"]                       # the reference's alignSubQueriesNew strings
    cp = os.path.join(OUT, "ref_checkpoint_alignments.txt.gz")
    success, ignored, retry = [], [], []
    order = []
    for n, (qn, text) in enumerate(merged.items()):
        q = ExtendedAlignment([])
        q[qn] = text
        callback_queryAlignment(success, ignored, retry, 0, q, n, qn, cp)
        order.append(qn)
    first = order[0]
    callback_queryAlignment(success, ignored, retry, 0, ExtendedAlignment([]), 900, "empty_query", cp)   # ignored
    callback_queryAlignment(success, ignored, retry, 1, None, 901, "retried_query", cp)                  # retry
    q = ExtendedAlignment([])
    q[first] = merged[first].replace("-", "", 1) + "-"        # the same taxon again: the later line wins on read
    callback_queryAlignment(success, ignored, retry, 0, q, 902, first, cp)
    meta["checkpoint_case"] = "example_ehmm"
    meta["ref_callback"] = {"n_success": len(success), "ignored": ignored, "retry": retry, "order": order + [first]}
    # the reference reads what the repo wrote
    qas = []
    for qn, text in merged.items():
        a = QueryAlignment()
        a[qn] = text
        qas.append(a)
    tabbed = QueryAlignment()
    tabbed["name\twith tab"] = "acGT-x"
    qas.insert(3, tabbed)
    qas.insert(5, "skipped")
    rc = os.path.join(tmp, "checkpoint_alignments.txt.gz")
    gcmm.writeCheckpointAlignments(qas, rc)
    pool = ThreadPoolExecutor(max_workers=2)

    class _Lock:
        def acquire(self):
            pass

        def release(self):
            pass
    got = readCheckpointAlignments(rc, pool, _Lock())
    pool.shutdown()
    meta["ref_read_of_repo_checkpoint"] = {t: [a[t], [int(x) for x in a._col_labels]] for t, a in got.items()}
    with gzip.open(os.path.join(OUT, "wire.json.gz"), "wt") as f:
        json.dump(meta, f, separators=(",", ":"), sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(ref[0], ignore_errors=True)
    print("wire: weights %d taxa (reference wrote) / %d (reference read); checkpoint %d members written by the "
          "reference, %d taxa read by the reference" % (len(ref_weights), len(back), len(success), len(got)))
    print(open(os.path.join(OUT, "ref_weights.txt")).readline()[:200])


if __name__ == "__main__":
    main()
