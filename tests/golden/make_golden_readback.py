#!/usr/bin/env python3
"""a1 fidelity check with the reference's OWN reader.

RUNS ONLY IN THE BUILD CONTAINER.  witch_amd.gcmm.search writes the hmmsearch.results.* files of a
golden case in the reference's chunk layout; the reference's readHMMSearch
(witch_msa/gcmm/loader.py:277-294: `find` + eval of every result file of a subset directory) reads
them back, and the reference's ranking (loader.py:325-330) orders them.  What the reference read is
stored as data in tests/golden/readback.json.gz; tests/test_gcmm_host.py checks that it is exactly
what the engine holds.
"""
import gzip
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    ref = mg.import_reference()
    Configs = ref[1]
    from witch_msa.gcmm.loader import readHMMSearch
    Configs.log = staticmethod(lambda *a, **k: None)
    sys.path.insert(0, os.path.dirname(HERE))
    from tests.conftest import load_case
    from tests.test_gcmm_host import _engine_from_golden
    from witch_amd import gcmm
    case = load_case("dna_hmmbuild")
    _engine_from_golden(case)
    tmp = tempfile.mkdtemp(prefix="readback_")
    dirs = {i: os.path.join(tmp, "root", "A_0_%d" % i) for i in case.hmm_index}
    files, _ = gcmm.search(dirs, num_cpus=6)           # lcm(8, 6) // 8 = 3 chunks

    class _Lock:
        def acquire(self):
            pass

        def release(self):
            pass

    class _Subset:
        def __init__(self, index, d):
            self.index, self.alignment_dir = index, d
    ranks = {}
    for i in case.hmm_index:                            # arrival order = index order
        for taxon, pairs in readHMMSearch(_Lock(), len(dirs), _Subset(i, dirs[i])).items():
            ranks.setdefault(taxon, []).extend(pairs)
    ranked = {t: [[int(i), float(s)] for i, s in sorted(v, key=lambda x: x[1], reverse=True)] for t, v in ranks.items()}
    with gzip.open(os.path.join(HERE, "readback.json.gz"), "wt") as f:
        json.dump({"case": "dna_hmmbuild", "num_cpus": 6, "n_files": len(files), "ranked": ranked}, f, separators=(",", ":"), sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(ref[0], ignore_errors=True)
    print("readback: %d files written, %d taxa read back by the reference" % (len(files), len(ranked)))


if __name__ == "__main__":
    main()
