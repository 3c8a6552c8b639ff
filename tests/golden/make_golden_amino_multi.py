#!/usr/bin/env python3
"""HMMER-pinned protein multidomain pairs.

RUNS ONLY IN THE BUILD CONTAINER.  BASELINE config 5 (`aa_50k_x500`: 50 000 protein queries of 50-2000
residues x 500 models) has a multidomain region in a quarter of its pairs - the class HMMER resolves with
200 stochastic tracebacks.  This takes the FIRST 64 queries of that seeded workload and
every 25th of its 500 models (20 models, the files bench.py generates) through HMMER-bin with the
reference's exact command lines and the reference's own evalHMMSearchOutput / calculateWeights /
getBackbones (make_golden.Case.finish) -> tests/golden/amino_multidomain (models gzipped).
"""
import gzip
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

sys.path.insert(0, mg.REPO)


def main():
    import bench
    from witch_amd import synth
    ref = mg.import_reference()
    tmp = tempfile.mkdtemp(prefix="aamulti_")
    fam, se, names, seqs, k = bench.make_workload("aa_50k_x500", tmp, 2000, None)
    c = mg.Case("amino_multidomain", "amino", k)
    for pos in range(0, 500, 25):
        hp = os.path.join(c.dir, "hmms", "A_0_%d.hmm" % se.index[pos])
        shutil.copyfile(se.paths[pos], hp)
        c.add_hmm(hp, int(se.index[pos]), int(se.nseq[pos]))
    c.qnames = list(names[:64])
    c.qseqs = [synth.to_text(s, "amino") for s in seqs[:64]]
    c.finish(ref)
    for hf in c.hmm_files:                                   # 20 x ~0.4 MB of text: keep them gzipped
        p = os.path.join(c.dir, hf)
        with open(p, "rb") as fi, gzip.GzipFile(p + ".gz", "wb", mtime=0) as fo:
            fo.write(fi.read())
        os.remove(p)
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(ref[0], ignore_errors=True)


if __name__ == "__main__":
    main()
