#!/usr/bin/env python3
"""End-to-end golden for the whole chain on a WITCH-shaped case.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).  The reference's example data
(examples/data: 500-sequence backbone alignment of 2574 columns, 500 fragment queries) through the
reference's own pipeline pieces, chained as witch_msa/gcmm/gcmm.py:219-248 chains them:

  eHMM      15 nested subsets of the backbone rows (BFS halves: 500 / 250 / 125 / 62-63 sequences),
            each built with the reference's hmmbuild command (gcmm/algorithm.py:463-470)
  search    hmmsearch --cpu 1 --noali -E 99999999 --max per HMM (algorithm.py:526-532), parsed by the
            reference's evalHMMSearchOutput (:579-605)
  rank      stable sort by score, arrival order = HMM index order (loader.py:325-330)
  weights   the reference's calculateWeights (weighting.py:58-74), k = 10
  align     the reference's getBackbones -> hmmalign per kept HMM (aligner.py:33-148)
  consensus the reference's alignSubQueriesNew (aligner.py:350-538)
  merge     the reference's mergeAlignmentsCollapsed (merger.py:40-131) -> <out>.fasta, <out>.masked.fasta

Only DATA is stored under tests/golden/example_e2e/: HMM text files (gzipped), the backbone alignment
and the queries (inputs of the reference's own example), the scores / weights / per-query strings and
the two final FASTA files.  No reference source is copied.
"""
import gzip
import hashlib
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from witch_amd import synth  # noqa: E402

N_HMMS, K = 15, 10


def main():
    ref = mg.import_reference()
    scratch, Configs, evalHMMSearchOutput, calculateWeights, getBackbones, alignSubQueriesNew, Alignment = ref
    from witch_msa.helpers.alignment_tools import ExtendedAlignment  # noqa: F401
    from witch_msa.gcmm.merger import mergeAlignmentsCollapsed
    out = os.path.join(HERE, "example_e2e")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "hmms"))
    tmp = tempfile.mkdtemp(prefix="golden_e2e_")
    # ---- inputs
    rows = []
    with gzip.open(os.path.join(mg.REF, "examples/data/backbone.aln.fasta.gz"), "rt") as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                rows.append([line[1:].split()[0], ""])
            elif line:
                rows[-1][1] += line.upper()          # the reference upper-cases on read (alignment_tools.py:730-731)
    names, seqs = [], []
    for line in open(os.path.join(mg.REF, "examples/data/unaligned_frag.fasta")):
        line = line.strip()
        if line.startswith(">"):
            names.append(line[1:].split()[0])
            seqs.append("")
        elif line:
            seqs[-1] += line.upper()
    B = len(rows[0][1])
    bpath = os.path.join(tmp, "backbone.fasta")
    with open(bpath, "w") as f:
        for n, t in rows:
            f.write(">%s\n%s\n" % (n, t))
    with gzip.open(os.path.join(out, "backbone.fasta.gz"), "wt", compresslevel=9) as f:
        f.write(open(bpath).read())
    qpath = os.path.join(out, "queries.fasta")
    with open(qpath, "w") as f:
        for n, t in zip(names, seqs):
            f.write(">%s\n%s\n" % (n, t))
    # ---- eHMM
    subsets = synth.bfs_subsets(len(rows), N_HMMS)
    hmm_files, hmm_index, nseq, retained, nongaps = [], [], [], {}, {}
    for idx, (lo, hi) in enumerate(subsets):
        fa = os.path.join(tmp, "sub.fasta")
        with open(fa, "w") as o:
            for n, t in rows[lo:hi]:
                o.write(">%s\n%s\n" % (n, t))
        hp = os.path.join(out, "hmms", "A_0_%d.hmm" % idx)
        mg.hmmbuild("dna", hp, fa)
        hmm_files.append("hmms/A_0_%d.hmm" % idx)
        hmm_index.append(idx)
        nseq.append(hi - lo)
        sub = Alignment()                      # algorithm.py:400-429
        for n, t in rows[lo:hi]:
            sub[n] = t
        retained[idx] = tuple(int(x) for x in sub.delete_all_gaps())
        cnt = [0] * sub.sequence_length()
        for t in sub.values():
            for c, ch in enumerate(t):
                cnt[c] += int(ch != '-')
        nongaps[idx] = tuple(cnt)
    # ---- search
    search = {}
    for hf in hmm_files:
        o = os.path.join(tmp, "s.out")
        mg.run([mg.HMMER + "/hmmsearch", "--cpu", "1", "--noali", "-E", "99999999", "-o", o, "--max",
                os.path.join(out, hf), qpath])
        search[hf] = {q: sc for q, (ev, sc) in evalHMMSearchOutput(o).items()}
    # ---- rank, weights, align, consensus (the reference's own functions)
    Configs.num_hmms = K
    Configs.use_weight = True
    Configs.hmmalignpath = mg.HMMER + "/hmmalign"
    Configs.outdir = tmp
    Configs.keeptemp = False
    Configs.log_path = None
    Configs.runtime_path = os.path.join(tmp, "runtime.txt")
    Configs.log = staticmethod(lambda *a, **k: None)
    Configs.runtime = staticmethod(lambda *a, **k: None)
    alignSubQueriesNew.subset_to_retained_columns = retained
    alignSubQueriesNew.subset_to_nongaps_per_column = nongaps

    class _Lock:
        def acquire(self):
            pass

        def release(self):
            pass

    class _Sub:
        pass
    index_to_hmm = {}
    for hf, idx in zip(hmm_files, hmm_index):
        s = _Sub()
        s.hmm_model_path = os.path.join(out, hf)
        index_to_hmm[idx] = s
    size_of = dict(zip(hmm_index, nseq))
    weights, merged, queries, ignored = {}, {}, [], []
    for qi, (qn, qs) in enumerate(zip(names, seqs)):
        scores = [(idx, search[hf][qn]) for hf, idx in zip(hmm_files, hmm_index) if qn in search[hf]]
        if not scores:
            ignored.append(qn)                 # results_handler.py:133-141: queries without weights are set aside
            continue
        ranked = sorted(scores, key=lambda x: x[1], reverse=True)
        idxs = [x[0] for x in ranked]
        w = calculateWeights((qn, idxs, [x[1] for x in ranked], [size_of[i] for i in idxs]))[qn]
        weights[qn] = [[int(i), float(x)] for i, x in w]
        q, _, _ = alignSubQueriesNew('unused', B, index_to_hmm, _Lock(), 120, qn, qs, w, qi)
        if len(q):
            merged[qn] = q[qn]
            queries.append(q)
        else:
            ignored.append(qn)
    # ---- final transitive merge
    Configs.output_path = os.path.join(tmp, "out.fasta")
    mergeAlignmentsCollapsed(bpath, queries, {}, None)
    final = {}
    for tag, p in (("full", os.path.join(tmp, "out.fasta")), ("masked", os.path.join(tmp, "out.masked.fasta"))):
        data = open(p, "rb").read()
        final[tag] = hashlib.sha256(data).hexdigest()
        with gzip.open(os.path.join(out, "merged.%s.fasta.gz" % tag), "wb", compresslevel=9) as f:
            f.write(data)
    for hf in hmm_files:
        hp = os.path.join(out, hf)
        with open(hp, "rb") as fi, gzip.open(hp + ".gz", "wb", compresslevel=9) as fo:
            fo.write(fi.read())
        os.remove(hp)
    gold = {"case": "example_e2e", "alphabet": "dna", "k": K, "hmm_files": hmm_files, "hmm_index": hmm_index,
            "nseq": nseq, "queries": names, "search": {hf: {q: {"score": sc} for q, sc in v.items()} for hf, v in search.items()},
            "weights": weights, "merged": merged, "ignored": ignored, "final_sha256": final,
            "retained": {str(k): list(v) for k, v in retained.items()},
            "nongaps": {str(k): list(v) for k, v in nongaps.items()}, "backbone_length": B,
            "align": {}, "search_nonull2": {}}
    with gzip.open(os.path.join(out, "golden.json.gz"), "wt") as f:
        json.dump(gold, f, separators=(",", ":"), sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(scratch, ignore_errors=True)
    print("example_e2e: %d HMMs x %d queries, %d reported pairs, %d queries aligned, %d ignored; final %s"
          % (len(hmm_files), len(names), sum(len(v) for v in search.values()), len(merged), len(ignored), final))


if __name__ == "__main__":
    main()
