#!/usr/bin/env python3
"""Time the REFERENCE's real CPU path on bench.py's inputs.

RUNS ONLY IN THE BUILD CONTAINER: it executes the HMMER 3.1b2 binaries bundled with the
reference (/root/reference/witch_msa/tools/magus/tools/hmmer) with the reference's exact command
lines and process scheme, on a subsample of the same seeded workload bench.py generates:

  search : one `hmmsearch --cpu 1 --noali -E 99999999 -o F --max HMM CHUNK` process per
           (HMM, query chunk), chunks = lcm(#HMMs, #cpus) // #HMMs, under a pool of <cpus>
           workers (witch_msa/gcmm/algorithm.py:280-284, 526-537), output parsed as
           evalHMMSearchOutput does (:579-605)
  weights: rank + calculateWeights + top-k (gcmm/loader.py:325-330, weighting.py:58-74)
  align  : one `hmmalign -o OUT HMM c1.fasta` process per (query, kept HMM) with the 0.999
           prefix rule (gcmm/aligner.py:58-63, 96-100), Stockholm row decoded (:126-142)

DP cost is linear in (query, HMM) pairs, so queries/s of the subsample is the figure of the
full workload (stated in the JSON).  Writes profiles/cpu_reference_<workload>.json, which
bench.py echoes as cpu_baseline.reference next to the float64 port it times live.

    python tests/tools/time_reference_cpu.py --workload dna_100k_x200 --nq 1000 --cpus 8
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HMMER = "/root/reference/witch_msa/tools/magus/tools/hmmer"


def _hmmsearch(args):
    hmm, chunk, out = args
    subprocess.run([HMMER + "/hmmsearch", "--cpu", "1", "--noali", "-E", "99999999", "-o", out, "--max", hmm, chunk],
                   check=True, stdout=subprocess.DEVNULL)
    from tests.refparse import evalHMMSearchOutput
    res = evalHMMSearchOutput(out)
    open(out, "w").write(str(res))           # algorithm.py:535-537
    return out


def _hmmalign(args):
    hmm, text, name, wd = args
    os.makedirs(wd, exist_ok=True)
    fa, out = os.path.join(wd, "c1.fasta"), os.path.join(wd, "hmmalign.out")
    open(fa, "w").write(">%s\n%s\n" % (name, text))
    subprocess.run([HMMER + "/hmmalign", "-o", out, hmm, fa], check=True, stdout=subprocess.DEVNULL)
    row = "".join(l.split()[1] for l in open(out) if l.strip() and not l.startswith("#") and l.strip() != "//")
    cols, reg = [], 0
    for ch in row.replace(".", "-"):          # aligner.py:126-142
        if ch == "-":
            reg += 1
        elif ch.islower():
            cols.append(-1)
        else:
            cols.append(reg)
            reg += 1
    return len(cols)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dna_100k_x200")
    ap.add_argument("--nq", type=int, default=1000, help="queries in the subsample (every (N/nq)-th query)")
    ap.add_argument("--cpus", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import bench
    from oracle import oracle as orc
    from witch_amd import synth
    from witch_amd.gcmm.algorithm import divide_to_equal_chunks, num_chunks_for
    wd = tempfile.mkdtemp(prefix="witch_refcpu_")
    try:
        fam, ehmm, names, seqs, k = bench.make_workload(args.workload, os.path.join(wd, "ehmm"))
        alph = bench.WORKLOADS[args.workload][0]
        n_total = len(seqs)
        step = max(1, n_total // args.nq)
        pick = list(range(0, n_total, step))[:args.nq]
        qn = [names[i] for i in pick]
        qt = [synth.to_text(seqs[i], alph) for i in pick]
        H = len(ehmm.paths)
        M = [h.M for h in ehmm.hmms]
        # ---- search
        chunks = [c for c in divide_to_equal_chunks(range(len(qn)), num_chunks_for(H, args.cpus)) if c]
        cpaths = []
        for i, c in enumerate(chunks):
            p = os.path.join(wd, "fragment_chunk_%d.fasta" % i)
            synth_names = [qn[j] for j in c]
            with open(p, "w") as f:
                for j in c:
                    f.write(">%s\n%s\n" % (qn[j], qt[j]))
            cpaths.append(p)
        jobs = [(hp, cp, os.path.join(wd, "hmmsearch.results.%d.fragment_chunk_%d" % (h, ci)))
                for h, hp in enumerate(ehmm.paths) for ci, cp in enumerate(cpaths)]
        t0 = time.time()
        with ProcessPoolExecutor(args.cpus) as pool:
            outs = list(pool.map(_hmmsearch, jobs))
        t_search = time.time() - t0
        # ---- rank + weights (loader.py:277-332, weighting.py:58-74)
        t0 = time.time()
        ranks = {}
        for (hp, cp, out), h in zip(jobs, [h for h in range(H) for _ in cpaths]):
            for taxon, sc in eval(open(out).read()).items():
                ranks.setdefault(taxon, []).append((ehmm.index[h], sc[1]))
        nseq_of = dict(zip(ehmm.index, ehmm.nseq))
        path_of = dict(zip(ehmm.index, ehmm.paths))
        weights = {}
        for taxon, sc in ranks.items():
            sc = sorted(sc, key=lambda x: x[1], reverse=True)
            idxs = [x[0] for x in sc]
            weights[taxon] = orc.calculate_weights(idxs, [x[1] for x in sc], [nseq_of[i] for i in idxs], k)
        t_weights = time.time() - t0
        # ---- align
        text_of = dict(zip(qn, qt))
        ajobs = []
        for qi, taxon in enumerate(qn):
            w = weights.get(taxon)
            if not w:
                continue
            for r, (i, _) in enumerate(w[:orc.adaptive_cut(w)]):
                ajobs.append((path_of[i], text_of[taxon], taxon, os.path.join(wd, "constraints", str(qi), str(r))))
        t0 = time.time()
        with ProcessPoolExecutor(args.cpus) as pool:
            n_res = sum(pool.map(_hmmalign, ajobs, chunksize=8))
        t_align = time.time() - t0
        total = t_search + t_weights + t_align
        rec = {
            "workload": args.workload, "kind": "reference",
            "what": "HMMER 3.1b2 binaries bundled with the reference, the reference's command lines and "
                    "one-process-per-task scheme under a %d-worker pool" % args.cpus,
            "host": "build container (%s), %d cores used" % (_cpu_model(), args.cpus),
            "cores": args.cpus,
            "n_queries_sampled": len(qn), "n_queries_workload": n_total, "n_hmms": H,
            "model_len_min": int(min(M)), "model_len_max": int(max(M)), "k": k,
            "pairs_scored": len(qn) * H, "pairs_reported": int(sum(len(v) for v in ranks.values())),
            "pairs_aligned": len(ajobs), "residues_aligned": int(n_res),
            "seconds": {"search": round(t_search, 2), "weights": round(t_weights, 2), "align": round(t_align, 2),
                        "total": round(total, 2)},
            "queries_per_s": round(len(qn) / total, 4),
            "queries_per_s_per_core": round(len(qn) / total / args.cpus, 4),
            "sample": "every %d-th query of the seeded workload (%d of %d) x all %d HMMs" % (step, len(qn), n_total, H),
        }
        out = args.out or os.path.join(ROOT, "profiles", "cpu_reference_%s.json" % args.workload)
        json.dump(rec, open(out, "w"), indent=1)
        print(json.dumps(rec))
    finally:
        shutil.rmtree(wd, ignore_errors=True)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


if __name__ == "__main__":
    main()
