#!/usr/bin/env python3
"""Regenerate the golden fixtures under tests/golden/.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).  It executes the
reference's bundled HMMER 3.1b2 binaries with the reference's exact command
lines and imports the reference's own Python functions (from a scratch copy under
/tmp, because importing witch_msa writes files next to the package), and stores
ONLY data: HMM text files produced by hmmbuild, query FASTA, and JSON with
scores / envelopes / weights / aligned columns.  No reference source is copied.

    hmmbuild  : witch_msa/gcmm/algorithm.py:463-470
    hmmsearch : witch_msa/gcmm/algorithm.py:526-532, parsed by evalHMMSearchOutput (:579-605)
    weights   : witch_msa/gcmm/weighting.py:58-74 (calculateWeights)
    hmmalign  : witch_msa/gcmm/aligner.py:33-148 (getBackbones, use_gcm=False)
"""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
HMMER = os.path.join(REF, "witch_msa/tools/magus/tools/hmmer")
sys.path.insert(0, REPO)

from witch_amd import synth  # noqa: E402


def import_reference():
    scratch = tempfile.mkdtemp(prefix="refcopy_")
    shutil.copytree(os.path.join(REF, "witch_msa"), os.path.join(scratch, "witch_msa"))
    stubs = os.path.join(scratch, "stubs", "dendropy")
    os.makedirs(os.path.join(stubs, "datamodel"))
    open(os.path.join(stubs, "__init__.py"), "w").write(
        "class Tree: pass\nclass Taxon: pass\nclass DataSet: pass\nclass treecalc: pass\n")
    open(os.path.join(stubs, "datamodel", "__init__.py"), "w").write("")
    open(os.path.join(stubs, "datamodel", "taxonmodel.py"), "w").write("class Taxon: pass\n")
    open(os.path.join(stubs, "datamodel", "treemodel.py"), "w").write(
        "class Tree: pass\nclass Node: pass\nclass Edge: pass\n")
    os.environ["HOME"] = os.path.join(scratch, "home")
    os.makedirs(os.environ["HOME"])
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(scratch, "stubs"))
    sys.path.insert(0, scratch)
    from witch_msa.configs import Configs
    from witch_msa.gcmm.algorithm import evalHMMSearchOutput
    from witch_msa.gcmm.weighting import calculateWeights
    from witch_msa.gcmm.aligner import getBackbones, alignSubQueriesNew
    from witch_msa.helpers.alignment_tools import Alignment
    return scratch, Configs, evalHMMSearchOutput, calculateWeights, getBackbones, alignSubQueriesNew, Alignment


def run(cmd):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def hmmbuild(mol, out_hmm, in_fasta):
    # algorithm.py:463-469
    run([HMMER + "/hmmbuild", "--cpu", "1", "--" + mol, "--ere", "0.59", "--symfrac", "0.0",
         "--informat", "afa", "-o", "/dev/null", out_hmm, in_fasta])
    # drop the DATE line so regenerated fixtures are byte-stable
    lines = [l for l in open(out_hmm) if not l.startswith("DATE")]
    open(out_hmm, "w").writelines(lines)


def parse_domtbl(path):
    dom = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.split()
        # target, acc, tlen, query, acc, qlen, E, score, bias, #, of, cE, iE, score, bias, hmmfrom, hmmto, alifrom, alito, envfrom, envto, acc
        dom.setdefault(f[0], []).append([int(f[19]), int(f[20]), float(f[13]), float(f[14])])
    return dom


class Case:
    def __init__(self, name, alphabet, k):
        self.name, self.alphabet, self.k = name, alphabet, k
        self.dir = os.path.join(HERE, name)
        shutil.rmtree(self.dir, ignore_errors=True)
        os.makedirs(os.path.join(self.dir, "hmms"))
        self.hmm_files, self.hmm_index, self.nseq = [], [], []
        self.qnames, self.qseqs = [], []
        self.backbone = None          # list of (name, aligned text) = the backbone alignment
        self.subset_rows = []         # per HMM: (lo, hi) rows of the backbone it was built from

    def add_hmm(self, path, index, nseq):
        self.hmm_files.append(os.path.relpath(path, self.dir))
        self.hmm_index.append(index)
        self.nseq.append(nseq)

    def finish(self, ref):
        scratch, Configs, evalHMMSearchOutput, calculateWeights, getBackbones, alignSubQueriesNew, Alignment = ref
        qpath = os.path.join(self.dir, "queries.fasta")
        synth.write_fasta(qpath, self.qnames, self.qseqs, self.alphabet)
        tmp = tempfile.mkdtemp(prefix="golden_")
        search, nonull2 = {}, {}
        for hf in self.hmm_files:
            hp = os.path.join(self.dir, hf)
            out, dt = os.path.join(tmp, "s.out"), os.path.join(tmp, "d.tbl")
            # the reference's command line (algorithm.py:526-532) + --domtblout for envelopes
            run([HMMER + "/hmmsearch", "--cpu", "1", "--noali", "-E", "99999999", "-o", out, "--max",
                 "--domtblout", dt, hp, qpath])
            # bias column read here; score/evalue through the reference's own parser
            bias = {}
            started = False
            for line in open(out):
                s = line.strip()
                if not started and s.startswith("E-value"):
                    started = True
                elif started and s == "":
                    break
                elif started and "--" not in s and len(s.split()) >= 9:
                    bias[s.split()[8]] = float(s.split()[2])
            res = evalHMMSearchOutput(out)
            dom = parse_domtbl(dt)
            search[hf] = {q: {"evalue": ev, "score": sc, "bias": bias[q], "dom": dom.get(q, [])}
                          for q, (ev, sc) in res.items()}
            run([HMMER + "/hmmsearch", "--cpu", "1", "--noali", "-E", "99999999", "-o", out, "--max",
                 "--nonull2", hp, qpath])
            nonull2[hf] = {q: sc for q, (ev, sc) in evalHMMSearchOutput(out).items()}
        # ranking + weights with the reference's calculateWeights (weighting.py:58-74)
        Configs.num_hmms = self.k
        Configs.use_weight = True
        Configs.hmmalignpath = HMMER + "/hmmalign"
        weights, align, merged = {}, {}, {}
        retained, nongaps = {}, {}
        if self.backbone is not None:
            # algorithm.py:400-429: retained columns / non-gap counts of each subset alignment
            for idx, (lo, hi) in zip(self.hmm_index, self.subset_rows):
                sub = Alignment()
                for name, text in self.backbone[lo:hi]:
                    sub[name] = text
                retained[idx] = tuple(int(x) for x in sub.delete_all_gaps())
                cnt = [0] * sub.sequence_length()
                for text in sub.values():
                    for c, ch in enumerate(text):
                        cnt[c] += int(ch != '-')
                nongaps[idx] = tuple(cnt)
            alignSubQueriesNew.subset_to_retained_columns = retained
            alignSubQueriesNew.subset_to_nongaps_per_column = nongaps
            Configs.outdir = tmp
            Configs.keeptemp = False
            Configs.log_path = None
            Configs.runtime_path = os.path.join(tmp, 'runtime.txt')

            class _Lock:
                def acquire(self):
                    pass

                def release(self):
                    pass

        class _Sub:
            pass
        index_to_hmm = {}
        for hf, idx in zip(self.hmm_files, self.hmm_index):
            s = _Sub()
            s.hmm_model_path = os.path.join(self.dir, hf)
            index_to_hmm[idx] = s
        size_of = dict(zip(self.hmm_index, self.nseq))
        for qi, (qn, qs) in enumerate(zip(self.qnames, self.qseqs)):
            scores = []
            for hf, idx in zip(self.hmm_files, self.hmm_index):
                if qn in search[hf]:
                    scores.append((idx, search[hf][qn]["score"]))
            if not scores:
                continue
            # loader.py:325-330 (stable sort desc); arrival order here = HMM list order
            ranked = sorted(scores, key=lambda x: x[1], reverse=True)
            idxs = [x[0] for x in ranked]
            bits = [x[1] for x in ranked]
            sizes = [size_of[i] for i in idxs]
            w = calculateWeights((qn, idxs, bits, sizes))[qn]
            weights[qn] = [[int(i), float(x)] for i, x in w]
            # aligner.py:33-148 with use_gcm=False
            q1 = os.path.join(tmp, "c1.fasta")
            text = qs if isinstance(qs, str) else synth.to_text(qs, self.alphabet)
            open(q1, "w").write(">%s\n%s\n" % (qn, text))
            ret = getBackbones(index_to_hmm, qn, qi, text, q1, w, os.path.join(tmp, "wd%d" % qi),
                               os.path.join(tmp, "bb"), use_gcm=False)
            ret_str, weights_map, cols = ret
            align[qn] = {"ret_str": ret_str.replace(self.dir, "."),
                         "cols": {str(i): [int(c) for c in v] for i, v in cols.items()},
                         "order": [int(i) for i in cols.keys()]}
            if self.backbone is not None:
                # aligner.py:350-538: weighted consensus DP + compressInsertions (next row #1)
                q, _, _ = alignSubQueriesNew('unused', len(self.backbone[0][1]), index_to_hmm, _Lock(), 120,
                                             qn, text, w, qi)
                merged[qn] = q[qn] if len(q) else None
        shutil.rmtree(tmp, ignore_errors=True)
        gold = {"case": self.name, "alphabet": self.alphabet, "k": self.k,
                "hmm_files": self.hmm_files, "hmm_index": self.hmm_index, "nseq": self.nseq,
                "queries": self.qnames, "search": search, "search_nonull2": nonull2,
                "weights": weights, "align": align, "merged": merged,
                "retained": {str(k): list(v) for k, v in retained.items()},
                "nongaps": {str(k): list(v) for k, v in nongaps.items()},
                "backbone_length": (len(self.backbone[0][1]) if self.backbone else 0)}
        with gzip.open(os.path.join(self.dir, "golden.json.gz"), "wt") as f:
            json.dump(gold, f, separators=(",", ":"), sort_keys=True)
        n_rep = sum(len(v) for v in search.values())
        print("%s: %d HMMs x %d queries, %d reported pairs, %d aligned queries" %
              (self.name, len(self.hmm_files), len(self.qnames), n_rep, len(align)))


def degenerate(seq_text, rng, alphabet, n):
    codes = "RYMKSWHBVDN" if alphabet != "amino" else "BJZOUX"
    s = list(seq_text)
    for p in rng.choice(len(s), size=min(n, len(s)), replace=False):
        s[p] = codes[int(rng.integers(len(codes)))]
    return "".join(s)


def family_case(ref, name, alphabet, seed, root_len, n_leaves, n_sub, sub_rate, indel_rate, k,
                n_homolog, qlen, use_hmmbuild):
    rng = np.random.default_rng(seed + 7)
    fam = synth.make_family(seed, root_len, n_leaves, alphabet, sub_rate, indel_rate)
    c = Case(name, alphabet, k)
    subs = synth.bfs_subsets(n_leaves, n_sub)
    sym = synth.symbols(alphabet) + '-'
    c.backbone = [(fam.names[i], ''.join(sym[int(x)] for x in fam.msa[i])) for i in range(n_leaves)]
    c.subset_rows = list(subs)
    for idx, (lo, hi) in enumerate(subs):
        hp = os.path.join(c.dir, "hmms", "A_0_%d.hmm" % idx)
        if use_hmmbuild:
            fa = os.path.join(c.dir, "sub.fasta")
            synth.write_msa_fasta(fa, fam, lo, hi)
            hmmbuild("amino" if alphabet == "amino" else "dna", hp, fa)
            os.remove(fa)
        else:
            synth.write_hmm(synth.build_hmm(fam, lo, hi, "A_0_%d" % idx), hp)
        c.add_hmm(hp, idx, hi - lo)
    names, seqs = synth.make_queries(fam, seed + 1, n_homolog, qlen, sub_rate=0.05)
    texts = [synth.to_text(s, alphabet) for s in seqs]
    K = 20 if alphabet == "amino" else 4
    bg = synth.background(alphabet)
    # degenerate residues
    for t in range(6):
        texts.append(degenerate(texts[t], rng, alphabet, 1 + t))
        names.append("deg%02d" % t)
    # unrelated random sequences
    for t in range(5):
        L = int(rng.integers(30, qlen[1] if isinstance(qlen, tuple) else qlen))
        texts.append(synth.to_text(rng.choice(K, size=L, p=bg).astype(np.int8), alphabet))
        names.append("rnd%02d" % t)
    # very short queries
    for t in range(4):
        L = int(rng.integers(4, 25))
        leaf = fam.leaf_seq(int(rng.integers(n_leaves)))
        s = int(rng.integers(0, len(leaf) - L))
        texts.append(synth.to_text(leaf[s:s + L], alphabet))
        names.append("short%02d" % t)
    # chimeras: two windows from different places joined by a random spacer (true multi-domain)
    for t in range(5):
        leaf = fam.leaf_seq(int(rng.integers(n_leaves)))
        w = min(40, len(leaf) // 3)
        a = int(rng.integers(len(leaf) // 2, len(leaf) - w))
        b = int(rng.integers(0, len(leaf) // 2 - w))
        spacer = rng.choice(K, size=int(rng.integers(5, 30)), p=bg).astype(np.int8)
        texts.append(synth.to_text(np.concatenate([leaf[a:a + w], spacer, leaf[b:b + w]]), alphabet))
        names.append("chim%02d" % t)
    if alphabet == "amino":
        # low-complexity / biased composition (drives null2)
        for t in range(4):
            leaf = fam.leaf_seq(int(rng.integers(n_leaves)))
            lc = "".join(rng.choice(list("QNSG"), size=40))
            texts.append(synth.to_text(leaf[:50], alphabet) + lc)
            names.append("lowc%02d" % t)
    c.qnames, c.qseqs = names, texts
    c.finish(ref)


def example_case(ref):
    """The reference's own example data: first 30 backbone sequences -> one hmmbuild model;
    first 120 example fragments as queries (examples/data/*)."""
    c = Case("example_sub30", "dna", 1)
    fa = os.path.join(c.dir, "sub.fasta")
    n = 0
    with gzip.open(os.path.join(REF, "examples/data/backbone.aln.fasta.gz"), "rt") as f, open(fa, "w") as o:
        for line in f:
            if line.startswith(">"):
                n += 1
                if n > 30:
                    break
            o.write(line)
    hp = os.path.join(c.dir, "hmms", "A_0_0.hmm")
    hmmbuild("dna", hp, fa)
    os.remove(fa)
    c.add_hmm(hp, 0, 30)
    names, seqs = [], []
    for line in open(os.path.join(REF, "examples/data/unaligned_frag.fasta")):
        line = line.strip()
        if line.startswith(">"):
            names.append(line[1:].split()[0])
            seqs.append("")
        elif line:
            seqs[-1] += line.upper()     # alignment_tools.py:730-731 upper-cases on read
    c.qnames, c.qseqs = names[:120], seqs[:120]
    c.finish(ref)


def example_ehmm_case(ref):
    """Mini-eHMM over the reference's own example backbone (500 sequences, 2574 columns): nested
    subsets of 500 / 125 / 62 sequences -> hmmbuild models of 2574 / ~1500 / ~1300 nodes; the first
    40 example fragments as queries; includes the weighted-consensus strings of alignSubQueriesNew.
    The HMM text files are stored gzipped (they are 0.4-0.6 MB each)."""
    c = Case("example_ehmm", "dna", 3)
    rows = []
    with gzip.open(os.path.join(REF, "examples/data/backbone.aln.fasta.gz"), "rt") as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                rows.append([line[1:].split()[0], ""])
            elif line:
                rows[-1][1] += line.upper()
    c.backbone = [(n, t) for n, t in rows]
    c.subset_rows = [(0, len(rows)), (0, 125), (0, 62)]
    for idx, (lo, hi) in enumerate(c.subset_rows):
        fa = os.path.join(c.dir, "sub.fasta")
        with open(fa, "w") as o:
            for n, t in rows[lo:hi]:
                o.write(">%s\n%s\n" % (n, t))
        hp = os.path.join(c.dir, "hmms", "A_0_%d.hmm" % idx)
        hmmbuild("dna", hp, fa)
        os.remove(fa)
        c.add_hmm(hp, idx, hi - lo)
    names, seqs = [], []
    for line in open(os.path.join(REF, "examples/data/unaligned_frag.fasta")):
        line = line.strip()
        if line.startswith(">"):
            names.append(line[1:].split()[0])
            seqs.append("")
        elif line:
            seqs[-1] += line.upper()
    c.qnames, c.qseqs = names[:40], seqs[:40]
    c.finish(ref)
    for hf in c.hmm_files:      # gzip the big model files
        hp = os.path.join(c.dir, hf)
        with open(hp, "rb") as fi, gzip.open(hp + ".gz", "wb", compresslevel=9) as fo:
            fo.write(fi.read())
        os.remove(hp)


def main():
    ref = import_reference()
    family_case(ref, "dna_hmmbuild", "dna", 11, 120, 32, 8, 0.04, 0.004, 4, 30, (60, 110), True)
    family_case(ref, "dna_synth", "dna", 12, 150, 16, 4, 0.03, 0.003, 4, 20, 100, False)
    family_case(ref, "amino_hmmbuild", "amino", 13, 90, 16, 4, 0.08, 0.004, 4, 20, (40, 120), True)
    example_case(ref)
    example_ehmm_case(ref)
    shutil.rmtree(ref[0], ignore_errors=True)


if __name__ == "__main__":
    main()
