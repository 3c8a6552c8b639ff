"""Synthetic sequence families, eHMMs and query sets for tests and bench.

This is workload *generation* (SURVEY.md section 8d, configs 2-5), not a
re-implementation of the reference's eHMM construction: WITCH builds its HMMs
with ``hmmbuild`` (witch_msa/gcmm/algorithm.py:463-470), which cannot travel to
the GPU box.  The generator below evolves a family on a balanced binary tree,
decomposes the leaves into nested subsets (the shape UPP's centroid
decomposition yields, witch_msa/gcmm/algorithm.py:99-108) and writes one
HMMER3/f text model per subset, so that the whole product path - including the
HMM text parser - is exercised on files an unmodified ``hmmsearch``/``hmmalign``
can read too.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

DNA = "ACGT"
AMINO = "ACDEFGHIKLMNPQRSTVWY"

# HMMER's amino background
AMINO_BG = np.array([
    .0787945, .0151600, .0535222, .0668298, .0397062, .0695071, .0229198,
    .0590092, .0594422, .0963728, .0237718, .0414386, .0482904, .0395639,
    .0540978, .0683364, .0540687, .0673417, .0114135, .0304133])
AMINO_BG = AMINO_BG / AMINO_BG.sum()


def background(alphabet: str) -> np.ndarray:
    if alphabet == "amino":
        return AMINO_BG.copy()
    return np.full(4, 0.25)


def symbols(alphabet: str) -> str:
    return AMINO if alphabet == "amino" else DNA


@dataclass
class Family:
    alphabet: str                 # "dna" | "amino"
    msa: np.ndarray               # int8 [n_leaves, n_cols], -1 = gap
    names: list

    @property
    def n_leaves(self):
        return self.msa.shape[0]

    def leaf_seq(self, i: int) -> np.ndarray:
        row = self.msa[i]
        return row[row >= 0]


def make_family(seed: int, root_len: int, n_leaves: int, alphabet: str = "dna",
                sub_rate: float = 0.03, indel_rate: float = 0.002) -> Family:
    """Evolve ``n_leaves`` (power of two) sequences down a balanced binary tree.

    Per branch every residue is substituted with probability ``sub_rate`` and
    every site suffers a single-site deletion or insertion with probability
    ``indel_rate`` (half each).  Homology is tracked with sortable column keys
    so the true alignment falls out at the end.
    """
    assert n_leaves & (n_leaves - 1) == 0, "n_leaves must be a power of two"
    rng = np.random.default_rng(seed)
    bg = background(alphabet)
    K = len(bg)
    keys = np.arange(root_len, dtype=np.float64)
    res = rng.choice(K, size=root_len, p=bg).astype(np.int8)
    level = [(keys, res)]
    while len(level) < n_leaves:
        nxt = []
        for keys, res in level:
            for _child in range(2):
                k, r = keys.copy(), res.copy()
                n = len(r)
                # substitutions
                hit = rng.random(n) < sub_rate
                if hit.any():
                    r[hit] = rng.choice(K, size=int(hit.sum()), p=bg)
                # deletions
                dele = rng.random(n) < indel_rate * 0.5
                if dele.any() and (~dele).sum() > 8:
                    k, r = k[~dele], r[~dele]
                    n = len(r)
                # insertions (after position p)
                ins = np.nonzero(rng.random(n) < indel_rate * 0.5)[0]
                if len(ins):
                    nk = np.empty(len(ins))
                    for t, p in enumerate(ins):
                        hi = k[p + 1] if p + 1 < n else k[p] + 1.0
                        nk[t] = k[p] + (hi - k[p]) * (0.25 + 0.5 * rng.random())
                    nr = rng.choice(K, size=len(ins), p=bg).astype(np.int8)
                    k = np.concatenate([k, nk])
                    r = np.concatenate([r, nr])
                    order = np.argsort(k, kind="stable")
                    k, r = k[order], r[order]
                nxt.append((k, r))
        level = nxt
    allkeys = np.unique(np.concatenate([k for k, _ in level]))
    msa = np.full((n_leaves, len(allkeys)), -1, dtype=np.int8)
    for i, (k, r) in enumerate(level):
        msa[i, np.searchsorted(allkeys, k)] = r
    names = ["L%05d" % i for i in range(n_leaves)]
    return Family(alphabet, msa, names)


def bfs_subsets(n_leaves: int, n_subsets: int, min_size: int = 2):
    """Nested leaf subsets in BFS order over the balanced tree: [0,n), halves, ..."""
    out, queue = [], [(0, n_leaves)]
    while queue and len(out) < n_subsets:
        lo, hi = queue.pop(0)
        out.append((lo, hi))
        if hi - lo >= 2 * min_size:
            mid = (lo + hi) // 2
            queue.append((lo, mid))
            queue.append((mid, hi))
    return out


@dataclass
class SynthHMM:
    name: str
    alphabet: str
    nseq: int
    M: int
    mat: np.ndarray        # [M+1, K] match emission probabilities (row 0 unused)
    t: np.ndarray          # [M+1, 7] MM MI MD IM II DM DD  (row 0 = begin node)
    map_cols: np.ndarray   # [M+1] 1-based alignment column of each match state
    nongaps: np.ndarray    # [M] non-gap count per retained column
    cons: str = ""


def _entropy_scale(counts: np.ndarray, bg: np.ndarray, target_bits: float) -> float:
    """Scale factor on counts so that mean relative entropy/column ~= target."""
    def mean_re(alpha):
        c = counts * alpha + bg[None, :] * 1.0
        p = c / c.sum(1, keepdims=True)
        return float(np.mean(np.sum(p * np.log2(p / bg[None, :]), axis=1)))
    if mean_re(1.0) <= target_bits:
        return 1.0
    lo, hi = 1e-4, 1.0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if mean_re(mid) > target_bits:
            hi = mid
        else:
            lo = mid
    return 0.5 * (lo + hi)


def build_hmm(fam: Family, lo: int, hi: int, name: str, ere_bits: float | None = None) -> SynthHMM:
    """Profile HMM from leaves [lo,hi): every non-all-gap column is a match state
    (the reference runs hmmbuild with --symfrac 0.0, algorithm.py:463-469)."""
    sub = fam.msa[lo:hi]
    bg = background(fam.alphabet)
    K = len(bg)
    if ere_bits is None:
        ere_bits = 0.59 if K == 20 else 0.59   # the reference passes --ere 0.59 for every molecule
    keep = np.nonzero((sub >= 0).any(axis=0))[0]
    a = sub[:, keep]
    n, M = a.shape
    present = a >= 0
    counts = np.zeros((M, K))
    for x in range(K):
        counts[:, x] = (a == x).sum(axis=0)
    alpha = _entropy_scale(counts, bg, ere_bits)
    c = counts * alpha + bg[None, :]
    mat = np.zeros((M + 1, K))
    mat[1:] = c / c.sum(1, keepdims=True)
    # transitions between consecutive match columns (no insert columns exist)
    t = np.zeros((M + 1, 7))
    p_now, p_nxt = present[:, :-1], present[:, 1:]
    mm = (p_now & p_nxt).sum(0) * alpha
    md = (p_now & ~p_nxt).sum(0) * alpha
    dm = (~p_now & p_nxt).sum(0) * alpha
    dd = (~p_now & ~p_nxt).sum(0) * alpha
    mi_prior, ii, im = 0.0032, 0.769, 0.231
    for k in range(1, M):
        tot_m = mm[k - 1] + md[k - 1] + 1.0
        pm = (mm[k - 1] + 0.9936) / tot_m
        pd = (md[k - 1] + 0.0032) / tot_m
        s = pm + pd + mi_prior
        t[k, 0], t[k, 1], t[k, 2] = pm / s, mi_prior / s, pd / s
        t[k, 3], t[k, 4] = im, ii
        tot_d = dm[k - 1] + dd[k - 1] + 1.0
        t[k, 5] = (dm[k - 1] + 0.7) / tot_d
        t[k, 6] = (dd[k - 1] + 0.3) / tot_d
    # begin node: B->M1, B->I0, B->D1, I0->M1, I0->I0, (D0->M1 = 1, D0->D1 = 0)
    first_gap = float((~present[:, 0]).sum()) * alpha
    tot = n * alpha + 1.0
    pd0 = (first_gap + 0.0032) / tot
    t[0] = [1.0 - pd0 - mi_prior, mi_prior, pd0, im, ii, 1.0, 0.0]
    # end node M: M->E, M->I, (M->D = 0), I->E..., D->E = 1
    t[M] = [1.0 - mi_prior, mi_prior, 0.0, im, ii, 1.0, 0.0]
    sym = symbols(fam.alphabet)
    cons = "".join(sym[int(np.argmax(mat[k]))].lower() for k in range(1, M + 1))
    map_cols = np.zeros(M + 1, dtype=np.int64)
    map_cols[1:] = keep + 1
    return SynthHMM(name, fam.alphabet, n, M, mat, t, map_cols, present.sum(0).astype(np.int64), cons)


def _fmt(p: float) -> str:
    if p <= 0.0:
        return "      *"
    return "%7.5f" % (-np.log(p) + 0.0)


def write_hmm(h: SynthHMM, path: str) -> None:
    """HMMER3/f text: all numbers are -ln p, '*' = 0."""
    sym = symbols(h.alphabet)
    K = len(sym)
    bg = background(h.alphabet)
    L = []
    L.append("HMMER3/f [3.1b2 | February 2015]")
    L.append("NAME  %s" % h.name)
    L.append("LENG  %d" % h.M)
    L.append("MAXL  %d" % (h.M + 64))
    L.append("ALPH  %s" % ("amino" if h.alphabet == "amino" else "DNA"))
    L.append("RF    no")
    L.append("MM    no")
    L.append("CONS  yes")
    L.append("CS    no")
    L.append("MAP   yes")
    L.append("DATE  Sat Oct  3 00:00:00 2026")
    L.append("NSEQ  %d" % h.nseq)
    L.append("EFFN  %f" % float(h.nseq))
    L.append("CKSUM 0")
    L.append("STATS LOCAL MSV      -10.0000  0.70000")
    L.append("STATS LOCAL VITERBI  -11.0000  0.70000")
    L.append("STATS LOCAL FORWARD   -5.0000  0.70000")
    L.append("HMM     " + "".join("     %s   " % c for c in sym))
    L.append("            m->m     m->i     m->d     i->m     i->i     d->m     d->d")
    compo = h.mat[1:].mean(axis=0)
    L.append("  COMPO   " + "  ".join(_fmt(p) for p in compo))
    ins = "          " + "  ".join(_fmt(p) for p in bg)
    L.append(ins)
    L.append("          " + "  ".join(_fmt(p) for p in h.t[0]))
    for k in range(1, h.M + 1):
        L.append("%7d   " % k + "  ".join(_fmt(p) for p in h.mat[k]) +
                 " %6d %s - - -" % (h.map_cols[k], h.cons[k - 1]))
        L.append(ins)
        tk = h.t[k].copy()
        L.append("          " + "  ".join(_fmt(p) for p in tk))
    L.append("//")
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")


@dataclass
class SynthEHMM:
    family: Family
    hmms: list = field(default_factory=list)        # SynthHMM
    paths: list = field(default_factory=list)
    index: list = field(default_factory=list)       # HMM index as in A_0_<idx>
    nseq: list = field(default_factory=list)


def make_ehmm(fam: Family, n_subsets: int, outdir: str, witch_layout: bool = True) -> SynthEHMM:
    """Write ``n_subsets`` models.  With ``witch_layout`` the files land where
    the reference expects them:
    <outdir>/root/A_0_<idx>/hmmbuild.model.A_0_<idx>."""
    e = SynthEHMM(fam)
    subs = bfs_subsets(fam.n_leaves, n_subsets)
    for idx, (lo, hi) in enumerate(subs):
        h = build_hmm(fam, lo, hi, "A_0_%d" % idx)
        if witch_layout:
            d = os.path.join(outdir, "root", "A_0_%d" % idx)
            os.makedirs(d, exist_ok=True)
            p = os.path.join(d, "hmmbuild.model.A_0_%d" % idx)
        else:
            os.makedirs(outdir, exist_ok=True)
            p = os.path.join(outdir, "A_0_%d.hmm" % idx)
        write_hmm(h, p)
        e.hmms.append(h)
        e.paths.append(p)
        e.index.append(idx)
        e.nseq.append(hi - lo)
    return e


def make_queries(fam: Family, seed: int, n: int, length, sub_rate: float = 0.05,
                 flank_frac: float = 0.0):
    """``n`` query fragments: windows of leaf sequences with extra substitutions.

    ``length`` is an int (exact length, configs 2-4) or a (lo, hi) tuple
    (uniform lengths; long queries are built by concatenating family windows
    with random-background flanks, config 5).  Returns (names, list of int8
    arrays of residue codes 0..K-1).
    """
    rng = np.random.default_rng(seed)
    bg = background(fam.alphabet)
    K = len(bg)
    leaves = [fam.leaf_seq(i) for i in range(fam.n_leaves)]
    names, seqs = [], []
    for q in range(n):
        L = int(length) if np.isscalar(length) else int(rng.integers(length[0], length[1] + 1))
        parts, have = [], 0
        while have < L:
            leaf = leaves[int(rng.integers(len(leaves)))]
            want = L - have
            if flank_frac > 0 and rng.random() < flank_frac:
                w = int(min(want, rng.integers(5, 60)))
                parts.append(rng.choice(K, size=w, p=bg).astype(np.int8))
            else:
                w = int(min(want, len(leaf)))
                s = int(rng.integers(0, len(leaf) - w + 1))
                parts.append(leaf[s:s + w].copy())
            have += w
        seq = np.concatenate(parts)[:L]
        hit = rng.random(L) < sub_rate
        if hit.any():
            seq[hit] = rng.choice(K, size=int(hit.sum()), p=bg)
        names.append("q%06d" % q)
        seqs.append(seq.astype(np.int8))
    return names, seqs


def to_text(seq: np.ndarray, alphabet: str) -> str:
    sym = np.frombuffer(symbols(alphabet).encode(), dtype=np.uint8)
    return sym[np.asarray(seq, dtype=np.int64)].tobytes().decode()


def write_fasta(path: str, names, seqs, alphabet: str) -> None:
    with open(path, "w") as f:
        for n, s in zip(names, seqs):
            f.write(">%s\n%s\n" % (n, s if isinstance(s, str) else to_text(s, alphabet)))


def write_msa_fasta(path: str, fam: Family, lo: int, hi: int) -> None:
    """Aligned FASTA of leaves [lo,hi) (all-gap columns kept; hmmbuild drops them)."""
    sym = symbols(fam.alphabet)
    lut = np.frombuffer((sym + "-").encode(), dtype=np.uint8)
    with open(path, "w") as f:
        for i in range(lo, hi):
            row = fam.msa[i].astype(np.int64)
            row[row < 0] = len(sym)
            f.write(">%s\n%s\n" % (fam.names[i], lut[row].tobytes().decode()))
