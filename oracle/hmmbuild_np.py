"""TEST INFRASTRUCTURE (oracle/): a numpy restatement of `hmmbuild --<mol> --ere 0.59 --symfrac 0.0
--informat afa` (HMMER 3.1b2 p7_Builder; reference call site witch_msa/gcmm/algorithm.py:463-470), written
first and independently of the product's C++ (witch_amd/csrc/wh_build.cpp).  Only tests/ may import it.

Pinned on the HMM files the bundled hmmbuild binary produced for the golden cases
(tests/test_hmmbuild_host.py).  Steps: Henikoff position-based weights; fragments by the first..last residue
span (< 0.5 alen); match columns = columns with a residue (symfrac 0); weighted float32 counts along the
implied paths (nothing counted into or out of missing data); entropy weighting with Easel's bisection
(bracket tested before it is narrowed, tolerance 0.01) towards max(ere, (45 - log2(2/(M(M+1))))/M); posterior
mean under HMMER's default nucleic mixture Dirichlet priors.  DNA / RNA only.
"""
import math

import numpy as np

# Easel's nucleic alphabet "ACGT-RYMKSWHBVDN*~": K = 4 canonical, gap, 11 degenerate incl. N, '*', missing '~'
_DNA_SYMS = "ACGT-RYMKSWHBVDN*~"
_DNA_DEGEN = {"R": "AG", "Y": "CT", "M": "AC", "K": "GT", "S": "CG", "W": "AT", "H": "ACT", "B": "CGT",
              "V": "ACG", "D": "AGT", "N": "ACGT"}
_AMINO_SYMS = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~"
_AMINO_DEGEN = {"B": "ND", "J": "IL", "Z": "QE", "O": "K", "U": "C", "X": "ACDEFGHIKLMNPQRSTVWY"}


class Alphabet:
    def __init__(self, kind):
        self.kind = kind
        if kind in ("dna", "rna"):
            self.syms, degen, self.K, self.name = _DNA_SYMS, _DNA_DEGEN, 4, "DNA" if kind == "dna" else "RNA"
            syn = {"U": "T", "X": "N", "I": "A", "_": "-", ".": "-"}
        elif kind == "amino":
            self.syms, degen, self.K, self.name = _AMINO_SYMS, _AMINO_DEGEN, 20, "amino"
            syn = {"_": "-", ".": "-"}
        else:
            raise ValueError("unknown alphabet %r" % kind)
        self.Kp = len(self.syms)
        self.gap, self.missing = self.K, self.Kp - 1
        self.code = np.full(256, 255, dtype=np.uint8)
        for i, c in enumerate(self.syms):
            self.code[ord(c)] = i
            self.code[ord(c.lower())] = i
        for a, b in syn.items():
            self.code[ord(a)] = self.code[ord(b)]
            self.code[ord(a.lower())] = self.code[ord(b)]
        # degeneracy matrix [Kp][K] and counts
        self.degen = np.zeros((self.Kp, self.K), dtype=np.float64)
        for i in range(self.K):
            self.degen[i, i] = 1.0
        for c, members in degen.items():
            for m in members:
                self.degen[self.syms.index(c), self.syms.index(m)] = 1.0
        self.ndegen = self.degen.sum(axis=1)

    def digitize(self, rows):
        ax = np.stack([self.code[np.frombuffer(r.encode("ascii"), dtype=np.uint8)] for r in rows])
        if (ax == 255).any():
            raise ValueError("character outside the %s alphabet in the alignment" % self.name)
        return ax

    def is_residue(self, ax):      # canonical or degenerate (Easel: x < K or K < x < Kp-2)
        return (ax < self.K) | ((ax > self.K) & (ax < self.Kp - 2))


# HMMER's default priors (p7_prior_CreateNucleic / p7_prior_CreateAmino): (mixture coefficients, alphas)
_NUC_PRIOR = dict(
    tm=(np.array([1.0]), np.array([[2.0, 0.1, 0.1]])),
    ti=(np.array([1.0]), np.array([[0.06, 0.2]])),
    td=(np.array([1.0]), np.array([[0.1, 0.2]])),
    em=(np.array([0.24, 0.26, 0.08, 0.42]),
        np.array([[0.16, 0.45, 0.12, 0.39], [0.09, 0.03, 0.09, 0.04], [1.29, 0.40, 6.58, 0.51], [1.74, 1.49, 1.57, 1.95]])),
    ei=(np.array([1.0]), np.array([[1.0, 1.0, 1.0, 1.0]])),
)


def _mixdchlet_mean(counts, pq, alpha):
    """Posterior mean of a multinomial under a mixture Dirichlet prior (esl_mixdchlet_MPParameters):
    counts [n, K] float64 -> probabilities [n, K]."""
    from scipy.special import gammaln
    c = counts[:, None, :]                                   # [n, 1, K]
    a = alpha[None, :, :]                                    # [1, Q, K]
    # log P(c | alpha_q) up to a term that is the same for every component
    lp = (gammaln(a.sum(-1)) - gammaln((c + a).sum(-1)) + (gammaln(c + a) - gammaln(a)).sum(-1))   # [n, Q]
    lp = lp + np.log(pq)[None, :]
    lp -= lp.max(axis=1, keepdims=True)
    mix = np.exp(lp)
    mix /= mix.sum(axis=1, keepdims=True)
    tot = c.sum(-1) + a.sum(-1)                              # [n, Q]
    p = (mix[:, :, None] * (c + a) / tot[:, :, None]).sum(axis=1)
    return p / p.sum(axis=1, keepdims=True)


class CountHMM:
    """Counts (then parameters) in HMMER's layout: t[0..M][7] = MM MI MD IM II DM DD, mat[0..M][K], ins[0..M][K]."""
    MM, MI, MD, IM, II, DM, DD = range(7)


def pb_weights(ax, abc):
    nseq, alen = ax.shape
    canon = ax < abc.K
    wgt = np.zeros(nseq, dtype=np.float64)
    for x in range(abc.K):
        is_x = ax == x
        n_x = is_x.sum(axis=0)                               # per column
        if x == 0:
            ntypes = np.zeros(alen, dtype=np.int64)
        ntypes = ntypes + (n_x > 0)
    for x in range(abc.K):
        is_x = ax == x
        n_x = is_x.sum(axis=0).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            per_col = np.where(n_x > 0, 1.0 / (ntypes * n_x), 0.0)
        wgt += (is_x * per_col[None, :]).sum(axis=1)
    rlen = (ax < abc.K).sum(axis=1)                         # canonical residues only (probed with hmmbuild)
    wgt = np.where(rlen > 0, wgt / np.maximum(rlen, 1), wgt)
    del canon
    wgt = wgt / wgt.sum() * nseq
    return wgt


def mark_fragments(ax, abc, fragthresh=0.5):
    ax = ax.copy()
    nseq, alen = ax.shape
    for i in range(nseq):
        row = ax[i]
        res = np.nonzero(row != abc.gap)[0]
        lo, hi = (res[0], res[-1]) if len(res) else (alen, -1)
        if not (hi - lo + 1 <= fragthresh * alen):         # the span first..last residue, in columns
            continue
        row[:lo][row[:lo] == abc.gap] = abc.missing
        row[hi + 1:][row[hi + 1:] == abc.gap] = abc.missing
    return ax


def count_model(ax, wgt, abc, symfrac=0.0):
    """Fast model construction + weighted counting.  Returns (matcols [M] 0-based, t [M+1,7], mat [M+1,K], ins [M+1,K])
    as float32 arrays (HMMER counts in float)."""
    nseq, alen = ax.shape
    w32 = wgt.astype(np.float32)
    isres = abc.is_residue(ax)
    isgap = ax == abc.gap
    r = (isres * wgt[:, None]).sum(axis=0)
    tot = ((isres | isgap) * wgt[:, None]).sum(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        match = (r > 0) & (np.where(tot > 0, r / np.where(tot > 0, tot, 1.0), 0.0) >= symfrac)
    matcols = np.nonzero(match)[0]
    M = len(matcols)
    K = abc.K
    t = np.zeros((M + 1, 7), dtype=np.float32)
    mat = np.zeros((M + 1, K), dtype=np.float32)
    ins = np.zeros((M + 1, K), dtype=np.float32)
    kcol = np.cumsum(match)                                   # node index (1-based) at/after each column
    ST_M, ST_I, ST_D, ST_X = 0, 1, 2, 3
    for idx in range(nseq):
        row = ax[idx]
        wt = w32[idx]
        # state path: (state, k, residue)
        states = []
        for apos in range(alen):
            x = row[apos]
            k = int(kcol[apos])
            if isres[idx, apos]:
                states.append((ST_M if match[apos] else ST_I, k, x))
            elif match[apos] and x == abc.gap:
                states.append((ST_D, k, 255))
            elif x == abc.missing:
                if not states or states[-1][0] != ST_X:
                    states.append((ST_X, k, 255))
        # emissions
        for st, k, x in states:
            if st == ST_M:
                mat[k] += (wt * (abc.degen[x] / abc.ndegen[x])).astype(np.float32)
            elif st == ST_I:
                ins[k] += (wt * (abc.degen[x] / abc.ndegen[x])).astype(np.float32)
        # transitions: B = node 0 "match" state; E closes with the M->M / I->M / D->M slot of the last state
        prev = (ST_M, 0, 255)
        seq = states + [(ST_M, M + 1, 255)]                   # E as a pseudo match after node M
        for cur in seq:
            st2 = cur[0]
            st, k = prev[0], prev[1]
            if st2 == ST_X or st == ST_X:
                prev = cur
                continue
            if st == ST_M:
                t[k, CountHMM.MM if st2 == ST_M else CountHMM.MI if st2 == ST_I else CountHMM.MD] += wt
            elif st == ST_I:
                t[k, CountHMM.IM if st2 == ST_M else CountHMM.II] += wt     # I -> D does not exist in Plan 7
            else:
                t[k, CountHMM.DM if st2 == ST_M else CountHMM.DD] += wt     # D -> I neither
            prev = cur
    return matcols, t, mat, ins


def parameterize(t, mat, ins, prior):
    """Counts (float32) -> probabilities (float32), p7_ParameterEstimation."""
    M = t.shape[0] - 1
    tp = np.zeros_like(t)
    c = t.astype(np.float64)
    tp[:, 0:3] = _mixdchlet_mean(c[:, 0:3], *prior["tm"])
    tp[:, 3:5] = _mixdchlet_mean(c[:, 3:5], *prior["ti"])
    tp[:, 5:7] = _mixdchlet_mean(c[:, 5:7], *prior["td"])
    matp = np.zeros_like(mat)
    insp = np.zeros_like(ins)
    matp[1:] = _mixdchlet_mean(mat[1:].astype(np.float64), *prior["em"])
    insp[:] = _mixdchlet_mean(ins.astype(np.float64), *prior["ei"])
    # node 0: mat[0] = (1, 0, ...) by convention; D_0 does not exist; node M has no M->D / D->D
    matp[0] = 0.0
    matp[0, 0] = 1.0
    tp[0, 5], tp[0, 6] = 1.0, 0.0
    tp[M, 2] = 0.0
    s = tp[M, 0] + tp[M, 1]
    tp[M, 0] /= s
    tp[M, 1] /= s
    tp[M, 5], tp[M, 6] = 1.0, 0.0
    return tp.astype(np.float32), matp.astype(np.float32), insp.astype(np.float32)


def mean_match_relent(matp, bg):
    p = matp[1:].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        kl = np.where(p > 0, p * np.log2(p / bg[None, :]), 0.0).sum(axis=1)
    return float(kl.mean())


def entropy_weight(t, mat, ins, nseq, prior, bg, etarget):
    def f(neff):
        s = np.float32(neff / nseq)
        _, mp, _ = parameterize(t * s, mat * s, ins * s, prior)
        return mean_match_relent(mp, bg) - etarget
    if f(float(nseq)) <= 0.0:
        return float(nseq)
    # Easel's esl_root_Bisection, step for step: the bracket is tested BEFORE it is narrowed, and the
    # midpoint of the last tested bracket is the answer (absolute tolerance 0.01 on Neff, set by p7_EntropyWeight)
    xl, xr = 0.0, float(nseq)
    fxl = f(xl)
    x = xr
    for _ in range(100):
        x = (xl + xr) / 2.0
        fx = f(x)
        if fx == 0.0:
            break
        if (xr - xl) < 0.01 + 1e-12 * x or abs(fx) < 1e-12:
            break
        if (fxl > 0.0) == (fx > 0.0):
            xl, fxl = x, fx
        else:
            xr = x
    return x


def build(rows, molecule="dna", ere=0.59, symfrac=0.0, fragthresh=0.5, esigma=45.0):
    """rows: aligned sequences (equal length strings).  Returns a dict with the model."""
    abc = Alphabet(molecule)
    if abc.K != 4:
        raise NotImplementedError("amino priors are not restated yet")
    ax = abc.digitize(rows)
    nseq, alen = ax.shape
    wgt = pb_weights(ax, abc)
    ax = mark_fragments(ax, abc, fragthresh)
    matcols, t, mat, ins = count_model(ax, wgt, abc, symfrac)
    M = len(matcols)
    bg = np.full(abc.K, 1.0 / abc.K)
    etarget = max(ere, (esigma - math.log2(2.0 / (M * (M + 1.0)))) / M)
    neff = entropy_weight(t, mat, ins, nseq, _NUC_PRIOR, bg, etarget)
    s = np.float32(neff / nseq)
    tp, mp, ip = parameterize(t * s, mat * s, ins * s, _NUC_PRIOR)
    return dict(M=M, matcols=matcols, t=tp, mat=mp, ins=ip, nseq=nseq, neff=neff, abc=abc, wgt=wgt)
