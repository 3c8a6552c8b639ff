/*
 * p7_oracle.c - CPU oracle for the WITCH query-vs-eHMM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (witch_amd/, libwitch_hip.so)
 * may include, link or call this file; only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py use it, as the checker.
 *
 * What it restates.  The reference (c5shen/WITCH) obtains its numbers by spawning
 * HMMER 3.1b2 (February 2015) executables:
 *     hmmsearch --cpu 1 --noali -E 99999999 --max   witch_msa/gcmm/algorithm.py:526-532
 *     hmmalign -o OUT HMM QUERY                      witch_msa/gcmm/aligner.py:96-100
 * HMMER's source is NOT part of /root/reference (only prebuilt binaries under
 * witch_msa/tools/magus/tools/hmmer/), so this file restates HMMER 3.1b2's published
 * algorithm (Plan-7 local profile, Forward/Backward, posterior-heuristic domain
 * definition, null2 by expectation, optimal-accuracy alignment) as catalogued in
 * SURVEY.md Appendix A.1-A.7, in float64 probability space with per-row scaling.
 * The float32 rounding points of the final score assembly (A.6) are mimicked so the
 * "%6.1f" print that the reference parses (algorithm.py:597-599) can be reproduced.
 *
 * Parity pin: tests/golden/ holds outputs of the bundled HMMER binaries and of the
 * reference's own Python functions, produced in the build container by
 * tests/golden/make_golden.py; tests/test_oracle_golden.py checks this file against
 * them.  Regions that HMMER flags "multidomain" are resolved as HMMER does (A.4b: 200
 * stochastic tracebacks with Easel's seeded fast RNG + single-linkage clustering); on the
 * 191 golden multidomain pairs: reported mask 191/191, printed score 189/191 (two pairs one
 * deci-bit off, 0.002 and 0.005 bit from a rounding boundary: one sampled trace differing in
 * float rounding moves a per-residue null2 term by 1/200), envelopes 190/191.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_DNA   0
#define ORC_RNA   1
#define ORC_AMINO 2

#define ORC_FLAG_REPORTED 1   /* pair would appear in hmmsearch's per-sequence table   */
#define ORC_FLAG_MULTI    2   /* at least one region was multidomain (stochastic class) */
#define ORC_FLAG_OVERRIDE 4   /* reconstruction score overrode the Forward score (A.6)  */

enum { tMM = 0, tMI = 1, tMD = 2, tIM = 3, tII = 4, tDM = 5, tDD = 6 };

typedef struct {
  int M, K, Kp, alphabet, nseq;
  char name[256];
  double *t;      /* [(M+1)*7] probabilities as in the file, node 0..M           */
  double *mat;    /* [(M+1)*K] match emission probabilities (node 0 unused)       */
  int    *map;    /* [M+1] MAP annotation (alignment column, 1-based) or 0        */
  /* configured local profile (A.1); transitions of node 0 and node M are zero   */
  double *pt;     /* [(M+1)*7] */
  double *entry;  /* [M+2] B->M_k = occ_k / Z                                     */
  double *odds;   /* [Kp*(M+1)] e_k[a]/f[a]; degenerate = exp(weighted mean log) */
  double  bg[20];
} orc_hmm;

/* ----------------------------------------------------------------------------
 * Alphabets (Easel: DNA "ACGT-RYMKSWHBVDN*~", amino "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~")
 * -------------------------------------------------------------------------- */
static const char *DNA_SYM   = "ACGT-RYMKSWHBVDN*~";
static const char *AMINO_SYM = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";

static const double AMINO_BG[20] = {
  0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198,
  0.0590092, 0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639,
  0.0540978, 0.0683364, 0.0540687, 0.0673417, 0.0114135, 0.0304133 };

/* degen[x] = bitmask of canonical residues; for canonical x, 1<<x */
static void degen_masks(int alphabet, uint32_t *mask, int *K, int *Kp)
{
  int i;
  if (alphabet == ORC_AMINO) {
    *K = 20; *Kp = 29;
    for (i = 0; i < 29; i++) mask[i] = 0;
    for (i = 0; i < 20; i++) mask[i] = 1u << i;
    /* A0 C1 D2 E3 F4 G5 H6 I7 K8 L9 M10 N11 P12 Q13 R14 S15 T16 V17 W18 Y19 */
    mask[21] = (1u << 11) | (1u << 2);    /* B = N,D */
    mask[22] = (1u << 7)  | (1u << 9);    /* J = I,L */
    mask[23] = (1u << 13) | (1u << 3);    /* Z = Q,E */
    mask[24] = (1u << 8);                 /* O -> K  */
    mask[25] = (1u << 1);                 /* U -> C  */
    mask[26] = 0xFFFFFu;                  /* X = any */
  } else {
    *K = 4; *Kp = 18;
    for (i = 0; i < 18; i++) mask[i] = 0;
    for (i = 0; i < 4; i++) mask[i] = 1u << i;
    /* A0 C1 G2 T3 ; R5 Y6 M7 K8 S9 W10 H11 B12 V13 D14 N15 */
    mask[5]  = 1 | 4;        /* R = A,G */
    mask[6]  = 2 | 8;        /* Y = C,T */
    mask[7]  = 1 | 2;        /* M = A,C */
    mask[8]  = 4 | 8;        /* K = G,T */
    mask[9]  = 2 | 4;        /* S = C,G */
    mask[10] = 1 | 8;        /* W = A,T */
    mask[11] = 1 | 2 | 8;    /* H = A,C,T */
    mask[12] = 2 | 4 | 8;    /* B = C,G,T */
    mask[13] = 1 | 2 | 4;    /* V = A,C,G */
    mask[14] = 1 | 4 | 8;    /* D = A,G,T */
    mask[15] = 15;           /* N = any */
  }
}

/* Text residue -> digital code, or 255 for a character the alphabet rejects. */
int orc_digitize(int alphabet, const char *s, int n, uint8_t *dsq)
{
  const char *sym = (alphabet == ORC_AMINO) ? AMINO_SYM : DNA_SYM;
  int i;
  for (i = 0; i < n; i++) {
    int c = toupper((unsigned char) s[i]);
    if (alphabet != ORC_AMINO) {
      if (c == 'U') c = 'T';
      if (c == 'X') c = 'N';
      if (c == 'I') c = 'A';
    }
    if (c == '_' || c == '.') c = '-';
    const char *p = strchr(sym, c);
    dsq[i] = (p && c) ? (uint8_t) (p - sym) : 255;
  }
  return 0;
}

/* ----------------------------------------------------------------------------
 * HMMER3/f text reader
 * -------------------------------------------------------------------------- */
static double tok_prob(const char *tok)
{
  if (tok[0] == '*') return 0.0;
  return exp(-atof(tok));
}

void orc_hmm_free(orc_hmm *h)
{
  if (!h) return;
  free(h->t); free(h->mat); free(h->map); free(h->pt); free(h->entry); free(h->odds);
  free(h);
}

static int configure(orc_hmm *h);

orc_hmm *orc_hmm_read(const char *path)
{
  FILE *f = fopen(path, "r");
  char line[8192];
  orc_hmm *h;
  int k, x, in_body = 0;
  if (!f) return NULL;
  h = (orc_hmm *) calloc(1, sizeof(orc_hmm));
  h->alphabet = -1;
  if (!fgets(line, sizeof line, f) || strncmp(line, "HMMER3/", 7) != 0) goto fail;
  while (fgets(line, sizeof line, f)) {
    char key[64]; int n = 0;
    if (sscanf(line, "%63s%n", key, &n) != 1) continue;
    if (!strcmp(key, "NAME"))      sscanf(line + n, "%255s", h->name);
    else if (!strcmp(key, "LENG")) h->M = atoi(line + n);
    else if (!strcmp(key, "NSEQ")) h->nseq = atoi(line + n);
    else if (!strcmp(key, "ALPH")) {
      char a[64]; sscanf(line + n, "%63s", a);
      for (char *p = a; *p; p++) *p = (char) tolower((unsigned char) *p);
      if (!strcmp(a, "dna")) h->alphabet = ORC_DNA;
      else if (!strcmp(a, "rna")) h->alphabet = ORC_RNA;
      else if (!strcmp(a, "amino")) h->alphabet = ORC_AMINO;
      else goto fail;
    }
    else if (!strcmp(key, "HMM")) { in_body = 1; break; }
  }
  if (!in_body || h->M <= 0 || h->alphabet < 0) goto fail;
  {
    uint32_t mask[32];
    degen_masks(h->alphabet, mask, &h->K, &h->Kp);
  }
  h->t   = (double *) calloc((size_t) (h->M + 1) * 7, sizeof(double));
  h->mat = (double *) calloc((size_t) (h->M + 1) * h->K, sizeof(double));
  h->map = (int *) calloc((size_t) h->M + 1, sizeof(int));
  if (!fgets(line, sizeof line, f)) goto fail;            /* transition header */
  /* optional COMPO, then node-0 insert line + node-0 transition line */
  if (!fgets(line, sizeof line, f)) goto fail;
  {
    char key[64];
    sscanf(line, "%63s", key);
    if (!strcmp(key, "COMPO")) { if (!fgets(line, sizeof line, f)) goto fail; }
  }
  /* line = node-0 insert emissions (ignored: insert odds are hardwired to 1, A.1) */
  if (!fgets(line, sizeof line, f)) goto fail;            /* node-0 transitions */
  {
    char *p = strtok(line, " \t\n");
    for (x = 0; x < 7 && p; x++) { h->t[x] = tok_prob(p); p = strtok(NULL, " \t\n"); }
    if (x != 7) goto fail;
  }
  for (k = 1; k <= h->M; k++) {
    char *p;
    if (!fgets(line, sizeof line, f)) goto fail;
    p = strtok(line, " \t\n");
    if (!p || atoi(p) != k) goto fail;
    for (x = 0; x < h->K; x++) {
      p = strtok(NULL, " \t\n");
      if (!p) goto fail;
      h->mat[(size_t) k * h->K + x] = tok_prob(p);
    }
    p = strtok(NULL, " \t\n");
    if (p && p[0] != '-') h->map[k] = atoi(p);
    if (!fgets(line, sizeof line, f)) goto fail;          /* insert emissions: ignored */
    if (!fgets(line, sizeof line, f)) goto fail;
    p = strtok(line, " \t\n");
    for (x = 0; x < 7 && p; x++) { h->t[(size_t) k * 7 + x] = tok_prob(p); p = strtok(NULL, " \t\n"); }
    if (x != 7) goto fail;
  }
  fclose(f);
  if (configure(h) != 0) { orc_hmm_free(h); return NULL; }
  return h;
fail:
  fclose(f);
  orc_hmm_free(h);
  return NULL;
}

int orc_hmm_M(const orc_hmm *h)        { return h->M; }
int orc_hmm_K(const orc_hmm *h)        { return h->K; }
int orc_hmm_Kp(const orc_hmm *h)       { return h->Kp; }
int orc_hmm_nseq(const orc_hmm *h)     { return h->nseq; }
int orc_hmm_alphabet(const orc_hmm *h) { return h->alphabet; }
const char *orc_hmm_name(const orc_hmm *h) { return h->name; }
const int *orc_hmm_map(const orc_hmm *h) { return h->map; }
const double *orc_hmm_entry(const orc_hmm *h) { return h->entry; }
const double *orc_hmm_odds(const orc_hmm *h) { return h->odds; }
const double *orc_hmm_pt(const orc_hmm *h) { return h->pt; }

/* ----------------------------------------------------------------------------
 * A.1 profile configuration: local entry from match occupancy, local exit = 1,
 * odds ratios, degenerate residues by f-weighted mean of log-odds.
 * -------------------------------------------------------------------------- */
static int configure(orc_hmm *h)
{
  int M = h->M, K = h->K, Kp = h->Kp, k, x, a;
  uint32_t mask[32];
  double *occ = (double *) calloc((size_t) M + 2, sizeof(double));
  double Z = 0.0;
  degen_masks(h->alphabet, mask, &K, &Kp);
  if (h->alphabet == ORC_AMINO) { for (a = 0; a < 20; a++) h->bg[a] = AMINO_BG[a]; }
  else                          { for (a = 0; a < 4; a++)  h->bg[a] = 0.25; }
  h->pt    = (double *) calloc((size_t) (M + 1) * 7, sizeof(double));
  h->entry = (double *) calloc((size_t) M + 2, sizeof(double));
  h->odds  = (double *) calloc((size_t) Kp * (M + 1), sizeof(double));
  for (k = 1; k < M; k++)
    for (x = 0; x < 7; x++) h->pt[(size_t) k * 7 + x] = h->t[(size_t) k * 7 + x];
  occ[1] = h->t[tMI] + h->t[tMM];
  for (k = 2; k <= M; k++)
    occ[k] = occ[k - 1] * (h->t[(size_t) (k - 1) * 7 + tMM] + h->t[(size_t) (k - 1) * 7 + tMI])
           + (1.0 - occ[k - 1]) * h->t[(size_t) (k - 1) * 7 + tDM];
  for (k = 1; k <= M; k++) Z += occ[k] * (double) (M - k + 1);
  for (k = 1; k <= M; k++) h->entry[k] = occ[k] / Z;
  for (k = 1; k <= M; k++) {
    double sc[32];
    for (a = 0; a < K; a++) {
      double e = h->mat[(size_t) k * K + a];
      sc[a] = (e > 0.0) ? log(e / h->bg[a]) : -INFINITY;
      h->odds[(size_t) a * (M + 1) + k] = (e > 0.0) ? e / h->bg[a] : 0.0;
    }
    for (x = K; x < Kp; x++) {
      double num = 0.0, den = 0.0;
      if (mask[x] == 0) { h->odds[(size_t) x * (M + 1) + k] = 0.0; continue; }  /* gap, *, ~ */
      for (a = 0; a < K; a++) if (mask[x] & (1u << a)) { num += sc[a] * h->bg[a]; den += h->bg[a]; }
      h->odds[(size_t) x * (M + 1) + k] = exp(num / den);
    }
  }
  free(occ);
  return 0;
}

/* ----------------------------------------------------------------------------
 * DP matrices.  Row i (0..L), node k (0..M), states M,I,D; specials N,B,E,J,C.
 * Values in row i are the true values times exp(-lscale[i]).
 * -------------------------------------------------------------------------- */
typedef struct {
  int L, M;
  double *dp;      /* [(L+1)*(M+1)*3] */
  double *xs;      /* [(L+1)*5]  N B E J C */
  double *lscale;  /* [L+1] cumulative log scale of row i */
} orc_mx;

enum { sN = 0, sB = 1, sE = 2, sJ = 3, sC = 4 };
#define MX(m, i, k, s) ((m)->dp[((size_t) (i) * ((m)->M + 1) + (k)) * 3 + (s)])
#define XS(m, i, s)    ((m)->xs[(size_t) (i) * 5 + (s)])

static orc_mx *mx_new(int L, int M)
{
  orc_mx *m = (orc_mx *) calloc(1, sizeof(orc_mx));
  m->L = L; m->M = M;
  m->dp = (double *) calloc((size_t) (L + 1) * (M + 1) * 3, sizeof(double));
  m->xs = (double *) calloc((size_t) (L + 1) * 5, sizeof(double));
  m->lscale = (double *) calloc((size_t) L + 1, sizeof(double));
  return m;
}
static void mx_free(orc_mx *m) { if (m) { free(m->dp); free(m->xs); free(m->lscale); free(m); } }

typedef struct { double loop, move, EJ, EC; } orc_len;

/* A.1 length model.  HMMER computes pmove/ploop in float32; mimic that. */
static orc_len len_config(int Lcfg, int multihit)
{
  orc_len c;
  float nj = multihit ? 1.0f : 0.0f;
  float pmove = (2.0f + nj) / ((float) Lcfg + 2.0f + nj);
  float ploop = 1.0f - pmove;
  c.loop = ploop; c.move = pmove;
  c.EJ = multihit ? 0.5 : 0.0;
  c.EC = multihit ? 0.5 : 1.0;
  return c;
}

#define RESCALE_HI 1e60

/* A.2 Forward.  Returns ln P(x | profile) in nats (log-odds vs. the implicit
 * per-residue background, i.e. "fwd" of A.2), fills <fx>. */
static double forward(const orc_hmm *h, const uint8_t *dsq, int L, orc_len c, orc_mx *fx)
{
  int M = h->M, i, k;
  const double *pt = h->pt, *en = h->entry;
  double ls = 0.0;
  for (k = 0; k <= M; k++) { MX(fx, 0, k, 0) = MX(fx, 0, k, 1) = MX(fx, 0, k, 2) = 0.0; }
  XS(fx, 0, sN) = 1.0; XS(fx, 0, sB) = c.move; XS(fx, 0, sE) = 0.0; XS(fx, 0, sJ) = 0.0; XS(fx, 0, sC) = 0.0;
  fx->lscale[0] = 0.0;
  for (i = 1; i <= L; i++) {
    const double *od = h->odds + (size_t) dsq[i - 1] * (M + 1);
    double xB = XS(fx, i - 1, sB), xE = 0.0, xN, xJ, xC;
    MX(fx, i, 0, 0) = MX(fx, i, 0, 1) = MX(fx, i, 0, 2) = 0.0;
    for (k = 1; k <= M; k++) {
      const double *tp = pt + (size_t) (k - 1) * 7;
      const double *tk = pt + (size_t) k * 7;
      double m = od[k] * (MX(fx, i - 1, k - 1, 0) * tp[tMM] + MX(fx, i - 1, k - 1, 1) * tp[tIM]
                        + MX(fx, i - 1, k - 1, 2) * tp[tDM] + xB * en[k]);
      double d = MX(fx, i, k - 1, 0) * tp[tMD] + MX(fx, i, k - 1, 2) * tp[tDD];
      double ins = MX(fx, i - 1, k, 0) * tk[tMI] + MX(fx, i - 1, k, 1) * tk[tII];
      MX(fx, i, k, 0) = m; MX(fx, i, k, 1) = ins; MX(fx, i, k, 2) = d;
      xE += m + d;
    }
    xN = XS(fx, i - 1, sN) * c.loop;
    xC = XS(fx, i - 1, sC) * c.loop + xE * c.EC;
    xJ = XS(fx, i - 1, sJ) * c.loop + xE * c.EJ;
    if (xE > RESCALE_HI) {
      double r = 1.0 / xE;
      for (k = 1; k <= M; k++) { MX(fx, i, k, 0) *= r; MX(fx, i, k, 1) *= r; MX(fx, i, k, 2) *= r; }
      xN *= r; xC *= r; xJ *= r; ls += log(xE); xE = 1.0;
    }
    XS(fx, i, sN) = xN; XS(fx, i, sE) = xE; XS(fx, i, sJ) = xJ; XS(fx, i, sC) = xC;
    XS(fx, i, sB) = xJ * c.move + xN * c.move;
    fx->lscale[i] = ls;
  }
  return ls + log(XS(fx, L, sC) * c.move);
}

/* Backward, same scaling convention with its own scale factors. */
static double backward(const orc_hmm *h, const uint8_t *dsq, int L, orc_len c, orc_mx *bx)
{
  int M = h->M, i, k;
  const double *pt = h->pt, *en = h->entry;
  double ls = 0.0;
  /* row L */
  XS(bx, L, sC) = c.move; XS(bx, L, sJ) = 0.0; XS(bx, L, sN) = 0.0; XS(bx, L, sB) = 0.0;
  XS(bx, L, sE) = XS(bx, L, sC) * c.EC + XS(bx, L, sJ) * c.EJ;
  {
    double xE = XS(bx, L, sE);
    MX(bx, L, M, 0) = xE; MX(bx, L, M, 2) = xE; MX(bx, L, M, 1) = 0.0;
    for (k = M - 1; k >= 1; k--) {
      const double *tk = pt + (size_t) k * 7;
      MX(bx, L, k, 0) = xE + MX(bx, L, k + 1, 2) * tk[tMD];
      MX(bx, L, k, 2) = xE + MX(bx, L, k + 1, 2) * tk[tDD];
      MX(bx, L, k, 1) = 0.0;
    }
    MX(bx, L, 0, 0) = MX(bx, L, 0, 1) = MX(bx, L, 0, 2) = 0.0;
  }
  bx->lscale[L] = 0.0;
  for (i = L - 1; i >= 0; i--) {
    const double *od = h->odds + (size_t) dsq[i] * (M + 1);     /* residue x_{i+1} */
    double xB = 0.0, xE, xJ, xC, xN;
    for (k = 1; k <= M; k++) xB += MX(bx, i + 1, k, 0) * od[k] * en[k];
    xJ = XS(bx, i + 1, sJ) * c.loop + xB * c.move;
    xC = XS(bx, i + 1, sC) * c.loop;
    xN = XS(bx, i + 1, sN) * c.loop + xB * c.move;
    xE = xC * c.EC + xJ * c.EJ;
    if (i > 0) {
      MX(bx, i, M, 0) = xE; MX(bx, i, M, 2) = xE; MX(bx, i, M, 1) = 0.0;
      for (k = M - 1; k >= 1; k--) {
        const double *tk = pt + (size_t) k * 7;
        double mnext = MX(bx, i + 1, k + 1, 0) * od[k + 1];
        MX(bx, i, k, 0) = mnext * tk[tMM] + MX(bx, i + 1, k, 1) * tk[tMI] + MX(bx, i, k + 1, 2) * tk[tMD] + xE;
        MX(bx, i, k, 1) = mnext * tk[tIM] + MX(bx, i + 1, k, 1) * tk[tII];
        MX(bx, i, k, 2) = mnext * tk[tDM] + MX(bx, i, k + 1, 2) * tk[tDD] + xE;
      }
      MX(bx, i, 0, 0) = MX(bx, i, 0, 1) = MX(bx, i, 0, 2) = 0.0;
    } else {
      for (k = 0; k <= M; k++) { MX(bx, 0, k, 0) = MX(bx, 0, k, 1) = MX(bx, 0, k, 2) = 0.0; }
    }
    if (xB > RESCALE_HI || xN > RESCALE_HI) {
      double big = xB > xN ? xB : xN, r = 1.0 / big;
      if (i > 0) for (k = 1; k <= M; k++) { MX(bx, i, k, 0) *= r; MX(bx, i, k, 1) *= r; MX(bx, i, k, 2) *= r; }
      xB *= r; xJ *= r; xC *= r; xN *= r; xE *= r; ls += log(big);
    }
    XS(bx, i, sB) = xB; XS(bx, i, sJ) = xJ; XS(bx, i, sC) = xC; XS(bx, i, sN) = xN; XS(bx, i, sE) = xE;
    bx->lscale[i] = ls;
  }
  return ls + log(XS(bx, 0, sN));
}

/* ------------------------------------------------------------------------------------------
 * Extended-range twins of forward()/backward() for the hmmalign restatement.  A query with two
 * hits of which the first scores beyond ~1000 bits pushes the N state below the range of a
 * scaled double (and far below a scaled float); hmmalign itself notices the overflow in its
 * float32 Decoding and switches to its log-space "generic" code, which finds the best hit.  x87
 * long double (15-bit exponent) gives this CPU restatement the same reach with the same scaled
 * prob-space recurrences (checked against the hmmalign binary on multi-hit protein queries).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int L, M;
  long double *dp;
  long double *xs;
  long double *lscale;
} orc_mxx;
static orc_mxx *mxx_new(int L, int M)
{
  orc_mxx *m = (orc_mxx *) calloc(1, sizeof(orc_mxx));
  m->L = L; m->M = M;
  m->dp = (long double *) calloc((size_t) (L + 1) * (M + 1) * 3, sizeof(long double));
  m->xs = (long double *) calloc((size_t) (L + 1) * 5, sizeof(long double));
  m->lscale = (long double *) calloc((size_t) L + 1, sizeof(long double));
  return m;
}
static void mxx_free(orc_mxx *m) { if (m) { free(m->dp); free(m->xs); free(m->lscale); free(m); } }

static long double forward_x(const orc_hmm *h, const uint8_t *dsq, int L, orc_len c, orc_mxx *fx)
{
  int M = h->M, i, k;
  const double *pt = h->pt, *en = h->entry;
  long double ls = 0.0;
  for (k = 0; k <= M; k++) { MX(fx, 0, k, 0) = MX(fx, 0, k, 1) = MX(fx, 0, k, 2) = 0.0; }
  XS(fx, 0, sN) = 1.0; XS(fx, 0, sB) = c.move; XS(fx, 0, sE) = 0.0; XS(fx, 0, sJ) = 0.0; XS(fx, 0, sC) = 0.0;
  fx->lscale[0] = 0.0;
  for (i = 1; i <= L; i++) {
    const double *od = h->odds + (size_t) dsq[i - 1] * (M + 1);
    long double xB = XS(fx, i - 1, sB), xE = 0.0, xN, xJ, xC;
    MX(fx, i, 0, 0) = MX(fx, i, 0, 1) = MX(fx, i, 0, 2) = 0.0;
    for (k = 1; k <= M; k++) {
      const double *tp = pt + (size_t) (k - 1) * 7;
      const double *tk = pt + (size_t) k * 7;
      long double m = od[k] * (MX(fx, i - 1, k - 1, 0) * tp[tMM] + MX(fx, i - 1, k - 1, 1) * tp[tIM]
                        + MX(fx, i - 1, k - 1, 2) * tp[tDM] + xB * en[k]);
      long double d = MX(fx, i, k - 1, 0) * tp[tMD] + MX(fx, i, k - 1, 2) * tp[tDD];
      long double ins = MX(fx, i - 1, k, 0) * tk[tMI] + MX(fx, i - 1, k, 1) * tk[tII];
      MX(fx, i, k, 0) = m; MX(fx, i, k, 1) = ins; MX(fx, i, k, 2) = d;
      xE += m + d;
    }
    xN = XS(fx, i - 1, sN) * c.loop;
    xC = XS(fx, i - 1, sC) * c.loop + xE * c.EC;
    xJ = XS(fx, i - 1, sJ) * c.loop + xE * c.EJ;
    if (xE > RESCALE_HI) {
      long double r = 1.0 / xE;
      for (k = 1; k <= M; k++) { MX(fx, i, k, 0) *= r; MX(fx, i, k, 1) *= r; MX(fx, i, k, 2) *= r; }
      xN *= r; xC *= r; xJ *= r; ls += logl(xE); xE = 1.0;
    }
    XS(fx, i, sN) = xN; XS(fx, i, sE) = xE; XS(fx, i, sJ) = xJ; XS(fx, i, sC) = xC;
    XS(fx, i, sB) = xJ * c.move + xN * c.move;
    fx->lscale[i] = ls;
  }
  return ls + logl(XS(fx, L, sC) * c.move);
}

/* Backward in extended precision (see forward_x). */
static long double backward_x(const orc_hmm *h, const uint8_t *dsq, int L, orc_len c, orc_mxx *bx)
{
  int M = h->M, i, k;
  const double *pt = h->pt, *en = h->entry;
  long double ls = 0.0;
  /* row L */
  XS(bx, L, sC) = c.move; XS(bx, L, sJ) = 0.0; XS(bx, L, sN) = 0.0; XS(bx, L, sB) = 0.0;
  XS(bx, L, sE) = XS(bx, L, sC) * c.EC + XS(bx, L, sJ) * c.EJ;
  {
    long double xE = XS(bx, L, sE);
    MX(bx, L, M, 0) = xE; MX(bx, L, M, 2) = xE; MX(bx, L, M, 1) = 0.0;
    for (k = M - 1; k >= 1; k--) {
      const double *tk = pt + (size_t) k * 7;
      MX(bx, L, k, 0) = xE + MX(bx, L, k + 1, 2) * tk[tMD];
      MX(bx, L, k, 2) = xE + MX(bx, L, k + 1, 2) * tk[tDD];
      MX(bx, L, k, 1) = 0.0;
    }
    MX(bx, L, 0, 0) = MX(bx, L, 0, 1) = MX(bx, L, 0, 2) = 0.0;
  }
  bx->lscale[L] = 0.0;
  for (i = L - 1; i >= 0; i--) {
    const double *od = h->odds + (size_t) dsq[i] * (M + 1);     /* residue x_{i+1} */
    long double xB = 0.0, xE, xJ, xC, xN;
    for (k = 1; k <= M; k++) xB += MX(bx, i + 1, k, 0) * od[k] * en[k];
    xJ = XS(bx, i + 1, sJ) * c.loop + xB * c.move;
    xC = XS(bx, i + 1, sC) * c.loop;
    xN = XS(bx, i + 1, sN) * c.loop + xB * c.move;
    xE = xC * c.EC + xJ * c.EJ;
    if (i > 0) {
      MX(bx, i, M, 0) = xE; MX(bx, i, M, 2) = xE; MX(bx, i, M, 1) = 0.0;
      for (k = M - 1; k >= 1; k--) {
        const double *tk = pt + (size_t) k * 7;
        long double mnext = MX(bx, i + 1, k + 1, 0) * od[k + 1];
        MX(bx, i, k, 0) = mnext * tk[tMM] + MX(bx, i + 1, k, 1) * tk[tMI] + MX(bx, i, k + 1, 2) * tk[tMD] + xE;
        MX(bx, i, k, 1) = mnext * tk[tIM] + MX(bx, i + 1, k, 1) * tk[tII];
        MX(bx, i, k, 2) = mnext * tk[tDM] + MX(bx, i, k + 1, 2) * tk[tDD] + xE;
      }
      MX(bx, i, 0, 0) = MX(bx, i, 0, 1) = MX(bx, i, 0, 2) = 0.0;
    } else {
      for (k = 0; k <= M; k++) { MX(bx, 0, k, 0) = MX(bx, 0, k, 1) = MX(bx, 0, k, 2) = 0.0; }
    }
    if (xB > RESCALE_HI || xN > RESCALE_HI) {
      long double big = xB > xN ? xB : xN, r = 1.0L / big;
      if (i > 0) for (k = 1; k <= M; k++) { MX(bx, i, k, 0) *= r; MX(bx, i, k, 1) *= r; MX(bx, i, k, 2) *= r; }
      xB *= r; xJ *= r; xC *= r; xN *= r; xE *= r; ls += logl(big);
    }
    XS(bx, i, sB) = xB; XS(bx, i, sJ) = xJ; XS(bx, i, sC) = xC; XS(bx, i, sN) = xN; XS(bx, i, sE) = xE;
    bx->lscale[i] = ls;
  }
  return ls + logl(XS(bx, 0, sN));
}

/* ----------------------------------------------------------------------------
 * p7_FLogsum: table-driven float log-sum-exp (A.6)
 * -------------------------------------------------------------------------- */
static float flogsum_table[16000];
static int   flogsum_ready = 0;
static void flogsum_init(void)
{
  int i;
  if (flogsum_ready) return;
  for (i = 0; i < 16000; i++) flogsum_table[i] = (float) log(1. + exp((double) -i / 1000.0));
  flogsum_ready = 1;
}
static float flogsum(float a, float b)
{
  const float max = a > b ? a : b;
  const float min = a > b ? b : a;
  return (min == -INFINITY || (max - min) >= 15.7f) ? max : max + flogsum_table[(int) ((max - min) * 1000.0f)];
}

/* ----------------------------------------------------------------------------
 * Results
 * -------------------------------------------------------------------------- */
#define ORC_MAXENV 256   /* (HMMER has no limit; 256 is beyond any test) */
typedef struct {
  int    flags;
  int    nregions, nenv;
  int    env_i[ORC_MAXENV], env_j[ORC_MAXENV], env_multi[ORC_MAXENV];
  double envsc[ORC_MAXENV], domcorr[ORC_MAXENV];
  double fwd_nats, null_nats, seqbias_nats;
  double fwd_bits;        /* (fwd - null1)/ln2 in float64: the "Forward log-odds" */
  float  pre_score, seq_score, sum_score;
  int    decibits;
} orc_result;

static int decibits_of(float score)
{
  /* score*10 is exact in double (24+4 significant bits), rint = round-half-even, which is
   * what glibc's "%6.1f" does on the exact binary value (algorithm.py:597-599 parses it). */
  return (int) rint((double) score * 10.0);
}

/* A.5: posterior decoding on an isolated envelope + null2 by expectation. */
static void null2_by_expectation(const orc_hmm *h, const orc_mx *fx, const orc_mx *bx, int Ld,
                                 orc_len c, double fwdsc, float *null2 /*Kp*/)
{
  int M = h->M, K = h->K, Kp = h->Kp, i, k, a, x;
  double *fM = (double *) calloc((size_t) M + 1, sizeof(double));
  double *fI = (double *) calloc((size_t) M + 1, sizeof(double));
  double xfactor = 0.0, norm = 1.0 / (double) Ld;
  uint32_t mask[32];
  degen_masks(h->alphabet, mask, &K, &Kp);
  for (i = 1; i <= Ld; i++) {
    double sc = exp(fx->lscale[i] + bx->lscale[i] - fwdsc);
    for (k = 1; k <= M; k++) {
      fM[k] += MX(fx, i, k, 0) * MX(bx, i, k, 0) * sc;
      fI[k] += MX(fx, i, k, 1) * MX(bx, i, k, 1) * sc;
    }
    {
      double sc2 = exp(fx->lscale[i - 1] + bx->lscale[i] - fwdsc);
      xfactor += XS(fx, i - 1, sN) * XS(bx, i, sN) * c.loop * sc2;
      xfactor += XS(fx, i - 1, sJ) * XS(bx, i, sJ) * c.loop * sc2;
      xfactor += XS(fx, i - 1, sC) * XS(bx, i, sC) * c.loop * sc2;
    }
  }
  xfactor *= norm;
  for (a = 0; a < K; a++) {
    const double *od = h->odds + (size_t) a * (M + 1);
    double s = 0.0;
    for (k = 1; k <= M; k++) s += fM[k] * norm * od[k] + fI[k] * norm;
    null2[a] = (float) (s + xfactor);
  }
  /* degenerate residues: unweighted mean of the canonical odds (esl_abc_FAvgScVec) */
  for (x = K; x < Kp; x++) {
    if (mask[x] == 0) { null2[x] = 1.0f; continue; }
    float s = 0.0f; int n = 0;
    for (a = 0; a < K; a++) if (mask[x] & (1u << a)) { s += null2[a]; n++; }
    null2[x] = s / (float) n;
  }
  free(fM); free(fI);
}

/* ----------------------------------------------------------------------------
 * A.4b  Multidomain regions: stochastic traceback ensemble + clustering.
 *
 * HMMER 3.1b2 resolves a region flagged multidomain (p7_domaindef_ByPosteriorHeuristics ->
 * region_trace_ensemble) by: multihit Forward on the region's sub-sequence (length model of the
 * whole sequence); the pipeline's RNG re-initialised to its seed; 200 stochastic tracebacks
 * (p7_StochasticTrace); every sampled domain goes into a segment-pair ensemble and bumps the
 * per-residue null2 accumulators (p7_Null2_ByTrace); single-linkage clustering of the segments
 * (p7_spensemble_Cluster: >= 0.8 overlap of the smaller in sequence and model, both end diagonals
 * within 4, cluster in >= 25 % of the traces, end points at >= 2 % frequency); clusters dominated
 * by an overlapping, more probable one are dropped; every surviving cluster is an envelope.
 * The RNG is Easel's "fast" generator (esl_randomness_CreateFast(42) in p7_pipeline_Create,
 * identified in the bundled binary): x <- 69069 x + 1 (mod 2^32), seeded with Jenkins' mix3.
 * Restated from HMMER 3.1b2's published behaviour; pinned on the golden multidomain pairs
 * (tests/test_oracle_golden.py prints and bounds the residual).
 * -------------------------------------------------------------------------- */
typedef struct { uint32_t x; } orc_rng;

static uint32_t jenkins_mix3(uint32_t a, uint32_t b, uint32_t c)
{
  a -= b; a -= c; a ^= (c >> 13);
  b -= c; b -= a; b ^= (a << 8);
  c -= a; c -= b; c ^= (b >> 13);
  a -= b; a -= c; a ^= (c >> 12);
  b -= c; b -= a; b ^= (a << 16);
  c -= a; c -= b; c ^= (b >> 5);
  a -= b; a -= c; a ^= (c >> 3);
  b -= c; b -= a; b ^= (a << 10);
  c -= a; c -= b; c ^= (b >> 15);
  return c;
}
static void rng_init(orc_rng *r, uint32_t seed)
{
  r->x = jenkins_mix3(seed, 87654321u, 12345678u);
  if (r->x == 0) r->x = 42;
}
static double rng_next(orc_rng *r)
{
  r->x = r->x * 69069u + 1u;
  return (double) r->x / 4294967296.0;
}
/* esl_vec_FNorm + esl_rnd_FChoose on a short vector of float probabilities */
static int rng_choose(orc_rng *r, double *pd, int n)
{
  float p[4];
  double tot = 0.0, roll, sum = 0.0;
  int t;
  for (t = 0; t < n; t++) tot += pd[t];
  for (t = 0; t < n; t++) p[t] = tot > 0.0 ? (float) (pd[t] / tot) : 1.0f / (float) n;
  /* esl_rnd_FChoose of the bundled binary: first t with (p[0] + .. + p[t]) / norm > roll, in double */
  {
    double norm = 0.0;
    roll = rng_next(r);
    for (t = 0; t < n; t++) norm += p[t];
    for (t = 0; t < n; t++) { sum += p[t]; if (sum / norm > roll) return t; }
  }
  return n - 1;
}

typedef struct { int i, j, k, m, idx; float prob; } orc_seg;
enum { stM = 1, stD, stI, stN, stC, stJ, stE, stB, stS };

typedef struct {
  orc_seg *seg; int nseg, aseg;
} orc_ensemble;

static void ens_add(orc_ensemble *en, int idx, int i, int j, int k, int m)
{
  if (en->nseg == en->aseg) { en->aseg = en->aseg ? 2 * en->aseg : 256; en->seg = (orc_seg *) realloc(en->seg, sizeof(orc_seg) * (size_t) en->aseg); }
  en->seg[en->nseg].i = i; en->seg[en->nseg].j = j; en->seg[en->nseg].k = k; en->seg[en->nseg].m = m;
  en->seg[en->nseg].idx = idx; en->seg[en->nseg].prob = 0.0f;
  en->nseg++;
}

static int seg_linked(const orc_seg *h1, const orc_seg *h2)
{
  const float min_overlap = 0.8f;
  const int max_diagdiff = 4;
  int nov, n, d1, d2;
#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))
  nov = IMIN(h1->j, h2->j) - IMAX(h1->i, h2->i) + 1;
  n = IMIN(h1->j - h1->i + 1, h2->j - h2->i + 1);
  if ((float) nov / (float) n < min_overlap) return 0;
  nov = IMIN(h1->m, h2->m) - IMAX(h1->k, h2->k);
  n = IMIN(h1->m - h1->k + 1, h2->m - h2->k + 1);
  if ((float) nov / (float) n < min_overlap) return 0;
  /* start points OR end points on nearby diagonals (binary: link_spsamples returns TRUE at the first hit) */
  d1 = h1->i - h1->k; d2 = h2->i - h2->k; if (abs(d1 - d2) <= max_diagdiff) return 1;
  d1 = h1->j - h1->m; d2 = h2->j - h2->m; if (abs(d1 - d2) <= max_diagdiff) return 1;
  return 0;
}

/* esl_cluster_SingleLinkage's vertex order (stack a initialised in reverse, swap-with-last deletion) */
static int single_linkage(const orc_seg *seg, int n, int *assign)
{
  int *a = (int *) malloc(sizeof(int) * (size_t) (n + 1)), *b = (int *) malloc(sizeof(int) * (size_t) (n + 1));
  int na = n, nb, nc = 0, v, w, t;
  for (v = 0; v < n; v++) a[v] = n - v - 1;
  while (na > 0) {
    v = a[na - 1]; na--;
    b[0] = v; nb = 1;
    while (nb > 0) {
      v = b[nb - 1]; nb--;
      assign[v] = nc;
      for (t = na - 1; t >= 0; t--)
        if (seg_linked(seg + v, seg + a[t])) { w = a[t]; a[t] = a[na - 1]; na--; b[nb++] = w; }
    }
    nc++;
  }
  free(a); free(b);
  return nc;
}

static int iargmax(const int *v, int n) { int t, best = 0; for (t = 1; t < n; t++) if (v[t] > v[best]) best = t; return best; }

/* p7_spensemble_Cluster + the "dominated domain" removal of region_trace_ensemble.
 * Returns the number of envelopes; coordinates in sig[] sorted by start. */
static int cluster_ensemble(orc_ensemble *en, int nsamples, orc_seg *sig, int maxsig)
{
  const float min_posterior = 0.25f, min_endpointp = 0.02f;
  int n = en->nseg, nc, c, h, nsig = 0, d, d2;
  int *assign, *epc, *dominated;
  if (n == 0) return 0;
  assign = (int *) malloc(sizeof(int) * (size_t) n);
  nc = single_linkage(en->seg, n, assign);
  if (getenv("ORC_DBG")) {
    fprintf(stderr, "[oracle] nseg %d nc %d\n", n, nc);
    for (h = 0; h < n && h < atoi(getenv("ORC_DBG")); h++)
      fprintf(stderr, "   seg %d: t %d i %d j %d k %d m %d cluster %d\n", h, en->seg[h].idx, en->seg[h].i, en->seg[h].j, en->seg[h].k, en->seg[h].m, assign[h]);
  }
  for (c = 0; c < nc; c++) {
    int ninc = 0, idx_of_last = -1, imin = 0, imax = 0, jmin = 0, jmax = 0, kmin = 0, kmax = 0, mmin = 0, mmax = 0;
    int best_i, best_j, best_k, best_m, thr, span;
    for (h = 0; h < n; h++) if (assign[h] == c) { if (en->seg[h].idx != idx_of_last) ninc++; idx_of_last = en->seg[h].idx; }
    if ((float) ninc / (float) nsamples < min_posterior) continue;
    for (h = 0; h < n; h++) if (assign[h] == c) {
      const orc_seg *s = en->seg + h;
      if (imin == 0) { imin = imax = s->i; jmin = jmax = s->j; kmin = kmax = s->k; mmin = mmax = s->m; }
      else {
        imin = IMIN(imin, s->i); imax = IMAX(imax, s->i); jmin = IMIN(jmin, s->j); jmax = IMAX(jmax, s->j);
        kmin = IMIN(kmin, s->k); kmax = IMAX(kmax, s->k); mmin = IMIN(mmin, s->m); mmax = IMAX(mmax, s->m);
      }
    }
    thr = (int) ceilf((float) ninc * min_endpointp);
    span = IMAX(IMAX(imax - imin, jmax - jmin), IMAX(kmax - kmin, mmax - mmin)) + 1;
    epc = (int *) calloc((size_t) span, sizeof(int));
    for (h = 0; h < n; h++) if (assign[h] == c) epc[en->seg[h].i - imin]++;
    for (best_i = imin; best_i <= imax; best_i++) if (epc[best_i - imin] >= thr) break;
    if (best_i > imax) best_i = imin + iargmax(epc, imax - imin + 1);
    memset(epc, 0, sizeof(int) * (size_t) span);
    for (h = 0; h < n; h++) if (assign[h] == c) epc[en->seg[h].k - kmin]++;
    for (best_k = kmin; best_k <= kmax; best_k++) if (epc[best_k - kmin] >= thr) break;
    if (best_k > kmax) best_k = kmin + iargmax(epc, kmax - kmin + 1);
    memset(epc, 0, sizeof(int) * (size_t) span);
    for (h = 0; h < n; h++) if (assign[h] == c) epc[en->seg[h].j - jmin]++;
    for (best_j = jmax; best_j >= jmin; best_j--) if (epc[best_j - jmin] >= thr) break;
    if (best_j < jmin) best_j = jmin + iargmax(epc, jmax - jmin + 1);
    memset(epc, 0, sizeof(int) * (size_t) span);
    for (h = 0; h < n; h++) if (assign[h] == c) epc[en->seg[h].m - mmin]++;
    for (best_m = mmax; best_m >= mmin; best_m--) if (epc[best_m - mmin] >= thr) break;
    if (best_m < mmin) best_m = mmin + iargmax(epc, mmax - mmin + 1);
    free(epc);
    if (getenv("ORC_DBG")) fprintf(stderr, "[oracle] cluster %d: ninc %d thr %d i %d..%d j %d..%d k %d..%d m %d..%d best %d %d %d %d\n", c, ninc, thr, imin, imax, jmin, jmax, kmin, kmax, mmin, mmax, best_i, best_j, best_k, best_m);
    if (best_i > best_j || best_k > best_m) continue;
    if (nsig < maxsig) {
      sig[nsig].i = best_i; sig[nsig].j = best_j; sig[nsig].k = best_k; sig[nsig].m = best_m;
      sig[nsig].idx = c; sig[nsig].prob = (float) ninc / (float) nsamples;
      nsig++;
    }
  }
  free(assign);
  /* order by start point (stable) */
  for (d = 1; d < nsig; d++) {
    orc_seg t = sig[d];
    for (d2 = d - 1; d2 >= 0 && sig[d2].i > t.i; d2--) sig[d2 + 1] = sig[d2];
    sig[d2 + 1] = t;
  }
  /* drop clusters dominated by an overlapping, more probable one */
  dominated = (int *) calloc((size_t) nsig + 1, sizeof(int));
  for (d = 0; d < nsig; d++)
    for (d2 = d + 1; d2 < nsig; d2++) {
      int nov = IMIN(sig[d].j, sig[d2].j) - IMAX(sig[d].i, sig[d2].i) + 1, nn;
      if (nov == 0) break;
      nn = IMIN(sig[d].j - sig[d].i + 1, sig[d2].j - sig[d2].i + 1);
      if ((float) nov / (float) nn >= 0.8f) {
        if (sig[d].prob > sig[d2].prob) dominated[d2] = 1; else dominated[d] = 1;
      }
    }
  for (d = 0, d2 = 0; d2 < nsig; d2++) { if (dominated[d2]) continue; if (d != d2) sig[d] = sig[d2]; d++; }
  free(dominated);
  return d;
}

/* One region i..j (1-based, inclusive).  <fr>: multihit Forward matrix of the region's sub-sequence.
 * Sets n2sc[ireg..jreg]; returns the envelopes in sig[]. */
static int region_trace_ensemble(const orc_hmm *h, const uint8_t *dsq, int ireg, int jreg, const orc_mx *fr,
                                 orc_len c, float *n2sc, orc_seg *sig, int maxsig)
{
  const int nsamples = 200;
  const int M = h->M, K = h->K, Kp = h->Kp, Lr = jreg - ireg + 1;
  const int Q = ((M - 1) / 4 + 1) > 2 ? ((M - 1) / 4 + 1) : 2;       /* p7O_NQF: striped vectors of 4 floats */
  const double *pt = h->pt, *en = h->entry;
  const uint8_t *rs = dsq + (ireg - 1);                              /* rs[pos-1] = residue at region position pos */
  orc_rng rng;
  orc_ensemble ens = {0, 0, 0};
  int *cntM = (int *) calloc((size_t) M + 2, sizeof(int)), *cntI = (int *) calloc((size_t) M + 2, sizeof(int));
  int *usedk = (int *) malloc(sizeof(int) * (size_t) (2 * (Lr + M) + 8));
  uint32_t mask[32];
  int t, pos, nc, Kc = K, Kpc = Kp;
  /* per trace: domains found right-to-left */
  int adom = 64, *dfrom = (int *) malloc(sizeof(int) * 4 * (size_t) adom);
  float *dnull = (float *) malloc(sizeof(float) * 32 * (size_t) adom);
  degen_masks(h->alphabet, mask, &Kc, &Kpc);
  for (pos = ireg; pos <= jreg; pos++) n2sc[pos] = 0.0f;
  rng_init(&rng, 42u);
  for (t = 0; t < nsamples; t++) {
    int i = Lr, k = 0, s0 = stC, s1, ndom = 0, nused = 0, Ld = 0, sqto = 0, hmmto = 0, sqfrom = 0, hmmfrom = 0, d;
    while (s0 != stS) {
      double path[4] = {0.0, 0.0, 0.0, 0.0};
      switch (s0) {
      case stM:
        path[0] = XS(fr, i - 1, sB) * en[k];
        path[1] = MX(fr, i - 1, k - 1, 0) * pt[(size_t) (k - 1) * 7 + tMM];
        path[2] = MX(fr, i - 1, k - 1, 1) * pt[(size_t) (k - 1) * 7 + tIM];
        path[3] = MX(fr, i - 1, k - 1, 2) * pt[(size_t) (k - 1) * 7 + tDM];
        { static const int st[4] = { stB, stM, stI, stD }; s1 = st[rng_choose(&rng, path, 4)]; }
        k--; i--;
        break;
      case stD:
        path[0] = MX(fr, i, k - 1, 0) * pt[(size_t) (k - 1) * 7 + tMD];
        path[1] = MX(fr, i, k - 1, 2) * pt[(size_t) (k - 1) * 7 + tDD];
        s1 = rng_choose(&rng, path, 2) == 0 ? stM : stD;
        k--;
        break;
      case stI:
        path[0] = MX(fr, i - 1, k, 0) * pt[(size_t) k * 7 + tMI];
        path[1] = MX(fr, i - 1, k, 1) * pt[(size_t) k * 7 + tII];
        s1 = rng_choose(&rng, path, 2) == 0 ? stM : stI;
        i--;
        break;
      case stN: s1 = (i == 0) ? stS : stN; break;
      case stC:
        path[0] = XS(fr, i - 1, sC) * c.loop;
        path[1] = XS(fr, i, sE) * c.EC * exp(fr->lscale[i] - fr->lscale[i - 1]);
        s1 = rng_choose(&rng, path, 2) == 0 ? stC : stE;
        break;
      case stJ:
        path[0] = XS(fr, i - 1, sJ) * c.loop;
        path[1] = XS(fr, i, sE) * c.EJ * exp(fr->lscale[i] - fr->lscale[i - 1]);
        s1 = rng_choose(&rng, path, 2) == 0 ? stJ : stE;
        break;
      case stE: {
        /* on-the-fly FChoose over M(i,*) and D(i,*) in HMMER's striped order (q outer, r inner) */
        const double roll = rng_next(&rng), norm = 1.0 / XS(fr, i, sE);
        double sum = 0.0;
        int q, r, found = 0, guard = 0;
        s1 = stM;
        while (!found && guard++ < 4) {
          for (q = 0; q < Q && !found; q++) {
            for (r = 0; r < 4 && !found; r++) {
              const int kk = r * Q + q + 1;
              sum += (kk <= M) ? (double) (float) (MX(fr, i, kk, 0) * norm) : 0.0;
              if (roll < sum) { k = kk; s1 = stM; found = 1; }
            }
            for (r = 0; r < 4 && !found; r++) {
              const int kk = r * Q + q + 1;
              sum += (kk <= M) ? (double) (float) (MX(fr, i, kk, 2) * norm) : 0.0;
              if (roll < sum) { k = kk; s1 = stD; found = 1; }
            }
          }
        }
        if (!found) { k = 1; s1 = stM; }
        break;
      }
      case stB:
        path[0] = XS(fr, i, sN) * c.move;
        path[1] = XS(fr, i, sJ) * c.move;
        s1 = rng_choose(&rng, path, 2) == 0 ? stN : stJ;
        break;
      default: s1 = stS; break;
      }
      if (getenv("ORC_DBG") && atoi(getenv("ORC_DBG")) >= 1000 && t == atoi(getenv("ORC_DBG")) - 1000)
        fprintf(stderr, "   [trace %d] s0 %d -> s1 %d at i %d k %d  path %.9g %.9g %.9g %.9g rng %u\n", t, s0, s1, i, k, path[0], path[1], path[2], path[3], rng.x);
      /* the state just chosen sits at (k, i) */
      if (s1 == stE) { sqto = hmmto = 0; nused = 0; Ld = 0; }
      else if (s1 == stM) {
        if (sqto == 0) { sqto = i; hmmto = k; }
        sqfrom = i; hmmfrom = k;
        usedk[nused++] = k; Ld++;
      } else if (s1 == stI) { usedk[nused++] = -k; Ld++; }
      else if (s1 == stB) {
        /* domain complete: p7_Null2_ByTrace over its M and I states */
        float *nl;
        int x, a, u;
        if (ndom == adom) { adom *= 2; dfrom = (int *) realloc(dfrom, sizeof(int) * 4 * (size_t) adom); dnull = (float *) realloc(dnull, sizeof(float) * 32 * (size_t) adom); }
        dfrom[4 * ndom] = sqfrom; dfrom[4 * ndom + 1] = sqto; dfrom[4 * ndom + 2] = hmmfrom; dfrom[4 * ndom + 3] = hmmto;
        nl = dnull + 32 * (size_t) ndom;
        for (u = 0; u < nused; u++) { if (usedk[u] > 0) cntM[usedk[u]]++; else cntI[-usedk[u]]++; }
        {
          const float norm = 1.0f / (float) Ld;
          for (a = 0; a < K; a++) {
            const double *od = h->odds + (size_t) a * (M + 1);
            float lane[4] = { 0.f, 0.f, 0.f, 0.f };
            int q, r;
            for (q = 0; q < Q; q++)
              for (r = 0; r < 4; r++) {
                const int kk = r * Q + q + 1;
                if (kk > M) continue;
                if (cntM[kk]) lane[r] += ((float) cntM[kk] * norm) * (float) od[kk];
                if (cntI[kk]) lane[r] += (float) cntI[kk] * norm;
              }
            nl[a] = (lane[0] + lane[1]) + (lane[2] + lane[3]);
          }
          for (x = K; x < Kp; x++) {
            if (mask[x] == 0) { nl[x] = 1.0f; continue; }
            float sx = 0.0f; int nx = 0;
            for (a = 0; a < K; a++) if (mask[x] & (1u << a)) { sx += nl[a]; nx++; }
            nl[x] = sx / (float) nx;
          }
        }
        for (u = 0; u < nused; u++) { if (usedk[u] > 0) cntM[usedk[u]] = 0; else cntI[-usedk[u]] = 0; }
        ndom++;
      }
      if ((s1 == stN || s1 == stJ || s1 == stC) && s1 == s0) i--;
      s0 = s1;
    }
    /* domains were found right to left; the ensemble and the null2 bumps take them left to right */
    pos = 1;
    for (d = ndom - 1; d >= 0; d--) {
      const int *df = dfrom + 4 * d;
      const float *nl = dnull + 32 * (size_t) d;
      ens_add(&ens, t, df[0] + ireg - 1, df[1] + ireg - 1, df[2], df[3]);
      for (; pos <= df[0]; pos++) n2sc[ireg + pos - 1] += 1.0f;    /* sic: the domain's first residue is bumped by 1 too (matches the binary's scores) */
      for (; pos <= df[1]; pos++) n2sc[ireg + pos - 1] += nl[rs[pos - 1]];
    }
    for (; pos <= Lr; pos++) n2sc[ireg + pos - 1] += 1.0f;
  }
  for (pos = ireg; pos <= jreg; pos++) n2sc[pos] = logf(n2sc[pos] / (float) nsamples);
  if (getenv("ORC_DBG")) { float sm = 0.f; for (pos = ireg; pos <= jreg; pos++) sm += n2sc[pos]; fprintf(stderr, "[oracle] region n2sc sum %.6f; n2sc:", sm); for (pos = ireg; pos <= jreg; pos++) fprintf(stderr, " %.3f", n2sc[pos]); fprintf(stderr, "\n"); }
  nc = cluster_ensemble(&ens, nsamples, sig, maxsig);
  free(ens.seg); free(cntM); free(cntI); free(usedk); free(dfrom); free(dnull);
  return nc;
}

/* 1 (default): multidomain regions go through A.4b; 0: the round-1 behaviour (one envelope per region) */
int orc_resolve_multidomain = 1;
void orc_set_resolve_multidomain(int on) { orc_resolve_multidomain = on; }

/* Score one (query, HMM) pair the way "hmmsearch --max" does (A.2-A.6). */
int orc_score_pair(const orc_hmm *h, const uint8_t *dsq, int L, orc_result *r)
{
  int M = h->M, i, j, z, nclustered_env = 0;
  orc_mx *fx, *bx;
  orc_len cm = len_config(L, 1);
  double fwd, ov;
  double *btot, *etot, *mocc;
  float *n2sc;
  float nullsc, fwdsc_f;
  const double rt1 = 0.25, rt2 = 0.10, rt3 = 0.20;
  memset(r, 0, sizeof(*r));
  flogsum_init();
  if (L <= 0) return 0;
  fx = mx_new(L, M); bx = mx_new(L, M);
  fwd = forward(h, dsq, L, cm, fx);
  backward(h, dsq, L, cm, bx);
  ov = fwd;
  r->fwd_nats = fwd;
  {
    /* A.3 null1, float32 like p7_bg_SetLength/p7_bg_NullOne */
    float p1 = (float) L / (float) (L + 1);
    nullsc = (float) ((float) L * log(p1) + log(1. - p1));
    r->null_nats = nullsc;
    r->fwd_bits = (fwd - ((double) L * log((double) L / (L + 1.0)) + log(1.0 / (L + 1.0)))) / M_LN2;
  }
  fwdsc_f = (float) fwd;
  if (!isfinite(fwd)) { mx_free(fx); mx_free(bx); return 0; }

  /* A.4 domain decoding */
  btot = (double *) calloc((size_t) L + 1, sizeof(double));
  etot = (double *) calloc((size_t) L + 1, sizeof(double));
  mocc = (double *) calloc((size_t) L + 1, sizeof(double));
  n2sc = (float *) calloc((size_t) L + 1, sizeof(float));
  for (i = 1; i <= L; i++) {
    double pb = XS(fx, i - 1, sB) * XS(bx, i - 1, sB) * exp(fx->lscale[i - 1] + bx->lscale[i - 1] - ov);
    double pe = XS(fx, i, sE) * XS(bx, i, sE) * exp(fx->lscale[i] + bx->lscale[i] - ov);
    double sc = exp(fx->lscale[i - 1] + bx->lscale[i] - ov);
    double njcp = (XS(fx, i - 1, sN) * XS(bx, i, sN) + XS(fx, i - 1, sJ) * XS(bx, i, sJ)
                 + XS(fx, i - 1, sC) * XS(bx, i, sC)) * cm.loop * sc;
    btot[i] = btot[i - 1] + pb;
    etot[i] = etot[i - 1] + pe;
    mocc[i] = 1.0 - njcp;
  }

  /* region scan */
  {
    int triggered = 0;
    i = -1;
    for (j = 1; j <= L; j++) {
      if (!triggered) {
        if (mocc[j] - (btot[j] - btot[j - 1]) < rt2) i = j;
        else if (i == -1) i = j;
        if (mocc[j] >= rt1) triggered = 1;
      } else if (mocc[j] - (etot[j] - etot[j - 1]) < rt2) {
        /* region i..j */
        double mx = -1.0;
        int multi;
        r->nregions++;
        for (z = i; z <= j; z++) {
          double a = etot[z] - etot[i - 1], b = btot[j] - btot[z - 1];
          double e = a < b ? a : b;
          if (e > mx) mx = e;
        }
        multi = (mx >= rt3);
        if (multi) r->flags |= ORC_FLAG_MULTI;
        if (multi && orc_resolve_multidomain) {
          /* A.4b: the region is resolved into 0..n envelopes by the trace ensemble; n2sc of the whole
           * region comes from the traces and is NOT recomputed per envelope (null2_is_done) */
          const int Lr = j - i + 1;
          orc_seg sig[ORC_MAXENV];
          orc_mx *fr = mx_new(Lr, M);
          int nc, d;
          { double rf = forward(h, dsq + (i - 1), Lr, cm, fr); if (getenv("ORC_DBG")) fprintf(stderr, "[oracle] region forward %.12f\n", rf); }
          nc = region_trace_ensemble(h, dsq, i, j, fr, cm, n2sc, sig, ORC_MAXENV);
          mx_free(fr);
          nclustered_env += nc;
          for (d = 0; d < nc && r->nenv < ORC_MAXENV; d++) {
            const int i2 = sig[d].i, j2 = sig[d].j, Ld = j2 - i2 + 1;
            orc_len cu = len_config(L, 0);
            orc_mx *f2 = mx_new(Ld, M);
            double envsc = forward(h, dsq + (i2 - 1), Ld, cu, f2);
            float domcorr = 0.0f;
            int pos;
            for (pos = i2; pos <= j2; pos++) domcorr += n2sc[pos];
            r->env_i[r->nenv] = i2; r->env_j[r->nenv] = j2; r->env_multi[r->nenv] = 1;
            r->envsc[r->nenv] = (float) envsc; r->domcorr[r->nenv] = domcorr;
            r->nenv++;
            mx_free(f2);
          }
        } else if (r->nenv < ORC_MAXENV) {
          /* A.5 rescore the envelope in unihit mode, length model of the full sequence */
          int Ld = j - i + 1, pos;
          orc_len cu = len_config(L, 0);
          orc_mx *f2 = mx_new(Ld, M), *b2 = mx_new(Ld, M);
          float null2[32];
          double envsc = forward(h, dsq + (i - 1), Ld, cu, f2);
          float domcorr = 0.0f;
          backward(h, dsq + (i - 1), Ld, cu, b2);
          null2_by_expectation(h, f2, b2, Ld, cu, envsc, null2);
          for (pos = i; pos <= j; pos++) n2sc[pos] = logf(null2[dsq[pos - 1]]);
          for (pos = i; pos <= j; pos++) domcorr += n2sc[pos];
          r->env_i[r->nenv] = i; r->env_j[r->nenv] = j; r->env_multi[r->nenv] = multi;
          r->envsc[r->nenv] = (float) envsc; r->domcorr[r->nenv] = domcorr;
          r->nenv++;
          mx_free(f2); mx_free(b2);
        }
        i = -1; triggered = 0;
      }
    }
  }

  if (r->nregions > 0 && r->nenv > 0) {
    /* A.6 score assembly, float32 where HMMER is float32 */
    float seqbias = 0.0f, pre_score, seq_score, sum_score = 0.0f, sb2 = 0.0f;
    int Ld = 0, d;
    const double LOG2 = 0.69314718055994529;
    const float omega = 1.0f / 256.0f;
    for (i = 0; i <= L; i++) seqbias += n2sc[i];
    seqbias = flogsum(0.0f, (float) (log((double) omega) + seqbias));
    pre_score = (float) ((fwdsc_f - nullsc) / LOG2);
    seq_score = (float) ((fwdsc_f - (nullsc + seqbias)) / LOG2);
    r->seqbias_nats = seqbias;
    for (d = 0; d < r->nenv; d++) {
      if ((float) r->envsc[d] - (float) r->domcorr[d] > 0.0f) {
        sum_score += (float) r->envsc[d];
        Ld += r->env_j[d] - r->env_i[d] + 1;
        sb2 += (float) r->domcorr[d];
      }
    }
    sb2 = flogsum(0.0f, (float) (log((double) omega) + sb2));
    sum_score += (float) ((L - Ld) * log((float) L / (float) (L + 3)));
    {
      float pre2 = (float) ((sum_score - nullsc) / LOG2);
      sum_score = (float) ((sum_score - (nullsc + sb2)) / LOG2);
      r->sum_score = sum_score;
      if (Ld > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2; r->flags |= ORC_FLAG_OVERRIDE; }
    }
    r->pre_score = pre_score; r->seq_score = seq_score;
    r->decibits = decibits_of(seq_score);
    r->flags |= ORC_FLAG_REPORTED;
  }
  free(btot); free(etot); free(mocc); free(n2sc);
  mx_free(fx); mx_free(bx);
  return 0;
}

/* ----------------------------------------------------------------------------
 * A.7 hmmalign: unihit-local Forward/Backward, posterior decoding, optimal-accuracy
 * DP and traceback.  cols[r] = 0-based match column of residue r, or -1 when the
 * residue is emitted by an insert state or by the N/C flanks - exactly what the
 * reference derives from the Stockholm row (witch_msa/gcmm/aligner.py:126-142).
 * -------------------------------------------------------------------------- */
int orc_align_pair(const orc_hmm *h, const uint8_t *dsq, int L, int32_t *cols)
{
  int M = h->M, i, k;
  orc_len c = len_config(L, 0);
  orc_mxx *fx, *bx;
  float *ppM, *ppI, *ppN, *ppC, *ppJ;    /* posteriors, float32 like HMMER's matrices */
  float *oM, *oI, *oD, *oX;              /* OA matrices; oX: N B E J C per row         */
  const double *pt = h->pt, *en = h->entry;
  long double fwd;
  size_t W = (size_t) M + 1;
  for (i = 0; i < L; i++) cols[i] = -1;
  if (L <= 0) return 0;
  fx = mxx_new(L, M); bx = mxx_new(L, M);
  fwd = forward_x(h, dsq, L, c, fx);
  backward_x(h, dsq, L, c, bx);
  if (!isfinite((double) fwd)) { mxx_free(fx); mxx_free(bx); return -1; }
  ppM = (float *) calloc((size_t) (L + 1) * W, sizeof(float));
  ppI = (float *) calloc((size_t) (L + 1) * W, sizeof(float));
  ppN = (float *) calloc((size_t) L + 1, sizeof(float));
  ppC = (float *) calloc((size_t) L + 1, sizeof(float));
  ppJ = (float *) calloc((size_t) L + 1, sizeof(float));
  for (i = 1; i <= L; i++) {
    long double sc = expl(fx->lscale[i] + bx->lscale[i] - fwd);
    long double sc2 = expl(fx->lscale[i - 1] + bx->lscale[i] - fwd);
    for (k = 1; k <= M; k++) {
      ppM[i * W + k] = (float) (MX(fx, i, k, 0) * MX(bx, i, k, 0) * sc);
      ppI[i * W + k] = (float) (MX(fx, i, k, 1) * MX(bx, i, k, 1) * sc);
    }
    ppN[i] = (float) (XS(fx, i - 1, sN) * XS(bx, i, sN) * c.loop * sc2);
    ppJ[i] = (float) (XS(fx, i - 1, sJ) * XS(bx, i, sJ) * c.loop * sc2);
    ppC[i] = (float) (XS(fx, i - 1, sC) * XS(bx, i, sC) * c.loop * sc2);
  }
  mxx_free(fx); mxx_free(bx);

  oM = (float *) calloc((size_t) (L + 1) * W, sizeof(float));
  oI = (float *) calloc((size_t) (L + 1) * W, sizeof(float));
  oD = (float *) calloc((size_t) (L + 1) * W, sizeof(float));
  oX = (float *) calloc((size_t) (L + 1) * 5, sizeof(float));
#define GATE(t, v) (((t) > 0.0) ? (v) : 0.0f)
  {
    float tNl = c.loop > 0.0 ? 1.0f : 0.0f, tNm = c.move > 0.0 ? 1.0f : 0.0f;
    float tEJ = c.EJ > 0.0 ? 1.0f : 0.0f, tEC = c.EC > 0.0 ? 1.0f : 0.0f;
    /* row 0 as HMMER's optimal-accuracy fill initialises it: cells, E, J, C = -inf; N = B = 0 */
    oX[0 * 5 + sN] = 0.0f; oX[0 * 5 + sB] = 0.0f; oX[0 * 5 + sE] = -INFINITY;
    oX[0 * 5 + sJ] = -INFINITY; oX[0 * 5 + sC] = -INFINITY;
    for (k = 0; k <= M; k++) { oM[k] = -INFINITY; oI[k] = -INFINITY; oD[k] = -INFINITY; }
    for (i = 1; i <= L; i++) {
      float xB = oX[(i - 1) * 5 + sB], xE = -INFINITY;
      for (k = 1; k <= M; k++) {
        const double *tp = pt + (size_t) (k - 1) * 7;
        const double *tk = pt + (size_t) k * 7;
        float sv = GATE(en[k], xB), t;
        t = GATE(tp[tMM], oM[(i - 1) * W + k - 1]); if (t > sv) sv = t;
        t = GATE(tp[tIM], oI[(i - 1) * W + k - 1]); if (t > sv) sv = t;
        t = GATE(tp[tDM], oD[(i - 1) * W + k - 1]); if (t > sv) sv = t;
        sv += ppM[i * W + k];
        oM[i * W + k] = sv;
        if (sv > xE) xE = sv;
        {
          float a = GATE(tk[tMI], oM[(i - 1) * W + k]), b = GATE(tk[tII], oI[(i - 1) * W + k]);
          oI[i * W + k] = (a > b ? a : b) + ppI[i * W + k];
        }
        {
          float a = GATE(tp[tMD], oM[i * W + k - 1]), b = GATE(tp[tDD], oD[i * W + k - 1]);
          oD[i * W + k] = a > b ? a : b;
          if (oD[i * W + k] > xE) xE = oD[i * W + k];
        }
      }
      oX[i * 5 + sE] = xE;
      {
        float a = tNl * (oX[(i - 1) * 5 + sJ] + ppJ[i]), b = tEJ * xE;
        oX[i * 5 + sJ] = a > b ? a : b;
        a = tNl * (oX[(i - 1) * 5 + sC] + ppC[i]); b = tEC * xE;
        oX[i * 5 + sC] = a > b ? a : b;
        oX[i * 5 + sN] = tNl * (oX[(i - 1) * 5 + sN] + ppN[i]);
        a = tNm * oX[i * 5 + sN]; b = tNm * oX[i * 5 + sJ];
        oX[i * 5 + sB] = a > b ? a : b;
      }
    }
    /* traceback (first maximum wins, candidate order as in A.7) */
    {
      enum { stS, stN, stB, stM, stI, stD, stE, stJ, stC, stT };
      int s0 = stC, s1, guard = 4 * (L + M) + 16;
      i = L; k = 0;
      while (s0 != stS && guard-- > 0) {
        switch (s0) {
        case stC: {
          float a = tNl * (oX[(i - 1 < 0 ? 0 : i - 1) * 5 + sC] + ppC[i]), b = tEC * oX[i * 5 + sE];
          if (i == 0) s1 = stE; else s1 = (b > a) ? stE : stC;
          break; }
        case stJ: {
          float a = tNl * (oX[(i - 1 < 0 ? 0 : i - 1) * 5 + sJ] + ppJ[i]), b = tEJ * oX[i * 5 + sE];
          if (i == 0) s1 = stE; else s1 = (b > a) ? stE : stJ;
          break; }
        case stE: {
          /* argmax over M (">=": later wins) and D (">"), scanned in HMMER's striped order
           * k = r*Q + q + 1 (q outer over Q = max(2, ceil(M/4)) vectors, r inner over 4 lanes);
           * the order matters only when two cells tie exactly (A.7). */
          float mx = -INFINITY; int kmax = 0, smax = stM, kk, q, rr;
          int Q = (M - 1) / 4 + 1; if (Q < 2) Q = 2;
          for (q = 0; q < Q; q++) {
            for (rr = 0; rr < 4; rr++) { kk = rr * Q + q + 1; if (kk <= M && oM[i * W + kk] >= mx) { mx = oM[i * W + kk]; smax = stM; kmax = kk; } }
            for (rr = 0; rr < 4; rr++) { kk = rr * Q + q + 1; if (kk <= M && oD[i * W + kk] >  mx) { mx = oD[i * W + kk]; smax = stD; kmax = kk; } }
          }
          if (kmax == 0) { fprintf(stderr, "oracle: OA traceback found no cell at i=%d L=%d M=%d oE=%g fwd=%g\n", i, L, M, oX[i*5+sE], (double) fwd); guard = 0; s1 = stS; break; }
          k = kmax; s1 = smax;
          break; }
        case stM: {
          const double *tp = pt + (size_t) (k - 1) * 7;
          float path[4]; int best = 0, q;
          /* candidate order B, M, I, D (the order HMMER's vectorised transition table is
           * walked in); B first also makes the start cell (i=1 or k=1, all candidates 0) end in B */
          path[0] = GATE(en[k], oX[(i - 1) * 5 + sB]);
          path[1] = GATE(tp[tMM], oM[(i - 1) * W + k - 1]);
          path[2] = GATE(tp[tIM], oI[(i - 1) * W + k - 1]);
          path[3] = GATE(tp[tDM], oD[(i - 1) * W + k - 1]);
          for (q = 1; q < 4; q++) if (path[q] > path[best]) best = q;
          s1 = best == 0 ? stB : best == 1 ? stM : best == 2 ? stI : stD;
          cols[i - 1] = k - 1;
          k--; i--;
          break; }
        case stD: {
          const double *tp = pt + (size_t) (k - 1) * 7;
          float a = GATE(tp[tMD], oM[i * W + k - 1]), b = GATE(tp[tDD], oD[i * W + k - 1]);
          s1 = (b > a) ? stD : stM;
          k--;
          break; }
        case stI: {
          const double *tk = pt + (size_t) k * 7;
          float a = GATE(tk[tMI], oM[(i - 1) * W + k]), b = GATE(tk[tII], oI[(i - 1) * W + k]);
          s1 = (b > a) ? stI : stM;
          i--;
          break; }
        case stB: {
          float a = tNm * oX[i * 5 + sN], b = tNm * oX[i * 5 + sJ];
          s1 = (b > a) ? stJ : stN;
          break; }
        case stN:
          s1 = (i == 0) ? stS : stN;
          break;
        default: s1 = stS; break;
        }
        if ((s1 == stN || s1 == stJ || s1 == stC) && s1 == s0) i--;
        s0 = s1;
      }
    }
  }
#undef GATE
  free(ppM); free(ppI); free(ppN); free(ppC); free(ppJ);
  free(oM); free(oI); free(oD); free(oX);
  return 0;
}

/* ----------------------------------------------------------------------------
 * Batch drivers (OpenMP over pairs); residues are digital codes, CSR offsets.
 * -------------------------------------------------------------------------- */
int orc_score_batch(orc_hmm *const *hmms, int nh, const uint8_t *residues, const int64_t *offsets,
                    int64_t nq, int32_t *decibits, uint8_t *flags, double *fwd_bits, float *seq_score,
                    int nthreads)
{
  int64_t np = nq * nh, p;
  flogsum_init();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
  for (p = 0; p < np; p++) {
    int64_t q = p / nh; int hh = (int) (p % nh);
    orc_result r;
    orc_score_pair(hmms[hh], residues + offsets[q], (int) (offsets[q + 1] - offsets[q]), &r);
    decibits[p] = r.decibits;
    flags[p] = (uint8_t) r.flags;
    if (fwd_bits) fwd_bits[p] = r.fwd_bits;
    if (seq_score) seq_score[p] = r.seq_score;
  }
  return 0;
}

/* the same for a LIST of (query, model) pairs: the stratified samples of the headline-size tests */
int orc_score_pairs(orc_hmm *const *hmms, const uint8_t *residues, const int64_t *offsets,
                    const int64_t *pair_q, const int32_t *pair_h, int64_t npairs, int32_t *decibits, uint8_t *flags,
                    double *fwd_bits, float *seq_score, int nthreads)
{
  int64_t p;
  flogsum_init();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
  for (p = 0; p < npairs; p++) {
    int64_t q = pair_q[p];
    orc_result r;
    orc_score_pair(hmms[pair_h[p]], residues + offsets[q], (int) (offsets[q + 1] - offsets[q]), &r);
    decibits[p] = r.decibits;
    flags[p] = (uint8_t) r.flags;
    if (fwd_bits) fwd_bits[p] = r.fwd_bits;
    if (seq_score) seq_score[p] = r.seq_score;
  }
  return 0;
}

int orc_align_batch(orc_hmm *const *hmms, const uint8_t *residues, const int64_t *offsets,
                    const int64_t *pair_q, const int32_t *pair_h, int64_t npairs,
                    const int64_t *col_offsets, int32_t *cols, int nthreads)
{
  int64_t p;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (p = 0; p < npairs; p++) {
    int64_t q = pair_q[p];
    orc_align_pair(hmms[pair_h[p]], residues + offsets[q], (int) (offsets[q + 1] - offsets[q]),
                   cols + col_offsets[p]);
  }
  return 0;
}

int orc_result_size(void) { return (int) sizeof(orc_result); }

/* Design aid (tools/band_stats.py): for the FIRST envelope of a pair, the range of model nodes whose
 * lane block (Q nodes per block) holds a cell with posterior >= 2^log2eps on some envelope row -
 * out[0..1] from the multihit Forward/Backward of the whole sequence (what the scoring kernel knows
 * after its first two sweeps), out[2..3] from the unihit Forward/Backward of the envelope itself
 * (what the envelope sweeps need), out[4..5] = the envelope.  Block maxima over M, I, D. */
int orc_band_stats(const orc_hmm *h, const uint8_t *dsq, int L, int Q, double log2eps, int *out)
{
  int M = h->M, i, k, which;
  orc_result r;
  orc_len cm = len_config(L, 1), cu = len_config(L, 0);
  const double eps = exp2(log2eps);
  orc_score_pair(h, dsq, L, &r);
  if (r.nenv < 1) return 0;
  out[4] = r.env_i[0]; out[5] = r.env_j[0];
  for (which = 0; which < 2; which++) {
    const int i0 = which ? 1 : r.env_i[0], i1 = which ? r.env_j[0] - r.env_i[0] + 1 : r.env_j[0];
    const int Lx = which ? r.env_j[0] - r.env_i[0] + 1 : L;
    const uint8_t *x = which ? dsq + (r.env_i[0] - 1) : dsq;
    orc_mx *fx = mx_new(Lx, M), *bx = mx_new(Lx, M);
    double fwd = forward(h, x, Lx, which ? cu : cm, fx);
    int lo = M + 1, hi = 0;
    backward(h, x, Lx, which ? cu : cm, bx);
    for (i = i0; i <= i1; i++) {
      const double sc = exp(fx->lscale[i] + bx->lscale[i] - fwd);
      int b;
      for (b = 0; b * Q < M; b++) {
        double fm = 0.0, bm = 0.0;
        for (k = b * Q + 1; k <= (b + 1) * Q && k <= M; k++) {
          int st;
          for (st = 0; st < 3; st++) { if (MX(fx, i, k, st) > fm) fm = MX(fx, i, k, st); if (MX(bx, i, k, st) > bm) bm = MX(bx, i, k, st); }
        }
        if (fm * bm * sc >= eps) { if (b * Q + 1 < lo) lo = b * Q + 1; if ((b + 1) * Q > hi) hi = (b + 1) * Q; }
      }
    }
    out[2 * which] = lo; out[2 * which + 1] = hi;
    mx_free(fx); mx_free(bx);
  }
  return 1;
}
