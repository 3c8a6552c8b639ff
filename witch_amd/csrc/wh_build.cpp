// eHMM construction on the host: the model that
//     hmmbuild --cpu 1 --<mol> --ere <ere> --symfrac <symfrac> --informat afa
// (HMMER 3.1b2; the reference's call: witch_msa/gcmm/algorithm.py:463-470) writes for one subset
// alignment, as HMMER3/f text.  SURVEY.md section 8f #3.
//
// HMMER's source is not part of the reference checkout.  This file restates the published algorithm of
// p7_Builder for exactly that command line, keeping Easel's evaluation order and float/double types so that
// the printed five-decimal fields agree; it is pinned on the files the bundled hmmbuild binary produced for
// the golden cases (tests/test_hmmbuild_host.py: every probability field of 28 models, Neff to the digit):
//   1. relative weights: Henikoff position-based (--wpb, the default) over the canonical residues of every
//      column, divided by the sequence's count of canonical residues, scaled to sum to nseq (double);
//   2. fragments (--fragthresh 0.5): a sequence whose first..last residue span is shorter than half the
//      alignment has its leading and trailing gaps turned into missing data;
//   3. match columns (fast construction): every column with residue weight r > 0 and r / (r + gaps) >= symfrac;
//   4. weighted float32 counts of emissions and transitions along each sequence's implied path (gap in a
//      match column = delete, residue elsewhere = insert; nothing is counted into or out of missing data);
//   5. effective sequence number by entropy weighting (--eent): Neff <= nseq such that the mean match
//      relative entropy after the priors equals max(ere, (45 - log2(2 / (M (M+1)))) / M); Easel's bisection
//      step for step (bracket tested before it is narrowed, absolute tolerance 0.01);
//   6. parameters = posterior mean under HMMER's default mixture Dirichlet priors (constants below: nucleic
//      and amino; they are data of the published defaults);
//   7. composition (occupancy-weighted), consensus letters, MAP, and the text file.
// NOT written: the STATS lines (E-value calibration by simulation) and MAXL - this path's consumer
// (wh_ehmm_load) reads probabilities only; HMMER's own hmmsearch refuses a file without STATS.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/witch_hip.h"

namespace wh {
void set_error(const char *fmt, ...);
}

namespace {

enum { tMM = 0, tMI, tMD, tIM, tII, tDM, tDD };

struct Alphabet {
  int K = 0, Kp = 0;
  const char *syms = nullptr, *name = nullptr;
  uint8_t code[256];
  uint8_t degen[32][20];
  int ndegen[32];
  float bg[20];
  float cons_thresh = 0.f;
};

const double kAminoBgD[20] = {0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198,
                              0.0590092, 0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639,
                              0.0540978, 0.0683364, 0.0540687, 0.0673417, 0.0114135, 0.0304133};

bool make_alphabet(const char *mol, Alphabet &a) {
  memset(a.code, 255, sizeof a.code);
  memset(a.degen, 0, sizeof a.degen);
  memset(a.ndegen, 0, sizeof a.ndegen);
  struct D { char c; const char *m; };
  const D *degen = nullptr;
  int nd = 0;
  static const D dna_d[] = {{'R', "AG"}, {'Y', "CT"}, {'M', "AC"}, {'K', "GT"}, {'S', "CG"}, {'W', "AT"}, {'H', "ACT"},
                            {'B', "CGT"}, {'V', "ACG"}, {'D', "AGT"}, {'N', "ACGT"}};
  static const D aa_d[] = {{'B', "ND"}, {'J', "IL"}, {'Z', "QE"}, {'O', "K"}, {'U', "C"}, {'X', "ACDEFGHIKLMNPQRSTVWY"}};
  const char *syn = "";
  if (!strcmp(mol, "dna") || !strcmp(mol, "rna")) {
    a.syms = !strcmp(mol, "dna") ? "ACGT-RYMKSWHBVDN*~" : "ACGU-RYMKSWHBVDN*~";
    a.name = !strcmp(mol, "dna") ? "DNA" : "RNA";
    a.K = 4; degen = dna_d; nd = 11;
    syn = !strcmp(mol, "dna") ? "UTXNIA_-.-" : "TUXNIA_-.-";
    for (int x = 0; x < 4; x++) a.bg[x] = 0.25f;
    a.cons_thresh = 0.9f;
  } else if (!strcmp(mol, "amino")) {
    a.syms = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~"; a.name = "amino";
    a.K = 20; degen = aa_d; nd = 6;
    syn = "_-.-";
    for (int x = 0; x < 20; x++) a.bg[x] = (float)kAminoBgD[x];
    a.cons_thresh = 0.5f;
  } else return false;
  a.Kp = (int)strlen(a.syms);
  for (int i = 0; i < a.Kp; i++) {
    a.code[(unsigned char)a.syms[i]] = (uint8_t)i;
    a.code[(unsigned char)tolower(a.syms[i])] = (uint8_t)i;
  }
  for (const char *p = syn; p[0] && p[1]; p += 2) {
    a.code[(unsigned char)p[0]] = a.code[(unsigned char)p[1]];
    a.code[(unsigned char)tolower(p[0])] = a.code[(unsigned char)p[1]];
  }
  for (int x = 0; x < a.K; x++) { a.degen[x][x] = 1; a.ndegen[x] = 1; }
  for (int d = 0; d < nd; d++) {
    const int x = a.code[(unsigned char)degen[d].c];
    for (const char *m = degen[d].m; *m; m++) { a.degen[x][a.code[(unsigned char)*m]] = 1; a.ndegen[x]++; }
  }
  return true;
}

inline bool is_residue(const Alphabet &a, uint8_t x) { return x < a.K || (x > a.K && x < a.Kp - 2); }
inline bool is_gap(const Alphabet &a, uint8_t x) { return x == a.K; }
inline bool is_missing(const Alphabet &a, uint8_t x) { return x == a.Kp - 1; }

// ---- HMMER's default priors (p7_prior_CreateNucleic / p7_prior_CreateAmino)
struct Mix { int N, K; const double *pq; const double *alpha; const double *lg_alpha = nullptr; const double *lg_alpha_sum = nullptr; };

const double one[1] = {1.0};
const double nuc_tm[3] = {2.0, 0.1, 0.1}, nuc_ti[2] = {0.06, 0.2}, nuc_td[2] = {0.1, 0.2};
const double nuc_emq[4] = {0.24, 0.26, 0.08, 0.42};
const double nuc_em[16] = {0.16, 0.45, 0.12, 0.39, 0.09, 0.03, 0.09, 0.04, 1.29, 0.40, 6.58, 0.51, 1.74, 1.49, 1.57, 1.95};
const double nuc_ei[4] = {1.0, 1.0, 1.0, 1.0};
const double aa_tm[3] = {0.7939, 0.0278, 0.0135}, aa_ti[2] = {0.1551, 0.1331}, aa_td[2] = {0.9002, 0.5630};
const double aa_emq[9] = {0.178091, 0.056591, 0.0960191, 0.0781233, 0.0834977, 0.0904123, 0.114468, 0.0682132, 0.234585};
const double aa_em[180] = {
    0.270671, 0.039848, 0.017576, 0.016415, 0.014268, 0.131916, 0.012391, 0.022599, 0.020358, 0.030727, 0.015315, 0.048298, 0.053803, 0.020662, 0.023612, 0.216147, 0.147226, 0.065438, 0.003758, 0.009621,
    0.021465, 0.0103, 0.011741, 0.010883, 0.385651, 0.016416, 0.076196, 0.035329, 0.013921, 0.093517, 0.022034, 0.028593, 0.013086, 0.023011, 0.018866, 0.029156, 0.018153, 0.0361, 0.07177, 0.419641,
    0.561459, 0.045448, 0.438366, 0.764167, 0.087364, 0.259114, 0.21494, 0.145928, 0.762204, 0.24732, 0.118662, 0.441564, 0.174822, 0.53084, 0.465529, 0.583402, 0.445586, 0.22705, 0.02951, 0.12109,
    0.070143, 0.01114, 0.019479, 0.094657, 0.013162, 0.048038, 0.077, 0.032939, 0.576639, 0.072293, 0.02824, 0.080372, 0.037661, 0.185037, 0.506783, 0.073732, 0.071587, 0.042532, 0.011254, 0.028723,
    0.041103, 0.014794, 0.00561, 0.010216, 0.153602, 0.007797, 0.007175, 0.299635, 0.010849, 0.999446, 0.210189, 0.006127, 0.013021, 0.019798, 0.014509, 0.012049, 0.035799, 0.180085, 0.012744, 0.026466,
    0.115607, 0.037381, 0.012414, 0.018179, 0.051778, 0.017255, 0.004911, 0.796882, 0.017074, 0.285858, 0.075811, 0.014548, 0.015092, 0.011382, 0.012696, 0.027535, 0.088333, 0.94434, 0.004373, 0.016741,
    0.093461, 0.004737, 0.387252, 0.347841, 0.010822, 0.105877, 0.049776, 0.014963, 0.094276, 0.027761, 0.01004, 0.187869, 0.050018, 0.110039, 0.038668, 0.119471, 0.065802, 0.02543, 0.003215, 0.018742,
    0.452171, 0.114613, 0.06246, 0.115702, 0.284246, 0.140204, 0.100358, 0.55023, 0.143995, 0.700649, 0.27658, 0.118569, 0.09747, 0.126673, 0.143634, 0.278983, 0.358482, 0.66175, 0.061533, 0.199373,
    0.005193, 0.004039, 0.006722, 0.006121, 0.003468, 0.016931, 0.003647, 0.002184, 0.005019, 0.00599, 0.001473, 0.004158, 0.009055, 0.00363, 0.006583, 0.003172, 0.00369, 0.002967, 0.002772, 0.002686};
const double aa_ei[20] = {681, 120, 623, 651, 313, 902, 241, 371, 687, 676, 143, 548, 647, 415, 551, 926, 623, 505, 102, 269};

struct Prior { Mix tm, ti, td, em, ei; };
double esl_loggamma(double x);

Prior prior_for(int K) {
  // wh_hmmbuild is called from thread pools (ctypes releases the GIL): the tables are filled exactly once
  static double lg_nuc[16], lg_nuc_sum[4], lg_aa[180], lg_aa_sum[9];
  static std::once_flag once;
  std::call_once(once, [] {
    for (int q = 0; q < 4; q++) { double sum = 0.0; for (int x = 0; x < 4; x++) { lg_nuc[4 * q + x] = esl_loggamma(nuc_em[4 * q + x]); sum += nuc_em[4 * q + x]; } lg_nuc_sum[q] = esl_loggamma(sum); }
    for (int q = 0; q < 9; q++) { double sum = 0.0; for (int x = 0; x < 20; x++) { lg_aa[20 * q + x] = esl_loggamma(aa_em[20 * q + x]); sum += aa_em[20 * q + x]; } lg_aa_sum[q] = esl_loggamma(sum); }
  });
  if (K == 4) {
    Prior p{{1, 3, one, nuc_tm}, {1, 2, one, nuc_ti}, {1, 2, one, nuc_td}, {4, 4, nuc_emq, nuc_em}, {1, 4, one, nuc_ei}};
    p.em.lg_alpha = lg_nuc; p.em.lg_alpha_sum = lg_nuc_sum;
    return p;
  }
  Prior p{{1, 3, one, aa_tm}, {1, 2, one, aa_ti}, {1, 2, one, aa_td}, {9, 20, aa_emq, aa_em}, {1, 20, one, aa_ei}};
  p.em.lg_alpha = lg_aa; p.em.lg_alpha_sum = lg_aa_sum;
  return p;
}

// Easel's esl_stats_LogGamma (Lanczos, 11 coefficients, the constant ln sqrt(2 pi) to nine digits) in the
// evaluation order of the hmmbuild binary WITCH bundles (its compiler summed the series in two interleaved
// partial sums and formed the denominators from x + 10): the mixture coefficients of the match-emission
// prior are differences of ~30 such values, and a libm lgamma moves one printed digit in ~20 000 fields.
double esl_loggamma(double x) {
  static const double cof[11] = {46945.80336184385, -156060.5207784446, 206504.9568014106, -138893.4775095388,
                                 50317.96415085709, -9601.592329182778, 878.585593089525, -31.55153906098611,
                                 0.2908143421162229, -0.0002319827630494973, 1.251639670050933e-10};
  const double xx = x - 1.0;
  const double t10 = x + 10.0;
  double d0 = t10, d1 = t10 + -1.0, lo = 1.0, hi = 0.0;
  for (int i = 10; i >= 2; i -= 2) {
    lo += cof[i] / d0; hi += cof[i - 1] / d1;
    d0 += -2.0; d1 += -2.0;
  }
  double value = lo + hi;
  value = cof[0] / (t10 - 10.0) + value;
  value = std::log(value);
  const double tx = (11.0 + xx) + 0.5;
  return ((value + 0.918938533) + (xx + 0.5) * std::log(tx)) - tx;
}

// esl_mixdchlet_MPParameters: float counts in, float probabilities out
void mp_parameters(const float *cf, int K, const Mix &pri, float *pf) {
  double c[20], p[20], mix[16];
  for (int x = 0; x < K; x++) c[x] = cf[x];
  double totc = 0.0;
  for (int x = 0; x < K; x++) totc += c[x];
  if (pri.N > 1) {
    // (the terms that do not depend on the component, and the ones that depend on it alone, are evaluated once:
    // same values in the same sums as Easel's per-component loop, 25 instead of 60 LogGamma calls per node)
    double lg_c1[20], lg_sum3;
    {
      double sum3 = 0.0;
      for (int x = 0; x < K; x++) { sum3 += c[x]; lg_c1[x] = esl_loggamma(c[x] + 1.0); }
      lg_sum3 = esl_loggamma(sum3 + 1.0);
    }
    for (int q = 0; q < pri.N; q++) {
      const double *al = pri.alpha + (size_t)q * K;
      const double *lga = pri.lg_alpha + (size_t)q * K;
      double sum1 = 0.0, lnp = 0.0;
      for (int x = 0; x < K; x++) {
        sum1 += c[x] + al[x];
        const double a1 = esl_loggamma(al[x] + c[x]), a2 = lg_c1[x], a3 = lga[x];
        lnp += a1 - a2 - a3;
      }
      {
        const double a1 = esl_loggamma(sum1), a2 = pri.lg_alpha_sum[q], a3 = lg_sum3;
        lnp += a2 + a3 - a1;
      }
      mix[q] = pri.pq[q] > 0.0 ? lnp + std::log(pri.pq[q]) : -INFINITY;
    }
    double mx = mix[0];
    for (int q = 1; q < pri.N; q++) mx = std::max(mx, mix[q]);
    double s = 0.0;
    for (int q = 0; q < pri.N; q++) if (mix[q] > mx - 50.0) s += std::exp(mix[q] - mx);
    const double denom = mx + std::log(s);
    for (int q = 0; q < pri.N; q++) mix[q] = std::exp(mix[q] - denom);
    double ms = 0.0;
    for (int q = 0; q < pri.N; q++) ms += mix[q];
    for (int q = 0; q < pri.N; q++) mix[q] /= ms;
  } else mix[0] = 1.0;
  for (int x = 0; x < K; x++) p[x] = 0.0;
  for (int x = 0; x < K; x++)
    for (int q = 0; q < pri.N; q++) {
      const double *al = pri.alpha + (size_t)q * K;
      double tota = 0.0;
      for (int y = 0; y < K; y++) tota += al[y];
      p[x] += mix[q] * (c[x] + al[x]) / (totc + tota);
    }
  double ps = 0.0;
  for (int x = 0; x < K; x++) ps += p[x];
  for (int x = 0; x < K; x++) pf[x] = (float)(ps != 0.0 ? p[x] / ps : 1.0 / K);
}

struct Model {
  int M = 0, K = 0;
  std::vector<float> t, mat, ins;     // [M+1][7], [M+1][K], [M+1][K]
};

void fnorm(float *v, int n) {
  float s = 0.f;
  for (int i = 0; i < n; i++) s += v[i];
  if (s != 0.f) for (int i = 0; i < n; i++) v[i] /= s;
  else for (int i = 0; i < n; i++) v[i] = 1.0f / (float)n;
}

// p7_ParameterEstimation: counts -> probabilities in place
void parameter_estimation(Model &h, const Prior &pri) {
  const int M = h.M, K = h.K;
  for (int k = 0; k <= M; k++) mp_parameters(&h.t[(size_t)k * 7], 3, pri.tm, &h.t[(size_t)k * 7]);
  h.t[(size_t)M * 7 + tMD] = 0.f;
  fnorm(&h.t[(size_t)M * 7], 3);
  for (int k = 0; k <= M; k++) mp_parameters(&h.t[(size_t)k * 7 + 3], 2, pri.ti, &h.t[(size_t)k * 7 + 3]);
  for (int k = 1; k < M; k++) mp_parameters(&h.t[(size_t)k * 7 + 5], 2, pri.td, &h.t[(size_t)k * 7 + 5]);
  h.t[tDM] = 1.f; h.t[tDD] = 0.f;
  h.t[(size_t)M * 7 + tDM] = 1.f; h.t[(size_t)M * 7 + tDD] = 0.f;
  for (int k = 1; k <= M; k++) mp_parameters(&h.mat[(size_t)k * K], K, pri.em, &h.mat[(size_t)k * K]);
  for (int x = 0; x < K; x++) h.mat[x] = x == 0 ? 1.f : 0.f;
  for (int k = 0; k <= M; k++) mp_parameters(&h.ins[(size_t)k * K], K, pri.ei, &h.ins[(size_t)k * K]);
}

// p7_MeanMatchRelativeEntropy with esl_vec_FRelEntropy's float accumulation
double mean_match_relent(const Model &h, const float *bg) {
  double KL = 0.0;
  for (int k = 1; k <= h.M; k++) {
    const float *p = &h.mat[(size_t)k * h.K];
    float kl = 0.f;
    for (int i = 0; i < h.K; i++)
      if (p[i] > 0.f) kl += (float)((double)p[i] * std::log((double)(p[i] / bg[i])));
    KL += (double)(float)(1.44269504 * (double)kl);
  }
  return KL / (double)h.M;
}

#include "wh_calibrate.h"

// hmmbuild's MAXL line (p7_Builder_MaxLength at its default tail mass 1e-7): the length beyond which the core model,
// entered at match state 1 and left at its end, emits a sequence with probability below 1e-7.  Restated from the
// numbers: one unit of mass in M_1 at length 1; per step the delete states are closed over the current match masses,
// the next match / insert masses are formed, and the first length whose successors hold less than 1e-7 in total is
// the answer (+ 1).  This definition (among: uniform / unit starts over all match states, deletes counted as
// length, thresholds on the exited mass) is the one that gives the MAXL of all 38 nucleotide golden files, models of
// 13 .. 2 574 nodes (tests/test_hmmbuild_host.py).
int max_length(const Model &h) {
  const int M = h.M;
  std::vector<double> Mc((size_t)M + 2, 0.0), Ic((size_t)M + 2, 0.0), D((size_t)M + 2, 0.0), nM((size_t)M + 2, 0.0), nI((size_t)M + 2, 0.0);
  Mc[1] = 1.0;
  const int bound = 100000;
  for (int L = 1; L < bound; L++) {
    D[1] = 0.0;
    for (int k = 2; k <= M; k++) {
      const float *tp = &h.t[(size_t)(k - 1) * 7];
      D[(size_t)k] = Mc[(size_t)k - 1] * (double)tp[tMD] + D[(size_t)k - 1] * (double)tp[tDD];
    }
    double surv = 0.0;
    nM[1] = 0.0;
    for (int k = 2; k <= M; k++) {
      const float *tp = &h.t[(size_t)(k - 1) * 7];
      nM[(size_t)k] = Mc[(size_t)k - 1] * (double)tp[tMM] + Ic[(size_t)k - 1] * (double)tp[tIM] + D[(size_t)k - 1] * (double)tp[tDM];
      surv += nM[(size_t)k];
    }
    for (int k = 1; k < M; k++) {
      const float *tp = &h.t[(size_t)k * 7];
      nI[(size_t)k] = Mc[(size_t)k] * (double)tp[tMI] + Ic[(size_t)k] * (double)tp[tII];
      surv += nI[(size_t)k];
    }
    nI[(size_t)M] = 0.0;
    if (surv < 1e-7) return L + 1;
    Mc.swap(nM); Ic.swap(nI);
  }
  return bound;
}

void scale_model(Model &h, double scale) {
  const float s = (float)scale;
  for (float &v : h.t) v *= s;
  for (float &v : h.mat) v *= s;
  for (float &v : h.ins) v *= s;
}

void append(std::string &s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void append(std::string &s, const char *fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  const int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (n > 0) s.append(buf, (size_t)std::min<int>(n, (int)sizeof buf - 1));
}

void put_prob(std::string &s, float p) {
  if (p == 0.0f) append(s, " %8s", "*");
  else if (p == 1.0f) append(s, " %8.5f", 0.0);
  else append(s, " %8.5f", (double)-(float)std::log((double)p));   // a correctly rounded logf (the binary's libm; glibc's differs in rare last bits)
}

}  // namespace

extern "C" int wh_hmmbuild(const char *molecule, int32_t nseq, int64_t alen, const char *const *rows, const char *name,
                           double ere, double symfrac, double fragthresh, char **out_text, int64_t *out_len,
                           int32_t *out_M, double *out_neff) {
  return wh_hmmbuild2(molecule, nseq, alen, rows, name, ere, symfrac, fragthresh, 0, out_text, out_len, out_M, out_neff);
}

extern "C" int wh_hmmbuild2(const char *molecule, int32_t nseq, int64_t alen, const char *const *rows, const char *name,
                            double ere, double symfrac, double fragthresh, int32_t flags, char **out_text, int64_t *out_len,
                            int32_t *out_M, double *out_neff) {
  if (!molecule || !rows || !out_text || !out_len || nseq < 1 || alen < 1) { wh::set_error("wh_hmmbuild: bad argument"); return WH_EINVAL; }
  Alphabet abc;
  if (!make_alphabet(molecule, abc)) { wh::set_error("wh_hmmbuild: unknown molecule '%s' (dna, rna, amino)", molecule); return WH_EINVAL; }
  const int K = abc.K;
  // ---- digitize
  std::vector<uint8_t> ax((size_t)nseq * alen);
  for (int i = 0; i < nseq; i++) {
    if (!rows[i]) { wh::set_error("wh_hmmbuild: row %d is NULL", i); return WH_EINVAL; }
    for (int64_t p = 0; p < alen; p++) {
      const unsigned char ch = (unsigned char)rows[i][p];
      const uint8_t x = abc.code[ch];
      if (ch == 0 || x == 255) { wh::set_error("wh_hmmbuild: row %d, column %lld: character 0x%02x is not in the %s alphabet (or the row is short)", i, (long long)p + 1, ch, abc.name); return WH_EINVAL; }
      ax[(size_t)i * alen + p] = x;
    }
  }
  // ---- alignment checksum (Jenkins one-at-a-time over the digital residues), as hmmbuild's CKSUM line
  uint32_t cksum = 0;
  for (size_t z = 0; z < ax.size(); z++) { cksum += ax[z]; cksum += (cksum << 10); cksum ^= (cksum >> 6); }
  cksum += (cksum << 3); cksum ^= (cksum >> 11); cksum += (cksum << 15);
  // ---- 1. position-based weights
  std::vector<double> wgt((size_t)nseq, 0.0);
  {
    std::vector<int> nres((size_t)K);
    for (int64_t p = 0; p < alen; p++) {
      std::fill(nres.begin(), nres.end(), 0);
      for (int i = 0; i < nseq; i++) { const uint8_t x = ax[(size_t)i * alen + p]; if (x < K) nres[x]++; }
      int ntotal = 0;
      for (int x = 0; x < K; x++) if (nres[x] > 0) ntotal++;
      if (ntotal == 0) continue;
      for (int i = 0; i < nseq; i++) { const uint8_t x = ax[(size_t)i * alen + p]; if (x < K) wgt[i] += 1.0 / (double)(ntotal * nres[x]); }
    }
    for (int i = 0; i < nseq; i++) {
      int64_t rlen = 0;
      for (int64_t p = 0; p < alen; p++) rlen += ax[(size_t)i * alen + p] < K;   // canonical residues only (probed)
      if (rlen > 0) wgt[i] /= (double)rlen;
    }
    double sum = 0.0;
    for (int i = 0; i < nseq; i++) sum += wgt[i];
    if (sum != 0.0) for (int i = 0; i < nseq; i++) wgt[i] /= sum;
    else for (int i = 0; i < nseq; i++) wgt[i] = 1.0 / (double)nseq;
    for (int i = 0; i < nseq; i++) wgt[i] *= (double)nseq;
  }
  // ---- 2. fragments
  for (int i = 0; i < nseq; i++) {
    uint8_t *row = &ax[(size_t)i * alen];
    int64_t lo = 0, hi = alen - 1;
    while (lo < alen && is_gap(abc, row[lo])) lo++;
    while (hi >= 0 && is_gap(abc, row[hi])) hi--;
    const int64_t span = hi - lo + 1;
    if ((double)span <= fragthresh * (double)alen) {     // (<=: probed with hmmbuild at even and odd alignment lengths)
      for (int64_t p = 0; p < lo && p < alen; p++) row[p] = (uint8_t)(abc.Kp - 1);
      for (int64_t p = alen - 1; p > hi && p >= 0; p--) row[p] = (uint8_t)(abc.Kp - 1);
    }
  }
  // ---- 3. match columns
  std::vector<uint8_t> match((size_t)alen, 0);
  std::vector<int32_t> matcol;      // 1-based alignment column of node k (index k-1)
  for (int64_t p = 0; p < alen; p++) {
    double r = 0.0, tot = 0.0;
    for (int i = 0; i < nseq; i++) {
      const uint8_t x = ax[(size_t)i * alen + p];
      if (is_residue(abc, x)) { r += wgt[i]; tot += wgt[i]; }
      else if (is_gap(abc, x)) tot += wgt[i];
    }
    if (r > 0.0 && r / tot >= symfrac) { match[p] = 1; matcol.push_back((int32_t)(p + 1)); }
  }
  const int M = (int)matcol.size();
  if (M < 1) { wh::set_error("wh_hmmbuild: the alignment has no consensus column"); return WH_EINVAL; }
  // ---- 4. counts
  Model cnt;
  cnt.M = M; cnt.K = K;
  cnt.t.assign((size_t)(M + 1) * 7, 0.f);
  cnt.mat.assign((size_t)(M + 1) * K, 0.f);
  cnt.ins.assign((size_t)(M + 1) * K, 0.f);
  enum { sM = 0, sI, sD, sX };
  for (int i = 0; i < nseq; i++) {
    const uint8_t *row = &ax[(size_t)i * alen];
    const float wt = (float)wgt[i];
    int pst = sM, pk = 0, k = 0;        // previous state: B as the "match" of node 0
    auto emit = [&](std::vector<float> &tab, int node, uint8_t x) {
      float *ct = &tab[(size_t)node * K];
      if (x < K) ct[x] += wt;
      else for (int y = 0; y < K; y++) if (abc.degen[x][y]) ct[y] += wt / (float)abc.ndegen[x];
    };
    auto step = [&](int st, int node) {
      if (st != sX && pst != sX) {
        float *t = &cnt.t[(size_t)pk * 7];
        if (pst == sM) t[st == sM ? tMM : st == sI ? tMI : tMD] += wt;
        else if (pst == sI) t[st == sM ? tIM : tII] += wt;
        else t[st == sM ? tDM : tDD] += wt;
      }
      pst = st; pk = node;
    };
    for (int64_t p = 0; p < alen; p++) {
      const uint8_t x = row[p];
      if (match[p]) k++;
      if (is_residue(abc, x)) {
        if (match[p]) { emit(cnt.mat, k, x); step(sM, k); }
        else { emit(cnt.ins, k, x); step(sI, k); }
      } else if (match[p] && is_gap(abc, x)) step(sD, k);
      else if (is_missing(abc, x)) { if (pst != sX) step(sX, k); }
    }
    step(sM, M + 1);                    // E: counted in the M->M / I->M / D->M slot of the last state
  }
  // ---- 5. effective sequence number
  const Prior pri = prior_for(K);
  const double etarget = std::max(ere, (45.0 - std::log2(2.0 / ((double)M * (double)(M + 1)))) / (double)M);
  // (only the match emissions enter the relative entropy: the transition and insert estimates of HMMER's
  // target function are not needed to evaluate it)
  Model h2;
  h2.M = M; h2.K = K;
  auto target_f = [&](double neff) {
    const float sc = (float)(neff / (double)nseq);
    h2.mat = cnt.mat;
    for (float &v : h2.mat) v *= sc;
    for (int k = 1; k <= M; k++) mp_parameters(&h2.mat[(size_t)k * K], K, pri.em, &h2.mat[(size_t)k * K]);
    return mean_match_relent(h2, abc.bg) - etarget;
  };
  double neff = (double)nseq;
  if (target_f((double)nseq) > 0.0) {
    double xl = 0.0, xr = (double)nseq, fxl = target_f(xl), x = xr;
    for (int it = 0; it < 100; it++) {
      x = (xl + xr) / 2.0;
      const double fx = target_f(x);
      if (fx == 0.0) break;
      if ((xr - xl) < 0.01 + 1e-12 * x || std::fabs(fx) < 1e-12) break;
      if ((fxl > 0.0) == (fx > 0.0)) { xl = x; fxl = fx; } else xr = x;
    }
    neff = x;
  }
  Model h = cnt;
  scale_model(h, neff / (double)nseq);
  // ---- 6. parameters
  parameter_estimation(h, pri);
  // ---- 7. annotation: occupancy, composition, consensus
  std::vector<float> mocc((size_t)M + 1, 0.f), iocc((size_t)M + 1, 0.f), compo((size_t)K, 0.f);
  mocc[1] = h.t[tMI] + h.t[tMM];
  for (int k = 2; k <= M; k++) {
    const float *tp = &h.t[(size_t)(k - 1) * 7];
    mocc[k] = (float)(mocc[k - 1] * (tp[tMM] + tp[tMI]) + (1.0 - mocc[k - 1]) * tp[tDM]);     // (HMMER's literal 1.0 is a double)
  }
  iocc[0] = h.t[tMI] / h.t[tIM];
  for (int k = 1; k <= M; k++) iocc[k] = mocc[k] * h.t[(size_t)k * 7 + tMI] / h.t[(size_t)k * 7 + tIM];
  for (int x = 0; x < K; x++) compo[x] += h.ins[x] * iocc[0];
  for (int k = 1; k <= M; k++)
    for (int pass = 0; pass < 2; pass++)
      for (int x = 0; x < K; x++) compo[x] += (pass == 0 ? h.mat[(size_t)k * K + x] * mocc[k] : h.ins[(size_t)k * K + x] * iocc[k]);
  fnorm(compo.data(), K);
  // ---- text
  std::string s;
  s.reserve((size_t)M * (size_t)(3 * (K + 8) * 9) + 1024);
  s += "HMMER3/f [3.1b2 | February 2015]\n";
  append(s, "NAME  %s\n", name && *name ? name : "sub");
  append(s, "LENG  %d\n", M);
  if (K == 4) append(s, "MAXL  %d\n", max_length(h));      // nucleotide models only, as hmmbuild
  append(s, "ALPH  %s\n", abc.name);
  s += "RF    no\nMM    no\nCONS  yes\nCS    no\nMAP   yes\n";
  append(s, "NSEQ  %d\n", nseq);
  append(s, "EFFN  %f\n", neff);
  append(s, "CKSUM %u\n", cksum);
  if (flags & WH_BUILD_STATS) {
    // E-value calibration (wh_calibrate.h): what stock HMMER needs to accept the file; this path never reads it
    CalibModel cm;
    cm.M = M; cm.K = K; cm.t = h.t.data(); cm.mat = h.mat.data(); cm.bg = abc.bg;
    double ev[4];
    calibrate_model(cm, mean_match_relent(h, abc.bg), ev);
    // (the model keeps them as float32, and that is what hmmbuild prints)
    append(s, "STATS LOCAL MSV      %8.4f %8.5f\n", (double)(float)ev[1], (double)(float)ev[0]);
    append(s, "STATS LOCAL VITERBI  %8.4f %8.5f\n", (double)(float)ev[2], (double)(float)ev[0]);
    append(s, "STATS LOCAL FORWARD  %8.4f %8.5f\n", (double)(float)ev[3], (double)(float)ev[0]);
  }
  s += "HMM     ";
  for (int x = 0; x < K; x++) append(s, "     %c   ", abc.syms[x]);
  s += "\n";
  append(s, "        %8s %8s %8s %8s %8s %8s %8s\n", "m->m", "m->i", "m->d", "i->m", "i->i", "d->m", "d->d");
  s += "  COMPO ";
  for (int x = 0; x < K; x++) put_prob(s, compo[x]);
  s += "\n        ";
  for (int x = 0; x < K; x++) put_prob(s, h.ins[x]);
  s += "\n        ";
  for (int z = 0; z < 7; z++) put_prob(s, h.t[z]);
  s += "\n";
  for (int k = 1; k <= M; k++) {
    append(s, " %6d ", k);
    const float *mk = &h.mat[(size_t)k * K];
    int best = 0;
    for (int x = 1; x < K; x++) if (mk[x] > mk[best]) best = x;
    for (int x = 0; x < K; x++) put_prob(s, mk[x]);
    const char c = mk[best] >= abc.cons_thresh ? (char)toupper(abc.syms[best]) : (char)tolower(abc.syms[best]);
    append(s, " %6d %c %c %c %c\n        ", matcol[(size_t)k - 1], c, '-', '-', '-');
    for (int x = 0; x < K; x++) put_prob(s, h.ins[(size_t)k * K + x]);
    s += "\n        ";
    for (int z = 0; z < 7; z++) put_prob(s, h.t[(size_t)k * 7 + z]);
    s += "\n";
  }
  s += "//\n";
  char *buf = (char *)malloc(s.size() + 1);
  if (!buf) { wh::set_error("wh_hmmbuild: out of memory"); return WH_ENOMEM; }
  memcpy(buf, s.data(), s.size());
  buf[s.size()] = 0;
  *out_text = buf;
  *out_len = (int64_t)s.size();
  if (out_M) *out_M = M;
  if (out_neff) *out_neff = neff;
  return WH_OK;
}

extern "C" void wh_free_text(char *p) { free(p); }
