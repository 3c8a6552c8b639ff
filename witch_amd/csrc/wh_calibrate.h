// E-value calibration of a freshly built model: the three "STATS LOCAL" lines hmmbuild 3.1b2 writes
// (MSV mu, Viterbi mu, Forward tau, one lambda).  Restated from HMMER's published behaviour (p7_Calibrate:
// p7_Lambda, p7_MSVMu, p7_ViterbiMu, p7_Tau with the builder's defaults EmL = EvL = 200, EfL = 100, 200 sequences
// each, tail mass 0.04; the generator is re-seeded with 42 for every model) and pinned on the STATS lines of the
// model files under tests/golden (written by the reference's bundled hmmbuild).  WITCH never reads these numbers
// (hmmsearch runs with -E 99999999 and only bit scores are parsed, witch_msa/gcmm/algorithm.py:526-532,
// loader.py:293); they make a file written by wh_hmmbuild acceptable to stock HMMER (-p <hmmdir> reruns,
// witch_msa/gcmm/gcmm.py:163-171).
//
// What has to be reproduced exactly for the printed digits to come out:
//  * the random sequences: Easel's "fast" generator (x <- 69069 x + 1, Jenkins-mixed seed) and esl_rnd_FChoose over
//    the float background, 600 sequences drawn in one stream (MSV, then Viterbi, then Forward);
//  * the MSV filter's 8-bit arithmetic (third-bit units, base 190, saturating unsigned adds / subtracts) and the
//    Viterbi filter's 16-bit arithmetic (1/500-bit units, base 12000, saturating signed adds): both are integer
//    dynamic programmes, so a scalar restatement gives the striped SSE code's numbers;
//  * the Forward score only enters a maximum-likelihood Gumbel fit of 200 values: float64 here against HMMER's
//    float32 parser moves tau in its fifth decimal at most.
// Included by wh_build.cpp inside its unnamed namespace (uses its transition indices tMM .. tDD).
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

struct CalibModel {
  int M, K;
  const float *t;      // [M+1][7]  node 0 = begin
  const float *mat;    // [M+1][K]
  const float *bg;     // [K]
};

struct CalibRng {
  uint32_t x;
  static uint32_t mix3(uint32_t a, uint32_t b, uint32_t c) {
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
  explicit CalibRng(uint32_t seed) { x = mix3(seed, 87654321u, 12345678u); if (x == 0) x = 42; }
  double next() { x = x * 69069u + 1u; return (double)x / 4294967296.0; }
  // esl_rnd_FChoose over K float probabilities (first i whose running sum / total exceeds the roll, in double)
  int choose(const float *p, int K) {
    const double roll = next();
    double norm = 0.0, sum = 0.0;
    for (int i = 0; i < K; i++) norm += p[i];
    for (int i = 0; i < K; i++) { sum += p[i]; if (sum / norm > roll) return i; }
    return K - 1;
  }
};

// the float32 log-odds profile p7_ProfileConfig builds in local multihit mode (what the optimized profile is converted from)
struct CalibProfile {
  int M, K;
  std::vector<float> msc;   // [M+1][K]   match emission scores
  std::vector<float> tsc;   // [M][7]     node k -> k+1 transitions, k = 0 .. M-1 (node 0: all -inf)
  std::vector<float> bm;    // [M]        B -> M_{k+1}, stored at k
};

inline void calib_profile(const CalibModel &h, CalibProfile &gm) {
  const int M = h.M, K = h.K;
  const float ninf = -INFINITY;
  gm.M = M; gm.K = K;
  gm.msc.assign((size_t)(M + 1) * K, ninf);
  gm.tsc.assign((size_t)M * 7, ninf);
  gm.bm.assign((size_t)M, ninf);
  // p7_hmm_CalculateOccupancy
  std::vector<float> occ((size_t)M + 1, 0.f);
  occ[1] = h.t[tMI] + h.t[tMM];
  for (int k = 2; k <= M; k++) {
    const float *tp = h.t + (size_t)(k - 1) * 7;
    occ[k] = (float)(occ[k - 1] * (tp[tMM] + tp[tMI]) + (1.0 - occ[k - 1]) * tp[tDM]);     // (HMMER's literal 1.0 is a double)
  }
  float Z = 0.f;
  for (int k = 1; k <= M; k++) Z += occ[k] * (float)(M - k + 1);
  for (int k = 1; k <= M; k++) gm.bm[(size_t)k - 1] = (float)std::log((double)(occ[k] / Z));
  for (int k = 1; k < M; k++)
    for (int z = 0; z < 7; z++) gm.tsc[(size_t)k * 7 + z] = (float)std::log((double)h.t[(size_t)k * 7 + z]);
  for (int k = 1; k <= M; k++)
    for (int x = 0; x < K; x++) gm.msc[(size_t)k * K + x] = (float)std::log((double)h.mat[(size_t)k * K + x] / (double)h.bg[x]);
}

// p7_bg_NullOne at length L
inline float calib_nullone(int L) {
  const float p1 = (float)L / (float)(L + 1);
  return (float)((float)L * std::log((double)p1) + std::log(1. - (double)p1));
}

// ---- MSV filter (8-bit) ------------------------------------------------------------------------------------
struct CalibMSV {
  int M, K;
  float scale;
  uint8_t base, bias, tbm, tec, tjb;
  std::vector<uint8_t> rb;   // [K][M+1] biased match costs
  uint8_t unbiased(float sc) const { sc = -1.0f * roundf(scale * sc); return sc > 255.f ? 255 : (uint8_t)sc; }
  uint8_t biased(float sc) const { sc = -1.0f * roundf(scale * sc); return sc > (float)(255 - bias) ? 255 : (uint8_t)((uint8_t)sc + bias); }
};

inline void calib_msv_convert(const CalibProfile &gm, CalibMSV &om) {
  const int M = gm.M, K = gm.K;
  om.M = M; om.K = K;
  float mx = 0.0f;       // (the insert scores, all 0, are part of the maximum)
  for (int k = 1; k <= M; k++) for (int x = 0; x < K; x++) mx = std::max(mx, gm.msc[(size_t)k * K + x]);
  om.scale = (float)(3.0 / 0.69314718055994529);
  om.base = 190;
  om.bias = om.unbiased(-1.0f * mx);
  om.rb.assign((size_t)K * (M + 1), 255);
  for (int x = 0; x < K; x++) for (int k = 1; k <= M; k++) om.rb[(size_t)x * (M + 1) + k] = om.biased(gm.msc[(size_t)k * K + x]);
  om.tbm = om.unbiased(logf(2.0f / ((float)M * (float)(M + 1))));
  om.tec = om.unbiased(logf(0.5f));
  om.tjb = 0;
}

inline uint8_t sat_addu8(uint8_t a, uint8_t b) { const int s = (int)a + (int)b; return s > 255 ? 255 : (uint8_t)s; }
inline uint8_t sat_subu8(uint8_t a, uint8_t b) { return a > b ? (uint8_t)(a - b) : 0; }

// score in nats; overflow returns the filter's ceiling like p7_MSVMu does
inline float calib_msv(const CalibMSV &om, const uint8_t *dsq, int L, std::vector<uint8_t> &dp) {
  const int M = om.M;
  dp.assign((size_t)M + 1, 0);
  const uint8_t tjbm = (uint8_t)((int8_t)om.tjb + (int8_t)om.tbm);
  uint8_t xJ = 0, xB = sat_subu8(om.base, tjbm);
  for (int i = 1; i <= L; i++) {
    const uint8_t *rsc = &om.rb[(size_t)dsq[i] * (M + 1)];
    uint8_t xE = 0, prev = 0;          // prev = M(i-1, k-1); node 0 is -infinity (0)
    for (int k = 1; k <= M; k++) {
      uint8_t sv = std::max(prev, xB);
      sv = sat_addu8(sv, om.bias);
      sv = sat_subu8(sv, rsc[k]);
      xE = std::max(xE, sv);
      prev = dp[(size_t)k];
      dp[(size_t)k] = sv;
    }
    if (sat_addu8(xE, om.bias) == 255) return (float)(255 - om.base) / om.scale;
    xE = sat_subu8(xE, om.tec);
    xJ = std::max(xJ, xE);
    xB = std::max(om.base, xJ);
    xB = sat_subu8(xB, tjbm);
  }
  float sc = (float)((int)xJ - (int)om.tjb) - (float)om.base;
  sc /= om.scale;
  sc -= 3.0f;
  return sc;
}

// ---- Viterbi filter (16-bit) -------------------------------------------------------------------------------
struct CalibVit {
  int M, K;
  float scale;
  int16_t base;
  std::vector<int16_t> rw;    // [K][M+1]
  std::vector<int16_t> tw;    // [M+1][8]: BM MM IM DM (into node k, from k-1), MD MI II DD (out of node k)
  int16_t xE_loop, xE_move, xNCJ_move;
  int16_t wordify(float sc) const {
    sc = roundf(scale * sc);
    if (sc >= 32767.0f) return 32767;
    if (sc <= -32768.0f) return -32768;
    return (int16_t)sc;
  }
};
enum { vBM = 0, vMM, vIM, vDM, vMD, vMI, vII, vDD };

inline void calib_vit_convert(const CalibProfile &gm, CalibVit &om) {
  const int M = gm.M, K = gm.K;
  om.M = M; om.K = K;
  om.scale = (float)(500.0 / 0.69314718055994529);
  om.base = 12000;
  om.rw.assign((size_t)K * (M + 1), -32768);
  for (int x = 0; x < K; x++) for (int k = 1; k <= M; k++) om.rw[(size_t)x * (M + 1) + k] = om.wordify(gm.msc[(size_t)k * K + x]);
  om.tw.assign((size_t)(M + 1) * 8, -32768);
  auto cap = [](int16_t v, int16_t mx) { return v <= mx ? v : mx; };
  for (int k = 1; k <= M; k++) {
    int16_t *tp = &om.tw[(size_t)k * 8];
    const int kb = k - 1;      // the incoming transitions live at node k-1 of the profile
    tp[vBM] = cap(om.wordify(gm.bm[(size_t)kb]), 0);
    tp[vMM] = cap(om.wordify(gm.tsc[(size_t)kb * 7 + tMM]), 0);
    tp[vIM] = cap(om.wordify(gm.tsc[(size_t)kb * 7 + tIM]), 0);
    tp[vDM] = cap(om.wordify(gm.tsc[(size_t)kb * 7 + tDM]), 0);
    if (k < M) {
      tp[vMD] = cap(om.wordify(gm.tsc[(size_t)k * 7 + tMD]), 0);
      tp[vMI] = cap(om.wordify(gm.tsc[(size_t)k * 7 + tMI]), 0);
      tp[vII] = cap(om.wordify(gm.tsc[(size_t)k * 7 + tII]), -1);
      tp[vDD] = om.wordify(gm.tsc[(size_t)k * 7 + tDD]);
    }
  }
  om.xE_loop = om.wordify(-0.69314718055994529f);
  om.xE_move = om.wordify(-0.69314718055994529f);
  om.xNCJ_move = 0;
}

inline int16_t sat_add16(int16_t a, int16_t b) { const int s = (int)a + (int)b; return s > 32767 ? 32767 : s < -32768 ? -32768 : (int16_t)s; }

inline float calib_viterbi(const CalibVit &om, const uint8_t *dsq, int L, std::vector<int16_t> &mx) {
  const int M = om.M;
  mx.assign((size_t)3 * (M + 1), -32768);
  int16_t *Mx = mx.data(), *Ix = Mx + (M + 1), *Dx = Ix + (M + 1);
  int16_t xN = om.base, xB = (int16_t)((int)xN + (int)om.xNCJ_move), xJ = -32768, xC = -32768, xE;
  for (int i = 1; i <= L; i++) {
    const int16_t *rsc = &om.rw[(size_t)dsq[i] * (M + 1)];
    int16_t pm = -32768, pi_ = -32768, pd = -32768;     // row i-1 at node k-1
    int16_t dcv = -32768;                               // D(i,k): M(i,k-1) + MD, closed over DD below
    xE = -32768;
    for (int k = 1; k <= M; k++) {
      const int16_t *tp = &om.tw[(size_t)k * 8];
      int16_t sv = sat_add16(xB, tp[vBM]);
      sv = std::max(sv, sat_add16(pm, tp[vMM]));
      sv = std::max(sv, sat_add16(pi_, tp[vIM]));
      sv = std::max(sv, sat_add16(pd, tp[vDM]));
      sv = sat_add16(sv, rsc[k]);
      xE = std::max(xE, sv);
      pm = Mx[k]; pi_ = Ix[k]; pd = Dx[k];
      Mx[k] = sv;
      // D(i,k) = max(M(i,k-1) + MD(k-1), D(i,k-1) + DD(k-1)): the lazy-F passes of the SSE code reach this closure
      Dx[k] = dcv;
      const int16_t fromD = sat_add16(Dx[k], tp[vDD]);
      dcv = std::max(sat_add16(sv, tp[vMD]), fromD);
      Ix[k] = std::max(sat_add16(pm, tp[vMI]), sat_add16(pi_, tp[vII]));
    }
    if (xE >= 32767) return (32767.0f - (float)om.base) / om.scale;
    // NN = CC = JJ = 0 (the -3 nat approximation)
    xC = (int16_t)std::max((int)xC, (int)xE + (int)om.xE_move);
    xJ = (int16_t)std::max((int)xJ, (int)xE + (int)om.xE_loop);
    xB = (int16_t)std::max((int)xJ + (int)om.xNCJ_move, (int)xN + (int)om.xNCJ_move);
  }
  if (xC > -32768) {
    float sc = (float)xC + (float)om.xNCJ_move - (float)om.base;
    sc /= om.scale;
    sc -= 3.0f;
    return sc;
  }
  return -INFINITY;
}

// ---- Forward, local multihit, length model L (float64, scaled rows) ------------------------------------------
inline double calib_forward(const CalibProfile &gm, const uint8_t *dsq, int L, std::vector<double> &w) {
  const int M = gm.M, K = gm.K;
  // probabilities from the float scores, as the optimized profile holds them
  w.assign((size_t)8 * (M + 1) + (size_t)3 * (M + 1) * 2, 0.0);
  double *tMMp = w.data(), *tIMp = tMMp + (M + 1), *tDMp = tIMp + (M + 1), *tBMp = tDMp + (M + 1);
  double *tMDp = tBMp + (M + 1), *tMIp = tMDp + (M + 1), *tIIp = tMIp + (M + 1), *tDDp = tIIp + (M + 1);
  double *row0 = tDDp + (M + 1), *row1 = row0 + 3 * (M + 1);
  for (int k = 1; k <= M; k++) {
    const int kb = k - 1;
    tBMp[k] = std::exp((double)gm.bm[(size_t)kb]);
    tMMp[k] = std::exp((double)gm.tsc[(size_t)kb * 7 + tMM]);
    tIMp[k] = std::exp((double)gm.tsc[(size_t)kb * 7 + tIM]);
    tDMp[k] = std::exp((double)gm.tsc[(size_t)kb * 7 + tDM]);
    if (k < M) {
      tMDp[k] = std::exp((double)gm.tsc[(size_t)k * 7 + tMD]);
      tMIp[k] = std::exp((double)gm.tsc[(size_t)k * 7 + tMI]);
      tIIp[k] = std::exp((double)gm.tsc[(size_t)k * 7 + tII]);
      tDDp[k] = std::exp((double)gm.tsc[(size_t)k * 7 + tDD]);
    }
  }
  const float pmove_f = 3.0f / ((float)L + 3.0f), ploop_f = 1.0f - pmove_f;
  const double pmove = pmove_f, ploop = ploop_f;
  double xN = 1.0, xB = pmove, xJ = 0.0, xC = 0.0, logscale = 0.0;
  double *prev = row0, *cur = row1;
  for (int k = 0; k <= 3 * M + 2; k++) prev[k] = 0.0;
  for (int i = 1; i <= L; i++) {
    const int x = dsq[i];
    double *pM = prev, *pI = prev + (M + 1), *pD = pI + (M + 1);
    double *cM = cur, *cI = cur + (M + 1), *cD = cI + (M + 1);
    cM[0] = cI[0] = cD[0] = 0.0;
    double xE = 0.0;
    for (int k = 1; k <= M; k++) {
      const double e = std::exp((double)gm.msc[(size_t)k * K + x]);
      const double m = e * (xB * tBMp[k] + pM[k - 1] * tMMp[k] + pI[k - 1] * tIMp[k] + pD[k - 1] * tDMp[k]);
      cM[k] = m;
      cI[k] = k < M ? pM[k] * tMIp[k] + pI[k] * tIIp[k] : 0.0;
      cD[k] = k > 1 ? cM[k - 1] * tMDp[k - 1] + cD[k - 1] * tDDp[k - 1] : 0.0;
      xE += m + cD[k];
    }
    xJ = xJ * ploop + xE * 0.5;
    xC = xC * ploop + xE * 0.5;
    xN = xN * ploop;
    xB = (xN + xJ) * pmove;
    if (xE > 1e100 || (xE > 0.0 && xE < 1e-100) || xN < 1e-250) {
      const double s = 1.0 / std::max(std::max(xE, xN), std::max(xJ, xC));
      for (int k = 0; k <= 3 * M + 2; k++) cur[k] *= s;
      xN *= s; xB *= s; xJ *= s; xC *= s;
      logscale -= std::log(s);
    }
    std::swap(prev, cur);
  }
  return std::log(xC * pmove) + logscale;
}

// ---- Gumbel fits (Easel) -----------------------------------------------------------------------------------
inline double calib_fit_loc(const std::vector<double> &x, double lambda) {
  double esum = 0.0;
  for (double v : x) esum += std::exp(-lambda * v);
  return -std::log(esum / (double)x.size()) / lambda;
}

inline void calib_fit_complete(const std::vector<double> &x, double &mu, double &lambda) {
  const int n = (int)x.size();
  double sum = 0.0, sqsum = 0.0;
  for (double v : x) { sum += v; sqsum += v * v; }
  const double variance = (sqsum - sum * sum / (double)n) / ((double)n - 1.0);
  lambda = 3.14159265358979323846264338328 / std::sqrt(6. * variance);
  auto lawless416 = [&](double lam, double &f, double &df) {
    double esum = 0., xesum = 0., xxesum = 0., xsum = 0.;
    for (double v : x) {
      xsum += v;
      xesum += v * std::exp(-1. * lam * v);
      xxesum += v * v * std::exp(-1. * lam * v);
      esum += std::exp(-1. * lam * v);
    }
    f = (1. / lam) - (xsum / n) + (xesum / esum);
    df = ((xesum / esum) * (xesum / esum)) - (xxesum / esum) - (1. / (lam * lam));
  };
  double fx = 0., dfx = 0.;
  int i;
  for (i = 0; i < 100; i++) {
    lawless416(lambda, fx, dfx);
    if (std::fabs(fx) < 1e-5) break;
    lambda = lambda - fx / dfx;
    if (lambda <= 0.) lambda = 0.001;
  }
  if (i == 100) {      // Newton/Raphson failed: bisection, as Easel does
    double left = 0., right = 3.14159265358979323846264338328 / std::sqrt(6. * variance);
    lawless416(lambda, fx, dfx);
    while (fx > 0.) { right *= 2.; if (right > 100.) break; lawless416(right, fx, dfx); }
    for (i = 0; i < 100; i++) {
      const double mid = (left + right) / 2.;
      lawless416(mid, fx, dfx);
      if (std::fabs(fx) < 1e-5) { lambda = mid; break; }
      if (fx > 0.) left = mid; else right = mid;
      lambda = mid;
    }
  }
  double esum = 0.;
  for (double v : x) esum += std::exp(-lambda * v);
  mu = -std::log(esum / n) / lambda;
}

// lambda, MSV mu, Viterbi mu, Forward tau
inline void calibrate_model(const CalibModel &h, double meanrelent_bits, double out[4]) {
  const double LOG2 = 0.69314718055994529;
  const int EmL = 200, EmN = 200, EvL = 200, EvN = 200, EfL = 100, EfN = 200;
  const double Eft = 0.04;
  CalibProfile gm;
  calib_profile(h, gm);
  const double lambda = LOG2 + 1.44 / ((double)h.M * meanrelent_bits);
  CalibRng rng(42u);
  std::vector<uint8_t> dsq;
  auto draw = [&](int L) {
    dsq.assign((size_t)L + 2, 0);
    for (int i = 1; i <= L; i++) dsq[(size_t)i] = (uint8_t)rng.choose(h.bg, h.K);
  };
  std::vector<double> xv;
  // MSV
  {
    CalibMSV om;
    calib_msv_convert(gm, om);
    om.tjb = om.unbiased(logf(3.0f / (float)(EmL + 3)));
    const float nullsc = calib_nullone(EmL);
    std::vector<uint8_t> dp;
    xv.clear();
    for (int i = 0; i < EmN; i++) {
      draw(EmL);
      const float sc = calib_msv(om, dsq.data(), EmL, dp);
      xv.push_back((double)(sc - nullsc) / LOG2);
    }
  }
  const double mmu = calib_fit_loc(xv, lambda);
  // Viterbi
  {
    CalibVit om;
    calib_vit_convert(gm, om);
    om.xNCJ_move = om.wordify(logf(3.0f / ((float)EvL + 3.0f)));
    const float nullsc = calib_nullone(EvL);
    std::vector<int16_t> mx;
    xv.clear();
    for (int i = 0; i < EvN; i++) {
      draw(EvL);
      const float sc = calib_viterbi(om, dsq.data(), EvL, mx);
      xv.push_back((double)(sc - nullsc) / LOG2);
    }
  }
  const double vmu = calib_fit_loc(xv, lambda);
  // Forward
  {
    const float nullsc = calib_nullone(EfL);
    std::vector<double> w;
    xv.clear();
    for (int i = 0; i < EfN; i++) {
      draw(EfL);
      const float fsc = (float)calib_forward(gm, dsq.data(), EfL, w);
      xv.push_back((double)(fsc - nullsc) / LOG2);
    }
  }
  double gmu, glam;
  calib_fit_complete(xv, gmu, glam);
  const double tau = (gmu - std::log(-1. * std::log(1.0 - Eft)) / glam) + (std::log(Eft) / lambda);
  out[0] = lambda; out[1] = mmu; out[2] = vmu; out[3] = tau;
}
