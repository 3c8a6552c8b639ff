// Models of ANY size: scoring front end and alignment for the models the register-resident kernels cannot take
// (more than 3072 nodes: `--symfrac 0.0` makes every populated backbone column a node, so the root subsets of
// large backbones - 16S alignments of 5-12 k columns - get there; witch_msa/gcmm/algorithm.py:463-470).
//
// ONE wavefront per (query, model) pair, float64, lane r owns nodes r*Q+1 .. r*Q+Q for any Q, the DP rows in a
// per-wave HBM slab (wh_f64.h).  Slow beside the register kernels (every row goes through memory) and meant
// as the tail of an ensemble, not its body:
//  * generic_front_kernel: what the scoring kernels do up to the envelopes - multihit
//    Forward and Backward keeping only the special states, domain decoding, the region scan and the multidomain
//    test; single-domain regions are rescored here (unihit Forward with the rows kept, Backward with the
//    posterior expectations accumulated per node, null2 by expectation).  EVERY pair with a region is then
//    queued for resolve_kernel (wh_resolve.hip), which resolves the multidomain regions - it is generic in the
//    model size already - and assembles the score of the pair (A.6) from the queue record.
//  * generic_align_kernel: hmmalign's unihit Forward / Backward / posterior decoding / optimal-accuracy fill and
//    traceback (A.7; witch_msa/gcmm/aligner.py:96-142) in place in one slab: Forward rows -> posteriors ->
//    OA rows.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "wh_launch.h"
#include "wh_f64.h"

namespace wh {

namespace {

__device__ __forceinline__ double shfl_down_d(double v, int d) {
  const long long u = __double_as_longlong(v);
  const int lo = __shfl_down((int)(u & 0xFFFFFFFFll), d), hi = __shfl_down((int)(u >> 32), d);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_max_d(double x) {
  for (int m = 32; m >= 1; m >>= 1) {
    const long long u = __double_as_longlong(x);
    const int lo = __shfl_xor((int)(u & 0xFFFFFFFFll), m), hi = __shfl_xor((int)(u >> 32), m);
    const double o = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    x = o > x ? o : x;
  }
  return x;
}
__device__ __forceinline__ float wave_max_f(float x) { for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(x, m); x = o > x ? o : x; } return x; }

// value of array <arr> at node (lane, q+1): the lane's next node, or the first node of lane+1; 0 beyond the table
__device__ __forceinline__ double next_node(const double *base, int Q, int q, int lane) {
  if (q + 1 < Q) return __builtin_nontemporal_load(base + ofs2(q + 1, lane));
  if (lane < 63) return __builtin_nontemporal_load(base + ofs2(0, lane + 1));
  return 0.0;
}

// four nodes q0..q0+3 of a lane (two adjacent pairs: 16-byte accesses), and those plus the node that follows them
template <bool NT>
__device__ __forceinline__ void load4(const double *base, int q0, int lane, double *v) {
  if (!base) { v[0] = v[1] = v[2] = v[3] = 0.0; return; }
  const d2_t a = NT ? nt_load_d2(base + ofs2(q0, lane)) : ld_d2(base + ofs2(q0, lane));
  const d2_t b = NT ? nt_load_d2(base + ofs2(q0 + 2, lane)) : ld_d2(base + ofs2(q0 + 2, lane));
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
template <bool NT>
__device__ __forceinline__ void load5(const double *base, int Q, int q0, int lane, double *v) {
  if (!base) { v[0] = v[1] = v[2] = v[3] = v[4] = 0.0; return; }
  load4<NT>(base, q0, lane, v);
  v[4] = next_node(base, Q, q0 + 3, lane);
}
__device__ __forceinline__ void store4(double *base, int q0, int lane, const double *v) {
  st_d2(base + ofs2(q0, lane), v[0], v[1]);
  st_d2(base + ofs2(q0 + 2, lane), v[2], v[3]);
}

// Backward sweep (A.2), float64, over seq[0..L-1] with length model c.  Two alternating rows (brow[0], brow[1]:
// 3 x Q x 64 doubles each; row i in brow[i & 1]).
//   MODE 0: the special states of every row go to xs ((L+1) x xNSPEC doubles: N B E J C, xCLS = cumulated scale).
//   MODE 1: posterior expectations against the stored Forward rows <fmx> (rows 0..L, specials in the row tails):
//           accM/accI[node] += F x B x scale per row (arrays of Q x 64 doubles, zeroed by the caller); returns the
//           unnormalised flank factor sum_i P(N, J, C emit residue i).
//   MODE 2: posteriors in place: row i of <fmx> gets (float)ppM / (float)ppI in its state slots 0 / 1, xs gets
//           ppN, ppJ, ppC per row in slots 0..2.
template <int MODE>
__device__ double gbackward(const GModel &m, const uint8_t *seq, int L, GLen c, const GMx &fmx, double *brow0, double *brow1,
                            double *xs, double fwdsc, double *accM, double *accI, int lane, double *bwd_out = nullptr) {
  const int Q = m.Q, M = m.M;
  const size_t SQ = (size_t)Q * 64;
  const double *tf = m.tf;
  double ls = 0.0, xfactor = 0.0;
  // row L+1 does not exist: specials of "row L" start the recursion
  double nN = 0.0, nJ = 0.0, nC = c.move;          // specials of the row below the one being formed
  double xbsum_prev = 0.0;                         // sum_k M(i+1,k) od_{x_{i+1}}[k] entry[k], formed while row i+1 was written
  for (int i = L; i >= 0; i--) {
    double xBv, xJv, xCv, xNv, xEv;
    if (i == L) { xCv = c.move; xJv = 0.0; xNv = 0.0; xBv = 0.0; }
    else {
      xBv = wave_sum_d(xbsum_prev);
      xJv = nJ * c.loop + xBv * c.move;
      xCv = nC * c.loop;
      xNv = nN * c.loop + xBv * c.move;
    }
    xEv = xCv * c.EC + xJv * c.EJ;
    // the rescale of this row depends on its special states only: it is applied while the cells are formed
    double rs = 1.0;
    if (xBv > kRescaleHi || xNv > kRescaleHi) {
      const double big = xBv > xNv ? xBv : xNv;
      rs = 1.0 / big;
      xBv *= rs; xJv *= rs; xCv *= rs; xNv *= rs; xEv *= rs; ls += log(big);
    }
    if (i >= 1) {
      wave_mem_sync();     // row i+1 was written by other lanes of this wave
      const bool have_next = i < L;
      const double *nr = (i + 1) & 1 ? brow1 : brow0;
      double *cr = i & 1 ? brow1 : brow0;
      const double *odn = have_next ? m.te + (size_t)seq[i] * SQ : nullptr;     // residue x_{i+1}
      const double *odc = m.te + (size_t)seq[i - 1] * SQ;                       // residue x_i
      // Four nodes per step (Q is a multiple of 4), every load of the step requested before the first is used - a
      // step then costs one memory round trip instead of one per node.  "next" arrays (values at node k+1) are read
      // as the step's four nodes plus the one that follows them (the next step's first, or lane+1's first node).
      // pass 1, nodes of the lane from the last to the first: I, the M terms that do not need D, the D chain with
      // nothing entering from the right
      double dloc = 0.0, P = 1.0;
      for (int q0 = Q - 4; q0 >= 0; q0 -= 4) {
        double mn[5], on[5], tAn[5], tBn[5], tCn[5], tDn[5], inn[4], tII4[4], tMI4[4];
        load5<true>(have_next ? nr : nullptr, Q, q0, lane, mn);
        load5<true>(odn, Q, q0, lane, on);
        load5<true>(tf + gA * SQ, Q, q0, lane, tAn); load5<true>(tf + gB * SQ, Q, q0, lane, tBn);
        load5<true>(tf + gC * SQ, Q, q0, lane, tCn); load5<true>(tf + gD2 * SQ, Q, q0, lane, tDn);
        load4<true>(have_next ? nr + SQ : nullptr, q0, lane, inn);
        load4<false>(tf + gII * SQ, q0, lane, tII4); load4<false>(tf + gMI * SQ, q0, lane, tMI4);
        double oM4[4], oI4[4], oD4[4];
#pragma unroll
        for (int u = 3; u >= 0; u--) {
          const int k = lane * Q + q0 + u + 1;
          const double mnext = mn[u + 1] * on[u + 1] * rs, in = inn[u] * rs;
          double iv = mnext * tBn[u + 1] + in * tII4[u];
          double mpart = mnext * tAn[u + 1] + in * tMI4[u] + xEv;
          double av = mnext * tCn[u + 1] + xEv;
          if (k >= M) { iv = 0.0; if (k == M) { mpart = xEv; av = xEv; } else { mpart = 0.0; av = 0.0; } }
          dloc = k > M ? 0.0 : av + tDn[u + 1] * dloc;
          P = k > M ? 0.0 : P * tDn[u + 1];
          oM4[u] = mpart; oI4[u] = iv; oD4[u] = dloc;
        }
        store4(cr, q0, lane, oM4); store4(cr + SQ, q0, lane, oI4); store4(cr + 2 * SQ, q0, lane, oD4);
      }
      // cross-lane: D(first node of lane r) = dloc + P * D(first node of lane r+1), from the last lane down
      double Bv = dloc, Av = P;
      for (int d = 1; d < 64; d <<= 1) {
        const double Bo = shfl_down_d(Bv, d), Ao = shfl_down_d(Av, d);
        if (lane + d < 64) { Bv = Bv + Av * Bo; Av = Av * Ao; }
      }
      const double dn = shfl_down_d(Bv, 1);
      const double din = lane < 63 ? dn : 0.0;          // true D of the first node of lane+1
      // pass 2: true D, the M term through D(k+1), and what the row above needs from this one
      double Pq = 1.0, dnext = din, xbsum = 0.0;
      const double *fr = MODE != 0 ? fmx.row(i) : nullptr;
      double sc = 0.0;
      if (MODE != 0) sc = exp(__builtin_nontemporal_load(fr + 3 * SQ + xCLS) + ls - fwdsc);
      for (int q0 = Q - 4; q0 >= 0; q0 -= 4) {
        double tDn[5], tMDn[5], dl[4], mp[4], oc[4], te4[4], iv4[4], fM4[4], fI4[4], aM4[4], aI4[4];
        load5<true>(tf + gD2 * SQ, Q, q0, lane, tDn); load5<true>(tf + gD1 * SQ, Q, q0, lane, tMDn);
        load4<true>(cr + 2 * SQ, q0, lane, dl); load4<true>(cr, q0, lane, mp);
        load4<false>(odc, q0, lane, oc); load4<false>(tf + gE * SQ, q0, lane, te4);
        if (MODE != 0) { load4<true>(cr + SQ, q0, lane, iv4); load4<true>(fr, q0, lane, fM4); load4<true>(fr + SQ, q0, lane, fI4); }
        if (MODE == 1) { load4<true>(accM, q0, lane, aM4); load4<true>(accI, q0, lane, aI4); }
        double oM4[4], oD4[4];
#pragma unroll
        for (int u = 3; u >= 0; u--) {
          const int k = lane * Q + q0 + u + 1;
          Pq = k > M ? 0.0 : Pq * tDn[u + 1];
          double dv = dl[u] + Pq * din;
          double mv = mp[u] + (k < M ? dnext * tMDn[u + 1] : 0.0);
          if (k > M) { dv = 0.0; mv = 0.0; }
          oM4[u] = mv; oD4[u] = dv;
          dnext = dv;
          xbsum += mv * oc[u] * te4[u];
          if (MODE == 1) { aM4[u] = aM4[u] + fM4[u] * mv * sc; aI4[u] = aI4[u] + fI4[u] * iv4[u] * sc; }
          if (MODE == 2) { fM4[u] = (double)(float)(fM4[u] * mv * sc); fI4[u] = (double)(float)(fI4[u] * iv4[u] * sc); }
        }
        store4(cr, q0, lane, oM4); store4(cr + 2 * SQ, q0, lane, oD4);
        if (MODE == 1) { store4(accM, q0, lane, aM4); store4(accI, q0, lane, aI4); }
        if (MODE == 2) { double *fw = const_cast<double *>(fr); store4(fw, q0, lane, fM4); store4(fw + SQ, q0, lane, fI4); }
      }
      xbsum_prev = xbsum;
      if (MODE != 0) {
        // flank posteriors of residue i: F(i-1) x loop x B(i)
        const double *fp = fmx.row(i - 1) + 3 * SQ;
        const double sc2 = exp(__builtin_nontemporal_load(fp + xCLS) + ls - fwdsc);
        const double pn = __builtin_nontemporal_load(fp + xN) * xNv * c.loop * sc2;
        const double pj = __builtin_nontemporal_load(fp + xJ) * xJv * c.loop * sc2;
        const double pc = __builtin_nontemporal_load(fp + xC) * xCv * c.loop * sc2;
        if (MODE == 1) { xfactor += pn; xfactor += pj; xfactor += pc; }
        if (MODE == 2 && lane == 0) { double *t = xs + (size_t)i * xNSPEC; t[0] = (double)(float)pn; t[1] = (double)(float)pj; t[2] = (double)(float)pc; }
      }
    }
    if (MODE == 0 && lane == 0) { double *t = xs + (size_t)i * xNSPEC; t[xN] = xNv; t[xB] = xBv; t[xE] = xEv; t[xJ] = xJv; t[xC] = xCv; t[xCLS] = ls; }
    nN = xNv; nJ = xJv; nC = xCv;
    if (i == 0 && bwd_out) *bwd_out = ls + log(xNv);
  }
  wave_mem_sync();
  return xfactor;
}

// ------------------------------------------------------------------ log-space twins for the alignment's rare pass
// Two hits of thousands of bits each inside ONE unihit alignment leave the range of a scaled double (the flank
// state that carries the weaker hit underflows against its row's scale); hmmalign switches to its log-space
// "generic" code there.  Same layout, every value a natural-log probability, -inf
// for zero, no scaling; logs of the table entries are taken on the fly.  Speed does not matter here.
__device__ __forceinline__ double lse2(double a, double b) {
  const double mx = a > b ? a : b, mn = a > b ? b : a;
  if (mx == -INFINITY) return mx;
  return mx + log1p(exp(mn - mx));
}
__device__ __forceinline__ double llog(double p) { return p > 0.0 ? log(p) : -INFINITY; }
__device__ __forceinline__ double wave_lse(double x) {
  const double mx = wave_max_d(x);
  if (mx == -INFINITY) return mx;
  return mx + log(wave_sum_d(exp(x - mx)));
}

// Forward, rows 0..L kept (log M, I, D; log N B E J C in the row tails).  Returns log Z.
__device__ double gforward_log(const GModel &m, const uint8_t *seq, int L, GLen c, const GMx &mx, int lane) {
  const int Q = m.Q;
  const size_t SQ = (size_t)Q * 64;
  const double *tf = m.tf;
  const double lloop = llog(c.loop), lmove = llog(c.move), lEJ = llog(c.EJ), lEC = llog(c.EC);
  {
    double *r0 = mx.row(0);
    for (int q = 0; q < Q; q++) { const size_t o = ofs2(q, lane); r0[o] = -INFINITY; r0[SQ + o] = -INFINITY; r0[2 * SQ + o] = -INFINITY; }
    if (lane == 0) { double *s = r0 + 3 * SQ; s[xN] = 0.0; s[xB] = lmove; s[xE] = -INFINITY; s[xJ] = -INFINITY; s[xC] = -INFINITY; }
  }
  double pN = 0.0, pB = lmove, pJ = -INFINITY, pC = -INFINITY;
  double Psum = 0.0;                      // sum of log D->D over the lane's nodes 2..Q
  for (int q = 1; q < Q; q++) Psum += llog(tf[gD2 * SQ + ofs2(q, lane)]);
  const double d10 = llog(tf[gD1 * SQ + ofs2(0, lane)]), d20 = llog(tf[gD2 * SQ + ofs2(0, lane)]);
  for (int i = 1; i <= L; i++) {
    wave_mem_sync();
    const double *pr = mx.row(i - 1);
    double *cr = mx.row(i);
    const double *od = m.te + (size_t)seq[i - 1] * SQ;
    double pm1 = -INFINITY, pi1 = -INFINITY, pd1 = -INFINITY;
    if (lane > 0) { const size_t ol = ofs2(Q - 1, lane - 1); pm1 = __builtin_nontemporal_load(pr + ol); pi1 = __builtin_nontemporal_load(pr + SQ + ol); pd1 = __builtin_nontemporal_load(pr + 2 * SQ + ol); }
    double mprev = -INFINITY, dloc = -INFINITY, xe = -INFINITY;
    for (int q = 0; q < Q; q++) {
      const size_t o = ofs2(q, lane);
      const double oM = __builtin_nontemporal_load(pr + o), oI = __builtin_nontemporal_load(pr + SQ + o), oD = __builtin_nontemporal_load(pr + 2 * SQ + o);
      const double mm = llog(od[o]) + lse2(lse2(pm1 + llog(tf[gA * SQ + o]), pi1 + llog(tf[gB * SQ + o])),
                                           lse2(pd1 + llog(tf[gC * SQ + o]), pB + llog(tf[gE * SQ + o])));
      const double ins = lse2(oM + llog(tf[gMI * SQ + o]), oI + llog(tf[gII * SQ + o]));
      dloc = q > 0 ? lse2(mprev + llog(tf[gD1 * SQ + o]), dloc + llog(tf[gD2 * SQ + o])) : -INFINITY;
      cr[o] = mm; cr[SQ + o] = ins; cr[2 * SQ + o] = dloc;
      xe = lse2(xe, mm);
      pm1 = oM; pi1 = oI; pd1 = oD; mprev = mm;
    }
    const double mup = shfl_up_d(mprev, 1);
    const double mleft = lane > 0 ? mup : -INFINITY;
    double Bv = lse2(dloc, Psum + d10 + mleft), Av = Psum + d20;
    for (int d = 1; d < 64; d <<= 1) {
      const double Bo = shfl_up_d(Bv, d), Ao = shfl_up_d(Av, d);
      if (lane >= d) { Bv = lse2(Bv, Av + Bo); Av = Av + Ao; }
    }
    const double dup = shfl_up_d(Bv, 1);
    const double dleft = lane > 0 ? dup : -INFINITY;
    const double c0 = lane > 0 ? lse2(d10 + mleft, d20 + dleft) : -INFINITY;
    double Pq = 0.0;
    for (int q = 0; q < Q; q++) {
      const size_t o = ofs2(q, lane);
      if (q > 0) Pq += llog(tf[gD2 * SQ + o]);
      const double dv = lse2(__builtin_nontemporal_load(cr + 2 * SQ + o), Pq + c0);
      cr[2 * SQ + o] = dv;
      xe = lse2(xe, dv);
    }
    xe = wave_lse(xe);
    const double xn = pN + lloop, xc = lse2(pC + lloop, xe + lEC), xj = lse2(pJ + lloop, xe + lEJ);
    const double xb = lse2(xj + lmove, xn + lmove);
    if (lane == 0) { double *s = cr + 3 * SQ; s[xN] = xn; s[xB] = xb; s[xE] = xe; s[xJ] = xj; s[xC] = xc; }
    pN = xn; pB = xb; pJ = xj; pC = xc;
  }
  wave_mem_sync();
  return pC + lmove;
}

// Backward against the kept log Forward rows; posteriors in place like gbackward<2>.  Returns log Z of Backward.
__device__ double gbackward_log(const GModel &m, const uint8_t *seq, int L, GLen c, const GMx &fmx, double *brow0, double *brow1,
                                double *xs, double fwd, int lane) {
  const int Q = m.Q, M = m.M;
  const size_t SQ = (size_t)Q * 64;
  const double *tf = m.tf;
  const double lloop = llog(c.loop), lmove = llog(c.move), lEJ = llog(c.EJ), lEC = llog(c.EC);
  double nN = -INFINITY, nJ = -INFINITY, nC = lmove, xb_prev = -INFINITY, bwd = -INFINITY;
  for (int i = L; i >= 0; i--) {
    double xBv, xJv, xCv, xNv;
    if (i == L) { xCv = lmove; xJv = -INFINITY; xNv = -INFINITY; xBv = -INFINITY; }
    else {
      xBv = wave_lse(xb_prev);
      xJv = lse2(nJ + lloop, xBv + lmove);
      xCv = nC + lloop;
      xNv = lse2(nN + lloop, xBv + lmove);
    }
    const double xEv = lse2(xCv + lEC, xJv + lEJ);
    if (i >= 1) {
      wave_mem_sync();
      const bool have_next = i < L;
      const double *nr = (i + 1) & 1 ? brow1 : brow0;
      double *cr = i & 1 ? brow1 : brow0;
      const double *odn = have_next ? m.te + (size_t)seq[i] * SQ : nullptr;
      const double *odc = m.te + (size_t)seq[i - 1] * SQ;
      double dloc = -INFINITY, P = 0.0;
      for (int q = Q - 1; q >= 0; q--) {
        const int k = lane * Q + q + 1;
        const size_t o = ofs2(q, lane);
        double mnext = -INFINITY, in = -INFINITY;
        if (have_next) { mnext = next_node(nr, Q, q, lane) + llog(next_node(odn, Q, q, lane)); in = __builtin_nontemporal_load(nr + SQ + o); }
        const double lMM = llog(next_node(tf + gA * SQ, Q, q, lane)), lIM = llog(next_node(tf + gB * SQ, Q, q, lane));
        const double lDM = llog(next_node(tf + gC * SQ, Q, q, lane)), lDD = llog(next_node(tf + gD2 * SQ, Q, q, lane));
        double iv = lse2(mnext + lIM, in + llog(tf[gII * SQ + o]));
        double mpart = lse2(lse2(mnext + lMM, in + llog(tf[gMI * SQ + o])), xEv);
        double av = lse2(mnext + lDM, xEv);
        if (k >= M) { iv = -INFINITY; if (k == M) { mpart = xEv; av = xEv; } else { mpart = -INFINITY; av = -INFINITY; } }
        dloc = k > M ? -INFINITY : lse2(av, lDD + dloc);
        P = k > M ? -INFINITY : P + lDD;
        cr[o] = mpart; cr[SQ + o] = iv; cr[2 * SQ + o] = dloc;
      }
      double Bv = dloc, Av = P;
      for (int d = 1; d < 64; d <<= 1) {
        const double Bo = shfl_down_d(Bv, d), Ao = shfl_down_d(Av, d);
        if (lane + d < 64) { Bv = lse2(Bv, Av + Bo); Av = Av + Ao; }
      }
      const double dn = shfl_down_d(Bv, 1);
      const double din = lane < 63 ? dn : -INFINITY;
      double Pq = 0.0, dnext = din, xb = -INFINITY;
      double *fr = fmx.row(i);
      for (int q = Q - 1; q >= 0; q--) {
        const int k = lane * Q + q + 1;
        const size_t o = ofs2(q, lane);
        const double lDD = llog(next_node(tf + gD2 * SQ, Q, q, lane)), lMD = llog(next_node(tf + gD1 * SQ, Q, q, lane));
        Pq = k > M ? -INFINITY : Pq + lDD;
        double dv = lse2(__builtin_nontemporal_load(cr + 2 * SQ + o), Pq + din);
        double mv = __builtin_nontemporal_load(cr + o);
        if (k < M) mv = lse2(mv, dnext + lMD);
        if (k > M) { dv = -INFINITY; mv = -INFINITY; }
        cr[o] = mv; cr[2 * SQ + o] = dv;
        dnext = dv;
        xb = lse2(xb, mv + llog(odc[o]) + llog(tf[gE * SQ + o]));
        const double iv = __builtin_nontemporal_load(cr + SQ + o);
        const float pm = (float)exp(__builtin_nontemporal_load(fr + o) + mv - fwd), pi = (float)exp(__builtin_nontemporal_load(fr + SQ + o) + iv - fwd);
        fr[o] = (double)pm; fr[SQ + o] = (double)pi;
      }
      xb_prev = xb;
      const double *fp = fmx.row(i - 1) + 3 * SQ;
      const double pn = exp(__builtin_nontemporal_load(fp + xN) + xNv + lloop - fwd);
      const double pj = exp(__builtin_nontemporal_load(fp + xJ) + xJv + lloop - fwd);
      const double pc = exp(__builtin_nontemporal_load(fp + xC) + xCv + lloop - fwd);
      if (lane == 0) { double *t = xs + (size_t)i * xNSPEC; t[0] = (double)(float)pn; t[1] = (double)(float)pj; t[2] = (double)(float)pc; }
    }
    nN = xNv; nJ = xJv; nC = xCv;
    if (i == 0) bwd = xNv;
  }
  wave_mem_sync();
  return bwd;
}

// per-wave slab of the front kernel (doubles): rows 0..Lcap+3 | xsF | xsB | pb pe mocc btot etot | accM accI
__host__ __device__ inline size_t generic_rowlen(int Q) { return (size_t)3 * Q * 64 + xNSPEC; }

constexpr double kRt1 = 0.25, kRt2 = 0.10, kRt3 = 0.20;

}  // namespace

size_t generic_front_doubles(int Lcap, int Qmax) {
  return (size_t)(Lcap + 4) * generic_rowlen(Qmax) + 2 * (size_t)(Lcap + 2) * xNSPEC + 5 * (size_t)(Lcap + 4) + 2 * (size_t)Qmax * 64 + 8;
}
size_t generic_align_doubles(int Lcap, int Qmax) {
  return (size_t)(Lcap + 4) * generic_rowlen(Qmax) + 2 * (size_t)(Lcap + 2) * xNSPEC + 8;
}
size_t generic_lds_bytes(int Lcap) { return (size_t)(Lcap + 16) + 64 * 4 + kRextInts * WH_MAX_ENVELOPES * 4 + 64; }

__global__ __launch_bounds__(64, 3) void generic_front_kernel(GenericArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  const int lane = threadIdx.x;
  float *null2 = reinterpret_cast<float *>(lds_raw);                 // 32 floats (+ 32 spare)
  int *regs_lds = lds_raw + 64;                                      // WH_MAX_ENVELOPES regions x kRextInts (first row, last row, envsc, domcorr, multidomain)
  uint8_t *seq = reinterpret_cast<uint8_t *>(regs_lds + kRextInts * WH_MAX_ENVELOPES);
  // The region list of a pair: in LDS, WH_MAX_ENVELOPES entries - or, in the long-list pass (a.pair_list), in HBM with room
  // for every region a sequence of Lcap rows can hold.  Written by lane 0, read by every lane: ordered by hand (one
  // wavefront per workgroup, see wave_mem_sync; the LDS case goes through the same flat accesses, hence lgkmcnt).
  const bool listed = a.pair_list != nullptr;
  const int rcap = listed ? a.ext_cap : WH_MAX_ENVELOPES;
  auto list_sync = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); };
  const int64_t n_work = listed ? a.n_pairs : a.nq * (int64_t)a.n_list;
  const double LOG2 = 0.69314718055994529;
  double *slab = a.slab + (size_t)blockIdx.x * a.slab_stride;
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(a.counter, 1);
    item = __builtin_amdgcn_readfirstlane(__shfl(item, 0));
    if ((int64_t)item >= n_work) break;
    const int64_t pl = listed ? a.pair_list[item] : 0;
    const int h = listed ? (int)(pl % a.H) : a.hmm_list[item / a.nq];
    const int64_t qi = listed ? pl / a.H : item % a.nq;
    int *rl = listed ? a.rext + (size_t)item * a.rext_stride : regs_lds;
    const DevHMM hm = a.hmms[h];
    GModel m;
    m.tf = a.gtab + hm.gfw_off; m.te = a.gtab + hm.gem_off;
    m.Q = __builtin_amdgcn_readfirstlane(hm.Q); m.M = __builtin_amdgcn_readfirstlane(hm.M);
    const size_t SQ = (size_t)m.Q * 64;
    GMx mx;
    mx.Q = m.Q; mx.rowlen = generic_rowlen(m.Q); mx.p = slab;
    double *xsF = slab + (size_t)(a.Lcap + 4) * generic_rowlen(a.Qmax);
    double *xsB = xsF + (size_t)(a.Lcap + 2) * xNSPEC;
    const int Lp2 = (a.Lcap + 3) & ~1;       // (even: the accumulator arrays behind are moved in 16-byte pairs)
    double *pbv = xsB + (size_t)(a.Lcap + 2) * xNSPEC, *pev = pbv + Lp2, *moccv = pev + Lp2;
    double *btotv = moccv + Lp2, *etotv = btotv + Lp2;
    double *accM = etotv + Lp2, *accI = accM + (size_t)a.Qmax * 64;
    const int64_t off = a.offsets[qi];
    const int L = (int)(a.offsets[qi + 1] - off);
    const size_t out = (size_t)qi * a.H + h;
    int flags = 0;
    float fwd_bits_out = -INFINITY;
    wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + out : nullptr;
    if (dp) { if (!listed) dp->fwd_bits = -INFINITY; dp->seq_score = 0.f; dp->pre_score = 0.f; dp->seqbias_nats = 0.f; dp->nregions = 0; dp->nenv = 0; }
    bool queued = false;
    float fwdsc_rec = -INFINITY;
    int nreg_rec = 0, nenv_rec = 0;
    if (L > 0 && L <= a.Lcap) {
      for (int t = lane; t < L; t += 64) { const int r = a.residues[off + t]; seq[t] = (uint8_t)(r < a.Kp ? r : a.Kp - 1); }
      __builtin_amdgcn_wave_barrier();
      // ---------------- A.2 multihit Forward and Backward of the whole sequence, special states only
      const GLen cm = glen_config(L, true), cu = glen_config(L, false);
      const double fwd = gforward<false>(m, seq, L, cm, mx, lane, xsF);
      const float fwdsc = (float)fwd;
      const float p1 = (float)L / (float)(L + 1);
      const float nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
      fwd_bits_out = (float)((fwd - (double)nullsc) / LOG2);
      if (dp && !listed) dp->fwd_bits = fwd_bits_out;     // (long-list pass: the pair keeps the Forward log-odds of the kernel that scored it first)
      fwdsc_rec = fwdsc;
      if (isfinite(fwd)) {
        (void)gbackward<0>(m, seq, L, cm, mx, mx.row(0), mx.row(1), xsB, fwd, nullptr, nullptr, lane);
        // ---------------- A.4 domain decoding (lanes over rows), then the serial region scan 64 rows per fetch
        for (int i = 1 + lane; i <= L; i += 64) {
          const double *f0 = xsF + (size_t)(i - 1) * xNSPEC, *f1 = xsF + (size_t)i * xNSPEC;
          const double *b0 = xsB + (size_t)(i - 1) * xNSPEC, *b1 = xsB + (size_t)i * xNSPEC;
          const double pb = f0[xB] * b0[xB] * exp(f0[xCLS] + b0[xCLS] - fwd);
          const double pe = f1[xE] * b1[xE] * exp(f1[xCLS] + b1[xCLS] - fwd);
          const double sc = exp(f0[xCLS] + b1[xCLS] - fwd);
          const double njcp = (f0[xN] * b1[xN] + f0[xJ] * b1[xJ] + f0[xC] * b1[xC]) * cm.loop * sc;
          pbv[i] = pb; pev[i] = pe; moccv[i] = 1.0 - njcp;
        }
        if (lane == 0) { btotv[0] = 0.0; etotv[0] = 0.0; }
        wave_mem_sync();
        int nenv = 0, nreg = 0, i0 = -1;
        bool trig = false;
        double btot = 0.0, etot = 0.0;
        for (int j0 = 1; j0 <= L; j0 += 64) {
          const int jj = j0 + lane;
          const bool valid = jj <= L;
          const double pbl = valid ? __builtin_nontemporal_load(pbv + jj) : 0.0, pel = valid ? __builtin_nontemporal_load(pev + jj) : 0.0;
          const double mol = valid ? __builtin_nontemporal_load(moccv + jj) : 0.0;
          double bout = 0.0, eout = 0.0;
          const int cnt = L - j0 + 1 < 64 ? L - j0 + 1 : 64;
          for (int t = 0; t < cnt; t++) {
            const int j = j0 + t;
            const double mocc = readlane_d(mol, t);
            const double bold = btot, eold = etot;
            btot = btot + readlane_d(pbl, t);
            etot = etot + readlane_d(pel, t);
            if (lane == t) { bout = btot; eout = etot; }
            if (!trig) {
              if (mocc - (btot - bold) < kRt2) i0 = j;
              else if (i0 == -1) i0 = j;
              if (mocc >= kRt1) trig = true;
            } else if (mocc - (etot - eold) < kRt2) {
              if (nenv < rcap) { if (lane == 0) { rl[kRextInts * nenv] = i0; rl[kRextInts * nenv + 1] = j; } nenv++; }
              else flags |= WH_FLAG_TRUNC;
              nreg++;
              i0 = -1; trig = false;
            }
          }
          if (valid) { btotv[jj] = bout; etotv[jj] = eout; }
        }
        list_sync();
        __builtin_amdgcn_wave_barrier();
        // multidomain test: max_z min(etot[z]-etot[i-1], btot[j]-btot[z-1]) >= rt3
        int multi_mask = 0;       // the first 32 regions (the record of the hmm_list mode)
        for (int e = 0; e < nenv; e++) {
          const int ri = rl[kRextInts * e], rj = rl[kRextInts * e + 1];
          const double e0 = __builtin_nontemporal_load(etotv + ri - 1), bj = __builtin_nontemporal_load(btotv + rj);
          double mxv = -1.0;
          for (int z = ri + lane; z <= rj; z += 64) {
            const double u = __builtin_nontemporal_load(etotv + z) - e0, v = bj - __builtin_nontemporal_load(btotv + z - 1);
            const double w = u < v ? u : v;
            mxv = w > mxv ? w : mxv;
          }
          mxv = wave_max_d(mxv);
          const bool mu = mxv >= kRt3;
          if (mu) { flags |= WH_FLAG_MULTI; if (e < 32) multi_mask |= 1 << e; }
          if (lane == 0) rl[kRextInts * e + 4] = mu ? 1 : 0;
        }
        list_sync();
        nreg_rec = nreg; nenv_rec = nenv;
        if (dp) { dp->nregions = nreg; dp->nenv = nenv < WH_MAX_ENVELOPES ? nenv : WH_MAX_ENVELOPES; }
        if (nenv > 0) {
          // ---------------- A.5 single-domain regions: the envelope is the region
          for (int e = 0; e < nenv; e++) {
            if (rl[kRextInts * e + 4]) { if (lane == 0) { rl[kRextInts * e + 2] = 0; rl[kRextInts * e + 3] = 0; } continue; }
            const int ri = rl[kRextInts * e], rj = rl[kRextInts * e + 1], Ld = rj - ri + 1;
            const uint8_t *eseq = seq + (ri - 1);
            const double envsc = gforward<true>(m, eseq, Ld, cu, mx, lane);
            float domcorr = 0.f;
            if (isfinite(envsc)) {
              for (int q = 0; q < m.Q; q++) { accM[ofs2(q, lane)] = 0.0; accI[ofs2(q, lane)] = 0.0; }
              double xfactor = gbackward<1>(m, eseq, Ld, cu, mx, mx.row(a.Lcap + 2), mx.row(a.Lcap + 3), nullptr, envsc, accM, accI, lane);
              const double norm = 1.0 / (double)Ld;
              xfactor *= norm;
              // null2 by expectation: canonical residues, then the degenerate codes as unweighted means
              double si = 0.0;
              for (int q = 0; q < m.Q; q++) si += __builtin_nontemporal_load(accI + ofs2(q, lane)) * norm;
              for (int x = 0; x < a.K; x++) {
                const double *od = m.te + (size_t)x * SQ;
                double sm = 0.0;
                for (int q = 0; q < m.Q; q++) sm += __builtin_nontemporal_load(accM + ofs2(q, lane)) * norm * od[ofs2(q, lane)];
                const double tot = wave_sum_d(sm + si);
                if (lane == 0) null2[x] = (float)(tot + xfactor);
              }
              __builtin_amdgcn_wave_barrier();
              if (lane >= a.K && lane < a.Kp) {
                const uint32_t msk = a.degen[lane];
                float s = 0.f; int n = 0;
                for (int x = 0; x < a.K; x++) if (msk & (1u << x)) { s += null2[x]; n++; }
                null2[lane] = n > 0 ? s / (float)n : 1.0f;
              }
              __builtin_amdgcn_wave_barrier();
              float dc = 0.f;
              for (int t = lane; t < Ld; t += 64) dc += logf(null2[eseq[t]]);
              domcorr = wave_sum_f(dc);
              __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) { rl[kRextInts * e + 2] = __builtin_bit_cast(int, (float)envsc); rl[kRextInts * e + 3] = __builtin_bit_cast(int, domcorr); }
            if (dp && e < WH_MAX_ENVELOPES) { dp->env_i[e] = ri; dp->env_j[e] = rj; dp->envsc[e] = (float)envsc; dp->domcorr[e] = domcorr; }
          }
          list_sync();
          __builtin_amdgcn_wave_barrier();
          // every pair with a region is finished by resolve_kernel: multidomain regions and the score assembly
          int slot = 0;
          if (lane == 0 && !listed) slot = atomicAdd(a.rcount, 1);
          slot = __shfl(slot, 0);
          if (listed) {
            queued = true;          // (the record is written below, for every pair of the list)
          } else if (slot < a.rcap) {
            queued = true;
            if (lane == 0) {
              ResolveRec *rr = a.rrecs + slot;
              rr->q = qi; rr->h = h; rr->fwdsc = fwdsc; rr->fwd_bits = fwd_bits_out; rr->nreg = nreg; rr->nenv = nenv;
              rr->multi_mask = multi_mask; rr->flags = flags;
              for (int e = 0; e < nenv; e++) {
                rr->ri[e] = rl[kRextInts * e]; rr->rj[e] = rl[kRextInts * e + 1];
                rr->envsc[e] = __builtin_bit_cast(float, rl[kRextInts * e + 2]); rr->domcorr[e] = __builtin_bit_cast(float, rl[kRextInts * e + 3]);
              }
            }
          } else flags |= WH_FLAG_TRUNC;
        }
      }
    }
    if (listed) {
      // every pair of the list has its record (slot = list position): the resolver launch that follows writes its result,
      // its regions from the list at rl (a pair without a region comes back unreported)
      if (lane == 0) {
        ResolveRec *rr = a.rrecs + item;
        rr->q = qi; rr->h = h; rr->fwdsc = fwdsc_rec; rr->fwd_bits = fwd_bits_out; rr->nreg = nreg_rec; rr->nenv = queued ? nenv_rec : 0;
        rr->multi_mask = 0; rr->flags = flags;
      }
    } else if (lane == 0) {
      if (!queued) { a.decibits[out] = 0; a.flags[out] = (uint8_t)flags; }
      if (a.fwd_bits) a.fwd_bits[out] = fwd_bits_out;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------- alignment (A.7)
namespace {
__device__ __forceinline__ float gatef(double t, float v) { return t > 0.0 ? v : 0.0f; }
__device__ __forceinline__ float ldf(const double *p) { return (float)__builtin_nontemporal_load(p); }
}  // namespace

__global__ __launch_bounds__(64, 3) void generic_align_kernel(GenericAlignArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_raw[];
  const int lane = threadIdx.x;
  uint8_t *seq = reinterpret_cast<uint8_t *>(lds_raw);
  double *slab = a.slab + (size_t)blockIdx.x * a.slab_stride;
  enum { oN = 0, oB, oE, oJ, oC };
  for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(a.counter, 1);
    item = __builtin_amdgcn_readfirstlane(__shfl(item, 0));
    if (item >= a.n_items) break;
    const int p = a.items[item];
    const int h = a.pair_h[p];
    const int64_t qi = a.pair_q[p];
    const DevHMM hm = a.hmms[h];
    GModel m;
    m.tf = a.gtab + hm.gfw_off; m.te = a.gtab + hm.gem_off;
    m.Q = __builtin_amdgcn_readfirstlane(hm.Q); m.M = __builtin_amdgcn_readfirstlane(hm.M);
    const int Q = m.Q, M = m.M;
    const size_t SQ = (size_t)Q * 64;
    const double *tf = m.tf;
    GMx mx;
    mx.Q = Q; mx.rowlen = generic_rowlen(Q); mx.p = slab;
    double *pps = slab + (size_t)(a.Lcap + 4) * generic_rowlen(a.Qmax);      // ppN ppJ ppC per row
    double *oxs = pps + (size_t)(a.Lcap + 2) * xNSPEC;                        // OA special states per row
    const int64_t off = a.offsets[qi];
    const int L = (int)(a.offsets[qi + 1] - off);
    int32_t *cols = a.cols + a.col_off[p];
    for (int t = lane; t < L; t += 64) cols[t] = -1;
    if (L <= 0 || L > a.Lcap) continue;
    for (int t = lane; t < L; t += 64) { const int r = a.residues[off + t]; seq[t] = (uint8_t)(r < a.Kp ? r : a.Kp - 1); }
    __builtin_amdgcn_wave_barrier();
    const GLen c = glen_config(L, false);
    const double fwd = gforward<true>(m, seq, L, c, mx, lane);
    if (!isfinite(fwd)) { if (lane == 0 && a.status) a.status[p] = 1; continue; }
    double bwd = 0.0;
    (void)gbackward<2>(m, seq, L, c, mx, mx.row(a.Lcap + 2), mx.row(a.Lcap + 3), pps, fwd, nullptr, nullptr, lane, &bwd);
    // Two hits thousands of bits apart in ONE unihit alignment leave the range of a scaled double (the flank state
    // that carries the weaker hit underflows against the row's scale): Forward and Backward then disagree, and the
    // pair is redone in log space, as hmmalign does.
    if (!(fabs(fwd - bwd) <= 1e-6 * fabs(fwd) + 1e-3)) {
      const double lf = gforward_log(m, seq, L, c, mx, lane);
      const double lb = gbackward_log(m, seq, L, c, mx, mx.row(a.Lcap + 2), mx.row(a.Lcap + 3), pps, lf, lane);
      // (3: the two log-space scores disagree as well - never seen.  hmmalign makes no such comparison: it decodes with
      // the Forward score and aligns, and so does this kernel; the class is counted and reported, nothing is dropped)
      if (lane == 0 && a.status) a.status[p] = (fabs(lf - lb) <= 1e-6 * fabs(lf) + 1e-3) ? 4 : 3;
    }
    // ---------------- optimal-accuracy fill, in place: row i holds ppM / ppI on entry, oM / oI / oD on exit
    const float tNl = c.loop > 0.0 ? 1.0f : 0.0f, tNm = c.move > 0.0 ? 1.0f : 0.0f;
    const float tEJ = c.EJ > 0.0 ? 1.0f : 0.0f, tEC = c.EC > 0.0 ? 1.0f : 0.0f;
    {
      double *r0 = mx.row(0);
      for (int q = 0; q < Q; q++) { const size_t o = ofs2(q, lane); r0[o] = -INFINITY; r0[SQ + o] = -INFINITY; r0[2 * SQ + o] = -INFINITY; }
      if (lane == 0) { oxs[oN] = 0.0; oxs[oB] = 0.0; oxs[oE] = -INFINITY; oxs[oJ] = -INFINITY; oxs[oC] = -INFINITY; }
    }
    float pxN = 0.f, pxB = 0.f, pxJ = -INFINITY, pxC = -INFINITY;
    for (int i = 1; i <= L; i++) {
      wave_mem_sync();
      const double *pr = mx.row(i - 1);
      double *cr = mx.row(i);
      // node k-1 of my first node: lane-1's last node of the previous row
      float pm1 = -INFINITY, pi1 = -INFINITY, pd1 = -INFINITY;
      if (lane > 0) { const size_t ol = ofs2(Q - 1, lane - 1); pm1 = ldf(pr + ol); pi1 = ldf(pr + SQ + ol); pd1 = ldf(pr + 2 * SQ + ol); }
      float mprev = -INFINITY, dloc = -INFINITY, xE = -INFINITY;
      bool pass = true;          // every D->D gate of the lane's nodes 2..Q is open: an entering D reaches the last node
      for (int q = 0; q < Q; q++) {
        const int k = lane * Q + q + 1;
        const size_t o = ofs2(q, lane);
        const float oM1 = ldf(pr + o), oI1 = ldf(pr + SQ + o), oD1 = ldf(pr + 2 * SQ + o);
        const float ppm = ldf(cr + o), ppi = ldf(cr + SQ + o);
        float sv = gatef(tf[gE * SQ + o], pxB), t;
        // (node 1: its predecessors are node 0, whose transitions are zero: every gate closed, value 0)
        t = gatef(tf[gA * SQ + o], pm1); if (t > sv) sv = t;
        t = gatef(tf[gB * SQ + o], pi1); if (t > sv) sv = t;
        t = gatef(tf[gC * SQ + o], pd1); if (t > sv) sv = t;
        sv += ppm;
        const float av = gatef(tf[gMI * SQ + o], oM1), bv = gatef(tf[gII * SQ + o], oI1);
        float iv = (av > bv ? av : bv) + ppi;
        if (k > M) { sv = -INFINITY; iv = -INFINITY; }
        cr[o] = (double)sv; cr[SQ + o] = (double)iv;
        if (k <= M && sv > xE) xE = sv;
        // D chain inside the lane with -inf in its first node (the true first node comes from lane-1, below)
        if (q > 0) {
          const bool gd = tf[gD2 * SQ + o] > 0.0;
          const float a1 = gatef(tf[gD1 * SQ + o], mprev), b1 = gd ? dloc : 0.0f;
          dloc = a1 > b1 ? a1 : b1;
          pass = pass && gd;
        }
        pm1 = oM1; pi1 = oI1; pd1 = oD1; mprev = sv;
      }
      // cross-lane: D(first node of lane r) = max(gate(tMD, M_last(r-1)), tDD open ? D_last(r-1) : 0), and
      // D_last(r) = max(dloc, pass ? D_first(r) : -inf).  Serial over the lanes: this is the slow path.
      const float mlast_up = __shfl_up(mprev, 1);
      const float mleft = lane > 0 ? mlast_up : -INFINITY;
      const double t1 = tf[gD1 * SQ + ofs2(0, lane)], t2 = tf[gD2 * SQ + ofs2(0, lane)];
      float dfirst = -INFINITY, dlast_prev = -INFINITY;
      for (int r = 0; r < 64; r++) {
        const float a1 = gatef(t1, mleft), b1 = t2 > 0.0 ? dlast_prev : 0.0f;
        const float f = a1 > b1 ? a1 : b1;
        const float mine_last = pass ? (dloc > f ? dloc : f) : dloc;
        if (lane == r) dfirst = f;
        dlast_prev = __shfl(mine_last, r);
      }
      // pass 2: the lane's true D values
      {
        float dprev = -INFINITY, mp = -INFINITY;
        for (int q = 0; q < Q; q++) {
          const int k = lane * Q + q + 1;
          const size_t o = ofs2(q, lane);
          float dv;
          if (q == 0) dv = dfirst;
          else {
            const float a1 = gatef(tf[gD1 * SQ + o], mp), b1 = tf[gD2 * SQ + o] > 0.0 ? dprev : 0.0f;
            dv = a1 > b1 ? a1 : b1;
          }
          if (k > M) dv = -INFINITY;
          cr[2 * SQ + o] = (double)dv;
          if (k <= M && dv > xE) xE = dv;
          dprev = dv; mp = ldf(cr + o);
        }
      }
      xE = wave_max_f(xE);
      const float ppN = ldf(pps + (size_t)i * xNSPEC + 0), ppJ = ldf(pps + (size_t)i * xNSPEC + 1), ppC = ldf(pps + (size_t)i * xNSPEC + 2);
      float av = tNl * (pxJ + ppJ), bv = tEJ * xE;
      const float xJ = av > bv ? av : bv;
      av = tNl * (pxC + ppC); bv = tEC * xE;
      const float xC = av > bv ? av : bv;
      const float xN = tNl * (pxN + ppN);
      av = tNm * xN; bv = tNm * xJ;
      const float xBn = av > bv ? av : bv;
      if (lane == 0) { double *t = oxs + (size_t)i * xNSPEC; t[oN] = xN; t[oB] = xBn; t[oE] = xE; t[oJ] = xJ; t[oC] = xC; }
      pxN = xN; pxB = xBn; pxJ = xJ; pxC = xC;
    }
    wave_mem_sync();
    // ---------------- traceback (first maximum wins, candidate order as in A.7), wave-uniform
    {
      enum { tS, tN, tB, tM, tI, tD, tE, tJ, tC };
      auto cell = [&](int i, int k, int s) -> float {
        if (k <= 0) return -INFINITY;
        const int q = (k - 1) % Q, ln = (k - 1) / Q;
        return ldf(mx.row(i) + (size_t)s * SQ + ofs2(q, ln));
      };
      auto ox = [&](int i, int s) -> float { return ldf(oxs + (size_t)i * xNSPEC + s); };
      auto pp = [&](int i, int s) -> float { return ldf(pps + (size_t)i * xNSPEC + s); };
      int s0 = tC, s1, i = L, k = 0, guard = 4 * (L + M) + 16;
      while (s0 != tS && guard-- > 0) {
        switch (s0) {
          case tC: {
            const float av = tNl * (ox(i - 1 < 0 ? 0 : i - 1, oC) + pp(i, 2)), bv = tEC * ox(i, oE);
            s1 = i == 0 ? tE : (bv > av ? tE : tC);
            break;
          }
          case tJ: {
            const float av = tNl * (ox(i - 1 < 0 ? 0 : i - 1, oJ) + pp(i, 1)), bv = tEJ * ox(i, oE);
            s1 = i == 0 ? tE : (bv > av ? tE : tJ);
            break;
          }
          case tE: {
            // argmax over M (">=": later wins) and D (">") in HMMER's striped scan order k = r*Qs + qs + 1: the winner
            // is the LAST M cell at the maximum if any M cell reaches it, else the FIRST D cell at the maximum
            const int Qs = ((M - 1) / 4 + 1) > 2 ? ((M - 1) / 4 + 1) : 2;
            float mxv = -INFINITY;
            const double *row = mx.row(i);
            for (int q = 0; q < Q; q++) {
              const int kk = lane * Q + q + 1;
              if (kk <= M) { const float vm = ldf(row + ofs2(q, lane)), vd = ldf(row + 2 * SQ + ofs2(q, lane)); mxv = vm > mxv ? vm : mxv; mxv = vd > mxv ? vd : mxv; }
            }
            mxv = wave_max_f(mxv);
            int bestM = -1, bestD = 0x7FFFFFFF;
            for (int q = 0; q < Q; q++) {
              const int kk = lane * Q + q + 1;
              if (kk <= M) {
                const int pos = ((kk - 1) % Qs) * 8 + (kk - 1) / Qs;
                if (ldf(row + ofs2(q, lane)) == mxv && pos > bestM) bestM = pos;
                if (ldf(row + 2 * SQ + ofs2(q, lane)) == mxv && pos + 4 < bestD) bestD = pos + 4;
              }
            }
            bestM = wave_max_i(bestM); bestD = wave_min_i(bestD);
            if (mxv == -INFINITY || (bestM < 0 && bestD == 0x7FFFFFFF)) { guard = 0; s1 = tS; if (lane == 0 && a.status) a.status[p] = 2; break; }
            if (bestM >= 0) { k = (bestM & 3) * Qs + (bestM >> 3) + 1; s1 = tM; }
            else { const int pd = bestD - 4; k = (pd & 3) * Qs + (pd >> 3) + 1; s1 = tD; }
            break;
          }
          case tM: {
            const size_t o = m.at(0, k);     // offset of node k inside an array
            float path[4];
            path[0] = gatef(tf[gE * SQ + o], ox(i - 1, oB));
            path[1] = gatef(tf[gA * SQ + o], cell(i - 1, k - 1, 0));
            path[2] = gatef(tf[gB * SQ + o], cell(i - 1, k - 1, 1));
            path[3] = gatef(tf[gC * SQ + o], cell(i - 1, k - 1, 2));
            int best = 0;
            for (int u = 1; u < 4; u++) if (path[u] > path[best]) best = u;
            s1 = best == 0 ? tB : best == 1 ? tM : best == 2 ? tI : tD;
            if (lane == 0) cols[i - 1] = k - 1;
            k--; i--;
            break;
          }
          case tD: {
            const size_t o = m.at(0, k);
            const float av = gatef(tf[gD1 * SQ + o], cell(i, k - 1, 0)), bv = gatef(tf[gD2 * SQ + o], cell(i, k - 1, 2));
            s1 = bv > av ? tD : tM;
            k--;
            break;
          }
          case tI: {
            const size_t o = m.at(0, k);
            const float av = gatef(tf[gMI * SQ + o], cell(i - 1, k, 0)), bv = gatef(tf[gII * SQ + o], cell(i - 1, k, 1));
            s1 = bv > av ? tI : tM;
            i--;
            break;
          }
          case tB: {
            const float av = tNm * ox(i, oN), bv = tNm * ox(i, oJ);
            s1 = bv > av ? tJ : tN;
            break;
          }
          case tN: s1 = i == 0 ? tS : tN; break;
          default: s1 = tS; break;
        }
        s1 = __builtin_amdgcn_readfirstlane(s1);
        if ((s1 == tN || s1 == tJ || s1 == tC) && s1 == s0) i--;
        s0 = s1;
        if (i < 0 || k < 0 || k > M) break;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

hipError_t launch_generic_front(const GenericArgs &a, int blocks, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&generic_front_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(generic_front_kernel, dim3(blocks), dim3(64), lds, s, a);
  return hipGetLastError();
}

// Pairs whose result carries WH_FLAG_TRUNC after the scoring kernels and the resolver: their positions (q * H + h) go to
// <list> for the long-list pass.  <count> counts all of them, the list holds the first <cap> (in no particular order).
__global__ void trunc_list_kernel(const uint8_t *flags, int64_t npairs, int *count, int64_t *list, int cap) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npairs; p += stride)
    if (flags[p] & WH_FLAG_TRUNC) {
      const int t = atomicAdd(count, 1);
      if (t < cap) list[t] = p;
    }
}

hipError_t launch_trunc_list(const uint8_t *flags, int64_t npairs, int *count, int64_t *list, int cap, hipStream_t s) {
  const int blocks = (int)std::min<int64_t>((npairs + 255) / 256, 8192);
  if (blocks < 1) return hipSuccess;
  hipLaunchKernelGGL(trunc_list_kernel, dim3(blocks), dim3(256), 0, s, flags, npairs, count, list, cap);
  return hipGetLastError();
}

hipError_t launch_generic_align(const GenericAlignArgs &a, int blocks, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&generic_align_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(generic_align_kernel, dim3(blocks), dim3(64), lds, s, a);
  return hipGetLastError();
}

}  // namespace wh
