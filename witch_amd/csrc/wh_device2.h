// Two-queries-per-wavefront DP sweeps (packed float32 math, gfx950).
//
// Same execution model as wh_device.h (one wavefront sweeps a DP row held in VGPRs, lanes
// blocked over model nodes), but every DP value is a float2 holding the SAME cell of TWO
// independent (query, HMM) problems that share the model.  All cell arithmetic then
// compiles to v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with the model's transition
// broadcast through op_sel, so one instruction advances both problems and the LDS reads of
// the transition tables are shared: this halves both the VALU issue slots and the LDS
// traffic per (query, HMM) pair.  The two problems may have different lengths: the sweep
// runs to the longer one and each problem's results are latched at its own last row.
#pragma once
#include "wh_device.h"

namespace wh {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f fma2(float a, v2f b, v2f c) { return __builtin_elementwise_fma((v2f)(a), b, c); }
__device__ __forceinline__ v2f splat(float x) { return (v2f)(x); }

__device__ __forceinline__ v2f wave_shr1(v2f x) { v2f r; r.x = wave_shr1(x.x); r.y = wave_shr1(x.y); return r; }
__device__ __forceinline__ v2f wave_sum(v2f x) {
  // the two components are reduced with interleaved DPP steps (independent chains)
  x.x += dppf<0xB1>(0.f, x.x);  x.y += dppf<0xB1>(0.f, x.y);
  x.x += dppf<0x4E>(0.f, x.x);  x.y += dppf<0x4E>(0.f, x.y);
  x.x += dppf<0x141>(0.f, x.x); x.y += dppf<0x141>(0.f, x.y);
  x.x += dppf<0x140>(0.f, x.x); x.y += dppf<0x140>(0.f, x.y);
  v2f r;
  r.x = (readlane_f(x.x, 0) + readlane_f(x.x, 16)) + (readlane_f(x.x, 32) + readlane_f(x.x, 48));
  r.y = (readlane_f(x.y, 0) + readlane_f(x.y, 16)) + (readlane_f(x.y, 32) + readlane_f(x.y, 48));
  return r;
}
__device__ __forceinline__ v2f scan_apply(const ScanC &c, v2f B) {
  B.x = fmaf(c.s[0], dppf<0x111>(0.f, B.x), B.x); B.y = fmaf(c.s[0], dppf<0x111>(0.f, B.y), B.y);
  B.x = fmaf(c.s[1], dppf<0x112>(0.f, B.x), B.x); B.y = fmaf(c.s[1], dppf<0x112>(0.f, B.y), B.y);
  B.x = fmaf(c.s[2], dppf<0x114>(0.f, B.x), B.x); B.y = fmaf(c.s[2], dppf<0x114>(0.f, B.y), B.y);
  B.x = fmaf(c.s[3], dppf<0x118>(0.f, B.x), B.x); B.y = fmaf(c.s[3], dppf<0x118>(0.f, B.y), B.y);
  B.x = fmaf(c.s[4], dppf<0x142, 0xA>(0.f, B.x), B.x); B.y = fmaf(c.s[4], dppf<0x142, 0xA>(0.f, B.y), B.y);
  B.x = fmaf(c.s[5], dppf<0x143, 0xC>(0.f, B.x), B.x); B.y = fmaf(c.s[5], dppf<0x143, 0xC>(0.f, B.y), B.y);
  return B;
}

// One of the two problems a wavefront advances together.
struct Prob {
  const uint8_t *seq;   // LDS, residues of the (sub)sequence, readable up to the pair's max length
  int L;                // rows of this problem
  float *spec;          // this problem's per-row special-state arrays in LDS (stride SP)
  float *Fs;            // this problem's Forward slab (STORE)
};

struct LenCfg2 { v2f loop, move; float EJ, EC; };
__device__ __forceinline__ LenCfg2 len_config2(int L0, int L1, bool multihit) {
  const LenCfg a = len_config(L0, multihit), b = len_config(L1, multihit);
  LenCfg2 c;
  c.loop = (v2f){a.loop, b.loop}; c.move = (v2f){a.move, b.move}; c.EJ = a.EJ; c.EC = a.EC;
  return c;
}

// emission row pieces of the two problems, interleaved into float2 cells
template <int Q>
__device__ __forceinline__ void load_em2_fwd(v2f (&od)[Q], const float *emL, const float *emG, int x0, int x1, int K, int lane) {
  float a[Q], b[Q];
  load_em_fwd<Q>(a, emL, emG, x0, K, lane);
  load_em_fwd<Q>(b, emL, emG, x1, K, lane);
#pragma unroll
  for (int q = 0; q < Q; q++) od[q] = (v2f){a[q], b[q]};
}
template <int Q>
__device__ __forceinline__ void load_em2_rev(v2f (&od)[Q], const float *emL, const float *emG, int x0, int x1, int K, int lane) {
  float a[Q], b[Q];
  load_em_rev<Q>(a, emL, emG, x0, K, lane);
  load_em_rev<Q>(b, emL, emG, x1, K, lane);
#pragma unroll
  for (int q = 0; q < Q; q++) od[q] = (v2f){a[q], b[q]};
}

// per-component power-of-two rescale factor: 2^-exponent(x) where x > thr, else 1
__device__ __forceinline__ void rescale_factors(v2f big, v2f &r, v2i &e) {
  e.x = big.x > kRescaleHi ? f32_exponent(big.x) : 0;
  e.y = big.y > kRescaleHi ? f32_exponent(big.y) : 0;
  r.x = pow2f_int(-e.x);
  r.y = pow2f_int(-e.y);
}

// ------------------------------------------------------------------ Forward sweep, 2 problems
// Transition tables are read from LDS (TransTab<Q,false>).  Fills each problem's spec arrays for
// rows 0..L_c; with STORE spills M/I rows (sparse, see wh_device.h).  xC_out/ef_out are latched
// at each problem's own last row.
template <int Q, bool STORE>
__device__ __forceinline__ void forward_sweep2(const TransTab<Q, false> &T, const ScanC &sc, const float *emL,
                                               const float *emG, int K, const Prob &p0, const Prob &p1,
                                               LenCfg2 cfg, int SP, float keep_scale, int lane, v2f &xC_out,
                                               v2i &ef_out) {
  v2f Mp[Q], Ip[Q], Dp[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) { Mp[q] = splat(0.f); Ip[q] = splat(0.f); Dp[q] = splat(0.f); }
  v2f xN = splat(1.0f), xB = cfg.move, xJ = splat(0.f), xC = splat(0.f), xE = splat(0.f);
  v2i ef = {0, 0};
  xC_out = splat(0.f);
  ef_out = ef;
  if (lane < 2) {
    float *sp = lane == 0 ? p0.spec : p1.spec;
    const float mv = lane == 0 ? cfg.move.x : cfg.move.y;
    sp[SP_N * SP] = 1.0f; sp[SP_B * SP] = mv; sp[SP_E * SP] = 0.f; sp[SP_J * SP] = 0.f; sp[SP_C * SP] = 0.f;
    reinterpret_cast<int *>(sp)[SP_S * SP] = 0;
  }
  const int Lmax = p0.L > p1.L ? p0.L : p1.L;
#pragma unroll 1
  for (int i = 1; i <= Lmax; i++) {
    asm volatile("" ::: "memory");   // keep LDS table reads inside the row (no hoisting into VGPRs)
    // emission rows of the two problems stay in separate registers: pairing them would cost a
    // v_mov per cell, two scalar multiplies into the halves of the packed cell cost nothing extra.
    // Rows beyond a problem's own length compute garbage in its component only (valid residue 0).
    float od0[Q], od1[Q];
    load_em_fwd<Q>(od0, emL, emG, i <= p0.L ? p0.seq[i - 1] : 0, K, lane);
    load_em_fwd<Q>(od1, emL, emG, i <= p1.L ? p1.seq[i - 1] : 0, K, lane);
    const v2f mm1 = wave_shr1(Mp[Q - 1]), im1 = wave_shr1(Ip[Q - 1]), dm1 = wave_shr1(Dp[Q - 1]);
#pragma unroll
    for (int q4 = Q / 4 - 1; q4 >= 0; q4--) {
      const float4 A = T.ld(FW_A, q4), B = T.ld(FW_B, q4), C = T.ld(FW_C, q4), E = T.ld(FW_E, q4);
      const float4 MI = T.ld(FW_MI, q4), II = T.ld(FW_II, q4);
#pragma unroll
      for (int j = 3; j >= 0; j--) {
        const int q = 4 * q4 + j;
        const v2f pm = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mm1;
        const v2f pi = q > 0 ? Ip[q > 0 ? q - 1 : 0] : im1;
        const v2f pd = q > 0 ? Dp[q > 0 ? q - 1 : 0] : dm1;
        const v2f ni = fma2(f4get(MI, j), Mp[q], f4get(II, j) * Ip[q]);
        v2f acc = xB * f4get(E, j);
        acc = fma2(f4get(A, j), pm, acc);
        acc = fma2(f4get(B, j), pi, acc);
        acc = fma2(f4get(C, j), pd, acc);
        Mp[q].x = od0[q] * acc.x;
        Mp[q].y = od1[q] * acc.y;
        Ip[q] = ni;
      }
    }
    const v2f mn1 = wave_shr1(Mp[Q - 1]);
    v2f dprev = splat(0.f);
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      const float4 D1 = T.ld(FW_D1, q4), D2 = T.ld(FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
        const v2f src = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mn1;
        dprev = fma2(f4get(D2, j), dprev, f4get(D1, j) * src);
        Dp[q] = dprev;
      }
    }
    v2f carry = wave_shr1(scan_apply(sc, dprev));
    v2f es = splat(0.f);
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      const float4 D2 = T.ld(FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
        carry *= f4get(D2, j);
        Dp[q] += carry;
        es += Mp[q] + Dp[q];
      }
    }
    xE = wave_sum(es);
    xN = xN * cfg.loop;
    xC = fma2(xC, cfg.loop, xE * cfg.EC);
    xJ = fma2(xJ, cfg.loop, xE * cfg.EJ);
    if (xE.x > kRescaleHi || xE.y > kRescaleHi) {
      v2f r; v2i e;
      rescale_factors(xE, r, e);
#pragma unroll
      for (int q = 0; q < Q; q++) { Mp[q] *= r; Ip[q] *= r; Dp[q] *= r; }
      xN *= r; xC *= r; xJ *= r; xE *= r;
      ef += e;
    }
    xB = (xJ + xN) * cfg.move;
    if (lane < 2) {
      const bool z = lane == 0;
      float *sp = z ? p0.spec : p1.spec;
      if (i <= (z ? p0.L : p1.L)) {
        sp[SP_N * SP + i] = z ? xN.x : xN.y; sp[SP_B * SP + i] = z ? xB.x : xB.y; sp[SP_E * SP + i] = z ? xE.x : xE.y;
        sp[SP_J * SP + i] = z ? xJ.x : xJ.y; sp[SP_C * SP + i] = z ? xC.x : xC.y;
        reinterpret_cast<int *>(sp)[SP_S * SP + i] = z ? ef.x : ef.y;
      }
    }
    if (i == p0.L) { xC_out.x = xC.x; ef_out.x = ef.x; }
    if (i == p1.L) { xC_out.y = xC.y; ef_out.y = ef.y; }
    if (STORE) {
      v2f lmax = splat(0.f);
#pragma unroll
      for (int q = 0; q < Q; q++) lmax = __builtin_elementwise_max(lmax, __builtin_elementwise_max(Mp[q], Ip[q]));
      const bool keep0 = (lmax.x > keep_scale * xE.x) && i <= p0.L;
      const bool keep1 = (lmax.y > keep_scale * xE.y) && i <= p1.L;
      const unsigned long long m0 = __ballot(keep0), m1 = __ballot(keep1);
      if (lane < 2) {
        const bool z = lane == 0;
        float *sp = z ? p0.spec : p1.spec;
        const unsigned long long m = z ? m0 : m1;
        if (i <= (z ? p0.L : p1.L)) {
          reinterpret_cast<unsigned *>(sp)[SP_ML * SP + i] = (unsigned)(m & 0xFFFFFFFFull);
          reinterpret_cast<unsigned *>(sp)[SP_MH * SP + i] = (unsigned)(m >> 32);
        }
      }
      if (keep0) {
        float4 *row = reinterpret_cast<float4 *>(p0.Fs) + (size_t)i * (2 * (Q / 4) * kWave) + lane;
#pragma unroll
        for (int q4 = 0; q4 < Q / 4; q4++) {
          nt_store4(row + q4 * kWave, Mp[4 * q4].x, Mp[4 * q4 + 1].x, Mp[4 * q4 + 2].x, Mp[4 * q4 + 3].x);
          nt_store4(row + (Q / 4 + q4) * kWave, Ip[4 * q4].x, Ip[4 * q4 + 1].x, Ip[4 * q4 + 2].x, Ip[4 * q4 + 3].x);
        }
      }
      if (keep1) {
        float4 *row = reinterpret_cast<float4 *>(p1.Fs) + (size_t)i * (2 * (Q / 4) * kWave) + lane;
#pragma unroll
        for (int q4 = 0; q4 < Q / 4; q4++) {
          nt_store4(row + q4 * kWave, Mp[4 * q4].y, Mp[4 * q4 + 1].y, Mp[4 * q4 + 2].y, Mp[4 * q4 + 3].y);
          nt_store4(row + (Q / 4 + q4) * kWave, Ip[4 * q4].y, Ip[4 * q4 + 1].y, Ip[4 * q4 + 2].y, Ip[4 * q4 + 3].y);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ Backward row, 2 problems
template <int Q>
__device__ __forceinline__ void backward_cells2(const TransTab<Q, false> &T, const ScanC &sc, v2f (&Mb)[Q],
                                                v2f (&Ib)[Q], v2f xE) {
  v2f Dn[Q];
  const v2f gm1 = wave_shr1(Mb[Q - 1]);
  v2f dprev = splat(0.f);
#pragma unroll
  for (int p4 = 0; p4 < Q / 4; p4++) {
    const float4 DM = T.ld(BW_DM, p4), DD = T.ld(BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
      const v2f g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
      dprev = fma2(f4get(DD, j), dprev, fma2(f4get(DM, j), g, xE));
      Dn[p] = dprev;
    }
  }
  v2f carry = wave_shr1(scan_apply(sc, dprev));
#pragma unroll
  for (int p4 = 0; p4 < Q / 4; p4++) {
    const float4 DD = T.ld(BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
      carry *= f4get(DD, j);
      Dn[p] += carry;
    }
  }
  const v2f dm1 = wave_shr1(Dn[Q - 1]);
#pragma unroll
  for (int p4 = Q / 4 - 1; p4 >= 0; p4--) {
    const float4 MM = T.ld(BW_MM, p4), IM = T.ld(BW_IM, p4), MI = T.ld(BW_MI, p4), II = T.ld(BW_II, p4);
    const float4 MD = T.ld(BW_MD, p4);
#pragma unroll
    for (int j = 3; j >= 0; j--) {
      const int p = 4 * p4 + j;
      const v2f g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
      const v2f dn = p > 0 ? Dn[p > 0 ? p - 1 : 0] : dm1;
      v2f nm = fma2(f4get(MM, j), g, xE);
      nm = fma2(f4get(MI, j), Ib[p], nm);
      nm = fma2(f4get(MD, j), dn, nm);
      const v2f ni = fma2(f4get(IM, j), g, f4get(II, j) * Ib[p]);
      Mb[p] = nm;
      Ib[p] = ni;
    }
  }
}

// State of a paired Backward sweep (special states and scale exponents per problem).
struct Bck2 {
  v2f xC, xJ, xN, xB;
  v2i eb;
};

// Advance both problems to row i (called for i = Lmax .. 1 or 0).  A problem whose last row
// is i is (re)initialised here, so rows above its own length never leak into its results.
// Returns E(i).  With i >= 1 the cells of row i are produced in Mb/Ib.
template <int Q>
__device__ __forceinline__ v2f backward_row2(const TransTab<Q, false> &T, const ScanC &sc, const float *emL,
                                             const float *emG, int K, const Prob &p0, const Prob &p1, LenCfg2 cfg,
                                             int i, int lane, v2f (&Mb)[Q], v2f (&Ib)[Q], Bck2 &st, bool cells) {
  const int Lmax = p0.L > p1.L ? p0.L : p1.L;
  if (i < Lmax) {
    float od0[Q], od1[Q];
    load_em_rev<Q>(od0, emL, emG, i < p0.L ? p0.seq[i] : 0, K, lane);
    load_em_rev<Q>(od1, emL, emG, i < p1.L ? p1.seq[i] : 0, K, lane);
    v2f part = splat(0.f);
#pragma unroll
    for (int p4 = 0; p4 < Q / 4; p4++) {
      const float4 E = T.ld(BW_E, p4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int p = 4 * p4 + j;
        Mb[p].x *= od0[p];
        Mb[p].y *= od1[p];
        part = fma2(f4get(E, j), Mb[p], part);
      }
    }
    st.xB = wave_sum(part);
    st.xJ = fma2(st.xJ, cfg.loop, st.xB * cfg.move);
    st.xC = st.xC * cfg.loop;
    st.xN = fma2(st.xN, cfg.loop, st.xB * cfg.move);
  }
  // (re)initialise the problem(s) whose last row is i: the row above does not exist for them
  if (i == p0.L || i == p1.L) {
    const bool r0 = i == p0.L, r1 = i == p1.L;
#pragma unroll
    for (int p = 0; p < Q; p++) {
      if (r0) { Mb[p].x = 0.f; Ib[p].x = 0.f; }
      if (r1) { Mb[p].y = 0.f; Ib[p].y = 0.f; }
    }
    if (r0) { st.xC.x = cfg.move.x; st.xJ.x = 0.f; st.xN.x = 0.f; st.xB.x = 0.f; st.eb.x = 0; }
    if (r1) { st.xC.y = cfg.move.y; st.xJ.y = 0.f; st.xN.y = 0.f; st.xB.y = 0.f; st.eb.y = 0; }
  }
  v2f xE = fma2(st.xC, splat(cfg.EC), st.xJ * cfg.EJ);
  if (cells) backward_cells2<Q>(T, sc, Mb, Ib, xE);
  const v2f big = __builtin_elementwise_max(st.xB, st.xN);
  if (big.x > kRescaleHi || big.y > kRescaleHi) {
    v2f r; v2i e;
    rescale_factors(big, r, e);
#pragma unroll
    for (int p = 0; p < Q; p++) { Mb[p] *= r; Ib[p] *= r; }
    st.xB *= r; st.xJ *= r; st.xC *= r; st.xN *= r; xE *= r;
    st.eb += e;
  }
  return xE;
}

}  // namespace wh
