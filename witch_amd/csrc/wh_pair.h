// Two queries per wavefront: the DP sweeps of wh_device.h with the rows of TWO (query, HMM) pairs of the SAME model in
// one wavefront's registers.
//
// Why: the full-width sweeps of the one-pair-per-wave kernel are bound by the CU's LDS read port, not by the vector ALU -
// a DP row of 1 024 cells reads 37 KiB of transition / emission tables (37 ds_read_b128 per wave), twelve waves per CU
// keep the 256 B/clk LDS port busy 72 % of the time while each SIMD's VALU is busy 45-50 % (round-4 analysis).
// Two pairs on one model use the SAME table pieces: read once, applied to both rows, the LDS bytes per row fall from 37
// to 20.5 KiB.  The price is registers (2 x 3Q DP cells + 2 x Q emission odds), so the kernel runs two waves per SIMD
// instead of three - but every wave now carries two independent instruction streams, which is what a wave needs to issue
// back to back (a single stream stalls on its own D-chain, DPP scans and reductions).
//
// Per query the arithmetic is the single-pair sweep's, operation by operation in the same order: results are the same.
// The two queries may differ in length: rows are indexed per query, a query that has run out of rows keeps computing on
// its last residue (nothing of that is written anywhere).
#pragma once
#include "wh_device.h"

namespace wh {

// two independent affine scans (scan_apply), steps interleaved: the other scan's step stands in the wait states a DPP
// read of a just-written register needs (2), so the pair costs 6 x 3 instructions instead of 2 x 6 x 3
__device__ __forceinline__ void scan_apply2(const ScanC &c, float &B0, float &B1) {
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %5 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %5 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(B0), "+v"(B1)
      : "v"(c.s[0]), "v"(c.s[1]), "v"(c.s[2]), "v"(c.s[3]), "v"(c.s[4]), "v"(c.s[5]));
}

// two wave sums, butterflies interleaved (same order of additions as wave_sum for each)
__device__ __forceinline__ void wave_sum2(float &x0, float &x1) {
  x0 += dppf<0xB1>(0.f, x0);  x1 += dppf<0xB1>(0.f, x1);
  x0 += dppf<0x4E>(0.f, x0);  x1 += dppf<0x4E>(0.f, x1);
  x0 += dppf<0x141>(0.f, x0); x1 += dppf<0x141>(0.f, x1);
  x0 += dppf<0x140>(0.f, x0); x1 += dppf<0x140>(0.f, x1);
  x0 = (readlane_f(x0, 0) + readlane_f(x0, 16)) + (readlane_f(x0, 32) + readlane_f(x0, 48));
  x1 = (readlane_f(x1, 0) + readlane_f(x1, 16)) + (readlane_f(x1, 32) + readlane_f(x1, 48));
}

// one query's view of a pair sweep
struct PairQ {
  const uint8_t *seq;   // residues of the swept rows (row i reads seq[i-1])
  int L;                // rows
  float *spec;          // per-row special-state arrays (stride SP), LDS
  float *Fs;            // Forward-row slab (STORE)
};

// ------------------------------------------------------------------ Forward sweep, two queries
// forward_sweep (wh_device.h) for two queries; STORE uses the SLIM layout (mask words in the SP_B / SP_E slots).
template <int Q, bool STORE>
__device__ __forceinline__ void forward_sweep2(const TransTab<Q, false> &T, const ScanC &sc, const float *emL, const float *emG,
                                               int K, const PairQ (&pq)[2], const LenCfg (&cfg)[2], int SP, float keep_scale,
                                               int lane, float (&xC_out)[2], int (&ef_out)[2]) {
  float Mp[2][Q], Ip[2][Q], Dp[2][Q];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int q = 0; q < Q; q++) { Mp[n][q] = 0.f; Ip[n][q] = 0.f; Dp[n][q] = 0.f; }
  float xN[2], xB[2], xJ[2], xC[2], xE[2];
  int ef[2];
  unsigned long long umask[2];
#pragma unroll
  for (int n = 0; n < 2; n++) {
    xN[n] = 1.0f; xB[n] = cfg[n].move; xJ[n] = 0.f; xC[n] = 0.f; xE[n] = 0.f; ef[n] = 0; umask[n] = 0;
    xC_out[n] = 0.f; ef_out[n] = 0;
    if (lane == 0) {
      float *spec = pq[n].spec;
      spec[SP_N * SP] = xN[n]; spec[SP_B * SP] = xB[n]; spec[SP_E * SP] = 0.f; spec[SP_J * SP] = 0.f;
      spec[SP_C * SP] = 0.f; reinterpret_cast<int *>(spec)[SP_S * SP] = 0;
    }
  }
  const int Lmax = pq[0].L > pq[1].L ? pq[0].L : pq[1].L;
#pragma unroll 1
  for (int i = 1; i <= Lmax; i++) {
    asm volatile("" ::: "memory");   // keep LDS table reads inside the row (no hoisting into VGPRs)
    bool act[2];
    float od[2][Q];
    float mm1[2], im1[2], dm1[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
      act[n] = i <= pq[n].L;
      const int x = __builtin_amdgcn_readfirstlane((int)pq[n].seq[(act[n] ? i : pq[n].L) - 1]);
      load_em_fwd<Q>(od[n], emL, emG, x, K, lane);
      mm1[n] = wave_shr1(Mp[n][Q - 1]); im1[n] = wave_shr1(Ip[n][Q - 1]); dm1[n] = wave_shr1(Dp[n][Q - 1]);
    }
#pragma unroll
    for (int q4 = Q / 4 - 1; q4 >= 0; q4--) {
      const float4 A = T.ld(FW_A, q4), B = T.ld(FW_B, q4), C = T.ld(FW_C, q4), E = T.ld(FW_E, q4);
      const float4 MI = T.ld(FW_MI, q4), II = T.ld(FW_II, q4);
#pragma unroll
      for (int j = 3; j >= 0; j--) {
        const int q = 4 * q4 + j;
#pragma unroll
        for (int n = 0; n < 2; n++) {
          const float pm = q > 0 ? Mp[n][q > 0 ? q - 1 : 0] : mm1[n];
          const float pi = q > 0 ? Ip[n][q > 0 ? q - 1 : 0] : im1[n];
          const float pd = q > 0 ? Dp[n][q > 0 ? q - 1 : 0] : dm1[n];
          const float ni = fmaf(f4get(MI, j), Mp[n][q], f4get(II, j) * Ip[n][q]);
          float acc = xB[n] * f4get(E, j);
          acc = fmaf(f4get(A, j), pm, acc);
          acc = fmaf(f4get(B, j), pi, acc);
          acc = fmaf(f4get(C, j), pd, acc);
          Mp[n][q] = od[n][q] * acc;
          Ip[n][q] = ni;
        }
      }
    }
    // D rows: local chains, cross-lane scans, fix-up
    float mn1[2], dprev[2];
#pragma unroll
    for (int n = 0; n < 2; n++) { mn1[n] = wave_shr1(Mp[n][Q - 1]); dprev[n] = 0.f; }
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      const float4 D1 = T.ld(FW_D1, q4), D2 = T.ld(FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
#pragma unroll
        for (int n = 0; n < 2; n++) {
          const float src = q > 0 ? Mp[n][q > 0 ? q - 1 : 0] : mn1[n];
          dprev[n] = fmaf(f4get(D2, j), dprev[n], f4get(D1, j) * src);
          Dp[n][q] = dprev[n];
        }
      }
    }
    scan_apply2(sc, dprev[0], dprev[1]);
    float carry[2], es[2];
#pragma unroll
    for (int n = 0; n < 2; n++) { carry[n] = wave_shr1(dprev[n]); es[n] = 0.f; }
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      const float4 D2 = T.ld(FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
#pragma unroll
        for (int n = 0; n < 2; n++) {
          carry[n] *= f4get(D2, j);
          Dp[n][q] += carry[n];
          es[n] += Mp[n][q] + Dp[n][q];
        }
      }
    }
    wave_sum2(es[0], es[1]);
#pragma unroll
    for (int n = 0; n < 2; n++) {
      xE[n] = es[n];
      xN[n] = xN[n] * cfg[n].loop;
      xC[n] = fmaf(xC[n], cfg[n].loop, xE[n] * cfg[n].EC);
      xJ[n] = fmaf(xJ[n], cfg[n].loop, xE[n] * cfg[n].EJ);
      if (xE[n] > kRescaleHi) {
        const int e = f32_exponent(xE[n]);
        const float r = pow2f_int(-e);
#pragma unroll
        for (int q = 0; q < Q; q++) { Mp[n][q] *= r; Ip[n][q] *= r; Dp[n][q] *= r; }
        xN[n] *= r; xC[n] *= r; xJ[n] *= r; xE[n] *= r;
        ef[n] += e;
      }
      xB[n] = (xJ[n] + xN[n]) * cfg[n].move;
      if (act[n]) {
        float *spec = pq[n].spec;
        if (lane == 0) {
          spec[SP_N * SP + i] = xN[n];
          if (!STORE) { spec[SP_B * SP + i] = xB[n]; spec[SP_E * SP + i] = xE[n]; }
          spec[SP_J * SP + i] = xJ[n]; spec[SP_C * SP + i] = xC[n];
          reinterpret_cast<int *>(spec)[SP_S * SP + i] = ef[n];
        }
        if (i == pq[n].L) { xC_out[n] = xC[n]; ef_out[n] = ef[n]; }
        if (STORE) {
          float lmax = 0.f;
#pragma unroll
          for (int q = 0; q < Q; q += 2) lmax = fmaxf(lmax, fmaxf(fmaxf(Mp[n][q], Mp[n][q + 1]), fmaxf(Ip[n][q], Ip[n][q + 1])));
          const bool keep = keep_scale < 0.f || lmax > keep_scale * xE[n];
          const unsigned long long mask = __ballot(keep);
          umask[n] |= __ballot(lmax > 0.5f * xE[n]);
          if (lane == 0) {
            reinterpret_cast<unsigned *>(spec)[SP_B * SP + i] = (unsigned)(mask & 0xFFFFFFFFull);
            reinterpret_cast<unsigned *>(spec)[SP_E * SP + i] = (unsigned)(mask >> 32);
          }
          if (keep) {
            float4 *row = reinterpret_cast<float4 *>(pq[n].Fs) + (size_t)i * (2 * (Q / 4) * kWave);
#pragma unroll
            for (int q4 = 0; q4 < Q / 4; q4++) {
              nt_store4(row + fs_piece<Q>(lane, q4), Mp[n][4 * q4], Mp[n][4 * q4 + 1], Mp[n][4 * q4 + 2], Mp[n][4 * q4 + 3]);
              nt_store4(row + fs_piece<Q>(lane, Q / 4 + q4), Ip[n][4 * q4], Ip[n][4 * q4 + 1], Ip[n][4 * q4 + 2], Ip[n][4 * q4 + 3]);
            }
          }
        }
      }
    }
  }
  if (STORE && lane == 0) {
#pragma unroll
    for (int n = 0; n < 2; n++) {
      reinterpret_cast<unsigned *>(pq[n].spec)[SP_B * SP] = (unsigned)(umask[n] & 0xFFFFFFFFull);
      reinterpret_cast<unsigned *>(pq[n].spec)[SP_E * SP] = (unsigned)(umask[n] >> 32);
    }
  }
}

// ------------------------------------------------------------------ Backward building blocks, two queries
// backward_emit for two queries: G_k = o_k(x) * B_M_k in place and the partial B-state sums
template <int Q>
__device__ __forceinline__ void backward_emit2(const TransTab<Q, false> &T, const float *emL, const float *emG, const int (&x)[2],
                                               int K, int lane, float (&Mb)[2][Q], float (&part)[2]) {
  part[0] = 0.f; part[1] = 0.f;
  // the emission pieces of both queries come from the LDS copy unless a residue is a degenerate code (rare: L2)
  auto groups = [&](auto em0, auto em1) {
#pragma unroll
    for (int p4 = 0; p4 < Q / 4; p4++) {
      const float4 E = T.ld(BW_E, p4);
      const float4 O0 = em0(Q / 4 - 1 - p4), O1 = em1(Q / 4 - 1 - p4);
      Mb[0][4 * p4 + 0] *= O0.w; part[0] = fmaf(E.x, Mb[0][4 * p4 + 0], part[0]);
      Mb[1][4 * p4 + 0] *= O1.w; part[1] = fmaf(E.x, Mb[1][4 * p4 + 0], part[1]);
      Mb[0][4 * p4 + 1] *= O0.z; part[0] = fmaf(E.y, Mb[0][4 * p4 + 1], part[0]);
      Mb[1][4 * p4 + 1] *= O1.z; part[1] = fmaf(E.y, Mb[1][4 * p4 + 1], part[1]);
      Mb[0][4 * p4 + 2] *= O0.y; part[0] = fmaf(E.z, Mb[0][4 * p4 + 2], part[0]);
      Mb[1][4 * p4 + 2] *= O1.y; part[1] = fmaf(E.z, Mb[1][4 * p4 + 2], part[1]);
      Mb[0][4 * p4 + 3] *= O0.x; part[0] = fmaf(E.w, Mb[0][4 * p4 + 3], part[0]);
      Mb[1][4 * p4 + 3] *= O1.x; part[1] = fmaf(E.w, Mb[1][4 * p4 + 3], part[1]);
    }
  };
  const int x0 = x[0], x1 = x[1];
  if (x0 < K && x1 < K) {
    const LdsF4 e0(emL + (size_t)x0 * Q * kWave + 4 * (kWave - 1 - lane)), e1(emL + (size_t)x1 * Q * kWave + 4 * (kWave - 1 - lane));
    groups([&](int q4) { return e0[q4 * kWave]; }, [&](int q4) { return e1[q4 * kWave]; });
  } else {
    const float4 *e0 = reinterpret_cast<const float4 *>(emG + (size_t)x0 * Q * kWave) + (kWave - 1 - lane);
    const float4 *e1 = reinterpret_cast<const float4 *>(emG + (size_t)x1 * Q * kWave) + (kWave - 1 - lane);
    groups([&](int q4) { return e0[q4 * kWave]; }, [&](int q4) { return e1[q4 * kWave]; });
  }
}

// backward_cells for two queries
template <int Q>
__device__ __forceinline__ void backward_cells2(const TransTab<Q, false> &T, const ScanC &sc, float (&Mb)[2][Q], float (&Ib)[2][Q],
                                                const float (&xE)[2]) {
  float Dn[2][Q];
  float gm1[2], dprev[2];
#pragma unroll
  for (int n = 0; n < 2; n++) { gm1[n] = wave_shr1(Mb[n][Q - 1]); dprev[n] = 0.f; }
#pragma unroll
  for (int p4 = 0; p4 < Q / 4; p4++) {
    const float4 DM = T.ld(BW_DM, p4), DD = T.ld(BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
#pragma unroll
      for (int n = 0; n < 2; n++) {
        const float g = p > 0 ? Mb[n][p > 0 ? p - 1 : 0] : gm1[n];
        dprev[n] = fmaf(f4get(DD, j), dprev[n], fmaf(f4get(DM, j), g, xE[n]));
        Dn[n][p] = dprev[n];
      }
    }
  }
  scan_apply2(sc, dprev[0], dprev[1]);
  float carry[2];
#pragma unroll
  for (int n = 0; n < 2; n++) carry[n] = wave_shr1(dprev[n]);
#pragma unroll
  for (int p4 = 0; p4 < Q / 4; p4++) {
    const float4 DD = T.ld(BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
#pragma unroll
      for (int n = 0; n < 2; n++) {
        carry[n] *= f4get(DD, j);
        Dn[n][p] += carry[n];
      }
    }
  }
  float dm1[2];
#pragma unroll
  for (int n = 0; n < 2; n++) dm1[n] = wave_shr1(Dn[n][Q - 1]);
#pragma unroll
  for (int p4 = Q / 4 - 1; p4 >= 0; p4--) {
    const float4 MM = T.ld(BW_MM, p4), IM = T.ld(BW_IM, p4), MI = T.ld(BW_MI, p4), II = T.ld(BW_II, p4);
    const float4 MD = T.ld(BW_MD, p4);
#pragma unroll
    for (int j = 3; j >= 0; j--) {
      const int p = 4 * p4 + j;
#pragma unroll
      for (int n = 0; n < 2; n++) {
        const float g = p > 0 ? Mb[n][p > 0 ? p - 1 : 0] : gm1[n];
        const float dn = p > 0 ? Dn[n][p > 0 ? p - 1 : 0] : dm1[n];
        float nm = fmaf(f4get(MM, j), g, xE[n]);
        nm = fmaf(f4get(MI, j), Ib[n][p], nm);
        nm = fmaf(f4get(MD, j), dn, nm);
        const float ni = fmaf(f4get(IM, j), g, f4get(II, j) * Ib[n][p]);
        Mb[n][p] = nm;
        Ib[n][p] = ni;
      }
    }
  }
}

}  // namespace wh
