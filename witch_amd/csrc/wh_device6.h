// Register-resident sweeps: the 8 transition arrays of one orientation stay in VGPRs for a
// whole sweep (TransTab<Q, true>), so a DP row issues only its 4 emission reads to the LDS.
// The row loop is software-pipelined by hand: the residue of row i+1 is fetched at the top
// of row i and its emission row is requested as soon as row i's emissions have been consumed,
// so no LDS latency is exposed on the wave's critical path.  (Measured on MI355X: the LDS
// variant issues 36 ds_read_b128 per row, 1:8 against VALU, and both pipes throttle each
// other at that ratio.)
#pragma once
#include "wh_device.h"

namespace wh {

__device__ __forceinline__ int uniform_i(int x) { return __builtin_amdgcn_readfirstlane(x); }

template <int Q, bool STORE>
__device__ __forceinline__ void forward_sweep_r(const TransTab<Q, true> &T, const ScanC &sc, const float *emL,
                                                const float *emG, int K, const uint8_t *seq, int L, LenCfg cfg,
                                                float *spec, int SP, float *Fs, float keep_scale, int lane,
                                                float &xC_out, int &ef_out) {
  float Mp[Q], Ip[Q], Dp[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) { Mp[q] = 0.f; Ip[q] = 0.f; Dp[q] = 0.f; }
  float xN = 1.0f, xB = cfg.move, xJ = 0.f, xC = 0.f, xE = 0.f;
  int ef = 0;
  if (lane == 0) {
    spec[SP_N * SP] = xN; spec[SP_B * SP] = xB; spec[SP_E * SP] = 0.f; spec[SP_J * SP] = 0.f;
    spec[SP_C * SP] = 0.f; reinterpret_cast<int *>(spec)[SP_S * SP] = 0;
  }
  float od[Q];
  load_em_fwd<Q>(od, emL, emG, uniform_i(seq[0]), K, lane);
#pragma unroll 1
  for (int i = 1; i <= L; i++) {
    asm volatile("" ::: "memory");
    const int xn = seq[i < L ? i : L - 1];            // residue of the next row (requested now, used mid-row)
    const float mm1 = wave_shr1(Mp[Q - 1]), im1 = wave_shr1(Ip[Q - 1]), dm1 = wave_shr1(Dp[Q - 1]);
#pragma unroll
    for (int q4 = Q / 4 - 1; q4 >= 0; q4--) {
      const float4 A = T.ld(FW_A, q4), B = T.ld(FW_B, q4), C = T.ld(FW_C, q4), E = T.ld(FW_E, q4);
      const float4 MI = T.ld(FW_MI, q4), II = T.ld(FW_II, q4);
#pragma unroll
      for (int j = 3; j >= 0; j--) {
        const int q = 4 * q4 + j;
        const float pm = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mm1;
        const float pi = q > 0 ? Ip[q > 0 ? q - 1 : 0] : im1;
        const float pd = q > 0 ? Dp[q > 0 ? q - 1 : 0] : dm1;
        const float ni = fmaf(f4get(MI, j), Mp[q], f4get(II, j) * Ip[q]);
        float acc = xB * f4get(E, j);
        acc = fmaf(f4get(A, j), pm, acc);
        acc = fmaf(f4get(B, j), pi, acc);
        acc = fmaf(f4get(C, j), pd, acc);
        Mp[q] = od[q] * acc;
        Ip[q] = ni;
      }
    }
    // this row's emissions are consumed: request the next row's
    asm volatile("" ::: "memory");
    load_em_fwd<Q>(od, emL, emG, uniform_i(xn), K, lane);
    const float mn1 = wave_shr1(Mp[Q - 1]);
    float dprev = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      const float4 D1 = T.ld(FW_D1, q4), D2 = T.ld(FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
        const float src = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mn1;
        dprev = fmaf(f4get(D2, j), dprev, f4get(D1, j) * src);
        Dp[q] = dprev;
      }
    }
    float carry = wave_shr1(scan_apply(sc, dprev));
    float es = 0.f;
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
    xC = fmaf(xC, cfg.loop, xE * cfg.EC);
    xJ = fmaf(xJ, cfg.loop, xE * cfg.EJ);
    if (xE > kRescaleHi) {
      const int e = f32_exponent(xE);
      const float r = pow2f_int(-e);
#pragma unroll
      for (int q = 0; q < Q; q++) { Mp[q] *= r; Ip[q] *= r; Dp[q] *= r; }
      xN *= r; xC *= r; xJ *= r; xE *= r;
      ef += e;
    }
    xB = (xJ + xN) * cfg.move;
    if (lane == 0) {
      spec[SP_N * SP + i] = xN; spec[SP_B * SP + i] = xB; spec[SP_E * SP + i] = xE;
      spec[SP_J * SP + i] = xJ; spec[SP_C * SP + i] = xC;
      reinterpret_cast<int *>(spec)[SP_S * SP + i] = ef;
    }
    if (STORE) {
      float lmax = 0.f;
#pragma unroll
      for (int q = 0; q < Q; q += 2) lmax = fmaxf(lmax, fmaxf(fmaxf(Mp[q], Mp[q + 1]), fmaxf(Ip[q], Ip[q + 1])));
      const bool keep = lmax > keep_scale * xE;
      const unsigned long long mask = __ballot(keep);
      if (lane == 0) {
        reinterpret_cast<unsigned *>(spec)[SP_ML * SP + i] = (unsigned)(mask & 0xFFFFFFFFull);
        reinterpret_cast<unsigned *>(spec)[SP_MH * SP + i] = (unsigned)(mask >> 32);
      }
      if (keep) {
        float4 *row = reinterpret_cast<float4 *>(Fs) + (size_t)i * (2 * (Q / 4) * kWave) + lane;
#pragma unroll
        for (int q4 = 0; q4 < Q / 4; q4++) {
          nt_store4(row + q4 * kWave, Mp[4 * q4], Mp[4 * q4 + 1], Mp[4 * q4 + 2], Mp[4 * q4 + 3]);
          nt_store4(row + (Q / 4 + q4) * kWave, Ip[4 * q4], Ip[4 * q4 + 1], Ip[4 * q4 + 2], Ip[4 * q4 + 3]);
        }
      }
    }
  }
  xC_out = xC;
  ef_out = ef;
}

}  // namespace wh
