// Log-space Forward / Backward / posterior decoding for the alignment kernel's fallback pass.
//
// hmmalign runs its scaled float32 Forward/Backward and, when Decoding reports that the Forward
// and Backward scale factors have drifted apart by more than float32 can hold (eslERANGE: a
// sequence with two hits of which the later one is far stronger, ...), repeats the pair with its
// "generic" log-space implementation.  The prob-space sweeps of
// wh_device.h detect the same situation (clamp_backward fires) and the pair is then redone
// here.  Same wavefront layout as the prob-space code (lane r owns nodes r*Q+q+1; Backward in
// reversed node order), every value a natural-log probability, -inf for zero.  Speed does not
// matter (flagged pairs are rare): logs of the table entries are taken on the fly.
#pragma once
#include "wh_device.h"

namespace wh {

__device__ __forceinline__ float lsum2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return m;
  return m + __logf(1.0f + __expf(fminf(a, b) - m));
}
__device__ __forceinline__ float lsum3(float a, float b, float c) { return lsum2(lsum2(a, b), c); }
__device__ __forceinline__ float lsum4(float a, float b, float c, float d) { return lsum2(lsum2(a, b), lsum2(c, d)); }
__device__ __forceinline__ float llog(float p) { return p > 0.f ? __logf(p) : -INFINITY; }

// value of lane-1, -inf into lane 0
__device__ __forceinline__ float shr1_log(float x) { return dppf<0x138, 0xF, 0xF, false>(-INFINITY, x); }

// log of the sum over the 64 lanes of exp(x)
__device__ __forceinline__ float wave_lsum(float x) {
  const float m = wave_max(x);
  if (m == -INFINITY) return m;
  return m + __logf(wave_sum(__expf(x - m)));
}

// inclusive scan over lanes of the maps D -> lsum(B_r, A_r + D); the A-parts (sums of log D->D
// coefficients) are model-only
struct LogScanC { float s[6]; };
__device__ __forceinline__ LogScanC logscan_prepare(float A) {
  LogScanC c;
  c.s[0] = A; A += dppf<0x111>(0.f, A);
  c.s[1] = A; A += dppf<0x112>(0.f, A);
  c.s[2] = A; A += dppf<0x114>(0.f, A);
  c.s[3] = A; A += dppf<0x118>(0.f, A);
  c.s[4] = A; A += dppf<0x142, 0xA>(0.f, A);
  c.s[5] = A;
  return c;
}
__device__ __forceinline__ float logscan_apply(const LogScanC &c, float B) {
  B = lsum2(B, c.s[0] + dppf<0x111>(-INFINITY, B));
  B = lsum2(B, c.s[1] + dppf<0x112>(-INFINITY, B));
  B = lsum2(B, c.s[2] + dppf<0x114>(-INFINITY, B));
  B = lsum2(B, c.s[3] + dppf<0x118>(-INFINITY, B));
  B = lsum2(B, c.s[4] + dppf<0x142, 0xA>(-INFINITY, B));
  B = lsum2(B, c.s[5] + dppf<0x143, 0xC>(-INFINITY, B));
  return B;
}

// Forward, unihit or multihit by <cfg>; writes log N,B,E,J,C of rows 0..L to spec[slot*SP + i]
// (slots 0..4 like the prob-space sweep) and the log M / log I rows 1..L densely to <Fs>
// ([row][2][Q/4][64][4], forward node order).  Returns log Z (before the final C->T move).
template <int Q>
__device__ __forceinline__ float forward_sweep_log(const TransTab<Q, false> &T, const float *emL, const float *emG, int K,
                                                   const uint8_t *seq, int L, LenCfg cfg, float *spec, int SP, float *Fs,
                                                   int lane) {
  const float lloop = llog(cfg.loop), lmove = llog(cfg.move), lEC = llog(cfg.EC), lEJ = llog(cfg.EJ);
  float Asum = 0.f;
#pragma unroll
  for (int q4 = 0; q4 < Q / 4; q4++) {
    const float4 d = T.ld(FW_D2, q4);
    Asum += llog(d.x); Asum += llog(d.y); Asum += llog(d.z); Asum += llog(d.w);
  }
  const LogScanC sc = logscan_prepare(Asum);
  float Mp[Q], Ip[Q], Dp[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) { Mp[q] = -INFINITY; Ip[q] = -INFINITY; Dp[q] = -INFINITY; }
  float xN = 0.f, xB = lmove, xJ = -INFINITY, xC = -INFINITY, xE = -INFINITY;
  if (lane == 0) {
    spec[0 * SP] = xN; spec[1 * SP] = xB; spec[2 * SP] = xE; spec[3 * SP] = xJ; spec[4 * SP] = xC;
  }
#pragma unroll 1
  for (int i = 1; i <= L; i++) {
    asm volatile("" ::: "memory");
    float od[Q];
    load_em_fwd<Q>(od, emL, emG, seq[i - 1], K, lane);
    const float mm1 = shr1_log(Mp[Q - 1]), im1 = shr1_log(Ip[Q - 1]), dm1 = shr1_log(Dp[Q - 1]);
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
        const float ni = lsum2(llog(f4get(MI, j)) + Mp[q], llog(f4get(II, j)) + Ip[q]);
        const float acc = lsum4(xB + llog(f4get(E, j)), llog(f4get(A, j)) + pm, llog(f4get(B, j)) + pi, llog(f4get(C, j)) + pd);
        Mp[q] = llog(od[q]) + acc;
        Ip[q] = ni;
      }
    }
    const float mn1 = shr1_log(Mp[Q - 1]);
    float dprev = -INFINITY;
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      const float4 D1 = T.ld(FW_D1, q4), D2 = T.ld(FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
        const float src = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mn1;
        dprev = lsum2(llog(f4get(D2, j)) + dprev, llog(f4get(D1, j)) + src);
        Dp[q] = dprev;
      }
    }
    float carry = shr1_log(logscan_apply(sc, dprev));
    float es = -INFINITY;
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      const float4 D2 = T.ld(FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
        carry += llog(f4get(D2, j));
        Dp[q] = lsum2(Dp[q], carry);
        es = lsum3(es, Mp[q], Dp[q]);
      }
    }
    xE = wave_lsum(es);
    xN = xN + lloop;
    xC = lsum2(xC + lloop, xE + lEC);
    xJ = lsum2(xJ + lloop, xE + lEJ);
    xB = lsum2(xJ, xN) + lmove;
    if (lane == 0) {
      spec[0 * SP + i] = xN; spec[1 * SP + i] = xB; spec[2 * SP + i] = xE; spec[3 * SP + i] = xJ; spec[4 * SP + i] = xC;
    }
    float4 *row = reinterpret_cast<float4 *>(Fs) + (size_t)i * (2 * (Q / 4) * kWave) + lane;
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      nt_store4(row + q4 * kWave, Mp[4 * q4], Mp[4 * q4 + 1], Mp[4 * q4 + 2], Mp[4 * q4 + 3]);
      nt_store4(row + (Q / 4 + q4) * kWave, Ip[4 * q4], Ip[4 * q4 + 1], Ip[4 * q4 + 2], Ip[4 * q4 + 3]);
    }
  }
  return xC + lmove;
}

// Backward (reversed node order) + posterior decoding in place: slab row i goes from (log F_M,
// log F_I) to (P(M_k at i), P(I_k at i)); spec slots pn/pj/pc receive the posteriors of residue i
// being emitted by N / J / C.  <specN/J/C> are the Forward log rows (read at i-1).
template <int Q, typename LoadF, typename StoreP>
__device__ __forceinline__ void backward_posterior_log(const TransTab<Q, false> &T, const float *emL, const float *emG, int K,
                                                        const uint8_t *seq, int L, LenCfg cfg, float lZ, float *slab, int lane,
                                                        LoadF fwd_special, StoreP store_post) {
  const float lloop = llog(cfg.loop), lmove = llog(cfg.move), lEC = llog(cfg.EC), lEJ = llog(cfg.EJ);
  float Asum = 0.f;
#pragma unroll
  for (int p4 = 0; p4 < Q / 4; p4++) {
    const float4 d = T.ld(BW_DD, p4);
    Asum += llog(d.x); Asum += llog(d.y); Asum += llog(d.z); Asum += llog(d.w);
  }
  const LogScanC sc = logscan_prepare(Asum);
  float Mb[Q], Ib[Q];
#pragma unroll
  for (int p = 0; p < Q; p++) { Mb[p] = -INFINITY; Ib[p] = -INFINITY; }
  float xC = lmove, xJ = -INFINITY, xN = -INFINITY, xB = -INFINITY;
#pragma unroll 1
  for (int i = L; i >= 1; i--) {
    asm volatile("" ::: "memory");
    if (i < L) {
      float od[Q];
      load_em_rev<Q>(od, emL, emG, seq[i], K, lane);
      float part = -INFINITY;
#pragma unroll
      for (int p4 = 0; p4 < Q / 4; p4++) {
        const float4 E = T.ld(BW_E, p4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int p = 4 * p4 + j;
          Mb[p] += llog(od[p]);                       // G_k = o_k(x_{i+1}) B_M_k(i+1)
          part = lsum2(part, llog(f4get(E, j)) + Mb[p]);
        }
      }
      xB = wave_lsum(part);
      xJ = lsum2(xJ + lloop, xB + lmove);
      xC = xC + lloop;
      xN = lsum2(xN + lloop, xB + lmove);
    }
    const float xE = lsum2(xC + lEC, xJ + lEJ);
    {
      float Dn[Q];
      const float gm1 = shr1_log(Mb[Q - 1]);
      float dprev = -INFINITY;
#pragma unroll
      for (int p4 = 0; p4 < Q / 4; p4++) {
        const float4 DM = T.ld(BW_DM, p4), DD = T.ld(BW_DD, p4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int p = 4 * p4 + j;
          const float g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
          dprev = lsum3(llog(f4get(DD, j)) + dprev, llog(f4get(DM, j)) + g, xE);
          Dn[p] = dprev;
        }
      }
      float carry = shr1_log(logscan_apply(sc, dprev));
#pragma unroll
      for (int p4 = 0; p4 < Q / 4; p4++) {
        const float4 DD = T.ld(BW_DD, p4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int p = 4 * p4 + j;
          carry += llog(f4get(DD, j));
          Dn[p] = lsum2(Dn[p], carry);
        }
      }
      const float dm1 = shr1_log(Dn[Q - 1]);
#pragma unroll
      for (int p4 = Q / 4 - 1; p4 >= 0; p4--) {
        const float4 MM = T.ld(BW_MM, p4), IM = T.ld(BW_IM, p4), MI = T.ld(BW_MI, p4), II = T.ld(BW_II, p4);
        const float4 MD = T.ld(BW_MD, p4);
#pragma unroll
        for (int j = 3; j >= 0; j--) {
          const int p = 4 * p4 + j;
          const float g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
          const float dn = p > 0 ? Dn[p > 0 ? p - 1 : 0] : dm1;
          const float nm = lsum4(llog(f4get(MM, j)) + g, xE, llog(f4get(MI, j)) + Ib[p], llog(f4get(MD, j)) + dn);
          const float ni = lsum2(llog(f4get(IM, j)) + g, llog(f4get(II, j)) + Ib[p]);
          Mb[p] = nm;
          Ib[p] = ni;
        }
      }
    }
    // posteriors of row i, in place (reversed order: component 3-j of the forward-ordered vector is position 4*p4+j)
    float4 *row = reinterpret_cast<float4 *>(slab) + (size_t)i * (2 * (Q / 4) * kWave) + (kWave - 1 - lane);
#pragma unroll
    for (int p4 = 0; p4 < Q / 4; p4++) {
      float4 *pm = row + (Q / 4 - 1 - p4) * kWave, *pi = row + (Q / 4 + Q / 4 - 1 - p4) * kWave;
      const float4 fm = nt_load4(pm), fi = nt_load4(pi);
      nt_store4(pm, __expf(fm.x + Mb[4 * p4 + 3] - lZ), __expf(fm.y + Mb[4 * p4 + 2] - lZ),
                __expf(fm.z + Mb[4 * p4 + 1] - lZ), __expf(fm.w + Mb[4 * p4 + 0] - lZ));
      nt_store4(pi, __expf(fi.x + Ib[4 * p4 + 3] - lZ), __expf(fi.y + Ib[4 * p4 + 2] - lZ),
                __expf(fi.z + Ib[4 * p4 + 1] - lZ), __expf(fi.w + Ib[4 * p4 + 0] - lZ));
    }
    float fN, fJ, fC;
    fwd_special(i - 1, fN, fJ, fC);
    store_post(i, __expf(fN + xN + lloop - lZ), __expf(fJ + xJ + lloop - lZ), __expf(fC + xC + lloop - lZ));
  }
}

}  // namespace wh
