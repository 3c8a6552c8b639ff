// Scoring kernel for models of 3 073 - 12 288 nodes: SEVERAL wavefronts per (query, HMM) pair.
//
// `hmmbuild --symfrac 0.0` (witch_msa/gcmm/algorithm.py:463-470) makes every populated backbone column a node, so
// the upper subsets of a large backbone (16S: 5-12 k columns) exceed what ONE wavefront holds in registers
// (64 lanes x 48 cells).  Until round 3 such models ran through the float64 one-wave kernels of wh_generic.hip,
// every DP row through memory, 30 x below the register kernels per cell.  Here a workgroup of W = 3..8 wavefronts
// owns one pair: wave w, lane r holds the cells of nodes ((w*64 + r)*Q + q + 1), Q = 24, in registers, float32
// probability space with exact power-of-two rescaling - the arithmetic of wh_score7.hip (same five sweeps, same
// operation order inside a lane, HMMER's float32 score assembly).  What crosses a wavefront boundary goes through
// LDS, twice per row and sweep:
//   * the D->D chain: every wave scans its own lanes (DPP affine scan), publishes the value its last cell would
//     have with a zero carry-in, and after the barrier folds the maps of the waves in front of it (A_wave is
//     model-only, formed once per sweep) into its carry-in;
//   * the row sum E(i) (Forward) / the B-state sum (Backward), the cells of the last lane that the next wave's
//     first lane needs, formed in wave order so that the result does not depend on timing.
// The model's tables do not fit in LDS (2 x 8 arrays x 12 288 nodes x 4 B): they are read from L2 on every use, in
// the lane-blocked layout of wh_common.h over W*64 virtual lanes (a wave's access is one contiguous 1 KiB line);
// all workgroups work on the same model at a time (model-major work items), so the reads hit L2.  The per-row
// special states of the ONE pair a workgroup works on live in LDS.  Envelope Forward rows are spilled densely to
// the workgroup's slab (no sparse spill, no node window: those are refinements of the one-wave kernel).
// Pairs with a multidomain region are queued for resolve_kernel exactly as wh_score7.hip queues them.
#include <hip/hip_runtime.h>

#include "wh_device.h"
#include "wh_launch.h"

namespace wh {

// floats in front of the LDS emission copy of the TR scoring kernels (= the whole block of the others), 16-byte aligned
__host__ __device__ inline int wide_em_lds_offset_floats(int Lcap) {
  const int SP = (Lcap + 1 + 3) / 4 * 4;
  const int fl = 4 + 6 * SP + 8 * 8 + 32 + kRegsInts + (Lcap + 16 + 3) / 4;
  return (fl + 3) / 4 * 4;
}

namespace wide {

constexpr float kWideKeepScale = 5.9604645e-08f;   // 2^-24 of the row's E, as in wh_score7.hip
constexpr float kWideMassTol = 2e-5f;

// per-wave exchange slot in LDS (floats)
enum { X_BD = 0, X_ES, X_BM, X_BI, X_BDD, X_AT, X_T0, X_T1, X_N };

__device__ __forceinline__ float flogsum0_w(float b) {
  const float mx = b > 0.f ? b : 0.f, mn = b > 0.f ? 0.f : b;
  if (mn == -INFINITY || (mx - mn) >= 15.7f) return mx;
  const int idx = (int)((mx - mn) * 1000.0f);
  return mx + (float)log(1.0 + exp((double)-idx / 1000.0));
}

// global-address-space views with WAVE-UNIFORM bases: a table access is then one scalar base + the lane's 32-bit offset +
// an immediate (with flat per-lane pointers the compiler kept 60 64-bit addresses per row in scratch)
typedef float wv4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) wv4 gf4;
__device__ __forceinline__ const gf4 *uniform_global(const void *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const gf4 *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float4 ldg4(const gf4 *base, unsigned idx) { const wv4 v = base[idx]; return make_float4(v.x, v.y, v.z, v.w); }
// element <elem> (wave-uniform) of virtual lane <vl>: the uniform part goes into the SCALAR base (two SALU adds), every access of
// a lane shares ONE 32-bit VGPR offset.  <nl> is kept opaque per row (asm) by the callers: otherwise the compiler hoists the 60
// loop-invariant addresses of a row out of the row loop as 64-bit VGPR pairs and spills them
__device__ __forceinline__ float4 ldt(const gf4 *base, int elem, int nl, unsigned vl) {
  const gf4 *p = base + (size_t)(unsigned)(elem * nl);
  const wv4 v = p[vl];
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }

typedef __attribute__((address_space(3))) wv4 lf4;
template <int NLT>
struct WCtxT {
  const gf4 *fw, *bw, *em;      // uniform bases of the three table groups: element (arr, q4) of virtual lane vl at [(arr*Q4 + q4) * NL + vl]
  const lf4 *emL;               // TR kernels: the emission rows of the canonical residues in LDS (same layout), or null
  int vl;
  float *spec;                  // LDS: SP_NARR-2 arrays of SP floats (N, B, E, J, C, S), one pair per workgroup
  float *xch;                   // LDS: W slots of X_N floats
  gf4 *Fs;                      // slab of the workgroup (uniform base): [row][2][Q4][NL]
  unsigned long long *masks;    // sparse spill (scoring): [row][W] the lanes of each wave that stored their cells of the row; or null
  unsigned long long *rowstat;  // WH_STATS: cycles of the multihit Forward row by segment ([8..15] of the stats block), or null
  int NLr, SP, w, W, lane, K, Kp;
  // lanes over all waves of the workgroup: a compile-time constant in the production instantiations (table accesses then
  // are one scalar base + immediate offsets; with a run-time value the compiler kept 60 offsets per row in scratch)
  __device__ __forceinline__ int nl() const { return NLT > 0 ? NLT : NLr; }
};

__device__ __forceinline__ void wg_barrier() { __syncthreads(); }

// What every wave reads from the exchange slots after a barrier, without a loop of dependent LDS reads: lane v fetches the
// slot of wave v (ONE round of LDS reads), the sum / the fold then walks the lanes with v_readlane in wave order (the same
// operations in the same order as the loop over the slots: bit-identical).  With a compile-time wave count the lane
// numbers are immediates.
template <int NLT>
__device__ __forceinline__ float slots_sum(const float *xch, int slot, int W, int lane) {
  const float v = lane < W ? xch[lane * X_N + slot] : 0.f;
  float s = 0.f;
  if constexpr (NLT > 0) {
#pragma unroll
    for (int u = 0; u < NLT / 64; u++) s += readlane_f(v, u);
  } else {
    for (int u = 0; u < W; u++) s += readlane_f(v, u);
  }
  return s;
}
// cin of wave w = the D of the last cell of wave w-1 = the fold over the waves in front of D_last(v) = A_v * D_last(v-1) + B_v:
// an affine prefix scan over the (at most eight) wave slots, lane v holding wave v - three DPP steps for every wave alike
// instead of w dependent steps for wave w (the last wave's fold was the longest stretch between two barriers).  The A part
// is model-only: its step multipliers are formed once per sweep.
struct WaveScan { float s0, s1, s2; };
__device__ __forceinline__ WaveScan wave_scan_prepare(const float *xch, int W, int lane) {
  float A = lane < W ? xch[lane * X_N + X_AT] : 1.f;
  WaveScan c;
  c.s0 = A; A *= dppf<0x111>(1.f, A);
  c.s1 = A; A *= dppf<0x112>(1.f, A);
  c.s2 = A;
  return c;
}
__device__ __forceinline__ float slots_fold(const WaveScan &c, const float *xch, int W, int w, int lane) {
  float B = lane < W ? xch[lane * X_N + X_BD] : 0.f;
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s0));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s1));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s2));
  asm("s_nop 1" : "+v"(B));
  return w > 0 ? readlane_f(B, w - 1) : 0.f;
}

// inclusive product over lanes 0..lane of <a> (model-only; once per sweep)
__device__ __forceinline__ float lane_prefix_product(float a, int lane) {
  for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_up(a, d); if (lane >= d) a *= o; }
  return a;
}

// ------------------------------------------------------------------------------------------ Forward (P1 / P3)
// TR: the eight transition arrays of the sweep live in REGISTERS (8 x Q floats per lane: 12-cell lanes), loaded once per sweep,
// and the emission rows of the canonical residues are read from LDS - a row then issues no table load at all.  Without TR
// (24-cell lanes: the tables would need 192 registers) every use is an L2 read.
template <int Q>
__device__ __forceinline__ float4 emission_piece(const gf4 *emG, const lf4 *emL, int K, int x, int q4, int nlv, unsigned vl) {
  constexpr int Q4 = Q / 4;
  if (emL != nullptr && x < K) { const wv4 v = emL[(unsigned)((x * Q4 + q4) * nlv) + vl]; return make_float4(v.x, v.y, v.z, v.w); }
  return ldt(emG, x * Q4 + q4, nlv, vl);
}

// STORE: the M and I rows go to the slab.  With c.masks a lane's cells are written only when one of them exceeds
// keep_scale * E(row) (the sparse spill of wh_device.h: a read of 800 residues walks an eighth of a 6 000-node model, and the
// dense rows made P3 / P4 HBM-bound: 49 KB per row and workgroup), the kept lanes of every wave are recorded per row, and the
// envelope's posterior mass certifies the result (backward_null2_wide); keep_scale < 0 stores everything.
template <int Q, bool STORE, int NLT, bool TR = false>
__device__ __forceinline__ void forward_wide(const WCtxT<NLT> &c, const uint8_t *seq, int L, LenCfg cfg, float &xC_out, int &ef_out, float keep_scale = -1.0f) {
  constexpr int Q4 = Q / 4;
  const int lane = c.lane, NL = c.nl(), SP = c.SP, w = c.w;
  float *spec = c.spec, *xch = c.xch;
  int nlv = NL;
  float4 tf[TR ? FW_NARR : 1][TR ? Q4 : 1];
  if (TR) {
#pragma unroll
    for (int a = 0; a < FW_NARR; a++)
#pragma unroll
      for (int q4 = 0; q4 < Q4; q4++) tf[TR ? a : 0][TR ? q4 : 0] = ldg4(c.fw, (unsigned)((a * Q4 + q4) * NL + c.vl));
  }
  auto T = [&](int a, int q4) -> float4 { if constexpr (TR) return tf[a][q4]; else return ldt(c.fw, a * Q4 + q4, nlv, (unsigned)c.vl); };
  // model-only parts of the D scans
  float A = 1.f;
#pragma unroll
  for (int q4 = 0; q4 < Q4; q4++) { const float4 d = T(FW_D2, q4); A *= d.x; A *= d.y; A *= d.z; A *= d.w; }
  const ScanC sc = scan_prepare(A);
  const float Aincl = lane_prefix_product(A, lane);
  float Aexcl = __shfl_up(Aincl, 1);
  if (lane == 0) Aexcl = 1.f;
  if (lane == 63) xch[w * X_N + X_AT] = Aincl;
  // TR: the in-lane running products of the D->D transitions (what FW_P holds for the one-wave kernels): the carry then
  // enters every cell with ONE independent FMA instead of a chain of Q dependent multiplications - two waves per SIMD do
  // not hide a dependent chain
  float Pq[TR ? Q : 1];
  if (TR) {
    float run = 1.f;
#pragma unroll
    for (int q4 = 0; q4 < Q4; q4++) {
      const float4 d = T(FW_D2, q4);
      run *= d.x; Pq[TR ? 4 * q4 : 0] = run; run *= d.y; Pq[TR ? 4 * q4 + 1 : 0] = run;
      run *= d.z; Pq[TR ? 4 * q4 + 2 : 0] = run; run *= d.w; Pq[TR ? 4 * q4 + 3 : 0] = run;
    }
  }
  float Mp[Q], Ip[Q], Dp[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) { Mp[q] = 0.f; Ip[q] = 0.f; Dp[q] = 0.f; }
  float xN = 1.0f, xB = cfg.move, xJ = 0.f, xC = 0.f, xE = 0.f;
  float bM = 0.f, bI = 0.f, bD = 0.f;          // row i-1 cells of the previous wave's last lane (lane 0 uses them)
  int ef = 0;
  if (w == 0 && lane == 0) {
    spec[SP_N * SP] = xN; spec[SP_B * SP] = xB; spec[SP_E * SP] = 0.f; spec[SP_J * SP] = 0.f; spec[SP_C * SP] = 0.f;
    reinterpret_cast<int *>(spec)[SP_S * SP] = 0;
  }
  wg_barrier();                                // X_AT of every wave is visible
  const WaveScan scW = wave_scan_prepare(xch, c.W, lane);
  long long seg[7] = {0, 0, 0, 0, 0, 0, 0};
  const bool timed = !STORE && c.rowstat != nullptr;
  long long tr = timed ? __builtin_readcyclecounter() : 0;
  auto mark = [&](int k) { if (timed) { const long long t = __builtin_readcyclecounter(); seg[k] += t - tr; tr = t; } };
#pragma unroll 1
  for (int i = 1; i <= L; i++) {
    const int x = __builtin_amdgcn_readfirstlane((int)seq[i - 1]);
    nlv = opaque_s(NL);
    float mm1 = wave_shr1(Mp[Q - 1]), im1 = wave_shr1(Ip[Q - 1]), dm1 = wave_shr1(Dp[Q - 1]);
    if (lane == 0) { mm1 = bM; im1 = bI; dm1 = bD; }
#pragma unroll
    for (int q4 = Q4 - 1; q4 >= 0; q4--) {
      const float4 Aa = T(FW_A, q4), Bb = T(FW_B, q4), Cc = T(FW_C, q4), Ee = T(FW_E, q4);
      const float4 MI = T(FW_MI, q4), II = T(FW_II, q4);
      const float4 O = TR ? emission_piece<Q>(c.em, c.emL, c.K, x, q4, nlv, (unsigned)c.vl) : ldt(c.em, x * Q4 + q4, nlv, (unsigned)c.vl);
#pragma unroll
      for (int j = 3; j >= 0; j--) {
        const int q = 4 * q4 + j;
        const float pm = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mm1;
        const float pi = q > 0 ? Ip[q > 0 ? q - 1 : 0] : im1;
        const float pd = q > 0 ? Dp[q > 0 ? q - 1 : 0] : dm1;
        const float ni = fmaf(f4get(MI, j), Mp[q], f4get(II, j) * Ip[q]);
        float acc = xB * f4get(Ee, j);
        acc = fmaf(f4get(Aa, j), pm, acc);
        acc = fmaf(f4get(Bb, j), pi, acc);
        acc = fmaf(f4get(Cc, j), pd, acc);
        Mp[q] = f4get(O, j) * acc;
        Ip[q] = ni;
      }
    }
    // D row: chains inside the lane, scan over the lanes of this wave, maps of the waves in front.  The first cell of a
    // wave needs the NEW M of the cell in front of it (the last cell of the previous wave): one exchange of its own
    if (lane == 63) xch[w * X_N + X_T0] = Mp[Q - 1];
    mark(0);
    wg_barrier();                                                               // #0
    mark(1);
    float mn1 = wave_shr1(Mp[Q - 1]);
    if (lane == 0) mn1 = w > 0 ? xch[(w - 1) * X_N + X_T0] : 0.f;
    float dprev = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < Q4; q4++) {
      const float4 D1 = T(FW_D1, q4), D2 = T(FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
        const float src = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mn1;
        dprev = fmaf(f4get(D2, j), dprev, f4get(D1, j) * src);
        Dp[q] = dprev;
      }
    }
    const float loc = scan_apply(sc, dprev);        // D of my last cell if nothing entered the wave from the front
    if (lane == 63) xch[w * X_N + X_BD] = loc;
    mark(2);
    wg_barrier();                                                               // #1
    mark(3);
    // D of the last cell of the wave in front: D_last(v) = A_v * D_last(v-1) + B_v, folded in wave order
    const float cin = slots_fold(scW, xch, c.W, w, lane);
    float carry = wave_shr1(loc);
    carry = fmaf(Aexcl, cin, carry);
    if (lane == 0) carry = cin;
    float es = 0.f;
    if constexpr (TR) {
      float part[Q4];
#pragma unroll
      for (int q4 = 0; q4 < Q4; q4++) {
#pragma unroll
        for (int j = 0; j < 4; j++) { const int q = 4 * q4 + j; Dp[q] = fmaf(Pq[q], carry, Dp[q]); }
        part[q4] = ((Mp[4 * q4] + Dp[4 * q4]) + (Mp[4 * q4 + 1] + Dp[4 * q4 + 1])) + ((Mp[4 * q4 + 2] + Dp[4 * q4 + 2]) + (Mp[4 * q4 + 3] + Dp[4 * q4 + 3]));
      }
#pragma unroll
      for (int q4 = 0; q4 < Q4; q4++) es += part[q4];
    } else {
#pragma unroll
      for (int q4 = 0; q4 < Q4; q4++) {
        const float4 D2 = T(FW_D2, q4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int q = 4 * q4 + j;
          carry *= f4get(D2, j);
          Dp[q] += carry;
          es += Mp[q] + Dp[q];
        }
      }
    }
    const float esw = wave_sum(es);
    if (lane == 63) { xch[w * X_N + X_ES] = esw; xch[w * X_N + X_BM] = Mp[Q - 1]; xch[w * X_N + X_BI] = Ip[Q - 1]; xch[w * X_N + X_BDD] = Dp[Q - 1]; }
    mark(4);
    wg_barrier();                                                               // #2
    mark(5);
    xE = slots_sum<NLT>(xch, X_ES, c.W, lane);
    if (w > 0) { bM = xch[(w - 1) * X_N + X_BM]; bI = xch[(w - 1) * X_N + X_BI]; bD = xch[(w - 1) * X_N + X_BDD]; }
    xN = xN * cfg.loop;
    xC = fmaf(xC, cfg.loop, xE * cfg.EC);
    xJ = fmaf(xJ, cfg.loop, xE * cfg.EJ);
    if (xE > kRescaleHi) {
      const int e = f32_exponent(xE);
      const float r = pow2f_int(-e);
#pragma unroll
      for (int q = 0; q < Q; q++) { Mp[q] *= r; Ip[q] *= r; Dp[q] *= r; }
      xN *= r; xC *= r; xJ *= r; xE *= r;
      bM *= r; bI *= r; bD *= r;
      ef += e;
    }
    xB = (xJ + xN) * cfg.move;
    if (w == 0 && lane == 0) {
      spec[SP_N * SP + i] = xN; spec[SP_B * SP + i] = xB; spec[SP_E * SP + i] = xE;
      spec[SP_J * SP + i] = xJ; spec[SP_C * SP + i] = xC;
      reinterpret_cast<int *>(spec)[SP_S * SP + i] = ef;
    }
    if (STORE) {
      bool keep = true;
      if (c.masks != nullptr) {
        float lmax = 0.f;
#pragma unroll
        for (int q = 0; q < Q; q += 2) lmax = fmaxf(lmax, fmaxf(fmaxf(Mp[q], Mp[q + 1]), fmaxf(Ip[q], Ip[q + 1])));
        keep = keep_scale < 0.f || lmax > keep_scale * xE;
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) __builtin_nontemporal_store(mask, c.masks + (size_t)i * c.W + w);
      }
      if (keep) {
        gf4 *row = c.Fs + (size_t)i * (2 * Q4) * NL;
#pragma unroll
        for (int q4 = 0; q4 < Q4; q4++) {
          const wv4 vm = {Mp[4 * q4], Mp[4 * q4 + 1], Mp[4 * q4 + 2], Mp[4 * q4 + 3]};
          const wv4 vi = {Ip[4 * q4], Ip[4 * q4 + 1], Ip[4 * q4 + 2], Ip[4 * q4 + 3]};
          row[(unsigned)(q4 * NL + c.vl)] = vm;
          row[(unsigned)((Q4 + q4) * NL + c.vl)] = vi;
        }
      }
    }
    mark(6);
  }
  if (timed && lane == 0 && (w == 0 || w == c.W - 1)) for (int k = 0; k < 7; k++) atomicAdd(c.rowstat + (w == 0 ? 0 : 8) + k, (unsigned long long)seg[k]);
  xC_out = xC;
  ef_out = ef;
  wg_barrier();
}


// ------------------------------------------------------------------------------------------ Backward rows
// One Backward row for every wave of the workgroup (reversed node order: wave 0, lane 0 holds the LAST node).
// On entry Mb / Ib hold row i+1 (zeros for i = L).  <emit>: multiply in the emission odds of residue x (row i < L) and
// form the B-state sum over all waves; then the D chain with the carries of the waves in front, then M / I.
template <int Q>
struct BackState { float Mb[Q], Ib[Q]; };

// the Backward sweep's eight transition arrays in registers (TR) - or nothing, and every use is an L2 read
template <int Q, bool TR>
struct BackTab {
  static constexpr bool kP = TR && Q <= kWideQReg;     // (16-cell lanes: the registers go to the tables; the carry walks its chain)
  float4 v[TR ? BW_NARR : 1][TR ? Q / 4 : 1];
  float P[kP ? Q : 1];           // in-lane running products of the D->D transitions (see forward_wide)
  template <int NLT>
  __device__ __forceinline__ void load(const WCtxT<NLT> &c) {
    if (TR) {
#pragma unroll
      for (int a = 0; a < BW_NARR; a++)
#pragma unroll
        for (int q4 = 0; q4 < Q / 4; q4++) v[TR ? a : 0][TR ? q4 : 0] = ldg4(c.bw, (unsigned)((a * (Q / 4) + q4) * c.nl() + c.vl));
      if (kP) {
        float run = 1.f;
#pragma unroll
        for (int q4 = 0; q4 < Q / 4; q4++) {
          const float4 d = v[TR ? BW_DD : 0][TR ? q4 : 0];
          run *= d.x; P[kP ? 4 * q4 : 0] = run; run *= d.y; P[kP ? 4 * q4 + 1 : 0] = run;
          run *= d.z; P[kP ? 4 * q4 + 2 : 0] = run; run *= d.w; P[kP ? 4 * q4 + 3 : 0] = run;
        }
      }
    }
  }
};

template <int Q, int NLT, bool TR = false>
__device__ __forceinline__ float backward_emit_wide(const WCtxT<NLT> &c, const BackTab<Q, TR> &tb, int x, float (&Mb)[Q], float &gfront) {
  constexpr int Q4 = Q / 4;
  const int NL = c.nl(), w = c.w, lane = c.lane;
  float *xch = c.xch;
  // reversed order: my reversed cells 4*p4 .. 4*p4+3 are the forward piece (Q4-1-p4) of the forward lane (NL-1-vl), components reversed
  const int rv = NL - 1 - c.vl;              // the forward lane that holds my reversed cells
  const int nlv = opaque_s(NL);
  float part = 0.f;
#pragma unroll
  for (int p4 = 0; p4 < Q4; p4++) {
    float4 E, O;
    if constexpr (TR) { E = tb.v[BW_E][p4]; O = emission_piece<Q>(c.em, c.emL, c.K, x, Q4 - 1 - p4, nlv, (unsigned)rv); }
    else { E = ldt(c.bw, BW_E * Q4 + p4, nlv, (unsigned)c.vl); O = ldt(c.em, x * Q4 + (Q4 - 1 - p4), nlv, (unsigned)rv); }
    if constexpr (TR) {
      // (one short chain per piece instead of one of Q dependent FMAs)
      Mb[4 * p4 + 0] *= O.w; Mb[4 * p4 + 1] *= O.z; Mb[4 * p4 + 2] *= O.y; Mb[4 * p4 + 3] *= O.x;
      float pp = E.x * Mb[4 * p4 + 0];
      pp = fmaf(E.y, Mb[4 * p4 + 1], pp); pp = fmaf(E.z, Mb[4 * p4 + 2], pp); pp = fmaf(E.w, Mb[4 * p4 + 3], pp);
      part += pp;
    } else {
      Mb[4 * p4 + 0] *= O.w; part = fmaf(E.x, Mb[4 * p4 + 0], part);
      Mb[4 * p4 + 1] *= O.z; part = fmaf(E.y, Mb[4 * p4 + 1], part);
      Mb[4 * p4 + 2] *= O.y; part = fmaf(E.z, Mb[4 * p4 + 2], part);
      Mb[4 * p4 + 3] *= O.x; part = fmaf(E.w, Mb[4 * p4 + 3], part);
    }
  }
  const float pw = wave_sum(part);
  if (lane == 63) { xch[w * X_N + X_ES] = pw; xch[w * X_N + X_BM] = Mb[Q - 1]; }
  wg_barrier();                                                                 // #A
  const float xB = slots_sum<NLT>(xch, X_ES, c.W, lane);
  gfront = w > 0 ? xch[(w - 1) * X_N + X_BM] : 0.f;     // G of the cell in front of my first one
  return xB;
}

template <int Q, int NLT, bool TR = false>
__device__ __forceinline__ void backward_cells_wide(const WCtxT<NLT> &c, const BackTab<Q, TR> &tb, const ScanC &sc, const WaveScan &scW, float Aexcl, float (&Mb)[Q], float (&Ib)[Q], float xE, float gfront) {
  constexpr int Q4 = Q / 4;
  const int NL = c.nl(), w = c.w, lane = c.lane;
  float *xch = c.xch;
  const int nlv = opaque_s(NL);
  auto T = [&](int a, int q4) -> float4 { if constexpr (TR) return tb.v[a][q4]; else return ldt(c.bw, a * Q4 + q4, nlv, (unsigned)c.vl); };
  float Dn[Q];
  float gm1 = wave_shr1(Mb[Q - 1]);
  if (lane == 0) gm1 = gfront;
  float dprev = 0.f;
#pragma unroll
  for (int p4 = 0; p4 < Q4; p4++) {
    const float4 DM = T(BW_DM, p4), DD = T(BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
      const float g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
      dprev = fmaf(f4get(DD, j), dprev, fmaf(f4get(DM, j), g, xE));
      Dn[p] = dprev;
    }
  }
  const float loc = scan_apply(sc, dprev);
  if (lane == 63) xch[w * X_N + X_BD] = loc;
  wg_barrier();                                                                 // #B
  const float cin = slots_fold(scW, xch, c.W, w, lane);
  float carry = wave_shr1(loc);
  carry = fmaf(Aexcl, cin, carry);
  if (lane == 0) carry = cin;
  const float dfront = carry;                  // D of the cell in front of my first one (lane r: last cell of lane r-1)
  if constexpr (BackTab<Q, TR>::kP) {
#pragma unroll
    for (int p = 0; p < Q; p++) Dn[p] = fmaf(tb.P[p], carry, Dn[p]);
  } else {
#pragma unroll
    for (int p4 = 0; p4 < Q4; p4++) {
      const float4 DD = T(BW_DD, p4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int p = 4 * p4 + j;
        carry *= f4get(DD, j);
        Dn[p] += carry;
      }
    }
  }
#pragma unroll
  for (int p4 = Q4 - 1; p4 >= 0; p4--) {
    const float4 MM = T(BW_MM, p4), IM = T(BW_IM, p4), MI = T(BW_MI, p4), II = T(BW_II, p4);
    const float4 MD = T(BW_MD, p4);
#pragma unroll
    for (int j = 3; j >= 0; j--) {
      const int p = 4 * p4 + j;
      const float g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
      const float dn = p > 0 ? Dn[p > 0 ? p - 1 : 0] : dfront;
      float nm = fmaf(f4get(MM, j), g, xE);
      nm = fmaf(f4get(MI, j), Ib[p], nm);
      nm = fmaf(f4get(MD, j), dn, nm);
      const float ni = fmaf(f4get(IM, j), g, f4get(II, j) * Ib[p]);
      Mb[p] = nm;
      Ib[p] = ni;
    }
  }
}

// model-only parts of the Backward D scans of this sweep (X_AT published; the caller's first barrier makes it visible)
template <int Q, int NLT>
__device__ __forceinline__ void backward_prepare_wide(const WCtxT<NLT> &c, ScanC &sc, float &Aexcl, WaveScan &scW) {
  constexpr int Q4 = Q / 4;
  float A = 1.f;
#pragma unroll
  for (int q4 = 0; q4 < Q4; q4++) { const float4 d = ldg4(c.bw, (unsigned)((BW_DD * Q4 + q4) * c.nl() + c.vl)); A *= d.x; A *= d.y; A *= d.z; A *= d.w; }
  sc = scan_prepare(A);
  const float Aincl = lane_prefix_product(A, c.lane);
  Aexcl = __shfl_up(Aincl, 1);
  if (c.lane == 0) Aexcl = 1.f;
  if (c.lane == 63) c.xch[c.w * X_N + X_AT] = Aincl;
  wg_barrier();
  scW = wave_scan_prepare(c.xch, c.W, c.lane);
}

// ------------------------------------------------------------------------------------------ P2: multihit Backward + decoding
template <int Q, int NLT, bool TR = false>
__device__ __forceinline__ void backward_decode_wide(const WCtxT<NLT> &c, const uint8_t *seq, int L, LenCfg cm, float invZ, int ef_L) {
  const int SP = c.SP;
  float *spec = c.spec;
  ScanC sc;
  float Aexcl;
  WaveScan scW;
  backward_prepare_wide<Q, NLT>(c, sc, Aexcl, scW);
  BackTab<Q, TR> tb;
  tb.load(c);
  float Mb[Q], Ib[Q];
#pragma unroll
  for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; }
  float xC = cm.move, xJ = 0.f, xN = 0.f, xB = 0.f;
  int eb = 0;
#pragma unroll 1
  for (int i = L; i >= 0; i--) {
    float gfront = 0.f;
    if (i < L) {
      xB = backward_emit_wide<Q, NLT, TR>(c, tb, __builtin_amdgcn_readfirstlane((int)seq[i]), Mb, gfront);
      xJ = fmaf(xJ, cm.loop, xB * cm.move);
      xC = xC * cm.loop;
      xN = fmaf(xN, cm.loop, xB * cm.move);
    }
    float xE = fmaf(xC, cm.EC, xJ * cm.EJ);
    if (i >= 1) backward_cells_wide<Q, NLT, TR>(c, tb, sc, scW, Aexcl, Mb, Ib, xE, gfront);
    const float big = fmaxf(xB, xN);
    if (big > kRescaleHi) {
      const int e = f32_exponent(big);
      const float r = pow2f_int(-e);
#pragma unroll
      for (int p = 0; p < Q; p++) { Mb[p] *= r; Ib[p] *= r; }
      xB *= r; xJ *= r; xC *= r; xN *= r; xE *= r;
      eb += e;
    }
    const float s_i = ldexpf(invZ, reinterpret_cast<const int *>(spec)[SP_S * SP + i] + eb - ef_L);
    const float pe = spec[SP_E * SP + i] * xE * s_i;
    const float pb = spec[SP_B * SP + i] * xB * s_i;
    float njc = 0.f;
    if (i >= 1) {
      const float s_p = ldexpf(invZ, reinterpret_cast<const int *>(spec)[SP_S * SP + i - 1] + eb - ef_L);
      njc = spec[SP_N * SP + i - 1] * xN;
      njc = fmaf(spec[SP_J * SP + i - 1], xJ, njc);
      njc = fmaf(spec[SP_C * SP + i - 1], xC, njc);
      njc = njc * cm.loop * s_p;
    }
    // every wave holds the same values; the row's Forward states (read above by all) are overwritten by ONE lane after a
    // barrier - the next row's barriers order the write before anybody reads row i-1's neighbours (rows are disjoint)
    wg_barrier();
    if (c.w == 0 && c.lane == 0) { spec[SP_E * SP + i] = pe; spec[SP_B * SP + i] = pb; spec[SP_N * SP + i] = njc; }
  }
  wg_barrier();
}

// ------------------------------------------------------------------------------------------ P4: unihit Backward + posteriors -> null2
// Returns domcorr (every wave holds it) and, in <mass_out>, the posterior mass that reached the accumulators: with the sparse
// spill it certifies the result (it must reach Ld (1 - tol), wh_score7.hip); when it does not, the function returns early
// and the caller repeats the envelope with every row stored.
template <int Q, int NLT, bool TR = false>
__device__ __forceinline__ float backward_null2_wide(const WCtxT<NLT> &c, const uint8_t *eseq, int Ld, LenCfg cu, float invZe, float *n2tab, uint32_t degen, float mass_tol, float &mass_out) {
  constexpr int Q4 = Q / 4;
  const int SP = c.SP, NL = c.nl(), w = c.w, lane = c.lane;
  const float *spec = c.spec;
  float *xch = c.xch;
  ScanC sc;
  float Aexcl;
  WaveScan scW;
  backward_prepare_wide<Q, NLT>(c, sc, Aexcl, scW);
  BackTab<Q, TR> tb;
  tb.load(c);
  float Mb[Q], Ib[Q], fM[Q];
#pragma unroll
  for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; fM[p] = 0.f; }
  float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f, xfac = 0.f, fIs = 0.f;
  int S_next = 0;
  // my reversed cells in the forward-ordered slab: forward lane NL-1-vl, piece Q4-1-p4, components reversed
  const int rv = NL - 1 - c.vl;
  // sparse spill: the word of the forward wave that holds my reversed cells (W-1-w), my cells are bit 63-lane; lane t < W
  // reads the word of wave t ONE ROW AHEAD (a per-lane address: a vector load, coherent with the stores of this workgroup)
  const bool sparse = c.masks != nullptr;
  unsigned long long mv_next = 0;
  if (sparse && lane < c.W) mv_next = __builtin_nontemporal_load(c.masks + (size_t)Ld * c.W + lane);
#pragma unroll 1
  for (int i = Ld; i >= 1; i--) {
    const int S_i = reinterpret_cast<const int *>(spec)[SP_S * SP + i];
    const int dS = S_i - reinterpret_cast<const int *>(spec)[SP_S * SP + i - 1];
    bool have = true;
    if (sparse) {
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mv_next & 0xFFFFFFFFull), c.W - 1 - w);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mv_next >> 32), c.W - 1 - w);
      const int bit = 63 - lane;
      have = ((bit < 32 ? lo >> bit : hi >> (bit - 32)) & 1u) != 0;
      if (i > 1 && lane < c.W) mv_next = __builtin_nontemporal_load(c.masks + (size_t)(i - 1) * c.W + lane);
    }
    // TR (one workgroup per CU: nothing else covers an HBM round trip): the row's stored Forward cells are requested
    // before the row's arithmetic and barriers, not where they are used
    constexpr bool PF = TR && Q <= kWideQReg;            // (16-cell lanes: no registers left for a row in flight)
    float4 fmP[PF ? Q4 : 1], fiP[PF ? Q4 : 1];
    if (PF) {
      const gf4 *rowp = c.Fs + (size_t)i * (2 * Q4) * NL;
      const int nlp = opaque_s(NL);
      if (have) {
#pragma unroll
        for (int p4 = 0; p4 < Q4; p4++) { fmP[PF ? p4 : 0] = ldt(rowp, Q4 - 1 - p4, nlp, (unsigned)rv); fiP[PF ? p4 : 0] = ldt(rowp, Q4 + Q4 - 1 - p4, nlp, (unsigned)rv); }
      } else {
#pragma unroll
        for (int p4 = 0; p4 < Q4; p4++) { fmP[PF ? p4 : 0] = make_float4(0.f, 0.f, 0.f, 0.f); fiP[PF ? p4 : 0] = make_float4(0.f, 0.f, 0.f, 0.f); }
      }
    }
    float gfront = 0.f;
    if (i < Ld) {
      mirror_scale<Q>(S_next - S_i, Mb, Ib, xJ, xC, xN);
      xB = backward_emit_wide<Q, NLT, TR>(c, tb, __builtin_amdgcn_readfirstlane((int)eseq[i]), Mb, gfront);
      xJ = fmaf(xJ, cu.loop, xB * cu.move);
      xC = xC * cu.loop;
      xN = fmaf(xN, cu.loop, xB * cu.move);
    }
    const float xE = fmaf(xC, cu.EC, xJ * cu.EJ);
    backward_cells_wide<Q, NLT, TR>(c, tb, sc, scW, Aexcl, Mb, Ib, xE, gfront);
    clamp_backward<Q>(Mb, Ib, xB, xJ, xC, xN);
    const float s_i = invZe;
    const float s_p = ldexpf(invZe, -dS);
    const gf4 *row = c.Fs + (size_t)i * (2 * Q4) * NL;
    const int nlr = opaque_s(NL);
    if (have) {
      float idot = 0.f;
#pragma unroll
      for (int p4 = 0; p4 < Q4; p4++) {
        const float4 fm = PF ? fmP[PF ? p4 : 0] : ldt(row, Q4 - 1 - p4, nlr, (unsigned)rv);
        const float4 fi = PF ? fiP[PF ? p4 : 0] : ldt(row, Q4 + Q4 - 1 - p4, nlr, (unsigned)rv);
        fM[4 * p4 + 0] = fmaf(fm.w * Mb[4 * p4 + 0], s_i, fM[4 * p4 + 0]);
        fM[4 * p4 + 1] = fmaf(fm.z * Mb[4 * p4 + 1], s_i, fM[4 * p4 + 1]);
        fM[4 * p4 + 2] = fmaf(fm.y * Mb[4 * p4 + 2], s_i, fM[4 * p4 + 2]);
        fM[4 * p4 + 3] = fmaf(fm.x * Mb[4 * p4 + 3], s_i, fM[4 * p4 + 3]);
        idot = fmaf(fi.w, Ib[4 * p4 + 0], idot); idot = fmaf(fi.z, Ib[4 * p4 + 1], idot);
        idot = fmaf(fi.y, Ib[4 * p4 + 2], idot); idot = fmaf(fi.x, Ib[4 * p4 + 3], idot);
      }
      fIs = fmaf(idot, s_i, fIs);
    }
    float nj = spec[SP_N * SP + i - 1] * xN;
    nj = fmaf(spec[SP_J * SP + i - 1], xJ, nj);
    nj = fmaf(spec[SP_C * SP + i - 1], xC, nj);
    S_next = S_i;
    xfac = fmaf(nj * cu.loop, s_p, xfac);
  }
  // null2[a] = (sum_k fM_k o_k(a) + sum_k fI_k + f_NJC) / Ld over ALL waves, summed in wave order
  const float siw = wave_sum(fIs);
  float smw = 0.f;
#pragma unroll
  for (int p = 0; p < Q; p++) smw += fM[p];
  smw = wave_sum(smw);
  wg_barrier();
  if (lane == 0) { xch[w * X_N + X_T1] = siw; xch[w * X_N + X_T0] = smw; }
  wg_barrier();
  float si = 0.f, sm = 0.f;
  for (int v = 0; v < c.W; v++) { si += xch[v * X_N + X_T1]; sm += xch[v * X_N + X_T0]; }
  const float mass = sm + si + xfac;
  mass_out = mass;
  if (!(fabsf((float)Ld - mass) <= mass_tol * (float)Ld)) { wg_barrier(); return 0.f; }   // (every wave holds the same sums: a uniform exit)
  const float norm = 1.0f / (float)Ld;
  for (int x = 0; x < c.K; x++) {
    float s = 0.f;
#pragma unroll
    for (int p4 = 0; p4 < Q4; p4++) {
      const float4 O = ldg4(c.em, (unsigned)((x * Q4 + (Q4 - 1 - p4)) * NL + rv));
      s = fmaf(fM[4 * p4 + 0], O.w, s); s = fmaf(fM[4 * p4 + 1], O.z, s);
      s = fmaf(fM[4 * p4 + 2], O.y, s); s = fmaf(fM[4 * p4 + 3], O.x, s);
    }
    const float sw = wave_sum(s);
    wg_barrier();
    if (lane == 0) xch[w * X_N + X_T1] = sw;
    wg_barrier();
    float tot = 0.f;
    for (int v = 0; v < c.W; v++) tot += xch[v * X_N + X_T1];
    if (w == 0 && lane == 0) n2tab[x] = (tot + si) * norm + xfac * norm;
  }
  wg_barrier();
  if (w == 0) {
    float mine = 1.0f;
    if (lane >= c.K && lane < c.Kp) {
      float s = 0.f; int n = 0;
      for (int x = 0; x < c.K; x++) if (degen & (1u << x)) { s += n2tab[x]; n++; }
      mine = n > 0 ? s / (float)n : 1.0f;
    } else if (lane < c.K) mine = n2tab[lane];
    __builtin_amdgcn_wave_barrier();
    if (lane < c.Kp) n2tab[lane] = logf(mine);
    __builtin_amdgcn_wave_barrier();
    float dc = 0.f;
    for (int t = lane; t < Ld; t += kWave) dc += n2tab[eseq[t]];
    const float domcorr = wave_sum(dc);
    if (lane == 0) xch[X_T1] = domcorr;
  }
  wg_barrier();
  const float out = xch[X_T1];
  wg_barrier();
  return out;
}

// ------------------------------------------------------------------------------------------ the kernel
template <int Q, int NLT, bool TR>
__global__ __launch_bounds__(512) void score_wide_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
  const int NL = NLT > 0 ? NLT : W * 64, SP = a.SP, vl = w * 64 + lane;
  constexpr int Q4 = Q / 4;
  volatile int *s_item = reinterpret_cast<volatile int *>(smem);
  float *spec = smem + 4;                               // 6 arrays of SP floats
  float *xch = spec + 6 * SP;                           // W x X_N
  float *n2tab = xch + 8 * X_N;
  int *regs = reinterpret_cast<int *>(n2tab + 32);      // kRegsInts: regions, spare, envelope results
  uint8_t *seq = reinterpret_cast<uint8_t *>(regs + kRegsInts);
  // TR: behind the residues (16-byte aligned), the emission rows of the canonical residues of the CURRENT model
  wv4 *emL = reinterpret_cast<wv4 *>(smem + wide_em_lds_offset_floats(a.Lcap));
  int em_h = -1;
  const double LOG2 = 0.69314718055994529;
  WCtxT<NLT> c;
  c.spec = spec; c.xch = xch; c.NLr = NL; c.SP = SP; c.w = w; c.W = W; c.lane = lane; c.K = a.K; c.Kp = a.Kp;
  c.emL = (TR && a.em_lds) ? (const lf4 *)emL : nullptr;
  c.Fs = const_cast<gf4 *>(uniform_global(a.scratch + (size_t)blockIdx.x * a.scratch_stride));
  // the per-row lane masks of the sparse spill sit behind the rows of the slab
  c.masks = a.sparse ? reinterpret_cast<unsigned long long *>(a.scratch + (size_t)blockIdx.x * a.scratch_stride + (size_t)(a.Lcap + 1) * 2 * Q * NL) : nullptr;
  c.vl = vl;
  c.rowstat = a.stats ? a.stats + 8 : nullptr;
  uint32_t degen = 0;
  for (int t = 0; t < 32; t++) if (t == lane) degen = a.degen[t];
  long long cyc[5] = {0, 0, 0, 0, 0};
  const long long t_kernel = a.stats ? __builtin_readcyclecounter() : 0;

  for (;;) {
    if (threadIdx.x == 0) *s_item = atomicAdd(a.counter, 1);
    __syncthreads();
    const long long item = *s_item;
    __syncthreads();
    if (item >= (long long)a.n_list * a.nq) break;
    const int h = a.hmm_list[item / a.nq];
    const int64_t qi = a.qorder ? (int64_t)a.qorder[item % a.nq] : item % a.nq;
    const DevHMM *hm = a.hmms + h;
    c.fw = uniform_global(a.tables + hm->wfw_off);
    c.bw = uniform_global(a.tables + hm->wbw_off);
    c.em = uniform_global(a.tables + hm->wem_off);
    if (TR && a.em_lds && h != em_h) {                  // (work items are model-major: once per model and workgroup)
      const gf4 *src = c.em;
      for (int t = threadIdx.x; t < a.K * Q4 * NL; t += blockDim.x) emL[t] = src[t];
      em_h = h;
      __syncthreads();
    }
    const int64_t off = a.offsets[qi];
    const int L = (int)(a.offsets[qi + 1] - off);
    const size_t out = (size_t)qi * a.H + h;
    int flags = 0, decibits = 0;
    float fwd_bits_out = -INFINITY;
    wh_pair_detail *dp = (a.detail && threadIdx.x == 0) ? a.detail + out : nullptr;
    if (dp) { dp->fwd_bits = -INFINITY; dp->seq_score = 0.f; dp->pre_score = 0.f; dp->seqbias_nats = 0.f; dp->nregions = 0; dp->nenv = 0; }
    if (L > 0 && L <= a.Lcap) {
      for (int t = threadIdx.x; t < L; t += blockDim.x) { const int r = a.residues[off + t]; seq[t] = (uint8_t)(r < a.Kp ? r : a.Kp - 1); }
      __syncthreads();
      // ---------------- P1
      const LenCfg cm = len_config(L, true);
      float xC1; int ef1;
      long long tq = a.stats ? __builtin_readcyclecounter() : 0;
      auto lap = [&](int slot) { if (a.stats) { const long long t = __builtin_readcyclecounter(); cyc[slot] += t - tq; tq = t; } };
      forward_wide<Q, false, NLT, TR>(c, seq, L, cm, xC1, ef1);
      lap(0);
      const double fwd_nats = (double)ef1 * LOG2 + log((double)(xC1 * cm.move));
      const float fwdsc = (float)fwd_nats;
      const float p1 = (float)L / (float)(L + 1);
      const float nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
      fwd_bits_out = (float)((fwd_nats - (double)nullsc) / LOG2);
      if (dp) dp->fwd_bits = fwd_bits_out;
      if (xC1 > 0.f && isfinite(fwdsc)) {
        // ---------------- P2 + region scan (wave 0; the others wait)
        backward_decode_wide<Q, NLT, TR>(c, seq, L, cm, 1.0f / (xC1 * cm.move), ef1);
        lap(1);
        if (w == 0) {
          const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
          int nenv = 0, nreg = 0, fl = 0, i0 = -1;
          float btot = 0.f, etot = 0.f;
          bool trig = false;
          if (lane == 0) { spec[SP_J * SP] = 0.f; spec[SP_C * SP] = 0.f; }
          for (int j = 1; j <= L; j++) {
            const float mocc = 1.0f - spec[SP_N * SP + j];
            const float bold = btot, eold = etot;
            btot += spec[SP_B * SP + j - 1];
            etot += spec[SP_E * SP + j];
            if (lane == 0) { spec[SP_J * SP + j] = btot; spec[SP_C * SP + j] = etot; }
            if (!trig) {
              if (mocc - (btot - bold) < rt2) i0 = j;
              else if (i0 == -1) i0 = j;
              if (mocc >= rt1) trig = true;
            } else if (mocc - (etot - eold) < rt2) {
              if (nenv < WH_MAX_ENVELOPES) { if (lane == 0) { regs[2 * nenv] = i0; regs[2 * nenv + 1] = j; } nenv++; }
              else fl |= WH_FLAG_TRUNC;
              nreg++; i0 = -1; trig = false;
            }
          }
          __builtin_amdgcn_wave_barrier();
          int multi_mask = 0;
          for (int e = 0; e < nenv; e++) {
            const int ri = regs[2 * e], rj = regs[2 * e + 1];
            float mx = -1.0f;
            const float e0 = spec[SP_C * SP + ri - 1], bj = spec[SP_J * SP + rj];
            for (int z = ri + lane; z <= rj; z += kWave) {
              const float u = spec[SP_C * SP + z] - e0, v = bj - spec[SP_J * SP + z - 1];
              mx = fmaxf(mx, fminf(u, v));
            }
            mx = wave_max(mx);
            if (mx >= rt3) { fl |= WH_FLAG_MULTI; multi_mask |= 1 << e; }
          }
          if (lane == 0) { regs[2 * WH_MAX_ENVELOPES] = nenv; regs[2 * WH_MAX_ENVELOPES + 1] = nreg; regs[2 * WH_MAX_ENVELOPES + 2] = fl; regs[2 * WH_MAX_ENVELOPES + 3] = multi_mask; }
        }
        __syncthreads();
        lap(2);
        const int nenv = regs[2 * WH_MAX_ENVELOPES], nreg = regs[2 * WH_MAX_ENVELOPES + 1], multi_mask = regs[2 * WH_MAX_ENVELOPES + 3];
        flags |= regs[2 * WH_MAX_ENVELOPES + 2];
        if (dp) { dp->nregions = nreg; dp->nenv = nenv; }
        if (nenv > 0) {
          const LenCfg cu = len_config(L, false);
          float seqbias_sum = 0.f, sum_score = 0.f, sb2 = 0.f;
          int Ld_tot = 0;
          const bool queue_pair = multi_mask != 0 && a.rrecs != nullptr;
          float *envres = reinterpret_cast<float *>(regs + 3 * WH_MAX_ENVELOPES);
          for (int e = 0; e < nenv; e++) {
            if (queue_pair && ((multi_mask >> e) & 1)) { if (threadIdx.x == 0) { envres[e] = 0.f; envres[WH_MAX_ENVELOPES + e] = 0.f; } continue; }
            const int ri = regs[2 * e], rj = regs[2 * e + 1];
            const int Ld = rj - ri + 1;
            const uint8_t *eseq = seq + (ri - 1);
            float xC3 = 0.f, envsc = -INFINITY, domcorr = 0.f; int ef3 = 0;
#pragma unroll 1
            for (int attempt = 0; attempt < 2; attempt++) {
              const bool dense = attempt == 1 || !a.sparse;
              forward_wide<Q, true, NLT, TR>(c, eseq, Ld, cu, xC3, ef3, dense ? -1.0f : kWideKeepScale);
              __threadfence_block();
              lap(3);
              envsc = (float)((double)ef3 * LOG2 + log((double)(xC3 * cu.move)));
              domcorr = 0.f;
              if (!(xC3 > 0.f)) break;
              float mass = 0.f;
              domcorr = backward_null2_wide<Q, NLT, TR>(c, eseq, Ld, cu, 1.0f / (xC3 * cu.move), n2tab, degen, dense ? INFINITY : kWideMassTol, mass);
              lap(4);
              if (!dense && !(fabsf((float)Ld - mass) <= kWideMassTol * (float)Ld)) continue;     // the sparse rows lost mass: once more with every row
              if (attempt == 1) flags |= WH_FLAG_EXACT;
              break;
            }
            seqbias_sum += domcorr;
            if (envsc - domcorr > 0.0f) { sum_score += envsc; Ld_tot += Ld; sb2 += domcorr; }
            if (dp) { dp->env_i[e] = ri; dp->env_j[e] = rj; dp->envsc[e] = envsc; dp->domcorr[e] = domcorr; }
            if (queue_pair && threadIdx.x == 0) { envres[e] = envsc; envres[WH_MAX_ENVELOPES + e] = domcorr; }
          }
          if (queue_pair) {
            __syncthreads();
            if (threadIdx.x == 0) {
              const int slot = atomicAdd(a.rcount, 1);
              if (slot < a.rcap) {
                ResolveRec *rr = a.rrecs + slot;
                rr->q = qi; rr->h = h; rr->fwdsc = fwdsc; rr->fwd_bits = fwd_bits_out; rr->nreg = nreg; rr->nenv = nenv;
                rr->multi_mask = multi_mask; rr->flags = flags;
                for (int e = 0; e < nenv; e++) { rr->ri[e] = regs[2 * e]; rr->rj[e] = regs[2 * e + 1]; rr->envsc[e] = envres[e]; rr->domcorr[e] = envres[WH_MAX_ENVELOPES + e]; }
              }
            }
          } else {
            // ---------------- A.6 score assembly (float32 where HMMER is float32), as in wh_score7.hip
            const float lomega = (float)log(1.0 / 256.0);
            const float seqbias = flogsum0_w(lomega + seqbias_sum);
            float pre_score = (float)(((double)fwdsc - (double)nullsc) / LOG2);
            float seq_score = (float)(((double)fwdsc - (double)(nullsc + seqbias)) / LOG2);
            sb2 = flogsum0_w(lomega + sb2);
            sum_score += (float)((double)(L - Ld_tot) * log((double)((float)L / (float)(L + 3))));
            const float pre2 = (float)(((double)sum_score - (double)nullsc) / LOG2);
            sum_score = (float)(((double)sum_score - (double)(nullsc + sb2)) / LOG2);
            if (Ld_tot > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2; flags |= WH_FLAG_OVERRIDE; }
            decibits = (int)rint((double)seq_score * 10.0);
            flags |= WH_FLAG_REPORTED;
            if (dp) { dp->seq_score = seq_score; dp->pre_score = pre_score; dp->seqbias_nats = seqbias; }
          }
        }
      }
    }
    if (threadIdx.x == 0) {
      a.decibits[out] = decibits;
      a.flags[out] = (uint8_t)flags;
      if (a.fwd_bits) a.fwd_bits[out] = fwd_bits_out;
    }
    __syncthreads();
  }
  if (a.stats && threadIdx.x == 0) {
    for (int t = 0; t < 5; t++) atomicAdd(a.stats + t, (unsigned long long)cyc[t]);
    atomicAdd(a.stats + 5, (unsigned long long)(__builtin_readcyclecounter() - t_kernel));
  }
}

// ------------------------------------------------------------------------------------------ alignment (A.7)
// "hmmalign" for the models this file serves: the chain of wh_align.hip (unihit Forward with dense rows -> Backward +
// posterior decoding in place -> optimal-accuracy fill -> traceback with HMMER's candidate orders and striped E-state
// scan) with the rows split over the workgroup's waves like the scoring sweeps above.  Pairs that leave float32 range
// (clamp_backward fires: hmmalign itself switches to its log-space code there) are handed to the float64 any-size
// kernel through <status>; so is nothing else - a pair either gets its columns here or is redone there.
enum { WA_PN = 0, WA_B, WA_E, WA_PJ, WA_PC, WA_S, WA_ON, WA_OB, WA_OE, WA_OJ, WA_OC, WA_NARR };

__device__ __forceinline__ float gate_w(float t, float v) { return t > 0.f ? v : 0.f; }
__device__ __forceinline__ float scan_apply_max_w(const ScanC &c, float B) {
  B = fmaxf(B, c.s[0] * dppf<0x111>(0.f, B));
  B = fmaxf(B, c.s[1] * dppf<0x112>(0.f, B));
  B = fmaxf(B, c.s[2] * dppf<0x114>(0.f, B));
  B = fmaxf(B, c.s[3] * dppf<0x118>(0.f, B));
  B = fmaxf(B, c.s[4] * dppf<0x142, 0xA>(0.f, B));
  B = fmaxf(B, c.s[5] * dppf<0x143, 0xC>(0.f, B));
  return B;
}
__device__ __forceinline__ int wave_max_i32_w(int x) { for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(x, m); x = o > x ? o : x; } return x; }

template <int Q, int NLT, bool TR>
__global__ __launch_bounds__(512) void align_wide_kernel(WideAlignArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
  const int NL = NLT > 0 ? NLT : W * 64, SP = a.SP, vl = w * 64 + lane;
  constexpr int Q4 = Q / 4;
  volatile int *s_item = reinterpret_cast<volatile int *>(smem);
  float *spec = smem + 4;                               // WA_NARR arrays of SP floats
  float *xch = spec + WA_NARR * SP;                     // W x X_N
  uint8_t *seq = reinterpret_cast<uint8_t *>(xch + 8 * X_N);
  WCtxT<NLT> c;
  c.spec = spec; c.xch = xch; c.NLr = NL; c.SP = SP; c.w = w; c.W = W; c.lane = lane; c.K = a.K; c.Kp = a.Kp; c.vl = vl;
  c.emL = nullptr;
  c.masks = nullptr;
  c.rowstat = nullptr;
  gf4 *slabA = const_cast<gf4 *>(uniform_global(a.scratch + (size_t)blockIdx.x * a.scratch_stride));      // F -> posteriors: [row][2][Q4][NL]
  gf4 *slabB = slabA + (size_t)(a.Lcap + 1) * 2 * Q4 * NL;                                                 // OA rows: [row][3][Q4][NL]
  c.Fs = slabA;
  const int rv = NL - 1 - vl;

  for (;;) {
    if (threadIdx.x == 0) *s_item = atomicAdd(a.counter, 1);
    __syncthreads();
    const int item = *s_item;
    __syncthreads();
    if (item >= a.n_items) break;
    const int pair = a.items[item];
    const int h = a.pair_h[pair];
    const int64_t qi = a.pair_q[pair];
    const DevHMM *hm = a.hmms + h;
    const int M = hm->M;
    c.fw = uniform_global(a.tables + hm->wfw_off);
    c.bw = uniform_global(a.tables + hm->wbw_off);
    c.em = uniform_global(a.tables + hm->wem_off);
    const float *fwG = a.tables + hm->wfw_off;
    const int64_t off = a.offsets[qi];
    const int L = (int)(a.offsets[qi + 1] - off);
    int32_t *cols = a.cols + a.col_off[pair];
    for (int t = threadIdx.x; t < L; t += blockDim.x) cols[t] = -1;
    bool active = L > 0 && L <= a.Lcap;
    for (int t = threadIdx.x; active && t < L; t += blockDim.x) { const int r = a.residues[off + t]; seq[t] = (uint8_t)(r < a.Kp ? r : a.Kp - 1); }
    __syncthreads();
    const LenCfg cu = len_config(L > 0 ? L : 1, false);
    // ---------------- unihit Forward, rows to slab A (spec slots 0..5 = N, B, E, J, C, S)
    float xC_L = 0.f; int ef_L = 0;
    if (active) forward_wide<Q, true, NLT, TR>(c, seq, L, cu, xC_L, ef_L);
    __threadfence_block();
    if (!(xC_L > 0.f)) active = false;
    bool clamped = false;
    // ---------------- Backward + posterior decoding, in place over slab A
    if (active) {
      const float invZ = 1.0f / (xC_L * cu.move);
      ScanC sc;
      float Aexcl;
      WaveScan scW;
  backward_prepare_wide<Q, NLT>(c, sc, Aexcl, scW);
      BackTab<Q, TR> tb0;
      tb0.load(c);
      float Mb[Q], Ib[Q];
#pragma unroll
      for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; }
      float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f;
#pragma unroll 1
      for (int i = L; i >= 1; i--) {
        gf4 *row = slabA + (size_t)i * (2 * Q4) * NL;
        const int nlr = opaque_s(NL);
        float4 fm4[Q4], fi4[Q4];
#pragma unroll
        for (int p4 = 0; p4 < Q4; p4++) { fm4[p4] = ldt(row, Q4 - 1 - p4, nlr, (unsigned)rv); fi4[p4] = ldt(row, Q4 + Q4 - 1 - p4, nlr, (unsigned)rv); }
        float gfront = 0.f;
        if (i < L) {
          mirror_scale<Q>(reinterpret_cast<const int *>(spec)[WA_S * SP + i + 1] - reinterpret_cast<const int *>(spec)[WA_S * SP + i], Mb, Ib, xJ, xC, xN);
          xB = backward_emit_wide<Q, NLT, TR>(c, tb0, __builtin_amdgcn_readfirstlane((int)seq[i]), Mb, gfront);
          xJ = fmaf(xJ, cu.loop, xB * cu.move);
          xC = xC * cu.loop;
          xN = fmaf(xN, cu.loop, xB * cu.move);
        }
        const float xE = fmaf(xC, cu.EC, xJ * cu.EJ);
        backward_cells_wide<Q, NLT, TR>(c, tb0, sc, scW, Aexcl, Mb, Ib, xE, gfront);
        clamped |= clamp_backward<Q>(Mb, Ib, xB, xJ, xC, xN);
        const float s_i = invZ;
        const float s_p = ldexpf(invZ, reinterpret_cast<const int *>(spec)[WA_S * SP + i - 1] - reinterpret_cast<const int *>(spec)[WA_S * SP + i]);
#pragma unroll
        for (int p4 = 0; p4 < Q4; p4++) {
          // position 4*p4+j (reversed order) is component 3-j of the forward-ordered vector
          const wv4 pm = {(fm4[p4].x * Mb[4 * p4 + 3]) * s_i, (fm4[p4].y * Mb[4 * p4 + 2]) * s_i, (fm4[p4].z * Mb[4 * p4 + 1]) * s_i, (fm4[p4].w * Mb[4 * p4 + 0]) * s_i};
          const wv4 pi = {(fi4[p4].x * Ib[4 * p4 + 3]) * s_i, (fi4[p4].y * Ib[4 * p4 + 2]) * s_i, (fi4[p4].z * Ib[4 * p4 + 1]) * s_i, (fi4[p4].w * Ib[4 * p4 + 0]) * s_i};
          row[(unsigned)((Q4 - 1 - p4) * NL + rv)] = pm;
          row[(unsigned)((Q4 + Q4 - 1 - p4) * NL + rv)] = pi;
        }
        const float pn = spec[WA_PN * SP + i - 1] * xN * cu.loop * s_p;
        const float pj = spec[WA_PJ * SP + i - 1] * xJ * cu.loop * s_p;
        const float pc = spec[WA_PC * SP + i - 1] * xC * cu.loop * s_p;
        if (w == 0 && lane == 0) { spec[WA_PN * SP + i] = pn; spec[WA_PJ * SP + i] = pj; spec[WA_PC * SP + i] = pc; }
      }
    }
    __threadfence_block();
    __syncthreads();
    if (clamped) active = false;                        // float32 range left: the float64 kernel redoes the pair
    if (clamped && threadIdx.x == 0) a.status[pair] = 1;
    // ---------------- optimal-accuracy fill, rows to slab B
    const float tNl = cu.loop > 0.f ? 1.f : 0.f, tNm = cu.move > 0.f ? 1.f : 0.f;
    const float tEJ = cu.EJ > 0.f ? 1.f : 0.f, tEC = cu.EC > 0.f ? 1.f : 0.f;
    if (active) {
      int nlv = NL;
      float4 tf[TR ? FW_NARR : 1][TR ? Q4 : 1];
      if (TR) {
#pragma unroll
        for (int arr = 0; arr < FW_NARR; arr++)
#pragma unroll
          for (int q4 = 0; q4 < Q4; q4++) tf[TR ? arr : 0][TR ? q4 : 0] = ldg4(c.fw, (unsigned)((arr * Q4 + q4) * NL + vl));
      }
      auto T = [&](int arr, int q4) -> float4 { if constexpr (TR) return tf[arr][q4]; else return ldt(c.fw, arr * Q4 + q4, nlv, (unsigned)vl); };
      float allpass = 1.f;
#pragma unroll
      for (int q4 = 0; q4 < Q4; q4++) { const float4 d = T(FW_D2, q4); if (!(d.x > 0.f && d.y > 0.f && d.z > 0.f && d.w > 0.f)) allpass = 0.f; }
      const ScanC sc = scan_prepare(allpass);
      const float Gincl = lane_prefix_product(allpass, lane);
      float Gexcl = __shfl_up(Gincl, 1);
      if (lane == 0) Gexcl = 1.f;
      if (lane == 63) xch[w * X_N + X_AT] = Gincl;
      float Mp[Q], Ip[Q], Dp[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) { Mp[q] = -INFINITY; Ip[q] = -INFINITY; Dp[q] = -INFINITY; }
      float oN = 0.f, oB = 0.f, oJ = -INFINITY, oC = -INFINITY;
      float bM = -INFINITY, bI = -INFINITY, bD = -INFINITY;     // previous row, last cell of the wave in front
      if (w == 0 && lane == 0) {
        spec[WA_ON * SP] = 0.f; spec[WA_OB * SP] = 0.f; spec[WA_OE * SP] = -INFINITY; spec[WA_OJ * SP] = -INFINITY; spec[WA_OC * SP] = -INFINITY;
      }
      __syncthreads();
#pragma unroll 1
      for (int i = 1; i <= L; i++) {
        nlv = opaque_s(NL);
        const gf4 *prow = slabA + (size_t)i * (2 * Q4) * NL;
        float mm1 = wave_shr1(Mp[Q - 1]), im1 = wave_shr1(Ip[Q - 1]), dm1 = wave_shr1(Dp[Q - 1]);
        // (wave_shr1 hands lane 0 a ZERO: the one-wave kernel's first lane sees 0 there too, not -inf)
        if (lane == 0 && w > 0) { mm1 = bM; im1 = bI; dm1 = bD; }
#pragma unroll
        for (int q4 = Q4 - 1; q4 >= 0; q4--) {
          const float4 A = T(FW_A, q4), B = T(FW_B, q4), C = T(FW_C, q4), E = T(FW_E, q4);
          const float4 MI = T(FW_MI, q4), II = T(FW_II, q4);
          const float4 pm4 = ldt(prow, q4, nlv, (unsigned)vl), pi4 = ldt(prow, Q4 + q4, nlv, (unsigned)vl);
#pragma unroll
          for (int j = 3; j >= 0; j--) {
            const int q = 4 * q4 + j;
            const float pm = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mm1;
            const float pi_ = q > 0 ? Ip[q > 0 ? q - 1 : 0] : im1;
            const float pd = q > 0 ? Dp[q > 0 ? q - 1 : 0] : dm1;
            float sv = gate_w(f4get(E, j), oB);
            sv = fmaxf(sv, gate_w(f4get(A, j), pm));
            sv = fmaxf(sv, gate_w(f4get(B, j), pi_));
            sv = fmaxf(sv, gate_w(f4get(C, j), pd));
            const float ni = fmaxf(gate_w(f4get(MI, j), Mp[q]), gate_w(f4get(II, j), Ip[q])) + f4get(pi4, j);
            Mp[q] = sv + f4get(pm4, j);
            Ip[q] = ni;
          }
        }
        if (lane == 63) xch[w * X_N + X_T0] = Mp[Q - 1];
        __syncthreads();                                                          // #0
        float mn1 = wave_shr1(Mp[Q - 1]);
        if (lane == 0 && w > 0) mn1 = xch[(w - 1) * X_N + X_T0];
        float dprev = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < Q4; q4++) {
          const float4 D1 = T(FW_D1, q4), D2 = T(FW_D2, q4);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int q = 4 * q4 + j;
            const float src = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mn1;
            dprev = fmaxf(gate_w(f4get(D1, j), src), gate_w(f4get(D2, j), dprev));
            Dp[q] = dprev;
          }
        }
        const float loc = scan_apply_max_w(sc, dprev);
        if (lane == 63) xch[w * X_N + X_BD] = loc;
        __syncthreads();                                                          // #1
        float cin = 0.f;                               // D of the last cell in front of my wave (0 in front of the first wave, as wave_shr1 gives)
        for (int v = 0; v < w; v++) cin = fmaxf(xch[v * X_N + X_BD], xch[v * X_N + X_AT] * cin);
        float carry = wave_shr1(loc);
        if (w > 0) carry = lane == 0 ? cin : fmaxf(carry, Gexcl * cin);
        float rowmax = -INFINITY;
#pragma unroll
        for (int q4 = 0; q4 < Q4; q4++) {
          const float4 D2 = T(FW_D2, q4);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int q = 4 * q4 + j;
            carry = gate_w(f4get(D2, j), carry);
            Dp[q] = fmaxf(Dp[q], carry);
            if (vl * Q + q < M) rowmax = fmaxf(rowmax, fmaxf(Mp[q], Dp[q]));
          }
        }
        const float rmw = wave_max(rowmax);
        if (lane == 63) { xch[w * X_N + X_ES] = rmw; xch[w * X_N + X_BM] = Mp[Q - 1]; xch[w * X_N + X_BI] = Ip[Q - 1]; xch[w * X_N + X_BDD] = Dp[Q - 1]; }
        __syncthreads();                                                          // #2
        float xE = -INFINITY;
        for (int v = 0; v < W; v++) xE = fmaxf(xE, xch[v * X_N + X_ES]);
        if (w > 0) { bM = xch[(w - 1) * X_N + X_BM]; bI = xch[(w - 1) * X_N + X_BI]; bD = xch[(w - 1) * X_N + X_BDD]; }
        {
          const float a1 = tNl * (oJ + spec[WA_PJ * SP + i]), b1 = tEJ * xE;
          oJ = a1 > b1 ? a1 : b1;
          const float a2 = tNl * (oC + spec[WA_PC * SP + i]), b2 = tEC * xE;
          oC = a2 > b2 ? a2 : b2;
          oN = tNl * (oN + spec[WA_PN * SP + i]);
          const float a3 = tNm * oN, b3 = tNm * oJ;
          oB = a3 > b3 ? a3 : b3;
        }
        if (w == 0 && lane == 0) {
          spec[WA_ON * SP + i] = oN; spec[WA_OB * SP + i] = oB; spec[WA_OE * SP + i] = xE; spec[WA_OJ * SP + i] = oJ; spec[WA_OC * SP + i] = oC;
        }
        gf4 *orow = slabB + (size_t)i * (3 * Q4) * NL;
#pragma unroll
        for (int q4 = 0; q4 < Q4; q4++) {
          const wv4 vm = {Mp[4 * q4], Mp[4 * q4 + 1], Mp[4 * q4 + 2], Mp[4 * q4 + 3]};
          const wv4 vi = {Ip[4 * q4], Ip[4 * q4 + 1], Ip[4 * q4 + 2], Ip[4 * q4 + 3]};
          const wv4 vd = {Dp[4 * q4], Dp[4 * q4 + 1], Dp[4 * q4 + 2], Dp[4 * q4 + 3]};
          orow[(unsigned)(q4 * NL + vl)] = vm;
          orow[(unsigned)((Q4 + q4) * NL + vl)] = vi;
          orow[(unsigned)((2 * Q4 + q4) * NL + vl)] = vd;
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---------------- traceback by the first wave: first maximum wins, candidate orders as in SURVEY.md A.7
    if (active && w == 0) {
      const float *sB = reinterpret_cast<const float *>((const void *)slabB);
      auto cellB = [&](int row, int st, int k) -> float {      // OA cell (row, state, node k >= 1)
        const int pos = k - 1, ln = pos / Q, q = pos % Q;
        return __builtin_nontemporal_load(sB + (((size_t)(row * 3 + st) * Q4 + q / 4) * NL + ln) * 4 + (q % 4));
      };
      auto tabF = [&](int arr, int k) -> float {
        const int pos = k - 1, ln = pos / Q, q = pos % Q;
        return fwG[(((size_t)arr * Q4 + q / 4) * NL + ln) * 4 + (q % 4)];
      };
      enum { stS, stN, stB, stM, stI, stD, stE, stJ, stC };
      int s0 = stC, s1 = stS, i = L, k = 0;
      int guard = 4 * (L + M) + 16;
      const int Qh = (M - 1) / 4 + 1 < 2 ? 2 : (M - 1) / 4 + 1;   // HMMER's SSE stripe count
      while (s0 != stS && guard-- > 0) {
        switch (s0) {
          case stC: {
            const float av = tNl * (spec[WA_OC * SP + i - 1] + spec[WA_PC * SP + i]), bv = tEC * spec[WA_OE * SP + i];
            s1 = bv > av ? stE : stC;
            break;
          }
          case stJ: {
            const float av = tNl * (spec[WA_OJ * SP + i - 1] + spec[WA_PJ * SP + i]), bv = tEJ * spec[WA_OE * SP + i];
            s1 = bv > av ? stE : stJ;
            break;
          }
          case stE: {
            // argmax over M (">=": the later cell in HMMER's striped scan wins) and D (">"), over the slices of all waves
            const float vmax = spec[WA_OE * SP + i];        // the row maximum the fill recorded (same comparisons, same values)
            int bestM = -1, bestD = -1;
            for (int v = 0; v < W; v++) {
              const gf4 *orow = slabB + (size_t)i * (3 * Q4) * NL;
#pragma unroll
              for (int q4 = 0; q4 < Q4; q4++) {
                const float4 m4 = ldg4(orow, (unsigned)(q4 * NL + v * 64 + lane)), d4 = ldg4(orow, (unsigned)((2 * Q4 + q4) * NL + v * 64 + lane));
#pragma unroll
                for (int j = 0; j < 4; j++) {
                  const int kk = (v * 64 + lane) * Q + 4 * q4 + j + 1;
                  if (kk <= M) {
                    const int qh = (kk - 1) % Qh, rh = (kk - 1) / Qh;
                    if (f4get(m4, j) == vmax) { const int pos = qh * 8 + rh; bestM = pos > bestM ? pos : bestM; }
                    if (f4get(d4, j) == vmax) { const int pos = 0x3FFFFFFF - (qh * 8 + 4 + rh); bestD = pos > bestD ? pos : bestD; }
                  }
                }
              }
            }
            bestM = wave_max_i32_w(bestM);
            bestD = wave_max_i32_w(bestD);
            int pos;
            if (bestM >= 0) { pos = bestM; s1 = stM; }
            else { pos = 0x3FFFFFFF - bestD; s1 = stD; }
            k = (pos % 8 % 4) * Qh + pos / 8 + 1;
            break;
          }
          case stM: {
            float path[4];
            path[0] = gate_w(tabF(FW_E, k), spec[WA_OB * SP + i - 1]);
            if (i > 1 && k > 1) {
              path[1] = gate_w(tabF(FW_A, k), cellB(i - 1, 0, k - 1));
              path[2] = gate_w(tabF(FW_B, k), cellB(i - 1, 1, k - 1));
              path[3] = gate_w(tabF(FW_C, k), cellB(i - 1, 2, k - 1));
            } else if (k > 1) {
              path[1] = gate_w(tabF(FW_A, k), -INFINITY);
              path[2] = gate_w(tabF(FW_B, k), -INFINITY);
              path[3] = gate_w(tabF(FW_C, k), -INFINITY);
            } else { path[1] = 0.f; path[2] = 0.f; path[3] = 0.f; }
            int best = 0;
            if (path[1] > path[best]) best = 1;
            if (path[2] > path[best]) best = 2;
            if (path[3] > path[best]) best = 3;
            s1 = best == 0 ? stB : best == 1 ? stM : best == 2 ? stI : stD;
            if (lane == 0) cols[i - 1] = k - 1;
            k--; i--;
            break;
          }
          case stD: {
            const float av = k > 1 ? gate_w(tabF(FW_D1, k), cellB(i, 0, k - 1)) : 0.f;
            const float bv = k > 1 ? gate_w(tabF(FW_D2, k), cellB(i, 2, k - 1)) : 0.f;
            s1 = bv > av ? stD : stM;
            k--;
            break;
          }
          case stI: {
            const float pmv = i > 1 ? cellB(i - 1, 0, k) : -INFINITY;
            const float piv = i > 1 ? cellB(i - 1, 1, k) : -INFINITY;
            const float av = gate_w(tabF(FW_MI, k), pmv), bv = gate_w(tabF(FW_II, k), piv);
            s1 = bv > av ? stI : stM;
            i--;
            break;
          }
          case stB: {
            const float av = tNm * spec[WA_ON * SP + i], bv = tNm * spec[WA_OJ * SP + i];
            s1 = bv > av ? stJ : stN;
            break;
          }
          case stN: s1 = i == 0 ? stS : stN; break;
          default: s1 = stS; break;
        }
        if ((s1 == stN || s1 == stJ || s1 == stC) && s1 == s0) i--;
        if (i < 0 || k < 0 || (s1 == stM && (k < 1 || i < 1)) || ((s1 == stC || s1 == stJ) && i < 1)) break;   // defensive
        s0 = s1;
      }
    }
    __syncthreads();
  }
}

}  // namespace wide

size_t wide_align_lds_bytes(int Lcap) {
  const int SP = (Lcap + 1 + 3) / 4 * 4;
  return (size_t)(4 + wide::WA_NARR * SP + 8 * wide::X_N) * sizeof(float) + (size_t)(Lcap + 16);
}

template <int Q, int NLT, bool TR = false>
static hipError_t launch_walign_t(const WideAlignArgs &a, int blocks, int waves, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&wide::align_wide_kernel<Q, NLT, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((wide::align_wide_kernel<Q, NLT, TR>), dim3(blocks), dim3(waves * 64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_align_wide(int Q, const WideAlignArgs &a, int blocks, int waves, size_t lds, hipStream_t s) {
  if (waves < 1 || waves > kWideWavesMax) return hipErrorInvalidValue;
  if (Q == 4) return launch_walign_t<4, 0>(a, blocks, waves, lds, s);
  if (Q == kWideQReg) {
    switch (waves) {
      case 5: return launch_walign_t<kWideQReg, 320, true>(a, blocks, waves, lds, s);
      case 6: return launch_walign_t<kWideQReg, 384, true>(a, blocks, waves, lds, s);
      case 7: return launch_walign_t<kWideQReg, 448, true>(a, blocks, waves, lds, s);
      case 8: return launch_walign_t<kWideQReg, 512, true>(a, blocks, waves, lds, s);
      default: return launch_walign_t<kWideQReg, 0, true>(a, blocks, waves, lds, s);
    }
  }
  if (Q == kWideQReg2) {
    switch (waves) {
      case 7: return launch_walign_t<kWideQReg2, 448, true>(a, blocks, waves, lds, s);
      case 8: return launch_walign_t<kWideQReg2, 512, true>(a, blocks, waves, lds, s);
      default: return launch_walign_t<kWideQReg2, 0, true>(a, blocks, waves, lds, s);
    }
  }
  if (Q != kWideQ) return hipErrorInvalidValue;
  switch (waves) {
    case 3: return launch_walign_t<kWideQ, 192>(a, blocks, waves, lds, s);
    case 4: return launch_walign_t<kWideQ, 256>(a, blocks, waves, lds, s);
    case 5: return launch_walign_t<kWideQ, 320>(a, blocks, waves, lds, s);
    case 6: return launch_walign_t<kWideQ, 384>(a, blocks, waves, lds, s);
    case 7: return launch_walign_t<kWideQ, 448>(a, blocks, waves, lds, s);
    case 8: return launch_walign_t<kWideQ, 512>(a, blocks, waves, lds, s);
    default: return launch_walign_t<kWideQ, 0>(a, blocks, waves, lds, s);
  }
}

size_t wide_lds_bytes(int Lcap, size_t em_floats) {
  return (size_t)wide_em_lds_offset_floats(Lcap) * sizeof(float) + em_floats * sizeof(float);
}

template <int Q, int NLT, bool TR>
static hipError_t launch_wide_t(const WideArgs &a, int blocks, int waves, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&wide::score_wide_kernel<Q, NLT, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((wide::score_wide_kernel<Q, NLT, TR>), dim3(blocks), dim3(waves * 64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_score_wide(int Q, const WideArgs &a, int blocks, int waves, size_t lds, hipStream_t s) {
  if (waves < 1 || waves > kWideWavesMax) return hipErrorInvalidValue;
  if (Q == 4) return launch_wide_t<4, 0, false>(a, blocks, waves, lds, s);       // test hook (WH_FORCE_WIDE=4): any workgroup size
  if (Q == kWideQReg) {
    // 12 cells per lane, transition tables in registers: models of 3 073 - 6 144 nodes (five to eight waves)
    switch (waves) {
      case 5: return launch_wide_t<kWideQReg, 320, true>(a, blocks, waves, lds, s);
      case 6: return launch_wide_t<kWideQReg, 384, true>(a, blocks, waves, lds, s);
      case 7: return launch_wide_t<kWideQReg, 448, true>(a, blocks, waves, lds, s);
      case 8: return launch_wide_t<kWideQReg, 512, true>(a, blocks, waves, lds, s);
      default: return launch_wide_t<kWideQReg, 0, true>(a, blocks, waves, lds, s);   // 1 - 4 waves: WH_FORCE_WIDE=12 on small models
    }
  }
  if (Q == kWideQReg2) {
    // 16 cells per lane, transition tables in registers: models of 6 145 - 8 192 nodes (seven or eight waves)
    switch (waves) {
      case 7: return launch_wide_t<kWideQReg2, 448, true>(a, blocks, waves, lds, s);
      case 8: return launch_wide_t<kWideQReg2, 512, true>(a, blocks, waves, lds, s);
      default: return launch_wide_t<kWideQReg2, 0, true>(a, blocks, waves, lds, s);   // WH_FORCE_WIDE=16 on small models
    }
  }
  if (Q != kWideQ || a.em_lds) return hipErrorInvalidValue;
  switch (waves) {
    case 3: return launch_wide_t<kWideQ, 192, false>(a, blocks, waves, lds, s);
    case 4: return launch_wide_t<kWideQ, 256, false>(a, blocks, waves, lds, s);
    case 5: return launch_wide_t<kWideQ, 320, false>(a, blocks, waves, lds, s);
    case 6: return launch_wide_t<kWideQ, 384, false>(a, blocks, waves, lds, s);
    case 7: return launch_wide_t<kWideQ, 448, false>(a, blocks, waves, lds, s);
    case 8: return launch_wide_t<kWideQ, 512, false>(a, blocks, waves, lds, s);
    default: return launch_wide_t<kWideQ, 0, false>(a, blocks, waves, lds, s);   // 1, 2 waves: WH_FORCE_WIDE=24 on small models
  }
}

}  // namespace wh
