// Device-side building blocks shared by the scoring and alignment kernels (gfx950 only).
//
// Execution model: ONE wavefront (64 lanes) owns one (query, HMM) pair.  Lane r holds the
// DP cells of model nodes k = r*Q + q + 1 (q < Q) in VGPRs and sweeps the query row by row.
//  * the in-row D->D dependency is an affine recurrence D_k = s_k + c_k * D_{k-1}: solved
//    per lane serially, across lanes with a 6-step DPP prefix scan whose multiplicative
//    part depends only on the model and is precomputed (ScanC);
//  * row sums (the E state, B<-M_k) are DPP butterfly reductions;
//  * the Backward sweep runs in REVERSED node order (lane r owns the block 63-r, cells
//    descending) so that "k+1 -> k" is again "lane r-1 -> lane r" and reuses the same scan.
#pragma once
#include <hip/hip_runtime.h>

#include "wh_common.h"

namespace wh {

// ------------------------------------------------------------------ DPP primitives
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = false>
__device__ __forceinline__ float dppf(float old, float src) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                         ROW_MASK, BANK_MASK, BOUND));
}

// 16-byte streaming accesses that bypass the vector L1 (data written by one lane is re-read
// by another lane of the same wave later; it has no reuse in L1)
typedef float v4f_t __attribute__((ext_vector_type(4)));
// Layout of a stored Forward row (in float4 pieces; a row is 2 * Q/4 * 64 of them): the pieces of ONE lane block - its M
// cells, then its I cells - are contiguous (128 B at 16 cells per lane).  The rows are stored sparsely - one to three
// adjacent lane blocks of a row after its first ~25 - and piece-major rows (piece q of every lane side by side, a 1 KB
// store per instruction when all 64 lanes store) put the 8 pieces of a block into 8 different 128-B lines: the memory
// system moved 9.2 TB per headline step for 2.7 TB stored (profiles/r05_v3_traffic.json).
// The price: a row stored by MANY lanes takes one 128-B line per lane and instruction (the headline with every envelope's
// first rows at full width - WH_SPILL_BAND=0 - runs 2.1 x slower than with piece-major rows), so this layout needs the band
// of lane blocks (spill_band, wh_score7.hip); models of fewer than 8 cells per lane have no band and keep piece-major rows.
// 8 192 x 200 headline pairs: 397 -> 373 ms; SURVEY's family sketch 295 -> 283 ms; protein slice 893 -> 875 ms.
// BLK = false: piece-major rows - the long-query instantiations of the scoring kernels (special states in HBM, "SG"), whose
// first sweep keeps no mask to place a band with: their envelopes store every lane block that passes the keep rule, which on
// block-contiguous rows cost the 24-cell class of the reference's example data (real 16S fragments of ~400 nt) 3 x its time
// (19.5 -> 60 ms for 9 000 pairs; found at the end of round 5 by tools/bench_example.py across builds).
template <int Q, bool BLK = true>
__device__ __forceinline__ constexpr int fs_piece(int lane, int q4) { return (BLK && Q >= 8) ? lane * (2 * (Q / 4)) + q4 : q4 * kWave + lane; }
__device__ __forceinline__ float4 nt_load4(const float4 *p) {
  v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4 *p, float a, float b, float c, float d) {
  v4f_t v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<v4f_t *>(p));
}

// value of lane-1 (lane 0 receives 0): wave_shr:1
__device__ __forceinline__ float wave_shr1(float x) { return dppf<0x138, 0xF, 0xF, true>(0.f, x); }

__device__ __forceinline__ float readlane_f(float x, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

// sum over the 64 lanes, result uniform
__device__ __forceinline__ float wave_sum(float x) {
  x += dppf<0xB1>(0.f, x);    // quad_perm [1,0,3,2]
  x += dppf<0x4E>(0.f, x);    // quad_perm [2,3,0,1]
  x += dppf<0x141>(0.f, x);   // row_half_mirror
  x += dppf<0x140>(0.f, x);   // row_mirror: every lane holds its 16-lane row sum
  return (readlane_f(x, 0) + readlane_f(x, 16)) + (readlane_f(x, 32) + readlane_f(x, 48));
}

__device__ __forceinline__ float wave_max(float x) {
  x = fmaxf(x, dppf<0xB1>(x, x));
  x = fmaxf(x, dppf<0x4E>(x, x));
  x = fmaxf(x, dppf<0x141>(x, x));
  x = fmaxf(x, dppf<0x140>(x, x));
  return fmaxf(fmaxf(readlane_f(x, 0), readlane_f(x, 16)), fmaxf(readlane_f(x, 32), readlane_f(x, 48)));
}

// ------------------------------------------------------------------ per-row arrays in HBM, walked downwards
// The sweeps of long queries keep their per-row special states in a per-wave HBM region (SG sweeps).  Read one value
// at a time, every row paid a dependent HBM round trip per array; here 64 rows of each array arrive with ONE coalesced
// load (lane t holds row top - t, shifted by the array's <shift>) and a row's value is a v_readlane into a scalar
// register.  <off[n]> is the float offset of array n's row 0, <shift[n]> is 0 or -1 (the sweep reads row i - 1).
template <int N>
struct RowsDown {
  unsigned v[N];
  int top;
  __device__ __forceinline__ void load(const float *spec, const int (&off)[N], const int (&shift)[N], int i, int lane) {
    top = i;
#pragma unroll
    for (int n = 0; n < N; n++) {
      const int r = i - lane + shift[n];
      v[n] = r >= 0 ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(spec) + off[n] + r) : 0u;
    }
  }
  __device__ __forceinline__ bool spent(int i) const { return top - i >= kWave; }
  __device__ __forceinline__ unsigned u(int n, int i) const { return (unsigned)__builtin_amdgcn_readlane((int)v[n], top - i); }
  __device__ __forceinline__ int s(int n, int i) const { return __builtin_amdgcn_readlane((int)v[n], top - i); }
  __device__ __forceinline__ float f(int n, int i) const { return __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)v[n], top - i)); }
};

// ------------------------------------------------------------------ affine prefix scan
// Inclusive scan over lanes of maps D -> A_r * D + B_r.  The A-part is model-only, so the
// six per-step multipliers are computed once (scan_prepare) and each row costs 6 x (DPP + FMA).
struct ScanC { float s[6]; };

__device__ __forceinline__ ScanC scan_prepare(float A) {
  ScanC c;
  c.s[0] = A; A *= dppf<0x111>(1.f, A);                 // row_shr:1
  c.s[1] = A; A *= dppf<0x112>(1.f, A);                 // row_shr:2
  c.s[2] = A; A *= dppf<0x114>(1.f, A);                 // row_shr:4
  c.s[3] = A; A *= dppf<0x118>(1.f, A);                 // row_shr:8
  c.s[4] = A; A *= dppf<0x142, 0xA>(1.f, A);            // row_bcast:15 into rows 1,3
  c.s[5] = A;
  return c;
}

__device__ __forceinline__ float scan_apply(const ScanC &c, float B) {
#ifndef WH_PLAIN_DPP_SCAN
  // B += c * B[lane - n] as ONE v_fmac_f32_dpp per step (lanes without a source are not
  // written, which equals adding c * 0); the s_nop covers the VALU-write -> DPP-read hazard,
  // which the assembler does not fill in for inline code
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s[0]));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s[1]));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s[2]));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s[3]));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(B) : "v"(c.s[4]));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(B) : "v"(c.s[5]));
  asm("s_nop 1" : "+v"(B));
  return B;
#else
  B = fmaf(c.s[0], dppf<0x111>(0.f, B), B);
  B = fmaf(c.s[1], dppf<0x112>(0.f, B), B);
  B = fmaf(c.s[2], dppf<0x114>(0.f, B), B);
  B = fmaf(c.s[3], dppf<0x118>(0.f, B), B);
  B = fmaf(c.s[4], dppf<0x142, 0xA>(0.f, B), B);
  B = fmaf(c.s[5], dppf<0x143, 0xC>(0.f, B), B);        // row_bcast:31 into rows 2,3
  return B;
#endif
}

// A pointer the compiler cannot prove wave-uniform (an argument of a non-inlined function arrives in vector registers)
// read back from lane 0: it then lives in scalar registers, and loads through it take the <scalar base + vector offset>
// form instead of a 64-bit address per lane.
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}

// ------------------------------------------------------------------ the same primitives inside a DPP row (16 lanes)
// Four problems per wavefront, one per row of 16 lanes (round 5, sweep_backward_null2_quad): every cross-lane step stays
// inside the row - the row_* DPP controls never leave it, and lane 0 of a row has no left neighbour (bound_ctrl -> 0).
__device__ __forceinline__ float row_shr1(float x) { return dppf<0x111, 0xF, 0xF, true>(0.f, x); }   // row_shr:1, lane 0 of the row <- 0
__device__ __forceinline__ float row_sum(float x) {          // every lane ends with the sum over its row
  x += dppf<0xB1>(0.f, x);
  x += dppf<0x4E>(0.f, x);
  x += dppf<0x141>(0.f, x);
  x += dppf<0x140>(0.f, x);
  return x;
}
struct ScanR { float s[4]; };
__device__ __forceinline__ ScanR scan_prepare_row(float A) {
  ScanR c;
  c.s[0] = A; A *= dppf<0x111>(1.f, A);
  c.s[1] = A; A *= dppf<0x112>(1.f, A);
  c.s[2] = A; A *= dppf<0x114>(1.f, A);
  c.s[3] = A;
  return c;
}
__device__ __forceinline__ float scan_apply_row(const ScanR &c, float B) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s[0]));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s[1]));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s[2]));
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(B) : "v"(c.s[3]));
  asm("s_nop 1" : "+v"(B));
  return B;
}

// ------------------------------------------------------------------ transition tables
// TREG: the 8 arrays of one orientation live in VGPRs (reloaded per pass from L2);
// otherwise they are read from LDS in 16-byte pieces on every use.
template <int Q, bool TREG>
struct TransTab;

template <int Q>
struct TransTab<Q, true> {
  float4 v[FW_NARR][Q / 4];
  __device__ __forceinline__ void load(const float *g, const float * /*lds*/, int lane) {
    const float4 *p = reinterpret_cast<const float4 *>(g);
#pragma unroll
    for (int a = 0; a < FW_NARR; a++)
#pragma unroll
      for (int q4 = 0; q4 < Q / 4; q4++) v[a][q4] = p[(a * (Q / 4) + q4) * kWave + lane];
  }
  __device__ __forceinline__ float4 ld(int a, int q4) const { return v[a][q4]; }
};

template <int Q>
struct TransTab<Q, false> {
  const float4 *p;
  __device__ __forceinline__ void load(const float * /*g*/, const float *lds, int lane) {
    p = reinterpret_cast<const float4 *>(lds) + lane;
  }
  __device__ __forceinline__ float4 ld(int a, int q4) const { return p[(a * (Q / 4) + q4) * kWave]; }
};

__device__ __forceinline__ float f4get(const float4 &v, int j) {
  return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
}

// product over the lane's Q cells of array <arr> (the D->D coefficients)
template <int Q, bool TREG>
__device__ __forceinline__ float lane_product(const TransTab<Q, TREG> &T, int arr) {
  float A = 1.f;
#pragma unroll
  for (int q4 = 0; q4 < Q / 4; q4++) {
    float4 d = T.ld(arr, q4);
    A *= d.x; A *= d.y; A *= d.z; A *= d.w;
  }
  return A;
}

// ------------------------------------------------------------------ emission rows
// Canonical residues come from the LDS copy of the table, degenerate codes (rare) from L2.
// The two sources are kept in separate branches so that the LDS path compiles to ds_read_b128
// (a pointer select between LDS and global would degrade both to flat loads).
// The LDS branch reads through a pointer TYPED as LDS: with two generic pointers the compiler merges the branches
// into one flat_load behind a pointer select - half the LDS rate, and a flat load counts on vmcnt AND lgkmcnt, so
// every row waited for the row stores and row prefetches in flight (seen in the round-3 ISA of every sweep).
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) v4f_t lds_cv4_t;
struct LdsF4 {      // float4 pieces of an LDS-resident array
  lds_cv4_t *p;
  __device__ __forceinline__ LdsF4(const float *base) : p((lds_cv4_t *)base) {}
  __device__ __forceinline__ float4 operator[](int i) const { const v4f_t v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
};

template <int Q>
__device__ __forceinline__ void load_em_fwd(float (&od)[Q], const float *emL, const float *emG, int x, int K,
                                            int lane) {
  if (x < K) {
    const LdsF4 p(emL + (size_t)x * Q * kWave + 4 * lane);
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      float4 v = p[q4 * kWave];
      od[4 * q4] = v.x; od[4 * q4 + 1] = v.y; od[4 * q4 + 2] = v.z; od[4 * q4 + 3] = v.w;
    }
  } else {
    const float4 *p = reinterpret_cast<const float4 *>(emG + (size_t)x * Q * kWave) + lane;
#pragma unroll
    for (int q4 = 0; q4 < Q / 4; q4++) {
      float4 v = p[q4 * kWave];
      od[4 * q4] = v.x; od[4 * q4 + 1] = v.y; od[4 * q4 + 2] = v.z; od[4 * q4 + 3] = v.w;
    }
  }
}

// reversed node order: position (lane, p) is forward position (63-lane, Q-1-p)
template <int Q>
__device__ __forceinline__ void load_em_rev(float (&od)[Q], const float *emL, const float *emG, int x, int K,
                                            int lane) {
  if (x < K) {
    const LdsF4 p(emL + (size_t)x * Q * kWave + 4 * (kWave - 1 - lane));
#pragma unroll
    for (int p4 = 0; p4 < Q / 4; p4++) {
      float4 v = p[(Q / 4 - 1 - p4) * kWave];
      od[4 * p4] = v.w; od[4 * p4 + 1] = v.z; od[4 * p4 + 2] = v.y; od[4 * p4 + 3] = v.x;
    }
  } else {
    const float4 *p = reinterpret_cast<const float4 *>(emG + (size_t)x * Q * kWave) + (kWave - 1 - lane);
#pragma unroll
    for (int p4 = 0; p4 < Q / 4; p4++) {
      float4 v = p[(Q / 4 - 1 - p4) * kWave];
      od[4 * p4] = v.w; od[4 * p4 + 1] = v.z; od[4 * p4 + 2] = v.y; od[4 * p4 + 3] = v.x;
    }
  }
}

// power-of-two rescale helpers: exact, so scaling adds no rounding error
__device__ __forceinline__ int f32_exponent(float x) { return ((__builtin_bit_cast(int, x) >> 23) & 0xFF) - 127; }
__device__ __forceinline__ float pow2f_int(int e) { return __builtin_bit_cast(float, (e + 127) << 23); }

struct LenCfg { float loop, move, EJ, EC; };

// A.1 length model; HMMER evaluates it in float32
__device__ __forceinline__ LenCfg len_config(int Lcfg, bool multihit) {
  LenCfg c;
  float nj = multihit ? 1.0f : 0.0f;
  c.move = (2.0f + nj) / ((float)Lcfg + 2.0f + nj);
  c.loop = 1.0f - c.move;
  c.EJ = multihit ? 0.5f : 0.0f;
  c.EC = multihit ? 0.5f : 1.0f;
  return c;
}

// Per-wave LDS block: six float arrays of SP entries (special states per row) + small tables.
enum { SP_N = 0, SP_B, SP_E, SP_J, SP_C, SP_S, SP_ML, SP_MH, SP_NARR };   // SP_S: cumulative scale exponent (int bits);
                                                                  // SP_ML/MH: 64-bit mask of lanes whose row block was stored

constexpr float kRescaleHi = 1048576.0f;   // 2^20

// ------------------------------------------------------------------ region scan over special states in HBM
// A.4's scan over rows 1..L (thresholds rt1/rt2) when the per-row arrays live in global memory: 64 rows are
// fetched with one coalesced load per array and walked with v_readlane, instead of three dependent L2/HBM
// round trips per row; the running sums are formed in the same order, so the result is bit-identical to the
// row-by-row loop.  Writes the cumulative sums into SP_J / SP_C (coalesced) and the regions into regs[].
__device__ __forceinline__ void region_scan_global(float *spec, int SP, int L, int *regs, int lane, int &nenv,
                                                   int &nreg, int &flags) {
  const float rt1 = 0.25f, rt2 = 0.10f;
  float btot = 0.f, etot = 0.f;
  int i0 = -1;
  bool trig = false;
  if (lane == 0) { spec[SP_J * SP] = 0.f; spec[SP_C * SP] = 0.f; }
  for (int j0 = 1; j0 <= L; j0 += kWave) {
    const int jj = j0 + lane;
    const bool valid = jj <= L;
    const float nv = valid ? __builtin_nontemporal_load(spec + SP_N * SP + jj) : 0.f;
    const float bv = valid ? __builtin_nontemporal_load(spec + SP_B * SP + jj - 1) : 0.f;
    const float ev = valid ? __builtin_nontemporal_load(spec + SP_E * SP + jj) : 0.f;
    float jout = 0.f, cout = 0.f;
    const int cnt = L - j0 + 1 < kWave ? L - j0 + 1 : kWave;
    for (int t = 0; t < cnt; t++) {
      const int j = j0 + t;
      const float mocc = 1.0f - readlane_f(nv, t);
      const float bold = btot, eold = etot;
      btot += readlane_f(bv, t);
      etot += readlane_f(ev, t);
      if (lane == t) { jout = btot; cout = etot; }
      if (!trig) {
        if (mocc - (btot - bold) < rt2) i0 = j;
        else if (i0 == -1) i0 = j;
        if (mocc >= rt1) trig = true;
      } else if (mocc - (etot - eold) < rt2) {
        if (nenv < WH_MAX_ENVELOPES) {
          if (lane == 0) { regs[2 * nenv] = i0; regs[2 * nenv + 1] = j; }
          nenv++;
        } else flags |= WH_FLAG_TRUNC;
        nreg++;
        i0 = -1;
        trig = false;
      }
    }
    if (valid) { spec[SP_J * SP + jj] = jout; spec[SP_C * SP + jj] = cout; }
  }
}

// ------------------------------------------------------------------ Forward sweep
// Fills spec[SP_*][0..L]; with STORE also writes the M and I rows (1..L) to <Fs>:
// [row][2][Q/4][64 lanes][4] floats (forward node order; a wavefront's store instruction covers
// one contiguous 1 KiB line).  A lane's cells are written only when one of them exceeds
// keep_scale * E(row) (exec-masked stores: adjacent kept lanes still form contiguous segments);
// the 64-bit mask of kept lanes goes to spec[SP_ML/SP_MH].  keep_scale < 0 stores everything.  Returns C(L) and its
// scale exponent.
// SLIM (with STORE): the per-row B and E arrays of an envelope sweep are never read again, so their slots hold the two
// mask words instead (SP_B <- low word, SP_E <- high word) and a wave's block needs six arrays, not eight.
// UM (without STORE): the dominant-path mask alone - the union over every EIGHTH row of the lane blocks that hold a cell
// above E(row)/2 - written to um_out[0..1]; the multihit sweep uses it to place the node window of its Backward sweep.
// BLK (with STORE): the stored rows keep a lane block's pieces contiguous (fs_piece, above: the scoring kernels); without it
// piece-major rows (the alignment kernels, whose passes read and rewrite whole rows).
template <int Q, bool TREG, bool STORE, bool USEP = false, bool SLIM = false, bool UM = false, bool COUNT = false, bool BLK = false>
__device__ __forceinline__ void forward_sweep(const TransTab<Q, TREG> &T, const ScanC &sc, const float *emL,
                                              const float *emG, int K, const uint8_t *seq, int L, LenCfg cfg,
                                              float *spec, int SP, float *Fs, float keep_scale, int lane,
                                              float &xC_out, int &ef_out, unsigned *um_out = nullptr, int *nstored_out = nullptr, int keep_lanes = 63 << 8) {
  float Mp[Q], Ip[Q], Dp[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) { Mp[q] = 0.f; Ip[q] = 0.f; Dp[q] = 0.f; }
  float xN = 1.0f, xB = cfg.move, xJ = 0.f, xC = 0.f, xE = 0.f;
  int ef = 0;
  unsigned long long um_prev = 0, um_steady = 0;
  int nstored = 0;                   // COUNT: lane blocks stored over all rows (scalar: one s_bcnt1 + s_add per row)
  unsigned long long umask = 0;      // STORE: union over the rows of the lane block that holds a cell above E(row)/2: where the
                                     // envelope's dominant alignment runs.  The first rows set no bit: it takes ~25 nucleotides
                                     // until the true diagonal outweighs the chance matches among ~1000 others
  if (lane == 0) {
    spec[SP_N * SP] = xN; spec[SP_B * SP] = xB; spec[SP_E * SP] = 0.f; spec[SP_J * SP] = 0.f;
    spec[SP_C * SP] = 0.f; reinterpret_cast<int *>(spec)[SP_S * SP] = 0;
  }
#pragma unroll 1
  for (int i = 1; i <= L; i++) {
    asm volatile("" ::: "memory");   // keep LDS table reads inside the row (no hoisting into VGPRs)
    const int x = seq[i - 1];
    float od[Q];
    load_em_fwd<Q>(od, emL, emG, x, K, lane);
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
    // D row: local chains, cross-lane scan, fix-up
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
      const float4 D2 = T.ld(USEP ? FW_P : FW_D2, q4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = 4 * q4 + j;
        if (USEP) {
          Dp[q] = fmaf(f4get(D2, j), carry, Dp[q]);   // D2 holds the in-lane running product here
        } else {
          carry *= f4get(D2, j);
          Dp[q] += carry;
        }
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
      spec[SP_N * SP + i] = xN;
      if (!(STORE && SLIM)) { spec[SP_B * SP + i] = xB; spec[SP_E * SP + i] = xE; }
      spec[SP_J * SP + i] = xJ; spec[SP_C * SP + i] = xC;
      reinterpret_cast<int *>(spec)[SP_S * SP + i] = ef;
    }
    if (UM && !STORE) {
      if ((i & 7) == 0) {
        float lmax = 0.f;
#pragma unroll
        for (int q = 0; q < Q; q += 2) lmax = fmaxf(fmaxf(fmaxf(fmaxf(lmax, Mp[q]), Mp[q + 1]), Ip[q]), Ip[q + 1]);      // (two v_max3 per four cells)
        const unsigned long long dom = __ballot(lmax > 0.5f * xE);
        umask |= dom;
        // ... and the same union over the blocks whose dominance CARRIED OVER from the last sampled row (the same block or the
        // next one up): a chance diagonal that rivals the alignment on its first rows does not last eight rows
        const unsigned long long link = dom & (um_prev | (um_prev << 1));
        if (link) um_steady |= link | (um_prev & (link | (link >> 1)));
        um_prev = dom;
      }
    }
    if (STORE) {
      constexpr int ML = SLIM ? SP_B : SP_ML, MH = SLIM ? SP_E : SP_MH;
      float lmax = 0.f;
#pragma unroll
      for (int q = 0; q < Q; q += 2) lmax = fmaxf(fmaxf(fmaxf(fmaxf(lmax, Mp[q]), Mp[q + 1]), Ip[q]), Ip[q + 1]);      // (two v_max3 per four cells)
      // (keep_scale < 0 must store EVERY row: a row that has underflowed to zero everywhere gives 0 > -0 = false, and the
      // full-width alignment passes, which read rows without looking at the masks, then saw the previous pair's cells)
      const unsigned long long dom = __ballot(lmax > 0.5f * xE);
      umask |= dom;
      // keep_lanes: the band of lane blocks the caller thinks worth storing (lowest | highest << 8; the mass certificate of the
      // Backward sweep judges the choice like it judges keep_scale)
      // (bits 16..: a cap on the highest block that rises with the row - block <cap - 1 + i / Q>: the alignment cannot be
      // further up the model than its start plus the rows walked.  It thins the FIRST rows, where the keep rule holds nothing back)
      const int blo = keep_lanes & 255;
      int bhi = (keep_lanes >> 8) & 255;
      if (keep_lanes >> 16) bhi = min(bhi, (keep_lanes >> 16) - 1 + i / Q);
      const bool keep = keep_scale < 0.f || (lmax > keep_scale * xE && lane >= blo && lane <= bhi);
      const unsigned long long mask = __ballot(keep);
      if (COUNT) nstored += __builtin_popcountll(mask);
      if (lane == 0) {
        reinterpret_cast<unsigned *>(spec)[ML * SP + i] = (unsigned)(mask & 0xFFFFFFFFull);
        reinterpret_cast<unsigned *>(spec)[MH * SP + i] = (unsigned)(mask >> 32);
      }
      if (keep) {
        float4 *row = reinterpret_cast<float4 *>(Fs) + (size_t)i * (2 * (Q / 4) * kWave);
#pragma unroll
        for (int q4 = 0; q4 < Q / 4; q4++) {
          // streamed once and re-read once by another lane of this wave: keep it out of L1
          nt_store4(row + (BLK ? fs_piece<Q>(lane, q4) : q4 * kWave + lane), Mp[4 * q4], Mp[4 * q4 + 1], Mp[4 * q4 + 2], Mp[4 * q4 + 3]);
          nt_store4(row + (BLK ? fs_piece<Q>(lane, Q / 4 + q4) : (Q / 4 + q4) * kWave + lane), Ip[4 * q4], Ip[4 * q4 + 1], Ip[4 * q4 + 2], Ip[4 * q4 + 3]);
        }
      }
    }
  }
  if (STORE && lane == 0) {
    // row 0 of the two mask arrays is free (rows are 1..L): the dominant-path mask of the sweep
    reinterpret_cast<unsigned *>(spec)[(SLIM ? SP_B : SP_ML) * SP] = (unsigned)(umask & 0xFFFFFFFFull);
    reinterpret_cast<unsigned *>(spec)[(SLIM ? SP_E : SP_MH) * SP] = (unsigned)(umask >> 32);
  }
  if (UM && !STORE && lane == 0) {
    um_out[0] = (unsigned)(umask & 0xFFFFFFFFull); um_out[1] = (unsigned)(umask >> 32);
    // (one word in front: lowest | highest << 8 | 1 << 16 of the steady blocks, 0 when there are none)
    um_out[-1] = um_steady ? (unsigned)(__builtin_ctzll(um_steady) | ((63 - __builtin_clzll(um_steady)) << 8) | (1 << 16)) : 0u;
  }
  if (COUNT && nstored_out) *nstored_out = nstored;
  xC_out = xC;
  ef_out = ef;
}

// ------------------------------------------------------------------ envelope Backward scaling
// The unihit Backward sweep of an envelope carries the scale the Forward sweep left on the rows
// still to come, 2^-(ef_e - S(i)): F_s(i,k) * B_s(i,k) / Z_s is then the posterior with no per-row
// exponent, and every cell whose posterior matters stays in float32 range because its Forward
// partner is O(1).  (A normaliser of Backward's own - max(B, N), as the multihit sweep uses - lets
// a strong alternative hit that starts later push the cells of the best path below 2^-149 in
// envelopes that hold two unequal hits: lost posterior mass, 0 * inf in the accumulators.)
// mirror_scale applies the Forward rescale of row i+1 (dn = S(i+1) - S(i)) to the carried state;
// clamp_backward saturates values whose Forward partner has underflowed (posterior nil) instead of
// rescaling the row.  All factors are powers of two: results are bit-identical to any other exact
// scaling as long as nothing leaves the float32 range.
constexpr float kClampHi = 1.1529215e18f;   // 2^60

template <int Q>
__device__ __forceinline__ void mirror_scale(int dn, float (&Mb)[Q], float (&Ib)[Q], float &xJ, float &xC, float &xN) {
  if (dn != 0) {
    const float r = __builtin_bit_cast(float, (127 - dn) << 23);
#pragma unroll
    for (int p = 0; p < Q; p++) { Mb[p] *= r; Ib[p] *= r; }
    xJ *= r; xC *= r; xN *= r;
  }
}

template <int Q>
__device__ __forceinline__ bool clamp_backward(float (&Mb)[Q], float (&Ib)[Q], float &xB, float &xJ, float &xC, float &xN) {
  if (fmaxf(xB, fmaxf(xN, xC)) > kClampHi) {
#pragma unroll
    for (int p = 0; p < Q; p++) { Mb[p] = fminf(Mb[p], kClampHi); Ib[p] = fminf(Ib[p], kClampHi); }
    xB = fminf(xB, kClampHi); xN = fminf(xN, kClampHi); xC = fminf(xC, kClampHi); xJ = fminf(xJ, kClampHi);
    return true;      // the pair has left float32 range (hmmalign would switch to its log-space code here)
  }
  return false;
}

// G_k = o_k(x) * B_M_k in place (reversed node order) and the B-state sum  sum_k E_k G_k;
// the emission piece of each 4-cell group is fetched where it is used.
template <int Q, bool TREG>
__device__ __forceinline__ float backward_emit(const TransTab<Q, TREG> &T, const float *emL, const float *emG, int x,
                                               int K, int lane, float (&Mb)[Q]) {
  float part = 0.f;
  auto groups = [&](auto em_ld) {
#pragma unroll
    for (int p4 = 0; p4 < Q / 4; p4++) {
      const float4 E = T.ld(BW_E, p4);
      const float4 O = em_ld(Q / 4 - 1 - p4);      // forward-ordered piece; component 3-j is position 4*p4+j
      Mb[4 * p4 + 0] *= O.w; part = fmaf(E.x, Mb[4 * p4 + 0], part);
      Mb[4 * p4 + 1] *= O.z; part = fmaf(E.y, Mb[4 * p4 + 1], part);
      Mb[4 * p4 + 2] *= O.y; part = fmaf(E.z, Mb[4 * p4 + 2], part);
      Mb[4 * p4 + 3] *= O.x; part = fmaf(E.w, Mb[4 * p4 + 3], part);
    }
  };
  x = __builtin_amdgcn_readfirstlane(x);
  if (x < K) {
    const LdsF4 ep(emL + (size_t)x * Q * kWave + 4 * (kWave - 1 - lane));
    groups([&](int q4) { return ep[q4 * kWave]; });
  } else {
    const float4 *ep = reinterpret_cast<const float4 *>(emG + (size_t)x * Q * kWave) + (kWave - 1 - lane);
    groups([&](int q4) { return ep[q4 * kWave]; });
  }
  return part;
}

// Transition arrays of one orientation read from the lane-blocked LDS copy through PER-LANE piece slots (a lane of a
// quarter-wave window owns 16 nodes somewhere in the model: slot[p4] = float4 index of its piece p4 inside an array).
template <int Q>
struct TransTabAt {
  LdsF4 base;
  int slot[4];
  __device__ __forceinline__ TransTabAt(const float *lds) : base(lds) {}
  __device__ __forceinline__ float4 ld(int a, int p4) const { return base[a * (Q / 4) * kWave + slot[p4]]; }
};

// ------------------------------------------------------------------ Backward sweep core
// One Backward row in reversed node order.  On entry Mb/Ib hold row i+1 (or zeros for
// i = L) and <G> has been formed in place in Mb (G_k = o_k(x_{i+1}) * B_M_k(i+1)).
// Produces row i in Mb/Ib.  xE = E(i).
template <int Q, bool TREG, bool USEP = false>
__device__ __forceinline__ void backward_cells(const TransTab<Q, TREG> &T, const ScanC &sc, float (&Mb)[Q],
                                               float (&Ib)[Q], float xE) {
  float Dn[Q];
  const float gm1 = wave_shr1(Mb[Q - 1]);   // G of node k+1 across the lane boundary
  float dprev = 0.f;
#pragma unroll
  for (int p4 = 0; p4 < Q / 4; p4++) {
    const float4 DM = T.ld(BW_DM, p4), DD = T.ld(BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
      const float g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
      dprev = fmaf(f4get(DD, j), dprev, fmaf(f4get(DM, j), g, xE));
      Dn[p] = dprev;
    }
  }
  float carry = wave_shr1(scan_apply(sc, dprev));
#pragma unroll
  for (int p4 = 0; p4 < Q / 4; p4++) {
    const float4 DD = T.ld(USEP ? BW_P : BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
      if (USEP) {
        Dn[p] = fmaf(f4get(DD, j), carry, Dn[p]);
      } else {
        carry *= f4get(DD, j);
        Dn[p] += carry;
      }
    }
  }
  const float dm1 = wave_shr1(Dn[Q - 1]);
#pragma unroll
  for (int p4 = Q / 4 - 1; p4 >= 0; p4--) {
    const float4 MM = T.ld(BW_MM, p4), IM = T.ld(BW_IM, p4), MI = T.ld(BW_MI, p4), II = T.ld(BW_II, p4);
    const float4 MD = T.ld(BW_MD, p4);
#pragma unroll
    for (int j = 3; j >= 0; j--) {
      const int p = 4 * p4 + j;
      const float g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
      const float dn = p > 0 ? Dn[p > 0 ? p - 1 : 0] : dm1;
      float nm = fmaf(f4get(MM, j), g, xE);
      nm = fmaf(f4get(MI, j), Ib[p], nm);
      nm = fmaf(f4get(MD, j), dn, nm);
      const float ni = fmaf(f4get(IM, j), g, f4get(II, j) * Ib[p]);
      Mb[p] = nm;
      Ib[p] = ni;
    }
  }
}

// One Backward row for FOUR windows at once, one per DPP row: backward_cells' arithmetic on 16 cells per lane, the
// cross-lane steps (G / D of the neighbouring node, the D->D scan) inside the row of 16 lanes.  TT::ld(array, piece).
template <class TT>
__device__ __forceinline__ void backward_cells_row(const TT &T, const ScanR &sc, float (&Mb)[16], float (&Ib)[16], float xE) {
  float Dn[16];
  const float gm1 = row_shr1(Mb[15]);
  float dprev = 0.f;
#pragma unroll
  for (int p4 = 0; p4 < 4; p4++) {
    const float4 DM = T.ld(BW_DM, p4), DD = T.ld(BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
      const float g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
      dprev = fmaf(f4get(DD, j), dprev, fmaf(f4get(DM, j), g, xE));
      Dn[p] = dprev;
    }
  }
  float carry = row_shr1(scan_apply_row(sc, dprev));
#pragma unroll
  for (int p4 = 0; p4 < 4; p4++) {
    const float4 DD = T.ld(BW_DD, p4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int p = 4 * p4 + j;
      carry *= f4get(DD, j);
      Dn[p] += carry;
    }
  }
  const float dm1 = row_shr1(Dn[15]);
#pragma unroll
  for (int p4 = 3; p4 >= 0; p4--) {
    const float4 MM = T.ld(BW_MM, p4), IM = T.ld(BW_IM, p4), MI = T.ld(BW_MI, p4), II = T.ld(BW_II, p4);
    const float4 MD = T.ld(BW_MD, p4);
#pragma unroll
    for (int j = 3; j >= 0; j--) {
      const int p = 4 * p4 + j;
      const float g = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
      const float dn = p > 0 ? Dn[p > 0 ? p - 1 : 0] : dm1;
      float nm = fmaf(f4get(MM, j), g, xE);
      nm = fmaf(f4get(MI, j), Ib[p], nm);
      nm = fmaf(f4get(MD, j), dn, nm);
      const float ni = fmaf(f4get(IM, j), g, f4get(II, j) * Ib[p]);
      Mb[p] = nm;
      Ib[p] = ni;
    }
  }
}

}  // namespace wh
