// Float64 building blocks shared by the multidomain resolver (wh_resolve.hip) and the any-size kernels
// (wh_generic.hip): one wavefront per (query, model) pair, lane r owns nodes r*Q+1 .. r*Q+Q for ANY Q, the DP
// rows in a per-wave HBM slab (or in registers for models of up to 1 024 nodes), the D->D chain a cross-lane
// affine scan in double.
#pragma once
#include <hip/hip_runtime.h>

#include "wh_launch.h"

namespace wh {
namespace {

constexpr double kRescaleHi = 1e60;

enum { gA = 0, gB, gC, gE, gMI, gII, gD1, gD2, gNARR };
enum { xN = 0, xB, xE, xJ, xC, xLS, xCLS, xNSPEC = 8 };   // specials in a row's tail; xLS = ln of the rescale applied at this row,
                                                         // xCLS = the sum of those up to and including this row

__device__ __forceinline__ double shfl_up_d(double v, int d) {
  const long long u = __double_as_longlong(v);
  const int lo = __shfl_up((int)(u & 0xFFFFFFFFll), d), hi = __shfl_up((int)(u >> 32), d);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double shfl_d(double v, int l) {
  const long long u = __double_as_longlong(v);
  const int lo = __shfl((int)(u & 0xFFFFFFFFll), l), hi = __shfl((int)(u >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {      // l must be wave-uniform
  const long long u = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(u & 0xFFFFFFFFll), l), hi = __builtin_amdgcn_readlane((int)(u >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_d(double x) {
  for (int m = 32; m >= 1; m >>= 1) {
    const long long u = __double_as_longlong(x);
    const int lo = __shfl_xor((int)(u & 0xFFFFFFFFll), m), hi = __shfl_xor((int)(u >> 32), m);
    x += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  return x;
}
__device__ __forceinline__ float wave_sum_f(float x) {
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
  return x;
}
__device__ __forceinline__ int wave_min_i(int x) { for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(x, m); x = o < x ? o : x; } return x; }
__device__ __forceinline__ int wave_max_i(int x) { for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(x, m); x = o > x ? o : x; } return x; }
__device__ __forceinline__ int wave_sum_i(int x) { for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m); return x; }

// This kernel runs ONE wavefront per workgroup, and for such a workgroup the compiler emits nothing for
// a workgroup-scope fence (checked in the ISA: a store followed at once by the dependent load of
// another lane).  Global data handed from one lane to another inside the wave is therefore ordered by
// hand: every store of the wave acknowledged, then the vector L1 invalidated.
__device__ __forceinline__ void wave_mem_sync() {
  asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
}

struct GLen { double loop, move, EJ, EC; };
__device__ __forceinline__ GLen glen_config(int Lcfg, bool multihit) {
  // HMMER evaluates the length model in float32 (A.1)
  const float nj = multihit ? 1.0f : 0.0f;
  const float pmove = (2.0f + nj) / ((float)Lcfg + 2.0f + nj);
  const float ploop = 1.0f - pmove;
  GLen c;
  c.loop = ploop; c.move = pmove; c.EJ = multihit ? 0.5 : 0.0; c.EC = multihit ? 0.5 : 1.0;
  return c;
}

// One model's float64 tables, lane-blocked: value of array <arr> at node k = lane*Q + q + 1 sits at
// [(arr*Q + q)*64 + lane]; emission odds rows the same with arr = residue code.
struct GModel {
  const double *tf, *te;
  int Q, M;
  // node k = lane*Q + q + 1; inside an array the nodes 2j, 2j+1 of a lane are adjacent (ofs2 below)
  __device__ __forceinline__ size_t at(int arr, int k) const { const int q = (k - 1) % Q, ln = (k - 1) / Q; return (size_t)arr * Q * 64 + ((((size_t)(q >> 1) * 64 + ln) << 1) + (q & 1)); }
  __device__ __forceinline__ double t(int arr, int k) const { return tf[at(arr, k)]; }
};

// The per-wave matrix slab: row i = [3 states][Q/2][64 lanes][2] doubles + xNSPEC specials.
struct GMx {
  double *p;
  size_t rowlen;
  int Q;
  __device__ __forceinline__ double *row(int i) const { return p + (size_t)i * rowlen; }
  __device__ __forceinline__ double cell(int i, int k, int s) const {   // k = 0 reads as 0
    if (k <= 0) return 0.0;
    const int q = (k - 1) % Q, ln = (k - 1) / Q;
    return __builtin_nontemporal_load(p + (size_t)i * rowlen + (size_t)s * Q * 64 + ((((size_t)(q >> 1) * 64 + ln) << 1) + (q & 1)));
  }
  __device__ __forceinline__ double spec(int i, int s) const { return __builtin_nontemporal_load(p + (size_t)i * rowlen + (size_t)3 * Q * 64 + s); }
  // cached reads for the sampling walk: the matrix is complete and the vector L1 was invalidated after the fill
  // (wave_mem_sync); 200 traces revisit the same band of cells, which then stay in L1 / L2
  __device__ __forceinline__ double cellc(int i, int k, int s) const {
    if (k <= 0) return 0.0;
    const int q = (k - 1) % Q, ln = (k - 1) / Q;
    return p[(size_t)i * rowlen + (size_t)s * Q * 64 + ((((size_t)(q >> 1) * 64 + ln) << 1) + (q & 1))];
  }
  __device__ __forceinline__ double specc(int i, int s) const { return p[(size_t)i * rowlen + (size_t)3 * Q * 64 + s]; }
};

// Forward sweep (A.2), float64.  STORE keeps every row (row i at slab row i), otherwise rows
// alternate between slab rows 0 and 1.  Returns ln P in nats (Forward score).
// Within one array of Q x 64 values, node (q, lane) sits at ofs2(q, lane): the two nodes 2j and 2j+1 of a lane are
// adjacent, so the Forward sweep moves them with ONE 16-byte access (it is bound by the number of memory
// instructions: 60 -> 30 per 4-node step).
__device__ __forceinline__ size_t ofs2(int q, int lane) { return (((size_t)(q >> 1) * 64 + lane) << 1) + (q & 1); }
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2_t nt_load_d2(const double *p) { return __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(p)); }
__device__ __forceinline__ d2_t ld_d2(const double *p) { return *reinterpret_cast<const d2_t *>(p); }
__device__ __forceinline__ void st_d2(double *p, double a, double b) { d2_t v = {a, b}; *reinterpret_cast<d2_t *>(p) = v; }

// global-address-space views for code that is CALLED (see gforward_reg)
typedef __attribute__((address_space(1))) double gdbl;
typedef __attribute__((address_space(1))) d2_t gd2_t;
__device__ __forceinline__ gdbl *as_global(const double *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (gdbl *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ d2_t ldg_d2(const gdbl *p) { return *reinterpret_cast<const gd2_t *>(p); }
// the same for a model's transition arrays staged in LDS (resolve_kernel: one copy per workgroup)
typedef __attribute__((address_space(3))) double ldbl;
typedef __attribute__((address_space(3))) d2_t ld2_t;
__device__ __forceinline__ d2_t ldl_d2(const ldbl *p) { return *reinterpret_cast<const ld2_t *>(p); }
__device__ __forceinline__ const ldbl *as_lds(const ldbl *p) {
  typedef __attribute__((address_space(3))) char lds_c;
  const int v = __builtin_amdgcn_readfirstlane((int)(size_t)(const lds_c *)p);
  return (const ldbl *)(const lds_c *)(size_t)(unsigned)v;
}
__device__ __forceinline__ void stg_d2(gdbl *p, double a, double b) { d2_t v = {a, b}; *reinterpret_cast<gd2_t *>(p) = v; }

template <bool STORE>
__device__ double gforward(const GModel &m, const uint8_t *seq, int L, GLen c, const GMx &mx, int lane, double *xs = nullptr) {
  // xs: per-row copy of the special states ((L+1) x xNSPEC doubles) for callers that do not keep the rows
  const int Q = m.Q;
  const size_t SQ = (size_t)Q * 64;
  double ls = 0.0;
  {
    double *r0 = mx.row(0);
    for (int q = 0; q < Q; q++) { r0[(size_t)q * 64 + lane] = 0.0; r0[SQ + (size_t)q * 64 + lane] = 0.0; r0[2 * SQ + (size_t)q * 64 + lane] = 0.0; }
    if (lane == 0) {
      double *s = r0 + 3 * SQ; s[xN] = 1.0; s[xB] = c.move; s[xE] = 0.0; s[xJ] = 0.0; s[xC] = 0.0; s[xLS] = 0.0; s[xCLS] = 0.0;
      if (xs) { xs[xN] = 1.0; xs[xB] = c.move; xs[xE] = 0.0; xs[xJ] = 0.0; xs[xC] = 0.0; xs[xLS] = 0.0; xs[xCLS] = 0.0; }
    }
  }
  double pN = 1.0, pB = c.move, pJ = 0.0, pC = 0.0;
  // model-only part of the cross-lane D scan: A_r = (prod_{q>=1} D2_q) * D2_0 = product of the lane's D2
  double Alane = 1.0;
  for (int q = 0; q < Q; q++) Alane *= m.tf[(size_t)gD2 * SQ + ofs2(q, lane)];
  for (int i = 1; i <= L; i++) {
    wave_mem_sync();    // row i-1 was written by other lanes of this wave
    const double *pr = mx.row(STORE ? i - 1 : (i - 1) & 1);
    double *cr = mx.row(STORE ? i : i & 1);
    const double *od = m.te + (size_t)seq[i - 1] * SQ;
    // previous row at node k-1 for my first node: lane-1's last node
    double pm1 = 0.0, pi1 = 0.0, pd1 = 0.0;
    if (lane > 0) {
      const size_t ol = ofs2(Q - 1, lane - 1);
      pm1 = __builtin_nontemporal_load(pr + ol);
      pi1 = __builtin_nontemporal_load(pr + SQ + ol);
      pd1 = __builtin_nontemporal_load(pr + 2 * SQ + ol);
    }
    double mprev = 0.0, dloc = 0.0, P = 1.0, esum = 0.0;
    // four nodes per step (Q is a multiple of 4): all loads of the step are issued before any of them is
    // used, so a step costs one memory round trip instead of one per node
    for (int q0 = 0; q0 < Q; q0 += 4) {
      double pM[4], pI[4], pD[4], tA[4], tB[4], tC[4], tE[4], tMI[4], tII[4], tD1[4], tD2[4], em[4];
#pragma unroll
      for (int u2 = 0; u2 < 2; u2++) {
        const size_t o = ofs2(q0 + 2 * u2, lane);
        const d2_t vM = nt_load_d2(pr + o), vI = nt_load_d2(pr + SQ + o), vD = nt_load_d2(pr + 2 * SQ + o);
        const d2_t vA = ld_d2(m.tf + gA * SQ + o), vB = ld_d2(m.tf + gB * SQ + o), vC = ld_d2(m.tf + gC * SQ + o), vE = ld_d2(m.tf + gE * SQ + o);
        const d2_t vMI = ld_d2(m.tf + gMI * SQ + o), vII = ld_d2(m.tf + gII * SQ + o), vD1 = ld_d2(m.tf + gD1 * SQ + o), vD2 = ld_d2(m.tf + gD2 * SQ + o);
        const d2_t vem = ld_d2(od + o);
        pM[2 * u2] = vM.x; pM[2 * u2 + 1] = vM.y; pI[2 * u2] = vI.x; pI[2 * u2 + 1] = vI.y; pD[2 * u2] = vD.x; pD[2 * u2 + 1] = vD.y;
        tA[2 * u2] = vA.x; tA[2 * u2 + 1] = vA.y; tB[2 * u2] = vB.x; tB[2 * u2 + 1] = vB.y; tC[2 * u2] = vC.x; tC[2 * u2 + 1] = vC.y;
        tE[2 * u2] = vE.x; tE[2 * u2 + 1] = vE.y; tMI[2 * u2] = vMI.x; tMI[2 * u2 + 1] = vMI.y; tII[2 * u2] = vII.x; tII[2 * u2 + 1] = vII.y;
        tD1[2 * u2] = vD1.x; tD1[2 * u2 + 1] = vD1.y; tD2[2 * u2] = vD2.x; tD2[2 * u2 + 1] = vD2.y;
        em[2 * u2] = vem.x; em[2 * u2 + 1] = vem.y;
      }
      double oM[4], oI[4], oD[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int q = q0 + u;
        const double mm = em[u] * (pm1 * tA[u] + pi1 * tB[u] + pd1 * tC[u] + pB * tE[u]);
        const double ins = pM[u] * tMI[u] + pI[u] * tII[u];
        // D chain inside the lane with nothing entering from the left (the entering value is added below)
        dloc = q > 0 ? mprev * tD1[u] + dloc * tD2[u] : 0.0;
        if (q > 0) P *= tD2[u];
        oM[u] = mm; oI[u] = ins; oD[u] = dloc;
        esum += mm;
        pm1 = pM[u]; pi1 = pI[u]; pd1 = pD[u]; mprev = mm;
      }
#pragma unroll
      for (int u2 = 0; u2 < 2; u2++) {
        const size_t o = ofs2(q0 + 2 * u2, lane);
        st_d2(cr + o, oM[2 * u2], oM[2 * u2 + 1]); st_d2(cr + SQ + o, oI[2 * u2], oI[2 * u2 + 1]); st_d2(cr + 2 * SQ + o, oD[2 * u2], oD[2 * u2 + 1]);
      }
    }
    // cross-lane: Dlast(r) = [dloc_last + P * D1_0 * Mlast(r-1)] + [P * D2_0] * Dlast(r-1)
    // (shuffles stay outside conditionals: a lane that skips a ds_bpermute does not lend its value)
    const double mup = shfl_up_d(mprev, 1);
    const double mleft = lane > 0 ? mup : 0.0;
    const double d10 = m.tf[gD1 * SQ + ofs2(0, lane)], d20 = m.tf[gD2 * SQ + ofs2(0, lane)];
    double Bv = dloc + P * d10 * (lane > 0 ? mleft : 0.0), Av = Alane;
    for (int d = 1; d < 64; d <<= 1) {
      const double Bo = shfl_up_d(Bv, d), Ao = shfl_up_d(Av, d);
      if (lane >= d) { Bv = Bv + Av * Bo; Av = Av * Ao; }
    }
    const double dup = shfl_up_d(Bv, 1);
    const double dleft = lane > 0 ? dup : 0.0;                   // true D of lane-1's last node
    const double c0 = lane > 0 ? d10 * mleft + d20 * dleft : 0.0;  // true D of my first node
    double Pq = 1.0;
    for (int q0 = 0; q0 < Q; q0 += 4) {
      double dl[4], t2[4];
#pragma unroll
      for (int u2 = 0; u2 < 2; u2++) {
        const size_t o = ofs2(q0 + 2 * u2, lane);
        const d2_t vd = nt_load_d2(cr + 2 * SQ + o), vt = ld_d2(m.tf + gD2 * SQ + o);
        dl[2 * u2] = vd.x; dl[2 * u2 + 1] = vd.y; t2[2 * u2] = vt.x; t2[2 * u2 + 1] = vt.y;
      }
      double dv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (q0 + u > 0) Pq *= t2[u];
        dv[u] = dl[u] + Pq * c0;
        esum += dv[u];
      }
#pragma unroll
      for (int u2 = 0; u2 < 2; u2++) st_d2(cr + 2 * SQ + ofs2(q0 + 2 * u2, lane), dv[2 * u2], dv[2 * u2 + 1]);
    }
    double xe = wave_sum_d(esum);
    double xn = pN * c.loop, xc = pC * c.loop + xe * c.EC, xj = pJ * c.loop + xe * c.EJ, lsd = 0.0;
    if (xe > kRescaleHi) {
      const double r = 1.0 / xe;
      for (int q = 0; q < Q; q++) {
        const size_t o = (size_t)q * 64 + lane;      // every entry of the row once, in any order
        cr[o] = __builtin_nontemporal_load(cr + o) * r; cr[SQ + o] = __builtin_nontemporal_load(cr + SQ + o) * r; cr[2 * SQ + o] = __builtin_nontemporal_load(cr + 2 * SQ + o) * r;
      }
      xn *= r; xc *= r; xj *= r; lsd = log(xe); ls += lsd; xe = 1.0;
    }
    const double xb = xj * c.move + xn * c.move;
    if (lane == 0) {
      double *s = cr + 3 * SQ; s[xN] = xn; s[xB] = xb; s[xE] = xe; s[xJ] = xj; s[xC] = xc; s[xLS] = lsd; s[xCLS] = ls;
      if (xs) { double *t = xs + (size_t)i * xNSPEC; t[xN] = xn; t[xB] = xb; t[xE] = xe; t[xJ] = xj; t[xC] = xc; t[xLS] = lsd; t[xCLS] = ls; }
    }
    pN = xn; pB = xb; pJ = xj; pC = xc;
  }
  wave_mem_sync();
  return ls + log(pC * c.move);
}

// The same sweep for models of up to 64 * QC nodes with the DP row in REGISTERS (3 x QC doubles per lane + the
// running D2 products for the D fix-up): the version above reads row i-1 back from the slab - eight dependent
// memory round trips per row, ~29 000 cycles per row on the protein workload with 2 048 waves in flight - here
// the only loads are the model's tables (independent of the recurrence), rows are written out and never read
// back, and the neighbour lane's last node arrives by shuffle.  Same operations in the same order as gforward.
// TL: the eight transition arrays come from the workgroup's LDS copy <tl_> (same layout as tf_), the emission row of the
// residue from L2 as before.  The sweep was bound by its table reads - nine float64 arrays per cell, 72 B, from L2 once
// per row and wave; the LDS copy leaves one (the emission row) on that path.
template <int QC, bool TL = false>
__device__ __attribute__((noinline)) double gforward_reg(const double *tf_, const double *te_, int Q_, const uint8_t *seq, int L_, GLen c,
                                                         double *slab_, size_t rowlen, int lane, bool store_, const ldbl *tl_ = nullptr) {
  // (a called function receives its arguments in vector registers: what is wave-uniform is said again, and the
  // table / slab pointers are given their address space back, or every access becomes a flat_ instruction with a
  // per-lane address and every guard a divergent branch.  Inlined at its eight call sites the function cost the
  // kernel 491 spilled registers and ran 15 % slower than called.)
  // (the dispatch calls gforward_reg<QC> only with Q == QC: a compile-time Q leaves the row straight-line code, so the
  // table reads of all its 4-node groups can be requested together instead of one group per round trip)
  (void)Q_;
  constexpr int Q = QC;
  const int L = __builtin_amdgcn_readfirstlane(L_);
  const bool store = __builtin_amdgcn_readfirstlane((int)store_) != 0;
  const gdbl *tf = as_global(tf_), *te = as_global(te_);
  const ldbl *tl = TL ? as_lds(tl_) : nullptr;
  gdbl *slab = as_global(slab_);
  const size_t SQ = (size_t)Q * 64;
  auto tab2 = [&](int arr, size_t o) -> d2_t { if constexpr (TL) return ldl_d2(tl + (size_t)arr * SQ + o); else return ldg_d2(tf + (size_t)arr * SQ + o); };
  double M[QC], I[QC], D[QC], PQ[QC];
#pragma unroll
  for (int q = 0; q < QC; q++) { M[q] = 0.0; I[q] = 0.0; D[q] = 0.0; PQ[q] = 1.0; }
  double ls = 0.0;
  if (store) {
    gdbl *r0 = slab;
    for (int q = 0; q < Q; q++) { r0[(size_t)q * 64 + lane] = 0.0; r0[SQ + (size_t)q * 64 + lane] = 0.0; r0[2 * SQ + (size_t)q * 64 + lane] = 0.0; }
    if (lane == 0) { gdbl *s = r0 + 3 * SQ; s[xN] = 1.0; s[xB] = c.move; s[xE] = 0.0; s[xJ] = 0.0; s[xC] = 0.0; s[xLS] = 0.0; s[xCLS] = 0.0; }
  }
  double pN = 1.0, pB = c.move, pJ = 0.0, pC = 0.0;
  double Alane = 1.0;
  for (int q = 0; q < Q; q += 2) { const d2_t v = tab2(gD2, ofs2(q, lane)); Alane *= v.x; Alane *= v.y; }
  double lastM = 0.0, lastI = 0.0, lastD = 0.0;     // node Q of this lane in the previous row
  for (int i = 1; i <= L; i++) {
    // (nothing in the loop stores to LDS, so the compiler would hoist the 8 x QC table reads of the LDS variant out of it -
    // 192 registers at 12 cells per lane, reloaded from scratch 44 times per row: measured 20 % SLOWER than the L2 reads)
    if (TL) asm volatile("" ::: "memory");
    gdbl *cr = slab + (size_t)i * rowlen;
    const gdbl *od = te + (size_t)seq[i - 1] * SQ;
    const double um = shfl_up_d(lastM, 1), ui = shfl_up_d(lastI, 1), ud = shfl_up_d(lastD, 1);
    double pm1 = lane > 0 ? um : 0.0, pi1 = lane > 0 ? ui : 0.0, pd1 = lane > 0 ? ud : 0.0;
    double mprev = 0.0, dloc = 0.0, P = 1.0, esum = 0.0, d10 = 0.0, d20 = 0.0;
#pragma unroll
    for (int q0 = 0; q0 < QC; q0 += 4) {
      if (q0 < Q) {
        double tA[4], tB[4], tC[4], tE[4], tMI[4], tII[4], tD1[4], tD2[4], em[4];
#pragma unroll
        for (int u2 = 0; u2 < 2; u2++) {
          const size_t o = ofs2(q0 + 2 * u2, lane);
          const d2_t vA = tab2(gA, o), vB = tab2(gB, o), vC = tab2(gC, o), vE = tab2(gE, o);
          const d2_t vMI = tab2(gMI, o), vII = tab2(gII, o), vD1 = tab2(gD1, o), vD2 = tab2(gD2, o);
          const d2_t vem = ldg_d2(od + o);
          tA[2 * u2] = vA.x; tA[2 * u2 + 1] = vA.y; tB[2 * u2] = vB.x; tB[2 * u2 + 1] = vB.y; tC[2 * u2] = vC.x; tC[2 * u2 + 1] = vC.y;
          tE[2 * u2] = vE.x; tE[2 * u2 + 1] = vE.y; tMI[2 * u2] = vMI.x; tMI[2 * u2 + 1] = vMI.y; tII[2 * u2] = vII.x; tII[2 * u2 + 1] = vII.y;
          tD1[2 * u2] = vD1.x; tD1[2 * u2 + 1] = vD1.y; tD2[2 * u2] = vD2.x; tD2[2 * u2 + 1] = vD2.y;
          em[2 * u2] = vem.x; em[2 * u2 + 1] = vem.y;
        }
        if (q0 == 0) { d10 = tD1[0]; d20 = tD2[0]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int q = q0 + u;
          const double oM = M[q], oI = I[q], oD = D[q];
          const double mm = em[u] * (pm1 * tA[u] + pi1 * tB[u] + pd1 * tC[u] + pB * tE[u]);
          const double ins = oM * tMI[u] + oI * tII[u];
          dloc = q > 0 ? mprev * tD1[u] + dloc * tD2[u] : 0.0;
          if (q > 0) P *= tD2[u];
          PQ[q] = P;
          M[q] = mm; I[q] = ins; D[q] = dloc;
          esum += mm;
          pm1 = oM; pi1 = oI; pd1 = oD; mprev = mm;
        }
      }
    }
    const double mup = shfl_up_d(mprev, 1);
    const double mleft = lane > 0 ? mup : 0.0;
    double Bv = dloc + P * d10 * (lane > 0 ? mleft : 0.0), Av = Alane;
    for (int d = 1; d < 64; d <<= 1) {
      const double Bo = shfl_up_d(Bv, d), Ao = shfl_up_d(Av, d);
      if (lane >= d) { Bv = Bv + Av * Bo; Av = Av * Ao; }
    }
    const double dup = shfl_up_d(Bv, 1);
    const double dleft = lane > 0 ? dup : 0.0;
    const double c0 = lane > 0 ? d10 * mleft + d20 * dleft : 0.0;
#pragma unroll
    for (int q = 0; q < QC; q++)
      if (q < Q) { D[q] = D[q] + PQ[q] * c0; esum += D[q]; }
    double xe = wave_sum_d(esum);
    double xn = pN * c.loop, xc = pC * c.loop + xe * c.EC, xj = pJ * c.loop + xe * c.EJ, lsd = 0.0;
    if (xe > kRescaleHi) {
      const double r = 1.0 / xe;
#pragma unroll
      for (int q = 0; q < QC; q++) { M[q] *= r; I[q] *= r; D[q] *= r; }
      xn *= r; xc *= r; xj *= r; lsd = log(xe); ls += lsd; xe = 1.0;
    }
    const double xb = xj * c.move + xn * c.move;
#pragma unroll
    for (int q = 0; q < QC; q++)
      if (q == Q - 1) { lastM = M[q]; lastI = I[q]; lastD = D[q]; }
    if (store) {
#pragma unroll
      for (int q = 0; q < QC; q += 2)
        if (q < Q) {
          const size_t o = ofs2(q, lane);
          stg_d2(cr + o, M[q], M[q + 1]); stg_d2(cr + SQ + o, I[q], I[q + 1]); stg_d2(cr + 2 * SQ + o, D[q], D[q + 1]);
        }
      if (lane == 0) { gdbl *s = cr + 3 * SQ; s[xN] = xn; s[xB] = xb; s[xE] = xe; s[xJ] = xj; s[xC] = xc; s[xLS] = lsd; s[xCLS] = ls; }
    }
    pN = xn; pB = xb; pJ = xj; pC = xc;
  }
  wave_mem_sync();
  return ls + log(pC * c.move);
}

// tl != NULL: the model's eight transition arrays are staged in the workgroup's LDS at <tl> (models of up to 16 cells per lane)
template <bool STORE>
__device__ __forceinline__ double gforward_any(const GModel &m, const uint8_t *seq, int L, GLen c, const GMx &mx, int lane, const ldbl *tl = nullptr) {
  if (tl) {
    switch (m.Q) {
      case 4: return gforward_reg<4, true>(m.tf, m.te, m.Q, seq, L, c, mx.p, mx.rowlen, lane, STORE, tl);
      case 8: return gforward_reg<8, true>(m.tf, m.te, m.Q, seq, L, c, mx.p, mx.rowlen, lane, STORE, tl);
      case 12: return gforward_reg<12, true>(m.tf, m.te, m.Q, seq, L, c, mx.p, mx.rowlen, lane, STORE, tl);
      case 16: return gforward_reg<16, true>(m.tf, m.te, m.Q, seq, L, c, mx.p, mx.rowlen, lane, STORE, tl);
      default: break;
    }
  }
  switch (m.Q) {
    case 4: return gforward_reg<4>(m.tf, m.te, m.Q, seq, L, c, mx.p, mx.rowlen, lane, STORE);
    case 8: return gforward_reg<8>(m.tf, m.te, m.Q, seq, L, c, mx.p, mx.rowlen, lane, STORE);
    case 12: return gforward_reg<12>(m.tf, m.te, m.Q, seq, L, c, mx.p, mx.rowlen, lane, STORE);
    case 16: return gforward_reg<16>(m.tf, m.te, m.Q, seq, L, c, mx.p, mx.rowlen, lane, STORE);
    default: return gforward<STORE>(m, seq, L, c, mx, lane);
  }
}

}  // namespace
}  // namespace wh
