// Alignment kernel: what "hmmalign -o OUT HMM QUERY" + the reference's Stockholm decode
// produce for one (query, HMM) pair (witch_msa/gcmm/aligner.py:96-142; algorithm:
// SURVEY.md A.7): unihit-local Forward, Backward, posterior decoding, optimal-accuracy
// (MEA) fill and traceback.  Output per residue: 0-based match column, or -1 when the
// residue sits in an insert state or in the N/C flanks.
//
// One wavefront per pair.  The Forward rows, then the posteriors (in place), then the
// OA rows live in per-wave HBM slabs:
//   F write 8 + read 8, posterior write 8 + read 8, OA write 12 bytes per DP cell
// at full width (HBM-bound).  A query short enough for it runs everything after the Forward sweep on the
// 256 / 512 nodes around its dominant path instead (align_window below: sparse Forward spill, compact rows,
// mass certificate, full-width fallback), and the traceback of both chains consumes whole M -> M runs per
// memory round trip (oa_traceback).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wh_device.h"
#include "wh_align_log.h"
#include "wh_launch.h"

namespace wh {

enum { AL_PN = 0, AL_B, AL_E, AL_PJ, AL_PC, AL_S, AL_ML, AL_MH, AL_ON, AL_OB, AL_OE, AL_OJ, AL_OC, AL_WPC, AL_NARR };
static_assert(AL_NARR == kAlignSpecArrays, "wh_launch.h sizes the special-state rows");

__device__ __forceinline__ float gate(float t, float v) { return t > 0.f ? v : 0.f; }

__device__ __forceinline__ float scan_apply_max(const ScanC &c, float B) {
  B = fmaxf(B, c.s[0] * dppf<0x111>(0.f, B));
  B = fmaxf(B, c.s[1] * dppf<0x112>(0.f, B));
  B = fmaxf(B, c.s[2] * dppf<0x114>(0.f, B));
  B = fmaxf(B, c.s[3] * dppf<0x118>(0.f, B));
  B = fmaxf(B, c.s[4] * dppf<0x142, 0xA>(0.f, B));
  B = fmaxf(B, c.s[5] * dppf<0x143, 0xC>(0.f, B));
  return B;
}

__device__ __forceinline__ int wave_max_i32(int x) {
  for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(x, m); x = o > x ? o : x; }
  return x;
}

template <int Q>
__device__ __forceinline__ float cell_load(const float *slab, int row, int nstate, int s, int k) {
  const int pos = k - 1, ln = pos / Q, q = pos % Q;
  const float *p = slab + ((size_t)(row * nstate + s) * (Q / 4) + q / 4) * (kWave * 4) + ln * 4 + (q % 4);
  return __builtin_nontemporal_load(p);
}

template <int Q>
__device__ __forceinline__ float tab_load(const float *tab, int arr, int k) {
  const int pos = k - 1, ln = pos / Q, q = pos % Q;
  return tab[((size_t)(arr * (Q / 4) + q / 4) * kWave + ln) * 4 + (q % 4)];
}

#define SPR(idx)  (SPECG ? __builtin_nontemporal_load(spec + (idx)) : spec[idx])
#define SPRI(idx) (SPECG ? __builtin_nontemporal_load(reinterpret_cast<const int *>(spec) + (idx)) : reinterpret_cast<const int *>(spec)[idx])

// ------------------------------------------------------------------ OA traceback
// First maximum wins, candidate orders as in SURVEY.md A.7.  Shared by the full-width passes and the node window:
// <oa(row, state, k)> reads an OA cell, <tab(arr, k)> a forward-orientation transition gate, <estate(i, s1, k)>
// resolves the E state of row i (argmax over the row in HMMER's striped scan order).  PJ / PC: the special-state
// arrays that hold the posteriors of J and C.
// A path mostly runs along a diagonal (M -> M): lane t evaluates the choice at cell (i - t, k - t), a ballot finds
// the first cell whose choice is not M, and the run up to it is consumed at once - one memory round trip per run
// instead of one per residue.  Same comparisons in the same order: the path is the serial walk's.
template <bool SPECG, int PJ, int PC, class OA, class TAB, class EST>
__device__ __forceinline__ void oa_traceback(float *spec, int SP, int L, int M, int lane, int32_t *cols, float tNl, float tNm,
                                             float tEJ, float tEC, OA oa, TAB tab, EST estate) {
  enum { stS, stN, stB, stM, stI, stD, stE, stJ, stC };
  int s0 = stC, s1 = stS, i = L, k = 0;
  int guard = 4 * (L + M) + 16;
  // row 0 of the OA matrix is -inf (never stored): handled by the i == 1 tests below
  while (s0 != stS && guard-- > 0) {
    switch (s0) {
      case stC: {
        const float av = tNl * (SPR(AL_OC * SP + i - 1) + SPR(PC * SP + i)), bv = tEC * SPR(AL_OE * SP + i);
        s1 = bv > av ? stE : stC;
        break;
      }
      case stJ: {
        const float av = tNl * (SPR(AL_OJ * SP + i - 1) + SPR(PJ * SP + i)), bv = tEJ * SPR(AL_OE * SP + i);
        s1 = bv > av ? stE : stJ;
        break;
      }
      case stE: estate(i, s1, k); break;
      case stM: {
        const int it = i - lane, kt = k - lane;     // lane t: the cell t steps down the diagonal
        const bool valid = it >= 1 && kt >= 1;
        int best = 0;
        if (valid) {
          float path[4];
          path[0] = gate(tab(FW_E, kt), SPR(AL_OB * SP + it - 1));
          if (it > 1 && kt > 1) {
            path[1] = gate(tab(FW_A, kt), oa(it - 1, 0, kt - 1));
            path[2] = gate(tab(FW_B, kt), oa(it - 1, 1, kt - 1));
            path[3] = gate(tab(FW_C, kt), oa(it - 1, 2, kt - 1));
          } else if (kt > 1) {   // previous row is row 0: -inf behind an open gate, 0 behind a closed one
            path[1] = gate(tab(FW_A, kt), -INFINITY);
            path[2] = gate(tab(FW_B, kt), -INFINITY);
            path[3] = gate(tab(FW_C, kt), -INFINITY);
          } else { path[1] = 0.f; path[2] = 0.f; path[3] = 0.f; }
          if (path[1] > path[best]) best = 1;
          if (path[2] > path[best]) best = 2;
          if (path[3] > path[best]) best = 3;
        }
        const unsigned long long stop = __ballot(!valid || best != 1);
        const int t = stop ? __builtin_ctzll(stop) : kWave - 1;      // last cell of this run
        if (valid && lane <= t) cols[it - 1] = kt - 1;
        const int bt = __shfl(best, t);
        s1 = stop == 0 ? stM : bt == 0 ? stB : bt == 1 ? stM : bt == 2 ? stI : stD;
        k -= t + 1; i -= t + 1;
        break;
      }
      case stD: {
        const float av = k > 1 ? gate(tab(FW_D1, k), oa(i, 0, k - 1)) : 0.f;
        const float bv = k > 1 ? gate(tab(FW_D2, k), oa(i, 2, k - 1)) : 0.f;
        s1 = bv > av ? stD : stM;
        k--;
        break;
      }
      case stI: {
        const float pmv = i > 1 ? oa(i - 1, 0, k) : -INFINITY;
        const float piv = i > 1 ? oa(i - 1, 1, k) : -INFINITY;
        const float av = gate(tab(FW_MI, k), pmv), bv = gate(tab(FW_II, k), piv);
        s1 = bv > av ? stI : stM;
        i--;
        break;
      }
      case stB: {
        const float av = tNm * SPR(AL_ON * SP + i), bv = tNm * SPR(AL_OJ * SP + i);
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

// argmax over the M (">=": the later cell in HMMER's striped scan wins) and D (">") cells of one OA row; this lane
// holds the NQ cells of nodes node0 + 1 .. node0 + NQ.  A D cell wins only if it comes before every tied M cell AND
// no tied M follows it; M uses ">=", so any tied M scanned after the D takes over: M wins whenever one exists.
// Returns true for M; k = the node.
template <int NQ>
__device__ __forceinline__ bool estate_argmax(const float (&om)[NQ], const float (&odd)[NQ], int node0, int M, int &k) {
  const int Qh = (M - 1) / 4 + 1 < 2 ? 2 : (M - 1) / 4 + 1;   // HMMER's SSE stripe count
  float vmax = -INFINITY;
#pragma unroll
  for (int q = 0; q < NQ; q++)
    if (node0 + q < M) vmax = fmaxf(vmax, fmaxf(om[q], odd[q]));
  vmax = wave_max(vmax);
  int bestM = -1, bestD = -1;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int kk = node0 + q + 1;
    if (kk <= M) {
      const int qh = (kk - 1) % Qh, rh = (kk - 1) / Qh;
      if (om[q] == vmax) { const int pos = qh * 8 + rh; bestM = pos > bestM ? pos : bestM; }
      if (odd[q] == vmax) { const int pos = 0x3FFFFFFF - (qh * 8 + 4 + rh); bestD = pos > bestD ? pos : bestD; }
    }
  }
  bestM = wave_max_i32(bestM);
  bestD = wave_max_i32(bestD);
  const int pos = bestM >= 0 ? bestM : 0x3FFFFFFF - bestD;
  k = (pos % 8 % 4) * Qh + pos / 8 + 1;
  return bestM >= 0;
}

// ------------------------------------------------------------------ alignment on a node window
// A fragment query aligns to a stretch of the model: after the full-width Forward sweep (which records, in row 0 of
// the two mask arrays, the lane blocks its dominant path runs through - wh_device.h) Backward, the posteriors, the
// OA fill and the traceback run on the 64 * QB nodes around that stretch only (QB = 4 or 8 nodes per lane,
// register-resident tables gathered once per pair, compact rows in slab B).  Paths that leave the window are
// dropped, so every posterior is a lower bound and their total over the rows is L minus the dropped mass: the
// window result is kept only when that total equals L within float32 noise (the rule of wh_score7.hip's envelope
// sweep), otherwise the caller runs the full-width sweeps.  The window posteriors of N/J/C go to AL_B/AL_E (which
// nothing reads after the Forward sweep) and AL_WPC.
// For this attempt the Forward sweep spills its rows sparsely (a lane block is written only when one of its cells
// exceeds 2^-24 of the row's E; the row's 64-bit mask of written blocks is in AL_ML/AL_MH): an unwritten block reads
// as zero here, and what that drops is part of the mass the certificate measures.
typedef __attribute__((address_space(3))) float awl_f;
typedef __attribute__((address_space(3))) uint8_t awl_u8;
typedef __attribute__((address_space(1))) float awg_f;

struct AlnWinCtx {              // <= 16 dwords: passed in registers to the non-inlined sweep
  awl_f *emL, *trL, *spec3;     // emission rows, both transition orientations, special-state rows (LDS)
  awl_u8 *seq;
  const awg_f *emG;             // all emission rows (L2)
  awg_f *slabA, *specg;         // Forward rows of this wave; special-state rows in HBM (SPECG)
  int SP, lane;
};

constexpr float kAlnWinTol = 3e-6f;
constexpr float kAlnKeepScale = 5.9604645e-08f;   // 2^-24, as in wh_score7.hip
enum { AW_PN = AL_B, AW_PJ = AL_E, AW_PC = AL_WPC };

// 1: columns written; 0: the window lost mass (or left float32 range), nothing was written
template <int QB, int Q, bool SPECG>
__device__ __noinline__ int align_window(const AlnWinCtx c, int32_t *cols, int L, int M, int m0, int Lcap, int Klds, LenCfg cu, float invZ, unsigned long long *wcyc) {
  long long t_last = wcyc ? (long long)__builtin_readcyclecounter() : 0;
#define AW_TICK(slot) do { if (wcyc) { const long long t_now = __builtin_readcyclecounter(); if (c.lane == 0) atomicAdd(wcyc + (slot), (unsigned long long)(t_now - t_last)); t_last = t_now; } } while (0)
  static_assert(Q % QB == 0 && QB % 4 == 0, "a window lane must stay inside one forward lane block");
  constexpr int Q4 = Q / 4, B4 = QB / 4, TBL = Q * kWave;
  constexpr int R = QB == 4 ? 6 : 2;   // rows in flight (more spill registers inside the row loops)
  L = __builtin_amdgcn_readfirstlane(L); M = __builtin_amdgcn_readfirstlane(M); m0 = __builtin_amdgcn_readfirstlane(m0);
  Lcap = __builtin_amdgcn_readfirstlane(Lcap); Klds = __builtin_amdgcn_readfirstlane(Klds);
  const int lane = c.lane, SP = __builtin_amdgcn_readfirstlane(c.SP);
  float *spec = SPECG ? (float *)c.specg : (float *)c.spec3;
  const uint8_t *seq = (const uint8_t *)c.seq;
  const float *trF = (const float *)c.trL, *trB = trF + FW_NARR * TBL;
  float4 *Wpp = reinterpret_cast<float4 *>((float *)c.slabA + (size_t)(Lcap + 1) * 2 * TBL);   // [row][M,I][B4][64]
  float4 *Woa = Wpp + (size_t)(Lcap + 1) * 2 * B4 * kWave;                                      // [row][M,I,D][B4][64]
  const int n0 = kWave * Q - m0 - kWave * QB;          // 0-based position of the window's first node

  // ---------------- Backward + posteriors, reversed node order (lane 0 holds the window's last nodes)
  {
    int fwd[B4], out[B4];
    TransTab<QB, true> T;
    {
      const float4 *bw4 = reinterpret_cast<const float4 *>(trB);
#pragma unroll
      for (int p4 = 0; p4 < B4; p4++) {
        const int m4 = (m0 >> 2) + lane * B4 + p4;
        const int rev = (m4 % Q4) * kWave + m4 / Q4;
        const int jf = 16 * Q - 1 - m4;
        fwd[p4] = (jf % Q4) * kWave + jf / Q4;
        const int jw = 16 * QB - 1 - (lane * B4 + p4);
        out[p4] = (jw % B4) * kWave + jw / B4;
#pragma unroll
        for (int a = 0; a < BW_NARR; a++) T.v[a][p4] = bw4[a * Q4 * kWave + rev];
      }
    }
    const ScanC sc = scan_prepare(lane_product<QB, true>(T, BW_DD));
    const LdsF4 em4L((const float *)c.emL);
    const float4 *em4G = reinterpret_cast<const float4 *>((const float *)c.emG);
    float Mb[QB], Ib[QB];
#pragma unroll
    for (int p = 0; p < QB; p++) { Mb[p] = 0.f; Ib[p] = 0.f; }
    float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f, acc = 0.f, accs = 0.f;
    bool clamped = false;
    int S_next = 0;
    // the stored Forward cells of a row are requested R rows ahead: at two waves per SIMD one HBM round trip takes
    // as long as ~10 window rows of arithmetic
    float4 ring[R][2 * B4];
    unsigned havebits = 0;      // bit r: ring slot r holds a written lane block
    const int lanef = (16 * Q - 1 - ((m0 >> 2) + lane * B4)) / Q4;     // forward lane block of my nodes
    // every row issues the same memory operations (requests are never skipped: an unwritten block is loaded and
    // discarded, a request past row 1 re-reads row 1), so the compiler's vmcnt bookkeeping is exact and a row waits
    // for ITS cells only - with a conditional request the waits collapse to the last one issued
    auto request_row = [&](float4 (&slot)[2 * B4], int rs, int r) {
      const unsigned mword = (unsigned)(lanef < 32 ? SPRI(AL_ML * SP + r) : SPRI(AL_MH * SP + r));
      const unsigned have = (mword >> (lanef & 31)) & 1u;
      havebits = (havebits & ~(1u << rs)) | (have << rs);
      const float4 *row = reinterpret_cast<const float4 *>((const float *)c.slabA) + (size_t)r * (2 * Q4 * kWave);
#pragma unroll
      for (int p4 = 0; p4 < B4; p4++) { slot[p4] = nt_load4(row + fwd[p4]); slot[B4 + p4] = nt_load4(row + Q4 * kWave + fwd[p4]); }
    };
    auto row2 = [&](int i, float4 (&slot)[2 * B4], int rs, auto more) {
      asm volatile("" ::: "memory");
      float4 fm_c[B4], fi_c[B4];
      const bool have = (havebits >> rs) & 1u;
#pragma unroll
      for (int p4 = 0; p4 < B4; p4++) {
        fm_c[p4] = have ? slot[p4] : make_float4(0.f, 0.f, 0.f, 0.f);
        fi_c[p4] = have ? slot[B4 + p4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if constexpr (decltype(more)::value) request_row(slot, rs, i - R >= 1 ? i - R : 1);
      const int S_i = SPRI(AL_S * SP + i);
      const int dS = S_i - SPRI(AL_S * SP + i - 1);
      if (i < L) {
        mirror_scale<QB>(S_next - S_i, Mb, Ib, xJ, xC, xN);
        const int x = __builtin_amdgcn_readfirstlane((int)seq[i]);
        float part = 0.f;
        // the consumer sits inside each branch: a value live across the merge would be one flat load (wh_device.h)
        auto emit = [&](auto em_ld) {
#pragma unroll
          for (int p4 = 0; p4 < B4; p4++) {
            const float4 E = T.v[BW_E][p4];
            const float4 O = em_ld(p4);
            Mb[4 * p4 + 0] *= O.w; part = fmaf(E.x, Mb[4 * p4 + 0], part);
            Mb[4 * p4 + 1] *= O.z; part = fmaf(E.y, Mb[4 * p4 + 1], part);
            Mb[4 * p4 + 2] *= O.y; part = fmaf(E.z, Mb[4 * p4 + 2], part);
            Mb[4 * p4 + 3] *= O.x; part = fmaf(E.w, Mb[4 * p4 + 3], part);
          }
        };
        if (x < Klds) emit([&](int p4) { return em4L[x * (Q * 16) + fwd[p4]]; });
        else emit([&](int p4) { return em4G[(size_t)x * (Q * 16) + fwd[p4]]; });
        xB = wave_sum(part);
        xJ = fmaf(xJ, cu.loop, xB * cu.move);
        xC = xC * cu.loop;
        xN = fmaf(xN, cu.loop, xB * cu.move);
      }
      const float xE = fmaf(xC, cu.EC, xJ * cu.EJ);
      backward_cells<QB, true, false>(T, sc, Mb, Ib, xE);
      clamped |= clamp_backward<QB>(Mb, Ib, xB, xJ, xC, xN);
      const float s_i = invZ;
      const float s_p = ldexpf(invZ, -dS);
      float4 *orow = Wpp + (size_t)i * (2 * B4 * kWave);
#pragma unroll
      for (int p4 = 0; p4 < B4; p4++) {
        // position 4*p4+j (reversed order) is component 3-j of the forward-ordered vector
        const float m0v = (fm_c[p4].x * Mb[4 * p4 + 3]) * s_i, m1v = (fm_c[p4].y * Mb[4 * p4 + 2]) * s_i;
        const float m2v = (fm_c[p4].z * Mb[4 * p4 + 1]) * s_i, m3v = (fm_c[p4].w * Mb[4 * p4 + 0]) * s_i;
        const float i0v = (fi_c[p4].x * Ib[4 * p4 + 3]) * s_i, i1v = (fi_c[p4].y * Ib[4 * p4 + 2]) * s_i;
        const float i2v = (fi_c[p4].z * Ib[4 * p4 + 1]) * s_i, i3v = (fi_c[p4].w * Ib[4 * p4 + 0]) * s_i;
        nt_store4(orow + out[p4], m0v, m1v, m2v, m3v);
        nt_store4(orow + B4 * kWave + out[p4], i0v, i1v, i2v, i3v);
        acc += (m0v + m1v) + (m2v + m3v);
        acc += (i0v + i1v) + (i2v + i3v);
      }
      const float pn = SPR(AL_PN * SP + i - 1) * xN * cu.loop * s_p;
      const float pj = SPR(AL_PJ * SP + i - 1) * xJ * cu.loop * s_p;
      const float pc = SPR(AL_PC * SP + i - 1) * xC * cu.loop * s_p;
      accs += pn + pj + pc;
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) { spec[AW_PN * SP + i] = pn; spec[AW_PJ * SP + i] = pj; spec[AW_PC * SP + i] = pc; }
      __builtin_amdgcn_wave_barrier();
      S_next = S_i;
    };
#pragma unroll
    for (int r = 0; r < R; r++) request_row(ring[r], r, L - r >= 1 ? L - r : 1);
    int ib = L;
#pragma unroll 1
    for (; ib - (R - 1) >= 1; ib -= R) {
#pragma unroll
      for (int r = 0; r < R; r++) row2(ib - r, ring[r], r, std::true_type{});
    }
#pragma unroll
    for (int r = 0; r < R; r++) if (ib - r >= 1) row2(ib - r, ring[r], r, std::false_type{});
    const float mass = wave_sum(acc) + accs;
    if (clamped || !(fabsf((float)L - mass) <= kAlnWinTol * (float)L)) return 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  AW_TICK(1);

  // ---------------- optimal-accuracy fill on the window (forward node order)
  const float tNl = cu.loop > 0.f ? 1.f : 0.f, tNm = cu.move > 0.f ? 1.f : 0.f;
  const float tEJ = cu.EJ > 0.f ? 1.f : 0.f, tEC = cu.EC > 0.f ? 1.f : 0.f;
  {
    TransTab<QB, true> T;
    {
      const float4 *fw4 = reinterpret_cast<const float4 *>(trF);
#pragma unroll
      for (int g = 0; g < B4; g++) {
        const int j = (n0 >> 2) + lane * B4 + g;
        const int slot = (j % Q4) * kWave + j / Q4;
#pragma unroll
        for (int a = 0; a < FW_NARR; a++) T.v[a][g] = fw4[a * Q4 * kWave + slot];
      }
    }
    float allpass = 1.f;
#pragma unroll
    for (int g = 0; g < B4; g++) {
      const float4 d = T.v[FW_D2][g];
      if (!(d.x > 0.f && d.y > 0.f && d.z > 0.f && d.w > 0.f)) allpass = 0.f;
    }
    const ScanC sc = scan_prepare(allpass);
    float Mp[QB], Ip[QB], Dp[QB];
#pragma unroll
    for (int q = 0; q < QB; q++) { Mp[q] = -INFINITY; Ip[q] = -INFINITY; Dp[q] = -INFINITY; }
    float oN = 0.f, oB = 0.f, oJ = -INFINITY, oC = -INFINITY;
    if (lane == 0) {
      spec[AL_ON * SP] = 0.f; spec[AL_OB * SP] = 0.f; spec[AL_OE * SP] = -INFINITY;
      spec[AL_OJ * SP] = -INFINITY; spec[AL_OC * SP] = -INFINITY;
    }
    float4 ring[R][2 * B4];
    auto request_pp = [&](float4 (&slot)[2 * B4], int r) {
      const float4 *prow = Wpp + (size_t)r * (2 * B4 * kWave) + lane;
#pragma unroll
      for (int g = 0; g < B4; g++) { slot[g] = nt_load4(prow + g * kWave); slot[B4 + g] = nt_load4(prow + (B4 + g) * kWave); }
    };
    auto row3 = [&](int i, float4 (&slot)[2 * B4], auto more) {
      asm volatile("" ::: "memory");
      float4 pm4[B4], pi4[B4];
#pragma unroll
      for (int g = 0; g < B4; g++) { pm4[g] = slot[g]; pi4[g] = slot[B4 + g]; }
      if constexpr (decltype(more)::value) request_pp(slot, i + R <= L ? i + R : L);   // same operations every row (see above)
      const float mm1 = wave_shr1(Mp[QB - 1]), im1 = wave_shr1(Ip[QB - 1]), dm1 = wave_shr1(Dp[QB - 1]);
#pragma unroll
      for (int g = B4 - 1; g >= 0; g--) {
        const float4 A = T.v[FW_A][g], B = T.v[FW_B][g], C = T.v[FW_C][g], E = T.v[FW_E][g];
        const float4 MI = T.v[FW_MI][g], II = T.v[FW_II][g];
#pragma unroll
        for (int j = 3; j >= 0; j--) {
          const int q = 4 * g + j;
          const float pm = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mm1;
          const float pi_ = q > 0 ? Ip[q > 0 ? q - 1 : 0] : im1;
          const float pd = q > 0 ? Dp[q > 0 ? q - 1 : 0] : dm1;
          float sv = gate(f4get(E, j), oB);
          sv = fmaxf(sv, gate(f4get(A, j), pm));
          sv = fmaxf(sv, gate(f4get(B, j), pi_));
          sv = fmaxf(sv, gate(f4get(C, j), pd));
          const float ni = fmaxf(gate(f4get(MI, j), Mp[q]), gate(f4get(II, j), Ip[q])) + f4get(pi4[g], j);
          Mp[q] = sv + f4get(pm4[g], j);
          Ip[q] = ni;
        }
      }
      const float mn1 = wave_shr1(Mp[QB - 1]);
      float dprev = 0.f;
#pragma unroll
      for (int g = 0; g < B4; g++) {
        const float4 D1 = T.v[FW_D1][g], D2 = T.v[FW_D2][g];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int q = 4 * g + j;
          const float src = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mn1;
          dprev = fmaxf(gate(f4get(D1, j), src), gate(f4get(D2, j), dprev));
          Dp[q] = dprev;
        }
      }
      float carry = wave_shr1(scan_apply_max(sc, dprev));
      float rowmax = -INFINITY;
#pragma unroll
      for (int g = 0; g < B4; g++) {
        const float4 D2 = T.v[FW_D2][g];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int q = 4 * g + j;
          carry = gate(f4get(D2, j), carry);
          Dp[q] = fmaxf(Dp[q], carry);
          if (n0 + lane * QB + q < M) rowmax = fmaxf(rowmax, fmaxf(Mp[q], Dp[q]));
        }
      }
      const float xE = wave_max(rowmax);
      {
        const float a1 = tNl * (oJ + SPR(AW_PJ * SP + i)), b1 = tEJ * xE;
        oJ = a1 > b1 ? a1 : b1;
        const float a2 = tNl * (oC + SPR(AW_PC * SP + i)), b2 = tEC * xE;
        oC = a2 > b2 ? a2 : b2;
        oN = tNl * (oN + SPR(AW_PN * SP + i));
        const float a3 = tNm * oN, b3 = tNm * oJ;
        oB = a3 > b3 ? a3 : b3;
      }
      if (lane == 0) {
        spec[AL_ON * SP + i] = oN; spec[AL_OB * SP + i] = oB; spec[AL_OE * SP + i] = xE;
        spec[AL_OJ * SP + i] = oJ; spec[AL_OC * SP + i] = oC;
      }
      float4 *orow = Woa + (size_t)i * (3 * B4 * kWave) + lane;
#pragma unroll
      for (int g = 0; g < B4; g++) {
        nt_store4(orow + g * kWave, Mp[4 * g], Mp[4 * g + 1], Mp[4 * g + 2], Mp[4 * g + 3]);
        nt_store4(orow + (B4 + g) * kWave, Ip[4 * g], Ip[4 * g + 1], Ip[4 * g + 2], Ip[4 * g + 3]);
        nt_store4(orow + (2 * B4 + g) * kWave, Dp[4 * g], Dp[4 * g + 1], Dp[4 * g + 2], Dp[4 * g + 3]);
      }
    };
#pragma unroll
    for (int r = 0; r < R; r++) request_pp(ring[r], 1 + r <= L ? 1 + r : L);
    int ib = 1;
#pragma unroll 1
    for (; ib + (R - 1) <= L; ib += R) {
#pragma unroll
      for (int r = 0; r < R; r++) row3(ib + r, ring[r], std::true_type{});
    }
#pragma unroll
    for (int r = 0; r < R; r++) if (ib + r <= L) row3(ib + r, ring[r], std::false_type{});
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  AW_TICK(2);

  // ---------------- traceback on the compact rows (a cell in front of the window reads as 0, which is what the
  // fill saw there)
  {
    const float *oaf = reinterpret_cast<const float *>(Woa);
    auto oa = [&](int row, int st, int k) -> float {
      const int pw = k - 1 - n0;
      if (pw < 0 || pw >= kWave * QB) return 0.f;
      const int ln = pw / QB, q = pw % QB;
      return __builtin_nontemporal_load(oaf + ((size_t)(row * 3 + st) * B4 + q / 4) * (kWave * 4) + ln * 4 + (q % 4));
    };
    auto tab = [&](int arr, int k) -> float { return tab_load<Q>(trF, arr, k); };
    auto estate = [&](int i, int &s1, int &k) {
      const float4 *orow = Woa + (size_t)i * (3 * B4 * kWave) + lane;
      float om[QB], odd[QB];
#pragma unroll
      for (int g = 0; g < B4; g++) {
        const float4 m4 = nt_load4(orow + g * kWave), d4 = nt_load4(orow + (2 * B4 + g) * kWave);
        om[4 * g] = m4.x; om[4 * g + 1] = m4.y; om[4 * g + 2] = m4.z; om[4 * g + 3] = m4.w;
        odd[4 * g] = d4.x; odd[4 * g + 1] = d4.y; odd[4 * g + 2] = d4.z; odd[4 * g + 3] = d4.w;
      }
      s1 = estate_argmax<QB>(om, odd, n0 + lane * QB, M, k) ? 3 : 5;   // stM : stD
    };
    oa_traceback<SPECG, AW_PJ, AW_PC>(spec, SP, L, M, lane, cols, tNl, tNm, tEJ, tEC, oa, tab, estate);
  }
  AW_TICK(3);
  return 1;
}
#undef AW_TICK

// SWAP (long models, Q > 24): only ONE transition orientation is resident in LDS; the waves of a
// workgroup run the three sweeps in lockstep and swap the tables between them (see wh_score_big.hip).
// LOGSP: the fallback pass for pairs that left float32 range (wh_align_log.h).
template <int Q, bool TREG, bool SPECG, bool SWAP, bool LOGSP = false>
__global__ __launch_bounds__(SWAP ? 256 : 512) void align_kernel(AlignArgs a) {
  // all LDS in ONE 16-byte aligned dynamic array: a static __shared__ object in front of it
  // would shift the base by 4 bytes and split every ds_read_b128 (measured: 13x LDS time)
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  volatile int *s_item_p = reinterpret_cast<volatile int *>(smem_raw);
  float *smem = smem_raw + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  constexpr int Q4 = Q / 4;
  float *emL = smem;
  const int Klds = a.Klds;   // emission rows staged in LDS (K, or 0 when they are read from L2)
  float *trL = smem + (size_t)Klds * TBL;
  float *wbase = trL + (TREG ? 0 : (SWAP ? 8 : 2 * FW_NARR) * TBL) + (size_t)wave * a.wave_lds;
  float *spec = SPECG ? a.spec_scratch + ((size_t)blockIdx.x * nwaves + wave) * a.spec_stride : wbase;   // AL_NARR * SP floats
  uint8_t *seq = reinterpret_cast<uint8_t *>(wbase + (SPECG ? 0 : AL_NARR * a.SP));
  const int SP = a.SP;
  float *slabA = a.scratch + ((size_t)blockIdx.x * nwaves + wave) * a.scratch_stride;   // F -> posteriors
  float *slabB = slabA + (size_t)(a.Lcap + 1) * 2 * TBL;                                // OA rows
  int cur_h = -1, cur_orient = -1;
  const DevHMM *hm = nullptr;
  const float *fwG = nullptr, *bwG = nullptr, *emG = nullptr;
  // SWAP: every thread of the workgroup calls this at the same points
  auto orient = [&](int o) {
    if (SWAP && cur_orient != o) {
      __syncthreads();
      const float4 *src = reinterpret_cast<const float4 *>(o ? bwG : fwG);
      float4 *dst = reinterpret_cast<float4 *>(trL);
      for (int t = threadIdx.x; t < 8 * TBL / 4; t += blockDim.x) dst[t] = src[t];
      __syncthreads();
      cur_orient = o;
    }
  };
  float *const trF = trL, *const trB = SWAP ? trL : trL + FW_NARR * TBL;

  for (;;) {
    if (threadIdx.x == 0) *s_item_p = atomicAdd(a.counter, 1);
    __syncthreads();
    const int item = *s_item_p;
    __syncthreads();
    if (item >= a.n_items) break;
    const int h = a.item_h[item];
    const int p_lo = a.item_start[item], p_hi = p_lo + a.item_count[item];
    if (h != cur_h) {
      hm = a.hmms + h;
      fwG = a.tables + hm->fw_off; bwG = a.tables + hm->bw_off; emG = a.tables + hm->em_off;
      const float4 *src = reinterpret_cast<const float4 *>(a.tables + hm->em_off);
      float4 *dst = reinterpret_cast<float4 *>(emL);
      for (int t = threadIdx.x; t < Klds * TBL / 4; t += blockDim.x) dst[t] = src[t];
      cur_orient = -1;
      if (!TREG && !SWAP) {
        const float4 *s1 = reinterpret_cast<const float4 *>(a.tables + hm->fw_off);
        const float4 *s2 = reinterpret_cast<const float4 *>(a.tables + hm->bw_off);
        float4 *d1 = reinterpret_cast<float4 *>(trL);
        for (int t = threadIdx.x; t < FW_NARR * TBL / 4; t += blockDim.x) { d1[t] = s1[t]; d1[FW_NARR * TBL / 4 + t] = s2[t]; }
      }
      cur_h = h;
      __syncthreads();
    }
    const int M = hm->M;

    for (int pbase = p_lo; pbase < p_hi; pbase += nwaves) {
      const int pi = pbase + wave;
      bool active = pi < p_hi;
      const int pair = active ? a.order[pi] : 0;
      const int64_t qi = active ? a.pair_q[pair] : 0;
      const int64_t off = a.offsets[qi];
      const int L = active ? (int)(a.offsets[qi + 1] - off) : 0;
      int32_t *cols = a.cols + (active ? a.col_offsets[pair] : 0);
      for (int t = lane; t < L; t += kWave) cols[t] = -1;
      if (L <= 0 || L > a.Lcap) active = false;
      for (int t = lane; active && t < L; t += kWave) {
        int c = a.residues[off + t];
        seq[t] = (uint8_t)(c < a.Kp ? c : a.Kp - 1);
      }
      __builtin_amdgcn_wave_barrier();
      const LenCfg cu = len_config(L > 0 ? L : 1, false);
      const long long t_pair = a.wcyc ? (long long)__builtin_readcyclecounter() : 0;

      // ---------------- unihit Forward, rows spilled to slab A: sparsely for the node-window attempt (a query short
      // enough to fit one), at full width otherwise and when the window result was not accepted
      float xC_L = 0.f, lZ = -INFINITY; int ef_L = 0;
      orient(0);
      constexpr bool kWindows = !SWAP && !LOGSP && !TREG && Q >= 8;
      const bool sparse = kWindows && !a.no_window && L <= kWave * (Q % 8 == 0 && Q > 8 ? 8 : 4);
      if (!LOGSP && !sparse && active && a.wstat && lane == 0) atomicAdd(a.wstat + 2, 1);
#pragma unroll 1
      for (int attempt = sparse ? 0 : 1; attempt < 2 && active; attempt++) {
        {
          TransTab<Q, TREG> T;
          T.load(fwG, trF, lane);
          if constexpr (LOGSP) {
            lZ = forward_sweep_log<Q>(T, emL, emG, Klds, seq, L, cu, spec, SP, slabA, lane);
            xC_L = lZ > -INFINITY ? 1.f : 0.f;
          } else {
            const ScanC sc = scan_prepare(lane_product<Q, TREG>(T, FW_D2));
            // forward_sweep uses spec slots 0..7 = N,B,E,J,C,S,ML,MH with stride SP (AL_PN..AL_MH coincide)
            forward_sweep<Q, TREG, true>(T, sc, emL, emG, Klds, seq, L, cu, spec, SP, slabA, attempt == 0 ? kAlnKeepScale : -1.0f, lane, xC_L, ef_L);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (!(xC_L > 0.f)) { active = false; break; }   // no alignment has non-zero probability: all residues stay -1
        if (attempt == 1) break;

        // ---------------- the three remaining passes on a node window, when the dominant path fits one
        if constexpr (kWindows) {
          const unsigned long long um = ((unsigned long long)(unsigned)SPRI(AL_MH * SP) << 32) | (unsigned)SPRI(AL_ML * SP);
          int done = 0, tried = 0, nodes_w = 0;
          if (um != 0) {
            int lo = __builtin_ctzll(um), hi = 63 - __builtin_clzll(um);
            lo = lo > 1 ? lo - 2 : 0; hi = hi < 63 ? hi + 1 : 63;
            const int nodes = (hi - lo + 1) * Q;
            nodes_w = nodes;
            AlnWinCtx c;
            c.emL = (awl_f *)emL; c.trL = (awl_f *)trL; c.spec3 = SPECG ? nullptr : (awl_f *)spec; c.seq = (awl_u8 *)seq;
            c.emG = (const awg_f *)emG; c.slabA = (awg_f *)slabA; c.specg = SPECG ? (awg_f *)spec : nullptr;
            c.SP = SP; c.lane = lane;
            const float invZ = 1.0f / (xC_L * cu.move);
            unsigned long long *wcyc = a.wcyc;
            if (wcyc && lane == 0) atomicAdd(wcyc, (unsigned long long)(__builtin_readcyclecounter() - t_pair));
            if (nodes <= 4 * kWave) {
              tried = 1;
              done = align_window<4, Q, SPECG>(c, cols, L, M, min((63 - hi) * Q, kWave * (Q - 4)), a.Lcap, Klds, cu, invZ, wcyc);
            } else if (Q % 8 == 0 && Q > 8 && nodes <= 8 * kWave) {
              tried = 1;
              done = align_window<(Q % 8 == 0 ? 8 : 4), Q, SPECG>(c, cols, L, M, min((63 - hi) * Q, kWave * (Q - 8)), a.Lcap, Klds, cu, invZ, wcyc);
            }
          }
          if (a.wstat && lane == 0) atomicAdd(a.wstat + (done ? (nodes_w > 4 * kWave ? 3 : 0) : tried ? 1 : 2), 1);
          if (a.wcyc && tried && lane == 0) {      // slack histogram (blocks of Q nodes between the path's span and the window), WH_STATS
            const int slack = ((nodes_w > 4 * kWave ? 8 : 4) * kWave - nodes_w) / Q;
            atomicAdd(a.wstat + 12 + (done ? 0 : 8) + (slack > 7 ? 7 : slack), 1);
          }
          if (done) active = false;   // columns are written
        }
      }

      // ---------------- Backward + posterior decoding, in place over slab A
      orient(1);
      if (LOGSP && active) {
        if constexpr (LOGSP) {
          TransTab<Q, TREG> T;
          T.load(bwG, trB, lane);
          backward_posterior_log<Q>(T, emL, emG, Klds, seq, L, cu, lZ, slabA, lane,
              [&](int r, float &fN, float &fJ, float &fC) { fN = SPR(AL_PN * SP + r); fJ = SPR(AL_PJ * SP + r); fC = SPR(AL_PC * SP + r); },
              [&](int r, float pn, float pj, float pc) {
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) { spec[AL_PN * SP + r] = pn; spec[AL_PJ * SP + r] = pj; spec[AL_PC * SP + r] = pc; }
                __builtin_amdgcn_wave_barrier();
              });
        }
      } else if (active) {
        bool clamped = false;
        const float invZ = 1.0f / (xC_L * cu.move);
        TransTab<Q, TREG> T;
        T.load(bwG, trB, lane);
        const ScanC sc = scan_prepare(lane_product<Q, TREG>(T, BW_DD));
        float Mb[Q], Ib[Q];
#pragma unroll
        for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; }
        float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f;
#pragma unroll 1
        for (int i = L; i >= 1; i--) {
          asm volatile("" ::: "memory");
          float4 *row = reinterpret_cast<float4 *>(slabA) + (size_t)i * (2 * Q4 * kWave) + (kWave - 1 - lane);
          float4 fm4[Q4], fi4[Q4];
#pragma unroll
          for (int p4 = 0; p4 < Q4; p4++) {
            fm4[p4] = nt_load4(row + (Q4 - 1 - p4) * kWave);
            fi4[p4] = nt_load4(row + (Q4 + Q4 - 1 - p4) * kWave);
          }
          if (i < L) {
            // mirrored scaling (wh_device.h, "envelope Backward scaling")
            mirror_scale<Q>(SPRI(AL_S * SP + i + 1) - SPRI(AL_S * SP + i), Mb, Ib, xJ, xC, xN);
            float od[Q];
            load_em_rev<Q>(od, emL, emG, seq[i], Klds, lane);
            float part = 0.f;
#pragma unroll
            for (int p4 = 0; p4 < Q4; p4++) {
              const float4 E = T.ld(BW_E, p4);
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const int p = 4 * p4 + j;
                Mb[p] *= od[p];
                part = fmaf(f4get(E, j), Mb[p], part);
              }
            }
            xB = wave_sum(part);
            xJ = fmaf(xJ, cu.loop, xB * cu.move);
            xC = xC * cu.loop;
            xN = fmaf(xN, cu.loop, xB * cu.move);
          }
          float xE = fmaf(xC, cu.EC, xJ * cu.EJ);
          backward_cells<Q, TREG>(T, sc, Mb, Ib, xE);
          clamped |= clamp_backward<Q>(Mb, Ib, xB, xJ, xC, xN);
          const float s_i = invZ;
          const float s_p = ldexpf(invZ, SPRI(AL_S * SP + i - 1) - SPRI(AL_S * SP + i));
#pragma unroll
          for (int p4 = 0; p4 < Q4; p4++) {
            // position 4*p4+j (reversed order) is component 3-j of the forward-ordered vector
            nt_store4(row + (Q4 - 1 - p4) * kWave, (fm4[p4].x * Mb[4 * p4 + 3]) * s_i, (fm4[p4].y * Mb[4 * p4 + 2]) * s_i,
                      (fm4[p4].z * Mb[4 * p4 + 1]) * s_i, (fm4[p4].w * Mb[4 * p4 + 0]) * s_i);
            nt_store4(row + (Q4 + Q4 - 1 - p4) * kWave, (fi4[p4].x * Ib[4 * p4 + 3]) * s_i, (fi4[p4].y * Ib[4 * p4 + 2]) * s_i,
                      (fi4[p4].z * Ib[4 * p4 + 1]) * s_i, (fi4[p4].w * Ib[4 * p4 + 0]) * s_i);
          }
          const float pn = SPR(AL_PN * SP + i - 1) * xN * cu.loop * s_p;
          const float pj = SPR(AL_PJ * SP + i - 1) * xJ * cu.loop * s_p;
          const float pc = SPR(AL_PC * SP + i - 1) * xC * cu.loop * s_p;
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) { spec[AL_PN * SP + i] = pn; spec[AL_PJ * SP + i] = pj; spec[AL_PC * SP + i] = pc; }
          __builtin_amdgcn_wave_barrier();
        }
        // float32 range left: queue the pair for the log-space pass (this pass still writes its columns)
        if (clamped && a.redo_list && lane == 0) a.redo_list[atomicAdd(a.redo_count, 1)] = pair;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

      // ---------------- optimal-accuracy fill (A.7), rows to slab B
      const float tNl = cu.loop > 0.f ? 1.f : 0.f, tNm = cu.move > 0.f ? 1.f : 0.f;
      const float tEJ = cu.EJ > 0.f ? 1.f : 0.f, tEC = cu.EC > 0.f ? 1.f : 0.f;
      orient(0);
      if (active) {
        TransTab<Q, TREG> T;
        T.load(fwG, trF, lane);
        float allpass = 1.f;
#pragma unroll
        for (int q4 = 0; q4 < Q4; q4++) {
          const float4 d = T.ld(FW_D2, q4);
          if (!(d.x > 0.f && d.y > 0.f && d.z > 0.f && d.w > 0.f)) allpass = 0.f;
        }
        const ScanC sc = scan_prepare(allpass);
        float Mp[Q], Ip[Q], Dp[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) { Mp[q] = -INFINITY; Ip[q] = -INFINITY; Dp[q] = -INFINITY; }
        float oN = 0.f, oB = 0.f, oJ = -INFINITY, oC = -INFINITY;
        if (lane == 0) {
          spec[AL_ON * SP] = 0.f; spec[AL_OB * SP] = 0.f; spec[AL_OE * SP] = -INFINITY;
          spec[AL_OJ * SP] = -INFINITY; spec[AL_OC * SP] = -INFINITY;
        }
#pragma unroll 1
        for (int i = 1; i <= L; i++) {
          asm volatile("" ::: "memory");
          const float4 *prow = reinterpret_cast<const float4 *>(slabA) + (size_t)i * (2 * Q4 * kWave) + lane;
          float4 pm4[Q4], pi4[Q4];
#pragma unroll
          for (int q4 = 0; q4 < Q4; q4++) { pm4[q4] = nt_load4(prow + q4 * kWave); pi4[q4] = nt_load4(prow + (Q4 + q4) * kWave); }
          const float mm1 = wave_shr1(Mp[Q - 1]), im1 = wave_shr1(Ip[Q - 1]), dm1 = wave_shr1(Dp[Q - 1]);
#pragma unroll
          for (int q4 = Q4 - 1; q4 >= 0; q4--) {
            const float4 A = T.ld(FW_A, q4), B = T.ld(FW_B, q4), C = T.ld(FW_C, q4), E = T.ld(FW_E, q4);
            const float4 MI = T.ld(FW_MI, q4), II = T.ld(FW_II, q4);
#pragma unroll
            for (int j = 3; j >= 0; j--) {
              const int q = 4 * q4 + j;
              const float pm = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mm1;
              const float pi_ = q > 0 ? Ip[q > 0 ? q - 1 : 0] : im1;
              const float pd = q > 0 ? Dp[q > 0 ? q - 1 : 0] : dm1;
              float sv = gate(f4get(E, j), oB);
              sv = fmaxf(sv, gate(f4get(A, j), pm));
              sv = fmaxf(sv, gate(f4get(B, j), pi_));
              sv = fmaxf(sv, gate(f4get(C, j), pd));
              const float ni = fmaxf(gate(f4get(MI, j), Mp[q]), gate(f4get(II, j), Ip[q])) + f4get(pi4[q4], j);
              Mp[q] = sv + f4get(pm4[q4], j);
              Ip[q] = ni;
            }
          }
          const float mn1 = wave_shr1(Mp[Q - 1]);
          float dprev = 0.f;
#pragma unroll
          for (int q4 = 0; q4 < Q4; q4++) {
            const float4 D1 = T.ld(FW_D1, q4), D2 = T.ld(FW_D2, q4);
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int q = 4 * q4 + j;
              const float src = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mn1;
              dprev = fmaxf(gate(f4get(D1, j), src), gate(f4get(D2, j), dprev));
              Dp[q] = dprev;
            }
          }
          float carry = wave_shr1(scan_apply_max(sc, dprev));
          float rowmax = -INFINITY;
#pragma unroll
          for (int q4 = 0; q4 < Q4; q4++) {
            const float4 D2 = T.ld(FW_D2, q4);
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int q = 4 * q4 + j;
              carry = gate(f4get(D2, j), carry);
              Dp[q] = fmaxf(Dp[q], carry);
              if (lane * Q + q < M) rowmax = fmaxf(rowmax, fmaxf(Mp[q], Dp[q]));
            }
          }
          const float xE = wave_max(rowmax);
          {
            const float a1 = tNl * (oJ + SPR(AL_PJ * SP + i)), b1 = tEJ * xE;
            oJ = a1 > b1 ? a1 : b1;
            const float a2 = tNl * (oC + SPR(AL_PC * SP + i)), b2 = tEC * xE;
            oC = a2 > b2 ? a2 : b2;
            oN = tNl * (oN + SPR(AL_PN * SP + i));
            const float a3 = tNm * oN, b3 = tNm * oJ;
            oB = a3 > b3 ? a3 : b3;
          }
          if (lane == 0) {
            spec[AL_ON * SP + i] = oN; spec[AL_OB * SP + i] = oB; spec[AL_OE * SP + i] = xE;
            spec[AL_OJ * SP + i] = oJ; spec[AL_OC * SP + i] = oC;
          }
          float4 *orow = reinterpret_cast<float4 *>(slabB) + (size_t)i * (3 * Q4 * kWave) + lane;
#pragma unroll
          for (int q4 = 0; q4 < Q4; q4++) {
            nt_store4(orow + q4 * kWave, Mp[4 * q4], Mp[4 * q4 + 1], Mp[4 * q4 + 2], Mp[4 * q4 + 3]);
            nt_store4(orow + (Q4 + q4) * kWave, Ip[4 * q4], Ip[4 * q4 + 1], Ip[4 * q4 + 2], Ip[4 * q4 + 3]);
            nt_store4(orow + (2 * Q4 + q4) * kWave, Dp[4 * q4], Dp[4 * q4 + 1], Dp[4 * q4 + 2], Dp[4 * q4 + 3]);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

      // ---------------- traceback
      if (active) {
        auto oa = [&](int row, int st, int k) -> float { return cell_load<Q>(slabB, row, 3, st, k); };
        auto tab = [&](int arr, int k) -> float { return tab_load<Q>(fwG, arr, k); };
        auto estate = [&](int i, int &s1, int &k) {
          const float4 *orow = reinterpret_cast<const float4 *>(slabB) + (size_t)i * (3 * Q4 * kWave) + lane;
          float om[Q], odd[Q];
#pragma unroll
          for (int q4 = 0; q4 < Q4; q4++) {
            const float4 m4 = nt_load4(orow + q4 * kWave), d4 = nt_load4(orow + (2 * Q4 + q4) * kWave);
            om[4 * q4] = m4.x; om[4 * q4 + 1] = m4.y; om[4 * q4 + 2] = m4.z; om[4 * q4 + 3] = m4.w;
            odd[4 * q4] = d4.x; odd[4 * q4 + 1] = d4.y; odd[4 * q4 + 2] = d4.z; odd[4 * q4 + 3] = d4.w;
          }
          s1 = estate_argmax<Q>(om, odd, lane * Q, M, k) ? 3 : 5;   // stM : stD
        };
        oa_traceback<SPECG, AL_PJ, AL_PC>(spec, SP, L, M, lane, cols, tNl, tNm, tEJ, tEC, oa, tab, estate);
      }
    }
  }
}

template <int Q, bool TREG, bool SPECG, bool SWAP, bool LOGSP = false>
static hipError_t launch_one(const AlignArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&align_kernel<Q, TREG, SPECG, SWAP, LOGSP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((align_kernel<Q, TREG, SPECG, SWAP, LOGSP>), dim3(blocks), dim3(threads), lds, s, a);
  return hipGetLastError();
}

// log-space pass
template <bool SPECG>
static hipError_t launch_align_log_q(int Q, const AlignArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  switch (Q) {
    case 4:  return launch_one<4, false, SPECG, false, true>(a, blocks, threads, lds, s);
    case 8:  return launch_one<8, false, SPECG, false, true>(a, blocks, threads, lds, s);
    case 12: return launch_one<12, false, SPECG, false, true>(a, blocks, threads, lds, s);
    case 16: return launch_one<16, false, SPECG, false, true>(a, blocks, threads, lds, s);
    case 20: return launch_one<20, false, SPECG, false, true>(a, blocks, threads, lds, s);
    case 24: return launch_one<24, false, SPECG, false, true>(a, blocks, threads, lds, s);
    // long models: pass-synchronous table swapping like the prob-space pass
    case 28: return launch_one<28, false, true, true, true>(a, blocks, threads, lds, s);
    case 32: return launch_one<32, false, true, true, true>(a, blocks, threads, lds, s);
    case 36: return launch_one<36, false, true, true, true>(a, blocks, threads, lds, s);
    case 40: return launch_one<40, false, true, true, true>(a, blocks, threads, lds, s);
    case 44: return launch_one<44, false, true, true, true>(a, blocks, threads, lds, s);
    case 48: return launch_one<48, false, true, true, true>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

template <bool SPECG>
static hipError_t launch_align_q(int Q, const AlignArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  switch (Q) {
    case 4:  return launch_one<4, false, SPECG, false>(a, blocks, threads, lds, s);
    case 8:  return launch_one<8, false, SPECG, false>(a, blocks, threads, lds, s);
    case 12: return launch_one<12, false, SPECG, false>(a, blocks, threads, lds, s);
    case 16: return launch_one<16, false, SPECG, false>(a, blocks, threads, lds, s);
    case 20: return launch_one<20, false, SPECG, false>(a, blocks, threads, lds, s);
    case 24: return launch_one<24, false, SPECG, false>(a, blocks, threads, lds, s);
    // long models: pass-synchronous table swapping, special states always in HBM
    case 28: return launch_one<28, false, true, true>(a, blocks, threads, lds, s);
    case 32: return launch_one<32, false, true, true>(a, blocks, threads, lds, s);
    case 36: return launch_one<36, false, true, true>(a, blocks, threads, lds, s);
    case 40: return launch_one<40, false, true, true>(a, blocks, threads, lds, s);
    case 44: return launch_one<44, false, true, true>(a, blocks, threads, lds, s);
    case 48: return launch_one<48, false, true, true>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

// protein models of 20/24 cells per lane do not fit both orientations beside 20 emission rows:
// they run the pass-synchronous variant too
static hipError_t launch_align_swap_mid(int Q, const AlignArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  if (a.logsp) {
    if (Q == 20) return launch_one<20, false, true, true, true>(a, blocks, threads, lds, s);
    if (Q == 24) return launch_one<24, false, true, true, true>(a, blocks, threads, lds, s);
  } else {
    if (Q == 20) return launch_one<20, false, true, true>(a, blocks, threads, lds, s);
    if (Q == 24) return launch_one<24, false, true, true>(a, blocks, threads, lds, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_align(int Q, const AlignArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  if (a.swap && Q <= 24) return launch_align_swap_mid(Q, a, blocks, threads, lds, s);
  if (a.logsp) return a.spec_scratch ? launch_align_log_q<true>(Q, a, blocks, threads, lds, s) : launch_align_log_q<false>(Q, a, blocks, threads, lds, s);
  return a.spec_scratch ? launch_align_q<true>(Q, a, blocks, threads, lds, s) : launch_align_q<false>(Q, a, blocks, threads, lds, s);
}

}  // namespace wh
