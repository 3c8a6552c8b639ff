// Scoring kernel, phase-call variant: the same five sweeps as wh_score.hip (P1 multihit Forward,
// P2 multihit Backward + domain decoding, region scan, per envelope P3 unihit Forward with the
// sparse row spill and P4 unihit Backward + posterior accumulation -> null2), but every sweep is
// a separate non-inlined device function.  Each sweep then gets a register allocation of its own
// (the fused body keeps values of all phases alive and spills), which lets the kernel run at
// THREE waves per SIMD (168 VGPRs) instead of two: measured on MI355X, the DP rows are bound by
// how many waves can issue, not by a pipe, so the third wave is worth more than wider registers.
// (hmmsearch --max per pair: witch_msa/gcmm/algorithm.py:526-532; algorithm SURVEY.md A.2-A.6.)
#include <hip/hip_runtime.h>

#include "wh_device.h"
#include "wh_launch.h"

// The file can be compiled a second time into the same library under another namespace and
// entry-point name (-DWH_K7NS=... -DWH_K7LAUNCH=...): the A/B slot for compiler-flag and source
// experiments (tools/ab_score.py compares both in one process).
#ifndef WH_K7NS
#define WH_K7NS k7
#define WH_K7LAUNCH launch_score7
#endif

namespace wh {
namespace WH_K7NS {


constexpr float kKeepScale7 = 5.9604645e-08f;   // 2^-24, see wh_score.hip
constexpr float kMassTol7 = 2e-5f;
// ... and of a window sweep: the deviation |Ld - mass| / Ld of FULL-WIDTH sweeps stays below 3e-6 (float32 noise of the
// sums: 1.6 million envelopes of the headline workload, none above); a window is accepted only inside that noise band,
// so what it loses cannot be told from rounding (0.05 % of the windows are rejected and redone at full width)
constexpr float kWinTol7 = 3e-6f;
// the spill certificate where an envelope's Forward rows are stored on a band of lane blocks (spill_band below): a band cuts
// posterior mass on purpose, so its sweeps are held to the noise band like a window's (with 2e-5 a 266-row envelope of the
// reference's example data lost 2.9e-3 nats of its null2 correction: the oracle comparison allows 1e-3)
__device__ __forceinline__ float spill_tol(bool banded) { return banded ? kWinTol7 : kMassTol7; }
// model classes that keep the FW_P / BW_P arrays in LDS (0 = none).  Measured at three waves per
// SIMD on the headline workload: 557 ms with the arrays (Q <= 16) vs 554 ms without - the saved
// multiplies do not show, so the arrays stay out of LDS.
#ifndef WH_K7_MAXQP
#define WH_K7_MAXQP 0
#endif
constexpr int kMaxQP = WH_K7_MAXQP;
// WH_SLIM_SPEC (the two-queries-per-wave kernel, wh_score9.hip): six per-row arrays per query instead of eight - an
// envelope's Forward sweep keeps its two mask words in the slots of the B and E rows, which nothing reads after it
#ifdef WH_SLIM_SPEC
constexpr bool kSlim = true;
constexpr int kSpML = SP_B, kSpMH = SP_E, kSpArr = 6;
#else
constexpr bool kSlim = false;
constexpr int kSpML = SP_ML, kSpMH = SP_MH, kSpArr = SP_NARR;
#endif
constexpr bool kMaskedAcc = true;   // P4: only lanes that own a stored Forward block accumulate

__device__ __forceinline__ float flogsum0_v7(float b) {
  const float mx = b > 0.f ? b : 0.f, mn = b > 0.f ? 0.f : b;
  if (mn == -INFINITY || (mx - mn) >= 15.7f) return mx;
  const int idx = (int)((mx - mn) * 1000.0f);
  return mx + (float)log(1.0 + exp((double)-idx / 1000.0));
}

// what every sweep needs to find its tables and its per-wave LDS block
// LDS pointers cross the call boundary with their address space in the type: inside a
// non-inlined function a plain float* would be a flat pointer and every table read a flat_load.
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(1))) float glb_f;   // same for HBM pointers: flat accesses would tie vmcnt to lgkmcnt

struct WaveCtx {
  lds_f *emL;                 // emission rows of the canonical residues (LDS)
  const glb_f *emG;           // all emission rows (L2; degenerate codes)
  lds_f *fwL, *bwL;           // transition arrays, forward / reversed orientation (LDS)
  lds_f *spec;                // SP_NARR per-row arrays, stride SP (LDS) ...
  glb_f *specg;               // ... or, for long queries (SG sweeps), in a per-wave HBM region
  lds_f *n2tab;               // 32 floats (LDS)
  glb_f *Fs;                  // Forward-row slab of this wave (HBM)
  int SP, lane;
  int alpha;                  // K | Kp << 8 | Klds << 16: alphabet size, with degenerate codes, emission rows staged in LDS
  uint32_t degen;             // this lane's degenerate-code mask (lane = residue code)
};
// The context travels to the non-inlined sweeps in argument registers while it flattens to at most
// 16 dwords of plain 4/8-byte members.  Every variant the compiler passed by reference through
// scratch instead (a 17th dword, byte-sized members) faulted on gfx950 with this toolchain
// (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION; not fully understood - the 16-byte LenCfg argument
// IS passed by reference and works).  Hence the packed <alpha> word; tests/test_abi_host.py checks
// the generated ISA.
#if defined(__HIP_DEVICE_COMPILE__)
static_assert(sizeof(WaveCtx) <= 80, "WaveCtx must stay register-passed (see comment)");
#endif
__device__ __forceinline__ int ctxK(const WaveCtx &c) { return c.alpha & 255; }
__device__ __forceinline__ int ctxKp(const WaveCtx &c) { return (c.alpha >> 8) & 255; }
__device__ __forceinline__ int ctxKlds(const WaveCtx &c) { return (c.alpha >> 16) & 255; }
struct P4Out { float mass, domcorr; };
struct RegOut { int nenv, nreg, flags; };   // flags: WH_FLAG_* | multidomain mask of the stored regions << 8 (12 bytes: stays in return registers)

struct FwdOut { float xC; int ef; int nst; };      // C(L), its scale exponent; STORE: lane blocks the sweep stored (rows x kept lanes)

// ---------------------------------------------------------------- P1 / P3
// (the sweep without STORE is the multihit one, P1: it also leaves the dominant-path mask in n2tab[30..31] for P2's window)
constexpr int kUmSlot = 30;
template <int Q, bool STORE, int TH, bool SG>
__device__ __noinline__ FwdOut sweep_forward(const WaveCtx c, lds_u8 *seq3, int L, LenCfg cfg, float keep_scale, int keep_lanes = 63 << 8) {
  const uint8_t *seq = (const uint8_t *)seq3;
  TransTab<Q, false> T;
  T.load(nullptr, (const float *)c.fwL, c.lane);
  const ScanC sc = scan_prepare(lane_product<Q, false>(T, FW_D2));
  FwdOut o;
  constexpr bool UM = !STORE && !SG && Q >= 8;
  o.nst = 0;
  forward_sweep<Q, false, STORE, (Q <= kMaxQP), kSlim, UM, STORE, !SG>(T, sc, (const float *)c.emL, (const float *)c.emG, ctxKlds(c), seq, L, cfg, SG ? (float *)c.specg : (float *)c.spec, c.SP, (float *)c.Fs, keep_scale, c.lane, o.xC, o.ef,
                                                                  reinterpret_cast<unsigned *>((float *)c.n2tab) + kUmSlot, &o.nst, keep_lanes);
  if (SG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the rows were written by lane 0, every lane reads them next
  return o;
}

// ---------------------------------------------------------------- P2: multihit Backward + decoding (A.4)
// Overwrites spec[SP_E], spec[SP_B], spec[SP_N] rows with the per-row posteriors pe, pb, njc.
template <int Q, int TH, bool SG>
__device__ __noinline__ void sweep_backward_decode(const WaveCtx c, lds_u8 *seq3, int L, LenCfg cm, float invZ, int ef_L) {
  const uint8_t *seq = (const uint8_t *)seq3;
  const float *emL = (const float *)c.emL;
  TransTab<Q, false> T;
  T.load(nullptr, (const float *)c.bwL, c.lane);
  const ScanC sc = scan_prepare(lane_product<Q, false>(T, BW_DD));
  const int lane = c.lane, SP = c.SP;
  float *spec = SG ? (float *)c.specg : (float *)c.spec;
  // long-query mode: L1-bypassing loads (the slots are rewritten for every pair)
  auto ldf = [&](int idx) -> float { return SG ? __builtin_nontemporal_load(spec + idx) : spec[idx]; };
  auto ldi = [&](int idx) -> int { return SG ? __builtin_nontemporal_load(reinterpret_cast<const int *>(spec) + idx) : reinterpret_cast<const int *>(spec)[idx]; };
  float Mb[Q], Ib[Q];
#pragma unroll
  for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; }
  float xC = cm.move, xJ = 0.f, xN = 0.f, xB = 0.f;
  int eb = 0;
  // long-query mode: the Forward rows are read 64 at a time (RowsDown), the posteriors leave 64 at a time
  RowsDown<7> R;
  const int r_off[7] = {SP_S * SP, SP_S * SP, SP_E * SP, SP_B * SP, SP_N * SP, SP_J * SP, SP_C * SP};
  const int r_sh[7] = {0, -1, 0, 0, -1, -1, -1};
  float wE = 0.f, wB = 0.f, wN = 0.f;
  auto flush = [&](int lo) {                   // rows R.top - lane down to <lo>
    const int r = R.top - lane;
    if (r >= lo) {
      __builtin_nontemporal_store(wE, spec + SP_E * SP + r);
      __builtin_nontemporal_store(wB, spec + SP_B * SP + r);
      __builtin_nontemporal_store(wN, spec + SP_N * SP + r);
    }
  };
  if (SG) R.load(spec, r_off, r_sh, L, lane);
#pragma unroll 1
  for (int i = L; i >= 0; i--) {
    asm volatile("" ::: "memory");
    if (SG && R.spent(i)) { flush(i + 1); R.load(spec, r_off, r_sh, i, lane); }
    if (i < L) {
      xB = wave_sum(backward_emit<Q, false>(T, emL, (const float *)c.emG, seq[i], ctxKlds(c), lane, Mb));
      xJ = fmaf(xJ, cm.loop, xB * cm.move);
      xC = xC * cm.loop;
      xN = fmaf(xN, cm.loop, xB * cm.move);
    }
    float xE = fmaf(xC, cm.EC, xJ * cm.EJ);
    if (i >= 1) backward_cells<Q, false, (Q <= kMaxQP)>(T, sc, Mb, Ib, xE);
    const float big = fmaxf(xB, xN);
    if (big > kRescaleHi) {
      const int e = f32_exponent(big);
      const float r = pow2f_int(-e);
#pragma unroll
      for (int p = 0; p < Q; p++) { Mb[p] *= r; Ib[p] *= r; }
      xB *= r; xJ *= r; xC *= r; xN *= r; xE *= r;
      eb += e;
    }
    const float s_i = ldexpf(invZ, (SG ? R.s(0, i) : ldi(SP_S * SP + i)) + eb - ef_L);
    const float pe = (SG ? R.f(2, i) : ldf(SP_E * SP + i)) * xE * s_i;
    const float pb = (SG ? R.f(3, i) : ldf(SP_B * SP + i)) * xB * s_i;
    float njc = 0.f;
    if (i >= 1) {
      const float s_p = ldexpf(invZ, (SG ? R.s(1, i) : ldi(SP_S * SP + i - 1)) + eb - ef_L);
      njc = (SG ? R.f(4, i) : ldf(SP_N * SP + i - 1)) * xN;
      njc = fmaf(SG ? R.f(5, i) : ldf(SP_J * SP + i - 1), xJ, njc);
      njc = fmaf(SG ? R.f(6, i) : ldf(SP_C * SP + i - 1), xC, njc);
      njc = njc * cm.loop * s_p;
    }
    if (SG) {
      if (lane == R.top - i) { wE = pe; wB = pb; wN = njc; }
    } else {
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) { spec[SP_E * SP + i] = pe; spec[SP_B * SP + i] = pb; spec[SP_N * SP + i] = njc; }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (SG) { flush(0); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
}

// ---------------------------------------------------------------- P4: unihit Backward + posterior -> null2
// Returns the null2 correction of the envelope (A.5) in *domcorr and the posterior mass that
// reached the accumulators (the certificate for the sparse spill, see wh_score.hip).
template <int Q, int TH, bool SG>
__device__ __noinline__ P4Out sweep_backward_null2(const WaveCtx c, lds_u8 *eseq3, int Ld, LenCfg cu, float invZe, int ef_e, float mass_tol) {
  const uint8_t *eseq = (const uint8_t *)eseq3;
  const float *emL = (const float *)c.emL;
  TransTab<Q, false> T;
  T.load(nullptr, (const float *)c.bwL, c.lane);
  const ScanC sc = scan_prepare(lane_product<Q, false>(T, BW_DD));
  const int lane = c.lane, SP = c.SP;
  const float *spec = SG ? (const float *)c.specg : (const float *)c.spec;
  auto ldf = [&](int idx) -> float { return SG ? __builtin_nontemporal_load(spec + idx) : spec[idx]; };
  auto ldi = [&](int idx) -> int { return SG ? __builtin_nontemporal_load(reinterpret_cast<const int *>(spec) + idx) : reinterpret_cast<const int *>(spec)[idx]; };
  auto ldu = [&](int idx) -> unsigned { return SG ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(spec) + idx) : reinterpret_cast<const unsigned *>(spec)[idx]; };
  float Mb[Q], Ib[Q], fM[Q];
#pragma unroll
  for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; fM[p] = 0.f; }
  float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f, xfac = 0.f, fIs = 0.f;
  const int src = kWave - 1 - lane;   // the forward-order lane that owns my (reversed) cells
  int S_next = 0;                     // S(i+1), carried so that every row reads its exponent once
  RowsDown<7> R;                      // long-query mode: 64 rows of the per-row arrays per coalesced load
  const int r_off[7] = {kSpML * SP, kSpMH * SP, SP_S * SP, SP_S * SP, SP_N * SP, SP_J * SP, SP_C * SP};
  const int r_sh[7] = {0, 0, 0, -1, -1, -1, -1};
  if (SG) R.load(spec, r_off, r_sh, Ld, lane);
#pragma unroll 1
  for (int i = Ld; i >= 1; i--) {
    asm volatile("" ::: "memory");
    if (SG && R.spent(i)) R.load(spec, r_off, r_sh, i, lane);
    auto mask_word = [&]() -> unsigned { return SG ? (src < 32 ? R.u(0, i) : R.u(1, i)) : (src < 32 ? ldu(kSpML * SP + i) : ldu(kSpMH * SP + i)); };
    // Forward row i: with three or more waves per SIMD the other waves cover the HBM latency,
    // so the row is requested only after the cell update (TH >= 768; saves 2*Q registers
    // across backward_cells); with two waves it is requested first.
    float4 fm4[Q / 4], fi4[Q / 4];
    auto request_row = [&]() {
      const unsigned mword = mask_word();
      const bool have = (mword >> (src & 31)) & 1u;
      const float4 *row = reinterpret_cast<const float4 *>((const float *)c.Fs) + (size_t)i * (2 * (Q / 4) * kWave);
      if (have) {
#pragma unroll
        for (int p4 = 0; p4 < Q / 4; p4++) {
          fm4[p4] = nt_load4(row + fs_piece<Q, !SG>(src, Q / 4 - 1 - p4));
          fi4[p4] = nt_load4(row + fs_piece<Q, !SG>(src, Q / 4 + Q / 4 - 1 - p4));
        }
      } else {
#pragma unroll
        for (int p4 = 0; p4 < Q / 4; p4++) { fm4[p4] = make_float4(0.f, 0.f, 0.f, 0.f); fi4[p4] = fm4[p4]; }
      }
    };
    if (TH < 768) request_row();
    // mirrored scaling (wh_device.h, "envelope Backward scaling")
    const int S_i = SG ? R.s(2, i) : ldi(SP_S * SP + i);
    const int dS = S_i - (SG ? R.s(3, i) : ldi(SP_S * SP + i - 1));      // Forward rescale at row i (>= 0)
    if (i < Ld) {
      mirror_scale<Q>(S_next - S_i, Mb, Ib, xJ, xC, xN);
      xB = wave_sum(backward_emit<Q, false>(T, emL, (const float *)c.emG, eseq[i], ctxKlds(c), lane, Mb));
      xJ = fmaf(xJ, cu.loop, xB * cu.move);
      xC = xC * cu.loop;
      xN = fmaf(xN, cu.loop, xB * cu.move);
    }
    float xE = fmaf(xC, cu.EC, xJ * cu.EJ);
    backward_cells<Q, false, (Q <= kMaxQP)>(T, sc, Mb, Ib, xE);
    clamp_backward<Q>(Mb, Ib, xB, xJ, xC, xN);
    const float s_i = invZe;
    const float s_p = ldexpf(invZe, -dS);
    if (TH >= 768 && kMaskedAcc) {
      // only the lanes that own a stored block run the accumulation (the others would add zeros)
      asm volatile("" ::: "memory");
      const unsigned mword = mask_word();
      if ((mword >> (src & 31)) & 1u) {
        const float4 *row = reinterpret_cast<const float4 *>((const float *)c.Fs) + (size_t)i * (2 * (Q / 4) * kWave);
        float idot = 0.f;
#pragma unroll
        for (int p4 = 0; p4 < Q / 4; p4++) {
          // reversed order: component 3-j of the forward-ordered vector is position 4*p4+j
          const float4 fm = nt_load4(row + fs_piece<Q, !SG>(src, Q / 4 - 1 - p4));
          const float4 fi = nt_load4(row + fs_piece<Q, !SG>(src, Q / 4 + Q / 4 - 1 - p4));
          fM[4 * p4 + 0] = fmaf(fm.w * Mb[4 * p4 + 0], s_i, fM[4 * p4 + 0]);
          fM[4 * p4 + 1] = fmaf(fm.z * Mb[4 * p4 + 1], s_i, fM[4 * p4 + 1]);
          fM[4 * p4 + 2] = fmaf(fm.y * Mb[4 * p4 + 2], s_i, fM[4 * p4 + 2]);
          fM[4 * p4 + 3] = fmaf(fm.x * Mb[4 * p4 + 3], s_i, fM[4 * p4 + 3]);
          idot = fmaf(fi.w, Ib[4 * p4 + 0], idot); idot = fmaf(fi.z, Ib[4 * p4 + 1], idot);
          idot = fmaf(fi.y, Ib[4 * p4 + 2], idot); idot = fmaf(fi.x, Ib[4 * p4 + 3], idot);
        }
        fIs = fmaf(idot, s_i, fIs);
      }
    } else {
      if (TH >= 768) { asm volatile("" ::: "memory"); request_row(); }
      float idot = 0.f;
#pragma unroll
      for (int p4 = 0; p4 < Q / 4; p4++) {
        fM[4 * p4 + 0] = fmaf(fm4[p4].w * Mb[4 * p4 + 0], s_i, fM[4 * p4 + 0]);
        fM[4 * p4 + 1] = fmaf(fm4[p4].z * Mb[4 * p4 + 1], s_i, fM[4 * p4 + 1]);
        fM[4 * p4 + 2] = fmaf(fm4[p4].y * Mb[4 * p4 + 2], s_i, fM[4 * p4 + 2]);
        fM[4 * p4 + 3] = fmaf(fm4[p4].x * Mb[4 * p4 + 3], s_i, fM[4 * p4 + 3]);
        idot = fmaf(fi4[p4].w, Ib[4 * p4 + 0], idot); idot = fmaf(fi4[p4].z, Ib[4 * p4 + 1], idot);
        idot = fmaf(fi4[p4].y, Ib[4 * p4 + 2], idot); idot = fmaf(fi4[p4].x, Ib[4 * p4 + 3], idot);
      }
      fIs = fmaf(idot, s_i, fIs);
    }
    float nj = (SG ? R.f(4, i) : ldf(SP_N * SP + i - 1)) * xN;
    nj = fmaf(SG ? R.f(5, i) : ldf(SP_J * SP + i - 1), xJ, nj);
    nj = fmaf(SG ? R.f(6, i) : ldf(SP_C * SP + i - 1), xC, nj);
    S_next = S_i;
    xfac = fmaf(nj * cu.loop, s_p, xfac);
  }
  // null2[a] = sum_k fM_k o_k(a) + sum_k fI_k + f_NJC, all / Ld
  float sm = 0.f;
#pragma unroll
  for (int p = 0; p < Q; p++) sm += fM[p];
  sm = wave_sum(sm);
  const float si = wave_sum(fIs);
  const float mass = sm + si + xfac;
  P4Out o;
  o.mass = mass; o.domcorr = 0.f;
  if (!(fabsf((float)Ld - mass) <= mass_tol * (float)Ld)) return o;   // certificate failed: the caller redoes the envelope densely
  const float norm = 1.0f / (float)Ld;
  float mine = 1.0f;
  for (int x = 0; x < ctxK(c); x++) {
    float od[Q];
    load_em_rev<Q>(od, emL, (const float *)c.emG, x, ctxKlds(c), lane);
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < Q; p++) s = fmaf(fM[p], od[p], s);
    s = wave_sum(s);
    if (lane == x) mine = (s + si) * norm + xfac * norm;
  }
  float *n2tab = (float *)c.n2tab;
  __builtin_amdgcn_wave_barrier();
  if (lane < ctxK(c)) n2tab[lane] = mine;
  __builtin_amdgcn_wave_barrier();
  if (lane >= ctxK(c) && lane < ctxKp(c)) {
    // degenerate codes: unweighted mean of the canonical ratios; gap/*/~ -> 1
    const uint32_t m = c.degen;
    float s = 0.f; int n = 0;
    for (int x = 0; x < ctxK(c); x++) if (m & (1u << x)) { s += n2tab[x]; n++; }
    mine = n > 0 ? s / (float)n : 1.0f;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < ctxKp(c)) n2tab[lane] = logf(mine);
  __builtin_amdgcn_wave_barrier();
  float dc = 0.f;
  for (int t = lane; t < Ld; t += kWave) dc += n2tab[eseq[t]];
  o.domcorr = wave_sum(dc);
  return o;
}

// ---------------------------------------------------------------- P4 on a node window
// The envelope's Backward sweep restricted to the window of 64*QB consecutive nodes that holds every lane block the
// Forward sweep stored (the union of its per-row masks): a 150-residue query walks ~200 of a model's ~900 nodes, so
// the window sweep does a quarter of the cells.  Window lane r owns the reversed nodes m0 + r*QB .. + QB-1 (m0 a
// multiple of QB, QB a divisor of Q, so a window lane's nodes lie in ONE forward lane block); its eight transition
// arrays are gathered ONCE from the lane-blocked LDS copy into registers (8 * QB/4 float4), emission pieces and
// the stored Forward rows are fetched per row through per-lane offsets.  Restricting the Backward sweep drops the
// paths that leave the window; the Forward rows are exact, so every accumulated posterior is a lower bound and the
// SAME mass certificate that guards the sparse spill guards the window: mass must reach Ld (1 - tol), otherwise the
// caller runs the full-width sweep on the same rows.
// BWG (staged launches, wh_staged.hip): the reversed transition arrays are gathered from the table buffer in L2
// (c.specg points at them there) instead of an LDS copy - the window sweep reads them once per envelope.
template <int QB, int Q, int TH, bool SG, bool BWG = false>
__device__ __noinline__ P4Out sweep_backward_null2_win(const WaveCtx c, lds_u8 *eseq3, int Ld, LenCfg cu, float invZe, float mass_tol, int m0) {
  static_assert(Q % QB == 0 && QB % 4 == 0, "a window lane must stay inside one forward lane block");
  static_assert(!(SG && BWG), "c.specg cannot be both");
  constexpr int Q4 = Q / 4, B4 = QB / 4;
  const uint8_t *eseq = (const uint8_t *)eseq3;
  const int lane = c.lane, SP = c.SP, Klds = ctxKlds(c);
  const float *spec = SG ? (const float *)c.specg : (const float *)c.spec;
  auto ldf = [&](int idx) -> float { return SG ? __builtin_nontemporal_load(spec + idx) : spec[idx]; };
  auto ldi = [&](int idx) -> int { return SG ? __builtin_nontemporal_load(reinterpret_cast<const int *>(spec) + idx) : reinterpret_cast<const int *>(spec)[idx]; };
  auto ldu = [&](int idx) -> unsigned { return SG ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(spec) + idx) : reinterpret_cast<const unsigned *>(spec)[idx]; };
  // float4 slots: reversed piece p4 of this lane is slot rev[p4] of a reversed-order array, and (components
  // reversed) slot fwd[p4] of a forward-order array (emission rows, stored Forward rows)
  int fwd[B4], frow[B4];       // (frow: the same piece in a stored Forward row, whose lane blocks are contiguous - fs_piece, wh_device.h)
  TransTab<QB, true> T;
  {
    const float4 *bw4 = BWG ? reinterpret_cast<const float4 *>((const float *)c.specg) : reinterpret_cast<const float4 *>((const float *)c.bwL);
#pragma unroll
    for (int p4 = 0; p4 < B4; p4++) {
      const int m4 = (m0 >> 2) + lane * B4 + p4;
      const int rev = (m4 % Q4) * kWave + m4 / Q4;
      const int jf = 16 * Q - 1 - m4;
      fwd[p4] = (jf % Q4) * kWave + jf / Q4;
      frow[p4] = fs_piece<Q, !SG>(jf / Q4, jf % Q4);
#pragma unroll
      for (int a = 0; a < BW_NARR; a++) T.v[a][p4] = bw4[a * Q4 * kWave + rev];
    }
  }
  const int lanef = (16 * Q - 1 - ((m0 >> 2) + lane * B4)) / Q4;     // forward lane block of my nodes
  const ScanC sc = scan_prepare(lane_product<QB, true>(T, BW_DD));
  const LdsF4 em4L((const float *)c.emL);
  const float4 *em4G = reinterpret_cast<const float4 *>((const float *)c.emG);
  float Mb[QB], Ib[QB], fM[QB];
#pragma unroll
  for (int p = 0; p < QB; p++) { Mb[p] = 0.f; Ib[p] = 0.f; fM[p] = 0.f; }
  float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f, xfac = 0.f, fIs = 0.f;
  int S_next = 0;
  // the stored Forward cells of a row are requested ONE ROW AHEAD (2 * QB registers: affordable here, not at full
  // width), so their HBM round trip runs beside a whole row of arithmetic
  float4 fm_n[B4], fi_n[B4];
  bool have_n = false;
  // long-query mode: 64 rows of the per-row arrays per coalesced load; the masks are read a row ahead, so their
  // chunk is reloaded one row early (arrays 0 / 1 hold rows top - 1 - t)
  RowsDown<7> R;
  const int r_off[7] = {kSpML * SP, kSpMH * SP, SP_S * SP, SP_S * SP, SP_N * SP, SP_J * SP, SP_C * SP};
  const int r_sh[7] = {-1, -1, 0, -1, -1, -1, -1};
  if (SG) R.load(spec, r_off, r_sh, Ld, lane);
  auto request_row = [&](int r, bool first) {
    const unsigned mword = SG ? (first ? (lanef < 32 ? ldu(kSpML * SP + r) : ldu(kSpMH * SP + r)) : (lanef < 32 ? R.u(0, r + 1) : R.u(1, r + 1)))
                              : (lanef < 32 ? ldu(kSpML * SP + r) : ldu(kSpMH * SP + r));
    have_n = (mword >> (lanef & 31)) & 1u;
    if (have_n) {
      const float4 *row = reinterpret_cast<const float4 *>((const float *)c.Fs) + (size_t)r * (2 * Q4 * kWave);
#pragma unroll
      for (int p4 = 0; p4 < B4; p4++) { fm_n[p4] = nt_load4(row + frow[p4]); fi_n[p4] = nt_load4(row + (SG ? Q4 * kWave : Q4) + frow[p4]); }
    }
  };
  request_row(Ld, true);
#pragma unroll 1
  for (int i = Ld; i >= 1; i--) {
    asm volatile("" ::: "memory");
    if (SG && R.spent(i)) R.load(spec, r_off, r_sh, i, lane);
    float4 fm_c[B4], fi_c[B4];
#pragma unroll
    for (int p4 = 0; p4 < B4; p4++) { fm_c[p4] = fm_n[p4]; fi_c[p4] = fi_n[p4]; }
    const bool have = have_n;
    if (i > 1) request_row(i - 1, false);
    const int S_i = SG ? R.s(2, i) : ldi(SP_S * SP + i);
    const int dS = S_i - (SG ? R.s(3, i) : ldi(SP_S * SP + i - 1));
    if (i < Ld) {
      mirror_scale<QB>(S_next - S_i, Mb, Ib, xJ, xC, xN);
      const int x = __builtin_amdgcn_readfirstlane((int)eseq[i]);
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
    clamp_backward<QB>(Mb, Ib, xB, xJ, xC, xN);
    const float s_i = invZe;
    const float s_p = ldexpf(invZe, -dS);
    if (have) {
      float idot = 0.f;
#pragma unroll
      for (int p4 = 0; p4 < B4; p4++) {
        const float4 fm = fm_c[p4];
        const float4 fi = fi_c[p4];
        fM[4 * p4 + 0] = fmaf(fm.w * Mb[4 * p4 + 0], s_i, fM[4 * p4 + 0]);
        fM[4 * p4 + 1] = fmaf(fm.z * Mb[4 * p4 + 1], s_i, fM[4 * p4 + 1]);
        fM[4 * p4 + 2] = fmaf(fm.y * Mb[4 * p4 + 2], s_i, fM[4 * p4 + 2]);
        fM[4 * p4 + 3] = fmaf(fm.x * Mb[4 * p4 + 3], s_i, fM[4 * p4 + 3]);
        idot = fmaf(fi.w, Ib[4 * p4 + 0], idot); idot = fmaf(fi.z, Ib[4 * p4 + 1], idot);
        idot = fmaf(fi.y, Ib[4 * p4 + 2], idot); idot = fmaf(fi.x, Ib[4 * p4 + 3], idot);
      }
      fIs = fmaf(idot, s_i, fIs);
    }
    float nj = (SG ? R.f(4, i) : ldf(SP_N * SP + i - 1)) * xN;
    nj = fmaf(SG ? R.f(5, i) : ldf(SP_J * SP + i - 1), xJ, nj);
    nj = fmaf(SG ? R.f(6, i) : ldf(SP_C * SP + i - 1), xC, nj);
    S_next = S_i;
    xfac = fmaf(nj * cu.loop, s_p, xfac);
  }
  float sm = 0.f;
#pragma unroll
  for (int p = 0; p < QB; p++) sm += fM[p];
  sm = wave_sum(sm);
  const float si = wave_sum(fIs);
  const float mass = sm + si + xfac;
  P4Out o;
  o.mass = mass; o.domcorr = 0.f;
  if (!(fabsf((float)Ld - mass) <= mass_tol * (float)Ld)) return o;   // the window lost mass: the caller runs the full-width sweep
  const float norm = 1.0f / (float)Ld;
  float mine = 1.0f;
  for (int x = 0; x < ctxK(c); x++) {
    float s = 0.f;
    auto dot = [&](auto em_ld) {
#pragma unroll
      for (int p4 = 0; p4 < B4; p4++) {
        const float4 O = em_ld(p4);
        s = fmaf(fM[4 * p4 + 0], O.w, s); s = fmaf(fM[4 * p4 + 1], O.z, s);
        s = fmaf(fM[4 * p4 + 2], O.y, s); s = fmaf(fM[4 * p4 + 3], O.x, s);
      }
    };
    if (x < Klds) dot([&](int p4) { return em4L[x * (Q * 16) + fwd[p4]]; });
    else dot([&](int p4) { return em4G[(size_t)x * (Q * 16) + fwd[p4]]; });
    s = wave_sum(s);
    if (lane == x) mine = (s + si) * norm + xfac * norm;
  }
  float *n2tab = (float *)c.n2tab;
  __builtin_amdgcn_wave_barrier();
  if (lane < ctxK(c)) n2tab[lane] = mine;
  __builtin_amdgcn_wave_barrier();
  if (lane >= ctxK(c) && lane < ctxKp(c)) {
    const uint32_t m = c.degen;
    float s = 0.f; int n = 0;
    for (int x = 0; x < ctxK(c); x++) if (m & (1u << x)) { s += n2tab[x]; n++; }
    mine = n > 0 ? s / (float)n : 1.0f;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < ctxKp(c)) n2tab[lane] = logf(mine);
  __builtin_amdgcn_wave_barrier();
  float dc = 0.f;
  for (int t = lane; t < Ld; t += kWave) dc += n2tab[eseq[t]];
  o.domcorr = wave_sum(dc);
  return o;
}

// ---------------------------------------------------------------- P4 for FOUR envelopes at once (round 5)
// The window sweep above keeps 4 cells per lane: of its ~170 instructions per row some 100 are per-ROW work (the D->D
// scan, the reductions, the special states, the row requests) that 64 lanes x 4 cells cannot amortise, and it is one
// dependent chain per row - latency-bound at three waves per SIMD.  Here a wavefront carries FOUR envelopes, one per DPP
// row of 16 lanes: lane r of quarter t owns the 16 reversed nodes m0_t + 16 r .. + 15 of envelope t's 256-node window,
// every cross-lane step stays inside the quarter (row_shr scans, row sums), and the per-row work is shared by four
// envelopes - a row costs what a full-width Backward row costs (~300 instructions, 40 table reads) and serves four pairs.
// The transition arrays cannot live in registers at 16 cells per lane; they are read from the LDS copy through per-lane
// piece slots (four consecutive lanes of a quarter read 64 contiguous bytes: conflict-free).  Each envelope's Forward
// rows come from ITS slab (c.Fs + t * slab_stride), its per-row special states from ITS copy in HBM (c.specg + t *
// spec_stride: the caller wrote the six arrays there after the envelope's Forward sweep), both requested a row ahead,
// the stored-lane mask two rows ahead.  The arithmetic per cell is the window sweep's; the sums over a row are formed in
// another order (16 lanes x 16 cells instead of 64 x 4), so results agree with it to float32 rounding, not bitwise - the
// same relation the window sweep has to the full-width one.  Same mass certificate.
// slots (LDS, 16 ints per envelope): [0] active, [1] Ld, [2] m0, [3] 1/Z (float), [4] loop, [5] move (float), [6] byte offset of
// the envelope's residues inside <seqs>; outputs [8] mass, [9] domcorr (float), [10..] the null2 table scratch is n2 + 32 t.
enum { QS_ACTIVE = 0, QS_LD, QS_M0, QS_INVZ, QS_LOOP, QS_MOVE, QS_SEQ, QS_PAD, QS_MASS, QS_DOMCORR, QS_INTS = 16 };
template <int Q, int TH>
__device__ __noinline__ void sweep_backward_null2_quad(const WaveCtx c, lds_i *slots3, lds_u8 *seqs3, lds_f *n2, int slab_stride, int spec_stride, float mass_tol) {
  static_assert(Q % 4 == 0, "float4 pieces");
  constexpr int Q4 = Q / 4;
  const int lane = c.lane, SP = c.SP, Klds = ctxKlds(c);
  const int t = lane >> 4, r = lane & 15;
  lds_i *slot = slots3 + t * QS_INTS;
  const bool active = slot[QS_ACTIVE] != 0;
  // (a slot without an envelope holds whatever the last one left: every value an address is formed from is pinned here)
  const int Ld = active ? min(max(slot[QS_LD], 0), SP - 1) : 0;
  const int m0 = active ? min(max(slot[QS_M0], 0), 16 * (Q - 4) * 4) & ~15 : 0;
  const float invZe = __builtin_bit_cast(float, slot[QS_INVZ]);
  LenCfg cu;
  cu.loop = __builtin_bit_cast(float, slot[QS_LOOP]); cu.move = __builtin_bit_cast(float, slot[QS_MOVE]); cu.EJ = 0.0f; cu.EC = 1.0f;
  lds_u8 *eseq = seqs3 + (active ? slot[QS_SEQ] : 0);
  // HBM addresses = a wave-uniform base in scalar registers + this lane's 32-bit byte offset.  The four envelopes walk
  // their rows TOGETHER from their last row down (step j: envelope t is on row Ld_t - j), so that the row part of every
  // address is uniform (- j rows from the base) and the lane's offset holds what differs: its quarter's slab / copy of
  // the arrays, its last row, its piece.  (Per-lane 64-bit row pointers cost 26 registers and spilled; a scratch reload
  // between two row requests counts on vmcnt like they do and serialised them.)
  typedef __attribute__((address_space(1))) unsigned glb_u;
  typedef __attribute__((address_space(1))) v4f_t glb_v4;
  typedef __attribute__((address_space(1))) char glb_c;
  const glb_c *FsU = (const glb_c *)uniform_ptr((const glb_f *)c.Fs);
  const glb_c *specU = (const glb_c *)uniform_ptr((const glb_f *)c.specg);
  constexpr int kRowBytes = 32 * Q4 * kWave;             // one Forward row of a slab: M and I cells of 64 lane blocks
  const unsigned qspec = 4u * (unsigned)(t * __builtin_amdgcn_readfirstlane(spec_stride) + Ld);
  const unsigned qslab = 4u * (unsigned)(t * __builtin_amdgcn_readfirstlane(slab_stride)) + (unsigned)Ld * kRowBytes;
  // word <arr * SP + (row Ld_t - back)> of my quarter's copy of the per-row arrays, <back> uniform
  auto ldu = [&](int arr_word, int back) -> unsigned {
    return __builtin_nontemporal_load((const glb_u *)(uniform_ptr(specU + 4 * ((long)__builtin_amdgcn_readfirstlane(arr_word) - (long)__builtin_amdgcn_readfirstlane(back))) + qspec));
  };
  auto ldf_ = [&](int arr_word, int back) -> float { return __builtin_bit_cast(float, ldu(arr_word, back)); };
  const unsigned qmask = qspec + 4u * (unsigned)(((kWave - 1 - ((m0 >> 4) + r)) < 32 ? kSpML : kSpMH) * SP);   // ... of the mask word that holds my lane block's bit
  // This lane's 16 nodes are ONE forward lane block (Q = 16, m0 a multiple of 16): reversed piece p4 is slot p4 * 64 + lr of a
  // reversed array, and slot (3 - p4) * 64 + lanef of a forward-ordered one - one register each, the rest are immediates.
  static_assert(Q == 16, "the quarter-wave sweep is written for 16 cells per lane");
  const int lr = (m0 >> 4) + r, lanef = kWave - 1 - lr;
  struct TabRow {
    LdsF4 base; int lr;
    __device__ __forceinline__ TabRow(const float *lds, int lr_) : base(lds), lr(lr_) {}
    __device__ __forceinline__ float4 ld(int a, int p4) const { return base[(a * 4 + p4) * kWave + lr]; }
  } T((const float *)c.bwL, lr);
  float A = 1.f;
#pragma unroll
  for (int p4 = 0; p4 < 4; p4++) { const float4 d = T.ld(BW_DD, p4); A *= d.x; A *= d.y; A *= d.z; A *= d.w; }
  const ScanR sc = scan_prepare_row(A);
  const LdsF4 em4L((const float *)c.emL);
  const float4 *em4G = reinterpret_cast<const float4 *>((const float *)c.emG);
  float Mb[16], Ib[16], fM[16];
#pragma unroll
  for (int p = 0; p < 16; p++) { Mb[p] = 0.f; Ib[p] = 0.f; fM[p] = 0.f; }
  float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f, xfac = 0.f, fIs = 0.f;
  // Every step REQUESTS what the next step needs - the Forward cells of my lane block, the row's special states, and the
  // stored-lane mask word of the step after - in wave-uniform control flow and for every lane alike (an envelope that has
  // run out of rows reads the rows in front of its slab: the caller leaves one slab / one array of slack in front of the
  // first), so that the loaded registers carry no copies across the loop edge and are waited for only where the row uses
  // them, a whole row of arithmetic later.  (Requested inside the per-envelope branch, every loaded value was copied at
  // the branch's merge point - which waits for it: one exposed HBM round trip per row.)
  auto mask_word = [&](int back) -> unsigned {             // the mask word holding my lane block's bit, my row Ld - back
    return __builtin_nontemporal_load((const glb_u *)(uniform_ptr(specU - 4 * (long)__builtin_amdgcn_readfirstlane(back)) + qmask));
  };
  float4 fm[4], fi[4];
  const unsigned foff = qslab + 128u * (unsigned)lanef;      // byte offset of my block's cells on my LAST row (fs_piece, wh_device.h); piece p4: + 16 (3 - p4), the I cells + 64
  auto ld4 = [&](const glb_c *rowp, unsigned off) -> float4 {
    const v4f_t v = __builtin_nontemporal_load((const glb_v4 *)(rowp + off));
    return make_float4(v.x, v.y, v.z, v.w);
  };
  auto request_row = [&](int back) {                        // my row Ld - back
#pragma unroll
    for (int p4 = 0; p4 < 4; p4++) {
      const glb_c *rowp = uniform_ptr(FsU - (long)__builtin_amdgcn_readfirstlane(back) * kRowBytes + 16 * (3 - p4));
      fm[p4] = ld4(rowp, foff);
      fi[p4] = ld4(uniform_ptr(rowp + 64), foff);
    }
  };
  int Lmax = Ld;
  Lmax = max(Lmax, __shfl_xor(Lmax, 16));
  Lmax = max(Lmax, __shfl_xor(Lmax, 32));
  // step j works on row i = Ld - j of my envelope with S(i), S(i-1), N/J/C(i-1) and the mask word of row i in registers
  // (every loop-carried value below is DEFINED by a load and nothing else: "x = y; y = load" would make the compiler copy
  // the loaded register at the loop edge - a copy waits for its load - so the scale exponent of a row is loaded twice)
  int S_i, S_p;
  float n_p, j_p, c_p;
  unsigned w_i;
  request_row(0);
  w_i = mask_word(0);
  S_i = (int)ldu(SP_S * SP, 0);
  S_p = (int)ldu(SP_S * SP, 1);
  n_p = ldf_(SP_N * SP, 1);
  j_p = ldf_(SP_J * SP, 1);
  c_p = ldf_(SP_C * SP, 1);
#pragma unroll 1
  for (int j = 0; j < Lmax; j++) {
    asm volatile("" ::: "memory");
    const int i = Ld - j;
    if (i >= 1) {
      if (j > 0) {
        const int x = eseq[i];
        float part = 0.f;
        auto emit = [&](auto em_ld) {
#pragma unroll
          for (int p4 = 0; p4 < 4; p4++) {
            const float4 E = T.ld(BW_E, p4);
            const float4 O = em_ld(p4);
            Mb[4 * p4 + 0] *= O.w; part = fmaf(E.x, Mb[4 * p4 + 0], part);
            Mb[4 * p4 + 1] *= O.z; part = fmaf(E.y, Mb[4 * p4 + 1], part);
            Mb[4 * p4 + 2] *= O.y; part = fmaf(E.z, Mb[4 * p4 + 2], part);
            Mb[4 * p4 + 3] *= O.x; part = fmaf(E.w, Mb[4 * p4 + 3], part);
          }
        };
        if (x < Klds) emit([&](int p4) { return em4L[x * (Q * 16) + (3 - p4) * kWave + lanef]; });
        else emit([&](int p4) { return em4G[(size_t)x * (Q * 16) + (3 - p4) * kWave + lanef]; });
        xB = row_sum(part);
        xJ = fmaf(xJ, cu.loop, xB * cu.move);
        xC = xC * cu.loop;
        xN = fmaf(xN, cu.loop, xB * cu.move);
      }
      const float xE = fmaf(xC, cu.EC, xJ * cu.EJ);
      backward_cells_row(T, sc, Mb, Ib, xE);
      clamp_backward<16>(Mb, Ib, xB, xJ, xC, xN);
      const float s_i = invZe;
      const float s_p = ldexpf(invZe, S_p - S_i);       // the Forward rescale between rows i-1 and i
      if ((w_i >> (lanef & 31)) & 1u) {                 // the Forward sweep stored my lane block on this row
        float idot = 0.f;
#pragma unroll
        for (int p4 = 0; p4 < 4; p4++) {
          const float4 fm_ = fm[p4], fi_ = fi[p4];
          fM[4 * p4 + 0] = fmaf(fm_.w * Mb[4 * p4 + 0], s_i, fM[4 * p4 + 0]);
          fM[4 * p4 + 1] = fmaf(fm_.z * Mb[4 * p4 + 1], s_i, fM[4 * p4 + 1]);
          fM[4 * p4 + 2] = fmaf(fm_.y * Mb[4 * p4 + 2], s_i, fM[4 * p4 + 2]);
          fM[4 * p4 + 3] = fmaf(fm_.x * Mb[4 * p4 + 3], s_i, fM[4 * p4 + 3]);
          idot = fmaf(fi_.w, Ib[4 * p4 + 0], idot); idot = fmaf(fi_.z, Ib[4 * p4 + 1], idot);
          idot = fmaf(fi_.y, Ib[4 * p4 + 2], idot); idot = fmaf(fi_.x, Ib[4 * p4 + 3], idot);
        }
        fIs = fmaf(idot, s_i, fIs);
      }
      float nj = n_p * xN;
      nj = fmaf(j_p, xJ, nj);
      nj = fmaf(c_p, xC, nj);
      xfac = fmaf(nj * cu.loop, s_p, xfac);
      // the Forward rescale between this row and the next one down, applied to the carried state NOW (the window sweep
      // applies it at the top of the next row: the same exact power of two on the same values, but here no row begins by
      // waiting for a scale exponent that was requested a moment ago)
      mirror_scale<16>(S_i - S_p, Mb, Ib, xJ, xC, xN);
    }
    // (uniform) the next step's row, for every lane
    if (j + 1 < Lmax) {
      request_row(j + 1);
      w_i = mask_word(j + 1);
      S_i = (int)ldu(SP_S * SP, j + 1);
      S_p = (int)ldu(SP_S * SP, j + 2);
      n_p = ldf_(SP_N * SP, j + 2);
      j_p = ldf_(SP_J * SP, j + 2);
      c_p = ldf_(SP_C * SP, j + 2);
    }
  }
  float sm = 0.f;
#pragma unroll
  for (int p = 0; p < 16; p++) sm += fM[p];
  sm = row_sum(sm);
  const float si = row_sum(fIs);
  const float mass = sm + si + xfac;
  const bool ok = active && fabsf((float)Ld - mass) <= mass_tol * (float)Ld;
  lds_f *n2tab = n2 + 32 * t;
  const int K = ctxK(c), Kp = ctxKp(c);
  // (every quarter walks the null2 steps; one without an accepted envelope computes on zeros and its result is not read)
  const float norm = 1.0f / (float)(Ld > 0 ? Ld : 1);
  for (int x = 0; x < K; x++) {
    float s_ = 0.f;
    auto dot = [&](auto em_ld) {
#pragma unroll
      for (int p4 = 0; p4 < 4; p4++) {
        const float4 O = em_ld(p4);
        s_ = fmaf(fM[4 * p4 + 0], O.w, s_); s_ = fmaf(fM[4 * p4 + 1], O.z, s_);
        s_ = fmaf(fM[4 * p4 + 2], O.y, s_); s_ = fmaf(fM[4 * p4 + 3], O.x, s_);
      }
    };
    if (x < Klds) dot([&](int p4) { return em4L[x * (Q * 16) + (3 - p4) * kWave + lanef]; });
    else dot([&](int p4) { return em4G[(size_t)x * (Q * 16) + (3 - p4) * kWave + lanef]; });
    s_ = row_sum(s_);
    if (r == 0) n2tab[x] = (s_ + si) * norm + xfac * norm;
  }
  __builtin_amdgcn_wave_barrier();
  const uint32_t mdeg = (uint32_t)__shfl((int)c.degen, min(K + r, kWave - 1));      // (all lanes active here) the mask of "my" degenerate code
  if (K + r < Kp) {
    // degenerate codes (at most 16 of them: one per lane of the quarter): unweighted mean of the canonical ratios; gap/*/~ -> 1
    float s_ = 0.f; int n = 0;
    for (int x = 0; x < K; x++) if (mdeg & (1u << x)) { s_ += n2tab[x]; n++; }
    n2tab[K + r] = n > 0 ? s_ / (float)n : 1.0f;
  }
  __builtin_amdgcn_wave_barrier();
  for (int d = r; d < Kp; d += 16) n2tab[d] = logf(n2tab[d]);
  __builtin_amdgcn_wave_barrier();
  float dc = 0.f;
  for (int u = r; u < Ld; u += 16) dc += n2tab[eseq[u]];
  const float domcorr = row_sum(dc);
  if (r == 0) {
    slot[QS_MASS] = __builtin_bit_cast(int, mass);
    slot[QS_DOMCORR] = __builtin_bit_cast(int, ok ? domcorr : 0.f);
  }
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- P2 on a node window (round 4)
// The multihit Backward sweep exists for per-ROW numbers only: the posterior of a domain beginning / ending at each row
// and of the row's residue being emitted by a flank state, which the region scan compares with 0.25 / 0.10 / 0.20.
// Restricted to the window of 64*QB nodes around the dominant alignment (placed by P1's mask) it costs a third - and
// its answers are LOWER BOUNDS with a known slack: the Forward side is the full-width sweep's, so a windowed posterior
// is the probability of its event AND of the rest of the path staying inside the window; what is missing is at most
// eps = 1 - Z_window / Z, the probability that a path leaves the window at all, and the sweep measures exactly that at
// row 0.  (eps is ~7e-4 on the headline workload: a junk second mini-domain anywhere in the model.)  The region scan
// below therefore takes every threshold decision with that slack and reports whether all of them were beyond doubt;
// only then are its regions used - they are then the regions of the full-width sweep, decision by decision - otherwise
// P2 runs at full width as before.  The windowed posteriors go to three arrays of their own (tmp = spec + kSpArr * SP):
// the Forward rows of P1 stay intact for the full-width sweep.
// INPL (staged launches, wh_staged.hip): the wave's block is a COPY of P1's rows (the originals stay in HBM), so the
// three posterior rows overwrite the E, B and N rows in place as the full-width sweep does - row i reads E(i), B(i) and
// N(i-1), and N(i) was read one row earlier.  BWG: as in sweep_backward_null2_win.
struct WinDec { float eps; };
// TMPG (four envelopes per wave, score_kernel7q): the three posterior rows go to the wave's HBM region (c.specg), not to
// LDS - the wave's LDS block then has room for four queries' residues and records.
template <int QB, int Q, int TH, bool INPL = false, bool BWG = false, bool TMPG = false>
__device__ __noinline__ WinDec sweep_backward_decode_win(const WaveCtx c, lds_u8 *seq3, int L, LenCfg cm, float invZ, int ef_L, int m0) {
  static_assert(Q % QB == 0 && QB % 4 == 0, "a window lane must stay inside one forward lane block");
  static_assert(!(BWG && TMPG) && !(INPL && TMPG), "c.specg serves one purpose per call");
  constexpr int Q4 = Q / 4, B4 = QB / 4;
  const uint8_t *seq = (const uint8_t *)seq3;
  const int lane = c.lane, SP = c.SP, Klds = ctxKlds(c);
  const float *spec = (const float *)c.spec;
  float *tmp = (float *)c.spec + kSpArr * SP;          // [0] pe, [1] pb, [2] njc rows of the window sweep
  float *tE = INPL ? (float *)c.spec + SP_E * SP : tmp, *tB = INPL ? (float *)c.spec + SP_B * SP : tmp + SP, *tN = INPL ? (float *)c.spec + SP_N * SP : tmp + 2 * SP;
  glb_f *gE = c.specg, *gB = c.specg + SP, *gN = c.specg + 2 * SP;
  const int *speci = reinterpret_cast<const int *>(spec);
  int fwd[B4];
  TransTab<QB, true> T;
  {
    const float4 *bw4 = BWG ? reinterpret_cast<const float4 *>((const float *)c.specg) : reinterpret_cast<const float4 *>((const float *)c.bwL);
#pragma unroll
    for (int p4 = 0; p4 < B4; p4++) {
      const int m4 = (m0 >> 2) + lane * B4 + p4;
      const int rev = (m4 % Q4) * kWave + m4 / Q4;
      const int jf = 16 * Q - 1 - m4;
      fwd[p4] = (jf % Q4) * kWave + jf / Q4;
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
  float xC = cm.move, xJ = 0.f, xN = 0.f, xB = 0.f;
  int eb = 0;
  float ratio = 0.f;
#pragma unroll 1
  for (int i = L; i >= 0; i--) {
    asm volatile("" ::: "memory");
    if (i < L) {
      const int x = __builtin_amdgcn_readfirstlane((int)seq[i]);
      float part = 0.f;
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
      xJ = fmaf(xJ, cm.loop, xB * cm.move);
      xC = xC * cm.loop;
      xN = fmaf(xN, cm.loop, xB * cm.move);
    }
    float xE = fmaf(xC, cm.EC, xJ * cm.EJ);
    if (i >= 1) backward_cells<QB, true, false>(T, sc, Mb, Ib, xE);
    const float big = fmaxf(xB, xN);
    if (big > kRescaleHi) {
      const int e = f32_exponent(big);
      const float r = pow2f_int(-e);
#pragma unroll
      for (int p = 0; p < QB; p++) { Mb[p] *= r; Ib[p] *= r; }
      xB *= r; xJ *= r; xC *= r; xN *= r; xE *= r;
      eb += e;
    }
    const float s_i = ldexpf(invZ, speci[SP_S * SP + i] + eb - ef_L);
    const float pe = spec[SP_E * SP + i] * xE * s_i;
    const float pb = spec[SP_B * SP + i] * xB * s_i;
    float njc = 0.f;
    if (i >= 1) {
      const float s_p = ldexpf(invZ, speci[SP_S * SP + i - 1] + eb - ef_L);
      njc = spec[SP_N * SP + i - 1] * xN;
      njc = fmaf(spec[SP_J * SP + i - 1], xJ, njc);
      njc = fmaf(spec[SP_C * SP + i - 1], xC, njc);
      njc = njc * cm.loop * s_p;
    } else {
      ratio = spec[SP_N * SP] * xN * s_i;               // N_F(0) N_B(0) / Z: the share of the paths that stay inside the window
    }
    __builtin_amdgcn_wave_barrier();
    if (TMPG) { if (lane == 0) { gE[i] = pe; gB[i] = pb; gN[i] = njc; } }
    else if (lane == 0) { tE[i] = pe; tB[i] = pb; tN[i] = njc; }
    __builtin_amdgcn_wave_barrier();
  }
  WinDec o;
  o.eps = 1.0f - ratio;
  return o;
}

// The region scan (region_scan_global + the multidomain test) over the windowed posteriors, every decision taken with
// the slack of the window (header above): a windowed posterior p_w stands for a true value in [p_w, p_w + eps].
// Returns the regions and, in bit 24 of flags, whether any decision was in doubt (the caller then discards the result).
// The cumulative sums go back into the pe row (etot) and the njc row (btot): both are read at the row they are written.
template <int TH, bool INPL = false, bool TMPG = false>
__device__ __noinline__ RegOut region_scan_cert(lds_f *spec3, int SP, int L, lds_i *regs3, int lane, float eps, glb_f *tmpg = nullptr) {
  float *tmp = TMPG ? (float *)tmpg : (float *)spec3 + kSpArr * SP;
  float *tE = INPL ? (float *)spec3 + SP_E * SP : tmp, *tB = INPL ? (float *)spec3 + SP_B * SP : tmp + SP, *tN = INPL ? (float *)spec3 + SP_N * SP : tmp + 2 * SP;
  if (TMPG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // the rows were stored by lane 0, every lane reads them next
  int *regs = (int *)regs3;
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
  const float slack = 2e-5f;                // float32 rounding of the sums, on top of eps
  const float e1 = eps + slack, e2 = 2.0f * eps + slack;
  int nenv = 0, nreg = 0, flags = 0, doubt = 0, doubt_md = 0;
  float btot = 0.f, etot = 0.f;
  int i0 = -1;
  bool trig = false;
  // (TMPG: the rows live in HBM and are rewritten for every pair - L1-bypassing accesses, as the long-query sweeps use)
  auto ldt = [&](const float *p_) -> float { return TMPG ? __builtin_nontemporal_load(p_) : *p_; };
  auto stt = [&](float *p_, float v_) { if (TMPG) __builtin_nontemporal_store(v_, p_); else *p_ = v_; };
  const float pb0 = ldt(tB);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) { stt(tE, 0.f); stt(tN, 0.f); }
  for (int j0 = 1; j0 <= L; j0 += kWave) {
    const int jj = j0 + lane;
    const bool valid = jj <= L;
    const float nv = valid ? ldt(tN + jj) : 0.f;
    const float bv = valid ? (jj - 1 == 0 ? pb0 : ldt(tB + jj - 1)) : 0.f;
    const float ev = valid ? ldt(tE + jj) : 0.f;
    float jout = 0.f, cout = 0.f;
    const int cnt = L - j0 + 1 < kWave ? L - j0 + 1 : kWave;
    for (int t = 0; t < cnt; t++) {
      const int j = j0 + t;
      const float mocc = 1.0f - readlane_f(nv, t);          // true value in [mocc - eps, mocc]
      const float bold = btot, eold = etot;
      btot += readlane_f(bv, t);
      etot += readlane_f(ev, t);
      if (lane == t) { jout = btot; cout = etot; }
      if (!trig) {
        const float d = mocc - (btot - bold);               // true value in [d - 2 eps, d]
        if (d >= rt2 - slack && d < rt2 + e2) doubt = 1;
        if (d < rt2) i0 = j;
        else if (i0 == -1) i0 = j;
        if (mocc >= rt1 - slack && mocc < rt1 + e1) doubt = 1;
        if (mocc >= rt1) trig = true;
      } else {
        const float d = mocc - (etot - eold);
        if (d >= rt2 - slack && d < rt2 + e2) doubt = 1;
        if (d < rt2) {
          if (nenv < WH_MAX_ENVELOPES) {
            if (lane == 0) { regs[2 * nenv] = i0; regs[2 * nenv + 1] = j; }
            nenv++;
          } else flags |= WH_FLAG_TRUNC;
          nreg++;
          i0 = -1;
          trig = false;
        }
      }
    }
    if (valid) { stt(tN + jj, jout); stt(tE + jj, cout); }
  }
  __builtin_amdgcn_wave_barrier();
  if (TMPG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // the cumulative sums were stored by other lanes of this wave
  // multidomain test: max_z min(etot[z]-etot[i-1], btot[j]-btot[z-1]) >= rt3; each sum of windowed posteriors over the
  // region's rows falls short of the true one by at most (rows) x eps
  int multi_mask = 0;
  for (int e = 0; e < nenv; e++) {
    const int ri = regs[2 * e], rj = regs[2 * e + 1];
    float mx = -1.0f;
    const float e0 = ldt(tE + ri - 1), bj = ldt(tN + rj);
    for (int z = ri + lane; z <= rj; z += kWave) {
      const float u = ldt(tE + z) - e0, v = bj - ldt(tN + z - 1);
      mx = fmaxf(mx, fminf(u, v));
    }
    mx = wave_max(mx);
    const float up = mx + (float)(rj - ri + 1) * eps + slack;
    if (mx >= rt3) { flags |= WH_FLAG_MULTI; multi_mask |= 1 << e; if (mx < rt3 + slack) doubt_md = 1; }
    else if (up >= rt3) doubt_md = 1;
  }
  RegOut o;
  o.nenv = nenv; o.nreg = nreg; o.flags = flags | (multi_mask << 8) | (doubt << 24) | (doubt_md << 25);
  return o;
}

// ---------------------------------------------------------------- region scan (A.4)
template <int TH, bool SG>
__device__ __noinline__ RegOut region_scan(lds_f *spec3, glb_f *specg, int SP, int L, lds_i *regs3, int lane) {
  float *spec = SG ? (float *)specg : (float *)spec3;
  auto ldf = [&](int idx) -> float { return SG ? __builtin_nontemporal_load(spec + idx) : spec[idx]; };
  int *regs = (int *)regs3;
  const float rt3 = 0.20f;
  int nenv = 0, nreg = 0, flags = 0;
  // 64 rows per fetch, walked with v_readlane (wh_device.h): in HBM mode one round trip per 64 rows, in LDS mode no
  // ds_read latency inside the serial recurrence; the sums are formed in the row-by-row order either way
  region_scan_global(spec, SP, L, regs, lane, nenv, nreg, flags);
  __builtin_amdgcn_wave_barrier();
  if (SG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // multidomain test: max_z min(etot[z]-etot[i-1], btot[j]-btot[z-1]) >= rt3
  int multi_mask = 0;
  for (int e = 0; e < nenv; e++) {
    const int ri = regs[2 * e], rj = regs[2 * e + 1];
    float mx = -1.0f;
    const float e0 = ldf(SP_C * SP + ri - 1), bj = ldf(SP_J * SP + rj);
    for (int z = ri + lane; z <= rj; z += kWave) {
      const float u = ldf(SP_C * SP + z) - e0, v = bj - ldf(SP_J * SP + z - 1);
      mx = fmaxf(mx, fminf(u, v));
    }
    mx = wave_max(mx);
    if (mx >= rt3) { flags |= WH_FLAG_MULTI; multi_mask |= 1 << e; }
  }
  RegOut o;
  o.nenv = nenv; o.nreg = nreg; o.flags = flags | (multi_mask << 8);
  return o;
}

struct EnvCounters { unsigned n_w256, n_w512, n_wfail, n_full; unsigned long long spill; };   // ... and bytes of Forward rows the envelope sweeps stored
// An envelope's Backward sweep + null2 once its Forward rows are in c.Fs and its per-row arrays in the wave's block: on a
// node window where one fits around the dominant alignment and passes the mass certificate, else at full width.
// <dense>: the Forward sweep stored every row (the redo after a failed spill certificate): full width, no tolerance.
// <skip_window>: the caller has tried the window already (the four-envelopes-per-wave sweep).  The caller checks the
// spill certificate |Ld - mass| <= spill_tol Ld on the result of a sparse sweep.
// The dominant-path mask of an envelope's Forward sweep, for placing the window of its Backward sweep: nothing outside the
// band was stored, so the window need not reach a block out there that a chance diagonal made dominant on the envelope's
// first rows (headline: 256-node windows 86.6 -> 95.5 % of the envelopes, full width 5.7 -> 1.8 %).
__device__ __forceinline__ unsigned long long mask_in_band(unsigned long long um, int band) {
  const int blo = band & 255, bhi = (band >> 8) & 255;
  const unsigned long long bm = (bhi >= 63 ? ~0ull : ((1ull << (bhi + 1)) - 1)) & ~((1ull << blo) - 1);
  return (um & bm) ? (um & bm) : um;
}
template <int Q, int TH, bool SG>
__device__ __forceinline__ P4Out envelope_backward(const ScoreArgs &a, const WaveCtx &c, const uint8_t *eseq, int Ld, LenCfg cu, const FwdOut &f3,
                                                   bool dense, bool skip_window, EnvCounters &ec, int lane, int band) {
  const int SP = c.SP;
  const bool banded = band != (63 << 8);
  const float tol = dense ? INFINITY : spill_tol(banded);
  P4Out p4;
  bool have4 = false;
  if constexpr (Q >= 8) {
    if (!dense && !a.no_window && !skip_window) {
      // the node window around the lane blocks the dominant alignment runs through (two blocks in
      // front: the envelope's first ~25 rows set no bit and lie that many nodes ahead; one block behind)
      const unsigned *su = reinterpret_cast<const unsigned *>(SG ? (const float *)c.specg : (const float *)c.spec);
      unsigned long long um = ((unsigned long long)su[kSpMH * SP] << 32) | su[kSpML * SP];
      um = mask_in_band(um, band);
      if (um != 0) {
        int lo = __builtin_ctzll(um), hi = 63 - __builtin_clzll(um);
        lo = lo > 1 ? lo - 2 : 0; hi = hi < 63 ? hi + 1 : 63;
        const int nodes = (hi - lo + 1) * Q;
        if (a.stats && lane == 0) { atomicAdd(a.stats + 12, (unsigned long long)(hi - lo + 1)); atomicAdd(a.stats + 14, 1ull); atomicAdd(a.stats + 15, (unsigned long long)__builtin_popcountll(um)); }
        if (nodes <= 4 * kWave) {
          const int m0 = min((63 - hi) * Q, kWave * (Q - 4));
          p4 = sweep_backward_null2_win<4, Q, TH, SG>(c, (lds_u8 *)eseq, Ld, cu, 1.0f / (f3.xC * cu.move), kWinTol7, m0);
          have4 = fabsf((float)Ld - p4.mass) <= kWinTol7 * (float)Ld;
          if (have4) ec.n_w256++; else ec.n_wfail++;
          if (a.stats && lane == 0) {
            atomicAdd(a.stats + (have4 ? 0 : 2), 1ull);
            const float dev = fabsf((float)Ld - p4.mass) / (float)Ld;
            atomicAdd(a.stats + (dev < 3e-7f ? 16 : dev < 1e-6f ? 17 : dev < 3e-6f ? 18 : dev < 1e-5f ? 19 : dev < 2e-5f ? 20 : 21), 1ull);
          }
        } else if (Q % 8 == 0 && Q > 8 && nodes <= 8 * kWave) {
          const int m0 = min((63 - hi) * Q, kWave * (Q - 8));
          p4 = sweep_backward_null2_win<(Q % 8 == 0 ? 8 : 4), Q, TH, SG>(c, (lds_u8 *)eseq, Ld, cu, 1.0f / (f3.xC * cu.move), kWinTol7, m0);
          have4 = fabsf((float)Ld - p4.mass) <= kWinTol7 * (float)Ld;
          if (have4) ec.n_w512++; else ec.n_wfail++;
          if (a.stats && lane == 0) atomicAdd(a.stats + (have4 ? 1 : 2), 1ull);
        }
      }
    }
  }
  if (!have4) {
    p4 = sweep_backward_null2<Q, TH, SG>(c, (lds_u8 *)eseq, Ld, cu, 1.0f / (f3.xC * cu.move), f3.ef, tol);
    ec.n_full++;
    if (a.stats && lane == 0) {
      atomicAdd(a.stats + 3, 1ull);
      const float dev = fabsf((float)Ld - p4.mass) / (float)Ld;
      atomicAdd(a.stats + (dev < 3e-7f ? 22 : dev < 1e-6f ? 23 : dev < 3e-6f ? 24 : dev < 1e-5f ? 25 : dev < 2e-5f ? 26 : 27), 1ull);
    }
  }
  return p4;
}

// ---------------- A.6 score assembly (float32 where HMMER is float32): the sums over a pair's envelopes -> deci-bits
__device__ __forceinline__ void assemble_score(int L, int Ld_tot, float seqbias_sum, float sum_score, float sb2, float fwdsc, float nullsc,
                                               wh_pair_detail *dp, int &flags, int &decibits) {
  const double LOG2 = 0.69314718055994529;
  const float lomega = (float)log(1.0 / 256.0);
  const float seqbias = flogsum0_v7(lomega + seqbias_sum);
  float pre_score = (float)(((double)fwdsc - (double)nullsc) / LOG2);
  float seq_score = (float)(((double)fwdsc - (double)(nullsc + seqbias)) / LOG2);
  sb2 = flogsum0_v7(lomega + sb2);
  sum_score += (float)((double)(L - Ld_tot) * log((double)((float)L / (float)(L + 3))));
  const float pre2 = (float)(((double)sum_score - (double)nullsc) / LOG2);
  sum_score = (float)(((double)sum_score - (double)(nullsc + sb2)) / LOG2);
  if (Ld_tot > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2; flags |= WH_FLAG_OVERRIDE; }
  decibits = (int)rint((double)seq_score * 10.0);
  flags |= WH_FLAG_REPORTED;
  if (dp) { dp->seq_score = seq_score; dp->pre_score = pre_score; dp->seqbias_nats = seqbias; }
}

// ---------------------------------------------------------------- envelopes + score assembly (A.5, A.6)
// Everything a pair needs after its regions are known: per envelope P3 (sparse spill) -> P4 on a node window / at full
// width / dense redo -> null2; then HMMER's float32 score assembly, or the pair's record for the multidomain resolver.
// Inlined into its two callers: the fused kernel below, and the envelope kernel of the staged launches (wh_staged.hip),
// whose P1 / P2 ran as launches of their own.
#define WH_TICK7(slot) do { if (a.stats) { const long long t_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(a.stats + (slot), (unsigned long long)(t_now - t_last)); t_last = t_now; } } while (0)
// The lane blocks an envelope's Forward sweep stores on its first attempt: those around the dominant path of the pair's
// multihit Forward sweep (<um1>, P1's mask: lane blocks that held a cell above E(row)/2 on a sampled row).  Without it the
// first rows of EVERY envelope are stored at full width - until ~25 residues have matched, local entry keeps all 64 lane
// blocks above keep_scale x E(row) - and those rows were two thirds of all spill bytes on the headline workload
// (5.99e11 -> 2.2e11 B per 8 192 queries; kernel time -7 %).  The margins are in nodes: the true diagonal is not dominant
// on the envelope's first rows, so the path starts BELOW the lowest sampled block (80 nodes; 64 nodes gave 5x the dense
// redos) and may run on past the highest (48 nodes).  A band that cuts posterior mass fails the spill certificate like a
// keep_scale that is too coarse does, and the envelope is redone with every row stored at full width.
constexpr int kAllLanes = 63 << 8;
struct P1Mask { unsigned long long um; unsigned steady; };      // the union mask; lowest | highest << 8 | 1 << 16 of the steady blocks (0: none)
template <int Q>
__device__ __forceinline__ int spill_band(const ScoreArgs &a, P1Mask pm) {
  unsigned long long um1 = pm.um;
  // the band goes around the STEADY blocks of P1's mask (forward_sweep: dominance that carried over between two sampled rows -
  // a chance diagonal of the first rows does not), around the plain union when there are none
  const bool steady = pm.steady && !(a.spill_band & 2);      // (WH_SPILL_BAND=3: the band around the plain union, no cap)
  if (steady) um1 = (1ull << (pm.steady & 255)) | (1ull << ((pm.steady >> 8) & 255));
  if (Q < 8 || um1 == 0 || !a.spill_band) return kAllLanes;
  int below = (80 + Q - 1) / Q, above = (48 + Q - 1) / Q;
  if (a.spill_band >= 1000) { below = (a.spill_band / 1000 + Q - 1) / Q; above = (a.spill_band % 1000 + Q - 1) / Q; }   // (development: margins in nodes, below * 1000 + above)
  const int lo = __builtin_ctzll(um1) - below, hi = 63 - __builtin_clzll(um1) + above;
  int band = (lo > 0 ? lo : 0) | ((hi < 63 ? hi : 63) << 8);
  // the cap that rises with the row (forward_sweep): the steady blocks begin where the alignment is ~25 rows old, so on row i
  // it cannot be above block <lowest + i / Q> by more than the deletions it holds: 32 nodes of slack (32 / 48 / 64 / 80: 366 / 367 /
  // 368 / 368 ms, without the cap 371; rejected windows 3 700 - 3 400 of 1.6 million)
  if (steady && !(a.spill_band & 4)) band |= (__builtin_ctzll(um1) + 1 + (32 + Q - 1) / Q) << 16;      // (WH_SPILL_BAND=5: no cap)
  return band;
}
__device__ __forceinline__ P1Mask p1_mask(const WaveCtx &c) {
  const unsigned *su = reinterpret_cast<const unsigned *>((const float *)c.n2tab) + kUmSlot;
  return P1Mask{((unsigned long long)su[1] << 32) | su[0], su[-1]};
}
// One envelope: Forward sweep (rows stored for the Backward sweep) -> Backward sweep + null2, by up to three attempts at
// what is stored.  0: the lane blocks of <band> that pass the keep rule, certificate at the noise band (a band cuts mass on
// purpose);  1: every lane block that passes the keep rule, certificate 2e-5 (the store of rounds 1-4: an envelope whose
// band failed gets exactly the result it had then - on SURVEY's family sketch 1 % of the pairs carry 1e-5 of their posterior
// mass on junk nodes far from the alignment, whatever the margins);  2: every row at full width, no certificate (WH_FLAG_EXACT).
// <first>: where to begin (0 with a band, 1 without; callers whose own first attempt failed pass the next one).
// Returns the null2 correction; <envsc> is the envelope's Forward score (the same from every attempt).
template <int Q, int TH, bool SG>
__device__ __forceinline__ float envelope_attempts(const ScoreArgs &a, WaveCtx &c, const uint8_t *eseq, int Ld, LenCfg cu, int band, int first,
                                                   EnvCounters &ec, int lane, int &flags, long long &t_last, float &envsc) {
  const double LOG2 = 0.69314718055994529;
  float domcorr = 0.f;
#pragma unroll 1
  for (int attempt = first; attempt < 3; attempt++) {
    const float keep_scale = attempt < 2 ? (a.keep_scale > 0.f ? a.keep_scale : kKeepScale7) : -1.0f;
    const FwdOut f3 = sweep_forward<Q, true, TH, SG>(c, (lds_u8 *)eseq, Ld, cu, keep_scale, attempt == 0 ? band : kAllLanes);
    ec.spill += (unsigned long long)f3.nst * (8 * Q);
    // the rows were written by other lanes of this wave: order the stores before the loads
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    envsc = (float)((double)f3.ef * LOG2 + log((double)(f3.xC * cu.move)));
    domcorr = 0.f;
    if (!(f3.xC > 0.f)) break;
    WH_TICK7(7);
    const P4Out p4 = envelope_backward<Q, TH, SG>(a, c, eseq, Ld, cu, f3, attempt == 2, false, ec, lane, attempt == 0 ? band : kAllLanes);
    domcorr = p4.domcorr;
    WH_TICK7(8);
    if (attempt < 2 && !(fabsf((float)Ld - p4.mass) <= spill_tol(attempt == 0) * (float)Ld)) continue;
    if (attempt == 2) flags |= WH_FLAG_EXACT;
    break;
  }
  return domcorr;
}

template <int Q, int TH, bool SG>
__device__ __forceinline__ void score_envelopes(const ScoreArgs &a, WaveCtx &c, uint8_t *seq, int *regs, int L, int lane, int h, int64_t qi, int nenv, int nreg,
                                                int multi_mask, float fwdsc, float nullsc, float fwd_bits_out, wh_pair_detail *dp, int &flags, int &decibits,
                                                EnvCounters &ec, long long &t_last, P1Mask um1) {
  const int band = spill_band<Q>(a, SG ? P1Mask{0ull, 0u} : um1);      // (long queries: P1 keeps no mask; the band gained nothing there, 2.5 % of the bytes on the protein slice)
  {
  // ---------------- envelopes
  const LenCfg cu = len_config(L, false);
  float seqbias_sum = 0.f, sum_score = 0.f, sb2 = 0.f;
  int Ld_tot = 0;
  // a pair with a multidomain region is finished by resolve_kernel (A.4b); its single-domain
  // regions are still scored here, their results staged in LDS for the pair's queue record
  const bool queue_pair = multi_mask != 0 && a.rrecs != nullptr;
  float *envres = reinterpret_cast<float *>(regs + 3 * WH_MAX_ENVELOPES);
  for (int e = 0; e < nenv; e++) {
    if (queue_pair && ((multi_mask >> e) & 1)) { if (lane == 0) { envres[e] = 0.f; envres[WH_MAX_ENVELOPES + e] = 0.f; } continue; }
    const int ri = regs[2 * e], rj = regs[2 * e + 1];
    const int Ld = rj - ri + 1;
    const uint8_t *eseq = seq + (ri - 1);
    float envsc = -INFINITY;
    const float domcorr = envelope_attempts<Q, TH, SG>(a, c, eseq, Ld, cu, band, band != kAllLanes ? 0 : 1, ec, lane, flags, t_last, envsc);
    seqbias_sum += domcorr;
    if (envsc - domcorr > 0.0f) { sum_score += envsc; Ld_tot += Ld; sb2 += domcorr; }
    if (dp) { dp->env_i[e] = ri; dp->env_j[e] = rj; dp->envsc[e] = envsc; dp->domcorr[e] = domcorr; }
    if (queue_pair && lane == 0) { envres[e] = envsc; envres[WH_MAX_ENVELOPES + e] = domcorr; }
  }
  if (queue_pair) {
    __builtin_amdgcn_wave_barrier();
    int slot = 0;
    if (lane == 0) slot = atomicAdd(a.rcount, 1);
    slot = __shfl(slot, 0);
    if (slot < a.rcap && lane == 0) {
      ResolveRec *rr = a.rrecs + slot;
      rr->q = qi; rr->h = h; rr->fwdsc = fwdsc; rr->fwd_bits = fwd_bits_out; rr->nreg = nreg; rr->nenv = nenv;
      rr->multi_mask = multi_mask; rr->flags = flags;
      for (int e = 0; e < nenv; e++) { rr->ri[e] = regs[2 * e]; rr->rj[e] = regs[2 * e + 1]; rr->envsc[e] = envres[e]; rr->domcorr[e] = envres[WH_MAX_ENVELOPES + e]; }
    }
    // provisional result: resolve_kernel writes the final score and flags of this pair
  } else assemble_score(L, Ld_tot, seqbias_sum, sum_score, sb2, fwdsc, nullsc, dp, flags, decibits);
  }
}

// ---------------------------------------------------------------- P2 + region scan of a pair (after its P1)
struct FrontState {               // per wave: the window heuristics of the multihit Backward sweep, and its path counters
  float eps_prev, eps_prev2;      // slack of the last two windows tried on the current model
  unsigned n_pairs_h;             // pairs of the current model this wave has scored
  unsigned n_p2w, n_p2rej;        // sweeps kept from a window / windows in doubt (redone at full width)
};
// TMPG (score_kernel7q: no room in the wave's LDS block for three more rows): P1's six per-row arrays are COPIED to the wave's
// HBM region <tmpg> (18 coalesced stores), the windowed sweep then writes its posteriors IN PLACE over the E, B and N rows
// (INPL, as in the staged launches) and the certified scan reads them there; only when the scan ends in doubt do the rows
// come back from the copy for the full-width sweep.  (Keeping the three posterior rows in HBM instead cost 80 000 cycles
// per pair in the scan: every 64-row chunk a round trip, two fences.)
template <int Q, int TH, bool SG, bool TMPG>
__device__ __forceinline__ RegOut score_regions(const ScoreArgs &a, const WaveCtx &c, uint8_t *seq, int *regs, int L, int lane, LenCfg cm, const FwdOut &f1,
                                                FrontState &fs, long long &t_last, glb_f *tmpg = nullptr) {
  const int SP = c.SP;
  bool saved = false;
  // ---------------- P2 + region scan: on a node window when every decision of the scan is then beyond doubt
  RegOut ro;
  bool have_ro = false;
  if constexpr (Q >= 8 && !SG) {
    // (the slack of a window is mostly the model's: a junk mini-domain costs what its weakest nodes allow.  Rows x
    // slack must stay below the 0.20 of the multidomain test for a single-domain region to be certified, so a wave
    // that has measured a slack too large for this query length TWICE in a row skips the window on the model's next
    // pairs and probes again every sixteenth)
    const bool try_win = fminf(fs.eps_prev, fs.eps_prev2) * (float)L < 0.17f || (fs.n_pairs_h & 15) == 0;
    fs.n_pairs_h++;
    if ((a.p2win == 1 || TMPG) && !a.no_window && try_win) {
      const unsigned *su = reinterpret_cast<const unsigned *>((const float *)c.n2tab) + kUmSlot;
      const unsigned long long um = ((unsigned long long)su[1] << 32) | su[0];
      if (um != 0) {
        int lo = __builtin_ctzll(um), hi = 63 - __builtin_clzll(um);
        lo = lo > 1 ? lo - 2 : 0; hi = hi < 63 ? hi + 1 : 63;
        const int nodes = (hi - lo + 1) * Q;
        WinDec wd;
        wd.eps = 1.0f;
        if (TMPG && nodes <= (Q % 8 == 0 && Q > 8 ? 8 : 4) * kWave) {
          const float *spec = (const float *)c.spec;
          for (int arr = 0; arr < kSpArr; arr++)
            for (int u = lane; u <= L; u += kWave) __builtin_nontemporal_store(spec[arr * SP + u], tmpg + arr * SP + u);
          saved = true;
        }
        if (nodes <= 4 * kWave) wd = sweep_backward_decode_win<4, Q, TH, TMPG, false, false>(c, (lds_u8 *)seq, L, cm, 1.0f / (f1.xC * cm.move), f1.ef, min((63 - hi) * Q, kWave * (Q - 4)));
        else if (Q % 8 == 0 && Q > 8 && nodes <= 8 * kWave) wd = sweep_backward_decode_win<(Q % 8 == 0 ? 8 : 4), Q, TH, TMPG, false, false>(c, (lds_u8 *)seq, L, cm, 1.0f / (f1.xC * cm.move), f1.ef, min((63 - hi) * Q, kWave * (Q - 8)));
        // (a dominant alignment too wide for any window says nothing about the MODEL's slack: only a measured slack is remembered)
        if (nodes <= (Q % 8 == 0 && Q > 8 ? 8 : 4) * kWave) { fs.eps_prev2 = fs.eps_prev; fs.eps_prev = fabsf(wd.eps); }
        if (wd.eps > -1e-4f && wd.eps < 0.01f) {
          ro = region_scan_cert<TH, TMPG, false>(c.spec, SP, L, (lds_i *)regs, lane, fmaxf(wd.eps, 0.f));
          have_ro = ((ro.flags >> 24) & 3) == 0;
          if (a.stats && lane == 0) {
            if ((ro.flags >> 24) & 1) atomicAdd(a.stats + 32, 1ull);
            if ((ro.flags >> 25) & 1) atomicAdd(a.stats + 33, 1ull);
            atomicAdd(a.stats + 34, (unsigned long long)(wd.eps * 1e9f));
            atomicAdd(a.stats + 35, 1ull);
          }
          ro.flags &= 0xFFFFFF;
        } else if (a.stats && lane == 0) atomicAdd(a.stats + 36, 1ull);
        if (have_ro) fs.n_p2w++; else fs.n_p2rej++;
      }
    }
  }
  if (!have_ro) {
    if (TMPG && saved) {
      // the window sweep wrote over P1's rows: back from the copy (the stores above are this wave's own: order them first)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      float *spec = (float *)c.spec;
      __builtin_amdgcn_wave_barrier();
      for (int arr = 0; arr < kSpArr; arr++)
        for (int u = lane; u <= L; u += kWave) spec[arr * SP + u] = __builtin_nontemporal_load(tmpg + arr * SP + u);
      __builtin_amdgcn_wave_barrier();
    }
    sweep_backward_decode<Q, TH, SG>(c, (lds_u8 *)seq, L, cm, 1.0f / (f1.xC * cm.move), f1.ef);
    WH_TICK7(5);
    ro = region_scan<TH, SG>(c.spec, c.specg, SP, L, (lds_i *)regs, lane);
  } else WH_TICK7(5);
  return ro;
}

#ifndef WH_SWEEPS_ONLY

template <int Q, int TH, bool SG>
__global__ __launch_bounds__(TH) void score_kernel7(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  volatile int *s_item_p = reinterpret_cast<volatile int *>(smem_raw);
  float *smem = smem_raw + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  float *emL = smem;
  float *trL = smem + (size_t)a.K * TBL;
  constexpr int NARR = FW_NARR + (Q <= kMaxQP ? 1 : 0);   // arrays per orientation in LDS
  float *wbase = trL + 2 * NARR * TBL + (size_t)wave * a.wave_lds;
  const int SP = a.SP;
  WaveCtx c;
  c.emL = (lds_f *)emL; c.fwL = (lds_f *)trL; c.bwL = (lds_f *)(trL + NARR * TBL);
  const int spArrAll = kSpArr + (a.p2win == 1 ? 3 : 0);          // + the three rows of the windowed P2 (ScoreArgs::p2win)
  c.spec = (lds_f *)wbase; c.n2tab = (lds_f *)(wbase + (SG ? 0 : spArrAll * SP));
  c.specg = SG ? (glb_f *)(a.spec_scratch + ((size_t)blockIdx.x * nwaves + wave) * a.spec_stride) : nullptr;
  c.degen = 0;
  for (int t = 0; t < 32; t++) if (t == lane) c.degen = a.degen[t];
  c.Fs = (glb_f *)(a.scratch + ((size_t)blockIdx.x * nwaves + wave) * a.scratch_stride);
  c.SP = SP; c.alpha = a.K | (a.Kp << 8) | (a.K << 16); c.lane = lane;
  int *regs = reinterpret_cast<int *>(wbase + (SG ? 0 : spArrAll * SP) + 32);
  uint8_t *seq = reinterpret_cast<uint8_t *>(regs + kRegsInts);
  const double LOG2 = 0.69314718055994529;
  int cur_h = -1;
  const DevHMM *hm = nullptr;
  EnvCounters ec = {0, 0, 0, 0, 0};                           // this wave's envelope Backward sweeps by path (wh_last_score_paths)
  FrontState fs = {0.f, 0.f, 0u, 0u, 0u};                     // window heuristics and path counters of the multihit Backward sweep

  for (;;) {
    __syncthreads();                       // every wave has left the previous item's deal: its counter may be reset
    if (threadIdx.x == 0) { *s_item_p = atomicAdd(a.counter, 1); s_item_p[1] = 0; }
    __syncthreads();
    const int item = *s_item_p;
    if (item >= a.n_items) break;
    const int h = a.hmm_list[item / a.n_qblocks];
    const int64_t q_lo = (int64_t)(item % a.n_qblocks) * a.QB;
    const int64_t q_hi = q_lo + a.QB < a.nq ? q_lo + a.QB : a.nq;
    if (h != cur_h) {
      hm = a.hmms + h;
      const float4 *src = reinterpret_cast<const float4 *>(a.tables + hm->em_off);
      float4 *dst = reinterpret_cast<float4 *>(emL);
      for (int t = threadIdx.x; t < a.K * TBL / 4; t += blockDim.x) dst[t] = src[t];
      const float4 *s1 = reinterpret_cast<const float4 *>(a.tables + hm->fw_off);
      const float4 *s2 = reinterpret_cast<const float4 *>(a.tables + hm->bw_off);
      float4 *d1 = reinterpret_cast<float4 *>(trL);
      for (int t = threadIdx.x; t < NARR * TBL / 4; t += blockDim.x) { d1[t] = s1[t]; d1[NARR * TBL / 4 + t] = s2[t]; }
      cur_h = h;
      fs.eps_prev = 0.f; fs.eps_prev2 = 0.f; fs.n_pairs_h = 0;
      __syncthreads();
    }
    c.emG = (const glb_f *)(a.tables + hm->em_off);

    // The queries of an item are DEALT to the waves one by one from a counter in LDS (round 5; until then in fixed turns,
    // wave w taking q_lo + w, + nwaves, ...): a pair costs 1.5 to 2.5 million cycles depending on its path - a fixed deal
    // left 8-15 % of the wave time waiting at the item's end for the wave that drew the expensive pairs (measured with
    // the per-wave cycle counters).  Mixed-length batches arrive in descending length order: longest first.
    for (;;) {
      int k_ = 0;
      if (lane == 0) k_ = atomicAdd(const_cast<int *>(s_item_p) + 1, 1);
      const int64_t qpos = q_lo + __builtin_amdgcn_readfirstlane(k_);
      if (qpos >= q_hi) break;
      const int64_t qi = a.qorder ? a.qorder[qpos] : qpos;
      const int64_t off = a.offsets[qi];
      const int L = (int)(a.offsets[qi + 1] - off);
      const size_t out = (size_t)qi * a.H + h;
      int flags = 0, decibits = 0;
      float fwd_bits_out = -INFINITY;
      wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + out : nullptr;
      if (dp) {
        dp->fwd_bits = -INFINITY; dp->seq_score = 0.f; dp->pre_score = 0.f; dp->seqbias_nats = 0.f;
        dp->nregions = 0; dp->nenv = 0;
      }
      if (L > 0 && L <= a.Lcap) {
        for (int t = lane; t < L; t += kWave) {
          int r = a.residues[off + t];
          seq[t] = (uint8_t)(r < a.Kp ? r : a.Kp - 1);
        }
        __builtin_amdgcn_wave_barrier();
        long long t_last = a.stats ? __builtin_readcyclecounter() : 0;
        // ---------------- P1
        const LenCfg cm = len_config(L, true);
        const FwdOut f1 = sweep_forward<Q, false, TH, SG>(c, (lds_u8 *)seq, L, cm, 0.f);
        const double fwd_nats = (double)f1.ef * LOG2 + log((double)(f1.xC * cm.move));
        const float fwdsc = (float)fwd_nats;
        const float p1 = (float)L / (float)(L + 1);
        const float nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
        fwd_bits_out = (float)((fwd_nats - (double)nullsc) / LOG2);
        if (dp) dp->fwd_bits = fwd_bits_out;
        if (f1.xC > 0.f && isfinite(fwdsc)) {
          WH_TICK7(4);
          RegOut ro;
          if constexpr (!SG && Q >= 20) {
            // (20- / 24-cell models: no LDS for three more rows - the window sweep in place, P1's rows backed up in HBM)
            if (a.p2win == 2) ro = score_regions<Q, TH, false, true>(a, c, seq, regs, L, lane, cm, f1, fs, t_last, (glb_f *)(a.p2_backup + ((size_t)blockIdx.x * nwaves + wave) * a.p2_backup_stride));
            else ro = score_regions<Q, TH, SG, false>(a, c, seq, regs, L, lane, cm, f1, fs, t_last);
          } else ro = score_regions<Q, TH, SG, false>(a, c, seq, regs, L, lane, cm, f1, fs, t_last);
          const int nenv = ro.nenv, nreg = ro.nreg, multi_mask = ro.flags >> 8;
          flags |= ro.flags & 0xFF;
          if (dp) { dp->nregions = nreg; dp->nenv = nenv; }
          if (nenv > 0) {
            WH_TICK7(6);
            score_envelopes<Q, TH, SG>(a, c, seq, regs, L, lane, h, qi, nenv, nreg, multi_mask, fwdsc, nullsc, fwd_bits_out, dp, flags, decibits, ec, t_last, p1_mask(c));
          }
        }
      }
      if (lane == 0) {
        a.decibits[out] = decibits;
        a.flags[out] = (uint8_t)flags;
        if (a.fwd_bits) a.fwd_bits[out] = fwd_bits_out;
      }
    }
  }
  if (a.paths && lane == 0) {
    if (ec.n_w256) atomicAdd(a.paths + 0, (unsigned long long)ec.n_w256);
    if (ec.n_w512) atomicAdd(a.paths + 1, (unsigned long long)ec.n_w512);
    if (ec.n_wfail) atomicAdd(a.paths + 2, (unsigned long long)ec.n_wfail);
    if (ec.n_full) atomicAdd(a.paths + 3, (unsigned long long)ec.n_full);
    if (ec.spill) atomicAdd(a.paths + 6, ec.spill);
    if (fs.n_p2w) atomicAdd(a.paths + 4, (unsigned long long)fs.n_p2w);
    if (fs.n_p2rej) atomicAdd(a.paths + 5, (unsigned long long)fs.n_p2rej);
  }
}

// ---------------------------------------------------------------- the fused kernel with FOUR envelopes per Backward sweep (round 5)
// score_kernel7 with one change of schedule: a wave takes its queries four at a time.  Each goes through P1, P2, the region
// scan and - when it has exactly one single-domain envelope whose dominant alignment fits a 256-node window, the case of
// ~87 % of the headline's pairs - through P3 into ITS OWN Forward slab (four per wave) with its per-row arrays copied to
// the wave's HBM region; then ONE sweep_backward_null2_quad serves the four envelopes, a quarter of the wave each, and
// the pairs are assembled.  Everything else (no envelope, several, a multidomain region, a wider window, a failed
// certificate) takes the one-pair path of score_kernel7 on the spot, through the same functions.
// Per-wave LDS block: [six per-row arrays][4 x 32 null2 floats][region list][4 x 16 slot ints][4 x 16 record words][4 x residues].
// a.scratch_stride = FIVE slabs per wave (one of slack in front); a.spec_scratch / a.spec_stride = per wave one array of slack,
// four copies of the six arrays and one copy of P1's (the windowed P2 works in place on the wave's block): 31 x SP floats.
enum { QR_QLO = 0, QR_QHI, QR_L, QR_NREG, QR_FLAGS, QR_FWDSC, QR_NULLSC, QR_FWDBITS, QR_ENVSC, QR_RI, QR_RJ, QR_XC3, QR_EF3, QR_BAND, QR_INTS = 16 };
template <int Q, int TH>
__global__ __launch_bounds__(TH) void score_kernel7q(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  volatile int *s_item_p = reinterpret_cast<volatile int *>(smem_raw);
  float *smem = smem_raw + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  float *emL = smem;
  float *trL = smem + (size_t)a.K * TBL;
  float *wbase = trL + 2 * FW_NARR * TBL + (size_t)wave * a.wave_lds;
  const int SP = a.SP;
  const int seqw = (a.Lcap + 3) / 4 + 4;                       // words per residue buffer
  WaveCtx c;
  c.emL = (lds_f *)emL; c.fwL = (lds_f *)trL; c.bwL = (lds_f *)(trL + FW_NARR * TBL);
  c.spec = (lds_f *)wbase; c.n2tab = (lds_f *)(wbase + kSpArr * SP);
  const size_t wid = (size_t)blockIdx.x * nwaves + wave;
  // (one array / one slab of slack in front: an envelope that has run out of rows keeps requesting the rows "above" its first)
  glb_f *specg0 = (glb_f *)(a.spec_scratch + wid * a.spec_stride) + SP;
  const int spec_stride1 = kSpArr * SP;                        // floats per copy of the six arrays
  glb_f *tmpg = specg0 + 4 * spec_stride1;                     // the copy of P1's rows while the windowed P2 works in place
  c.specg = nullptr;
  c.degen = 0;
  for (int t = 0; t < 32; t++) if (t == lane) c.degen = a.degen[t];
  const int slab1 = (int)(a.scratch_stride / 5);               // floats per slab: four + the slack
  glb_f *Fs0 = (glb_f *)(a.scratch + wid * a.scratch_stride) + slab1;
  glb_f *FsW = Fs0 - slab1;                                    // the slack slab doubles as the work slab of the one-pair paths (the four
  c.Fs = FsW;                                                  // slabs behind it hold waiting envelopes; what a run-out envelope reads from the slack is not used)
  c.SP = SP; c.alpha = a.K | (a.Kp << 8) | (a.K << 16); c.lane = lane;
  int *regs = reinterpret_cast<int *>(wbase + kSpArr * SP + 128);
  int *qslots = regs + kRegsInts;
  int *qrecs = qslots + 4 * QS_INTS;
  uint8_t *seqs = reinterpret_cast<uint8_t *>(qrecs + 4 * QR_INTS);
  const double LOG2 = 0.69314718055994529;
  int cur_h = -1;
  const DevHMM *hm = nullptr;
  EnvCounters ec = {0, 0, 0, 0, 0};
  FrontState fs = {0.f, 0.f, 0u, 0u, 0u};
  const long long t_kernel0 = a.stats ? (long long)__builtin_readcyclecounter() : 0;

  for (;;) {
    const long long t_bar0 = a.stats ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();                       // every wave has left the previous item's deal: its counter may be reset
    if (threadIdx.x == 0) { *s_item_p = atomicAdd(a.counter, 1); s_item_p[1] = 0; }
    __syncthreads();
    const int item = *s_item_p;
    if (a.stats && lane == 0) atomicAdd(a.stats + 39, (unsigned long long)((long long)__builtin_readcyclecounter() - t_bar0));
    if (item >= a.n_items) break;
    const int h = a.hmm_list[item / a.n_qblocks];
    const int64_t q_lo = (int64_t)(item % a.n_qblocks) * a.QB;
    const int64_t q_hi = q_lo + a.QB < a.nq ? q_lo + a.QB : a.nq;
    if (h != cur_h) {
      hm = a.hmms + h;
      const float4 *src = reinterpret_cast<const float4 *>(a.tables + hm->em_off);
      float4 *dst = reinterpret_cast<float4 *>(emL);
      for (int t = threadIdx.x; t < a.K * TBL / 4; t += blockDim.x) dst[t] = src[t];
      const float4 *s1 = reinterpret_cast<const float4 *>(a.tables + hm->fw_off);
      const float4 *s2 = reinterpret_cast<const float4 *>(a.tables + hm->bw_off);
      float4 *d1 = reinterpret_cast<float4 *>(trL);
      for (int t = threadIdx.x; t < FW_NARR * TBL / 4; t += blockDim.x) { d1[t] = s1[t]; d1[FW_NARR * TBL / 4 + t] = s2[t]; }
      cur_h = h;
      fs.eps_prev = 0.f; fs.eps_prev2 = 0.f; fs.n_pairs_h = 0;
      __syncthreads();
    }
    c.emG = (const glb_f *)(a.tables + hm->em_off);

    // (four consecutive queries per draw from the item's LDS counter: see score_kernel7)
    for (;;) {
      int k_ = 0;
      if (lane == 0) k_ = atomicAdd(const_cast<int *>(s_item_p) + 1, 4);
      const int64_t base = q_lo + __builtin_amdgcn_readfirstlane(k_);
      if (base >= q_hi) break;
      int waiting = 0;                                          // bit t: slot t waits for the four-envelope sweep
      if (lane < 4) qslots[lane * QS_INTS + QS_ACTIVE] = 0;
      __builtin_amdgcn_wave_barrier();
      // ------------------------------------------------ phase A: every query up to its envelope's Forward sweep
      for (int t = 0; t < 4; t++) {
        const int64_t qpos = base + t;
        if (qpos >= q_hi) break;
        const int64_t qi = a.qorder ? a.qorder[qpos] : qpos;
        const int64_t off = a.offsets[qi];
        const int L = (int)(a.offsets[qi + 1] - off);
        const size_t out = (size_t)qi * a.H + h;
        uint8_t *seq = seqs + t * seqw * 4;
        int flags = 0, decibits = 0;
        float fwd_bits_out = -INFINITY;
        bool deferred = false;
        wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + out : nullptr;
        if (dp) {
          dp->fwd_bits = -INFINITY; dp->seq_score = 0.f; dp->pre_score = 0.f; dp->seqbias_nats = 0.f;
          dp->nregions = 0; dp->nenv = 0;
        }
        if (L > 0 && L <= a.Lcap) {
          for (int u = lane; u < L; u += kWave) {
            int r = a.residues[off + u];
            seq[u] = (uint8_t)(r < a.Kp ? r : a.Kp - 1);
          }
          __builtin_amdgcn_wave_barrier();
          long long t_last = a.stats ? __builtin_readcyclecounter() : 0;
          const LenCfg cm = len_config(L, true);
          const FwdOut f1 = sweep_forward<Q, false, TH, false>(c, (lds_u8 *)seq, L, cm, 0.f);
          const double fwd_nats = (double)f1.ef * LOG2 + log((double)(f1.xC * cm.move));
          const float fwdsc = (float)fwd_nats;
          const float p1 = (float)L / (float)(L + 1);
          const float nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
          fwd_bits_out = (float)((fwd_nats - (double)nullsc) / LOG2);
          if (dp) dp->fwd_bits = fwd_bits_out;
          if (f1.xC > 0.f && isfinite(fwdsc)) {
            WH_TICK7(4);
            const RegOut ro = score_regions<Q, TH, false, true>(a, c, seq, regs, L, lane, cm, f1, fs, t_last, tmpg);
            const int nenv = ro.nenv, nreg = ro.nreg, multi_mask = ro.flags >> 8;
            flags |= ro.flags & 0xFF;
            if (dp) { dp->nregions = nreg; dp->nenv = nenv; }
            if (nenv == 1 && multi_mask == 0 && !a.no_window) {
              // ---- one single-domain envelope: its Forward sweep into slab t; the Backward sweep waits for the other three
              WH_TICK7(6);
              const int ri = regs[0], rj = regs[1];
              const int Ld = rj - ri + 1;
              const uint8_t *eseq = seq + (ri - 1);
              const LenCfg cu = len_config(L, false);
              c.Fs = Fs0 + (size_t)t * slab1;
              const float keep_scale = a.keep_scale > 0.f ? a.keep_scale : kKeepScale7;
              const int band = spill_band<Q>(a, p1_mask(c));
              const FwdOut f3 = sweep_forward<Q, true, TH, false>(c, (lds_u8 *)eseq, Ld, cu, keep_scale, band);
              ec.spill += (unsigned long long)f3.nst * (8 * Q);
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
              const float envsc = (float)((double)f3.ef * LOG2 + log((double)(f3.xC * cu.move)));
              WH_TICK7(7);
              float domcorr = 0.f;
              bool done = !(f3.xC > 0.f);
              if (!done) {
                const unsigned *su = reinterpret_cast<const unsigned *>((const float *)c.spec);
                const unsigned long long um = mask_in_band(((unsigned long long)su[kSpMH * SP] << 32) | su[kSpML * SP], band);
                int m0 = -1;
                if (um != 0) {
                  int lo = __builtin_ctzll(um), hi = 63 - __builtin_clzll(um);
                  lo = lo > 1 ? lo - 2 : 0; hi = hi < 63 ? hi + 1 : 63;
                  if ((hi - lo + 1) * Q <= 4 * kWave) m0 = min((63 - hi) * Q, kWave * (Q - 4));
                }
                if (m0 >= 0) {
                  // the six per-row arrays of this envelope -> the wave's HBM copy t (coalesced), the slot and the record
                  glb_f *dst = specg0 + (size_t)t * spec_stride1;
                  const float *spec = (const float *)c.spec;
                  for (int arr = 0; arr < kSpArr; arr++)
                    for (int u = lane; u <= Ld; u += kWave) __builtin_nontemporal_store(spec[arr * SP + u], dst + arr * SP + u);
                  if (lane == 0) {
                    int *sl = qslots + t * QS_INTS;
                    sl[QS_ACTIVE] = 1; sl[QS_LD] = Ld; sl[QS_M0] = m0;
                    sl[QS_INVZ] = __builtin_bit_cast(int, 1.0f / (f3.xC * cu.move));
                    sl[QS_LOOP] = __builtin_bit_cast(int, cu.loop); sl[QS_MOVE] = __builtin_bit_cast(int, cu.move);
                    sl[QS_SEQ] = t * seqw * 4 + (ri - 1);
                    int *qr = qrecs + t * QR_INTS;
                    qr[QR_QLO] = (int)(unsigned)(qi & 0xFFFFFFFF); qr[QR_QHI] = (int)(qi >> 32);
                    qr[QR_L] = L; qr[QR_NREG] = nreg; qr[QR_FLAGS] = flags;
                    qr[QR_FWDSC] = __builtin_bit_cast(int, fwdsc); qr[QR_NULLSC] = __builtin_bit_cast(int, nullsc);
                    qr[QR_FWDBITS] = __builtin_bit_cast(int, fwd_bits_out); qr[QR_ENVSC] = __builtin_bit_cast(int, envsc);
                    qr[QR_RI] = ri; qr[QR_RJ] = rj; qr[QR_XC3] = __builtin_bit_cast(int, f3.xC); qr[QR_EF3] = f3.ef; qr[QR_BAND] = band;
                  }
                  waiting |= 1 << t;
                  deferred = true;
                  WH_TICK7(10);
                } else {
                  // a wider window or none: this envelope's Backward sweep now, as score_envelopes runs it
                  const bool banded = band != kAllLanes;
                  const P4Out p4 = envelope_backward<Q, TH, false>(a, c, eseq, Ld, cu, f3, false, false, ec, lane, band);
                  domcorr = p4.domcorr;
                  if (!(fabsf((float)Ld - p4.mass) <= spill_tol(banded) * (float)Ld)) {
                    float envsc2;
                    domcorr = envelope_attempts<Q, TH, false>(a, c, eseq, Ld, cu, band, banded ? 1 : 2, ec, lane, flags, t_last, envsc2);
                  }
                  done = true;
                }
              }
              if (done) {
                WH_TICK7(8);
                float sum_score = 0.f, sb2 = 0.f; int Ld_tot = 0;
                if (envsc - domcorr > 0.0f) { sum_score = envsc; Ld_tot = Ld; sb2 = domcorr; }
                if (dp) { dp->env_i[0] = ri; dp->env_j[0] = rj; dp->envsc[0] = envsc; dp->domcorr[0] = domcorr; }
                assemble_score(L, Ld_tot, domcorr, sum_score, sb2, fwdsc, nullsc, dp, flags, decibits);
              }
              c.Fs = FsW;
            } else if (nenv > 0) {
              WH_TICK7(6);
              score_envelopes<Q, TH, false>(a, c, seq, regs, L, lane, h, qi, nenv, nreg, multi_mask, fwdsc, nullsc, fwd_bits_out, dp, flags, decibits, ec, t_last, p1_mask(c));
            }
          }
        }
        if (lane == 0) {
          if (!deferred) { a.decibits[out] = decibits; a.flags[out] = (uint8_t)flags; }
          if (a.fwd_bits) a.fwd_bits[out] = fwd_bits_out;
        }
      }
      // ------------------------------------------------ phase B: one Backward sweep for the waiting envelopes
      if (waiting) {
        long long t_last = a.stats ? __builtin_readcyclecounter() : 0;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // rows and per-row arrays of four envelopes, written by this wave
        WaveCtx cq = c;
        cq.specg = specg0;
        cq.Fs = Fs0;
        WH_TICK7(10);
        sweep_backward_null2_quad<Q, TH>(cq, (lds_i *)qslots, (lds_u8 *)seqs, (lds_f *)(wbase + kSpArr * SP), slab1, spec_stride1, kWinTol7);
        WH_TICK7(9);
        if (a.stats && lane == 0) atomicAdd(a.stats + 37, (unsigned long long)__builtin_popcount(waiting));
        // ---------------------------------------------- phase C: certificates, assembly
        for (int t = 0; t < 4; t++) {
          if (!((waiting >> t) & 1)) continue;
          const int *qr = qrecs + t * QR_INTS;
          const int *sl = qslots + t * QS_INTS;
          const int64_t qi = ((int64_t)qr[QR_QHI] << 32) | (unsigned)qr[QR_QLO];
          const size_t out = (size_t)qi * a.H + h;
          const int L = qr[QR_L], Ld = sl[QS_LD], ri = qr[QR_RI], rj = qr[QR_RJ];
          int flags = qr[QR_FLAGS], decibits = 0;
          const float fwdsc = __builtin_bit_cast(float, qr[QR_FWDSC]), nullsc = __builtin_bit_cast(float, qr[QR_NULLSC]);
          const float envsc = __builtin_bit_cast(float, qr[QR_ENVSC]);
          const float mass = __builtin_bit_cast(float, sl[QS_MASS]);
          float domcorr = __builtin_bit_cast(float, sl[QS_DOMCORR]);
          wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + out : nullptr;
          if (fabsf((float)Ld - mass) <= kWinTol7 * (float)Ld) ec.n_w256++;
          else {
            // the window lost mass: full width on the same rows (the per-row arrays back into the wave's block), then
            // the dense redo if the spill certificate fails too - score_envelopes' own sequence
            ec.n_wfail++;
            uint8_t *seq = seqs + t * seqw * 4;
            const uint8_t *eseq = seq + (ri - 1);
            const LenCfg cu = len_config(L, false);
            float *spec = (float *)c.spec;
            const glb_f *src = specg0 + (size_t)t * spec_stride1;
            __builtin_amdgcn_wave_barrier();
            for (int arr = 0; arr < kSpArr; arr++)
              for (int u = lane; u <= Ld; u += kWave) spec[arr * SP + u] = __builtin_nontemporal_load(src + arr * SP + u);
            __builtin_amdgcn_wave_barrier();
            c.Fs = Fs0 + (size_t)t * slab1;
            FwdOut f3; f3.xC = __builtin_bit_cast(float, qr[QR_XC3]); f3.ef = qr[QR_EF3];
            const int band = qr[QR_BAND];
            const bool banded = band != kAllLanes;
            const P4Out p4 = envelope_backward<Q, TH, false>(a, c, eseq, Ld, cu, f3, false, true, ec, lane, band);
            domcorr = p4.domcorr;
            if (!(fabsf((float)Ld - p4.mass) <= spill_tol(banded) * (float)Ld)) {
              float envsc2;
              domcorr = envelope_attempts<Q, TH, false>(a, c, eseq, Ld, cu, band, banded ? 1 : 2, ec, lane, flags, t_last, envsc2);
            }
            c.Fs = FsW;
          }
          float sum_score = 0.f, sb2 = 0.f; int Ld_tot = 0;
          if (envsc - domcorr > 0.0f) { sum_score = envsc; Ld_tot = Ld; sb2 = domcorr; }
          if (dp) { dp->env_i[0] = ri; dp->env_j[0] = rj; dp->envsc[0] = envsc; dp->domcorr[0] = domcorr; }
          assemble_score(L, Ld_tot, domcorr, sum_score, sb2, fwdsc, nullsc, dp, flags, decibits);
          if (lane == 0) { a.decibits[out] = decibits; a.flags[out] = (uint8_t)flags; }
        }
        WH_TICK7(11);
      }
    }
  }
  if (a.paths && lane == 0) {
    if (ec.n_w256) atomicAdd(a.paths + 0, (unsigned long long)ec.n_w256);
    if (ec.n_w512) atomicAdd(a.paths + 1, (unsigned long long)ec.n_w512);
    if (ec.n_wfail) atomicAdd(a.paths + 2, (unsigned long long)ec.n_wfail);
    if (ec.n_full) atomicAdd(a.paths + 3, (unsigned long long)ec.n_full);
    if (ec.spill) atomicAdd(a.paths + 6, ec.spill);
    if (fs.n_p2w) atomicAdd(a.paths + 4, (unsigned long long)fs.n_p2w);
    if (fs.n_p2rej) atomicAdd(a.paths + 5, (unsigned long long)fs.n_p2rej);
  }
  if (a.stats && lane == 0) atomicAdd(a.stats + 38, (unsigned long long)(__builtin_readcyclecounter() - t_kernel0));
}

template <int Q, int TH>
static hipError_t launch7q(const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_kernel7q<Q, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((score_kernel7q<Q, TH>), dim3(blocks), dim3(threads), lds, s, a);
  return hipGetLastError();
}

template <int Q, int TH, bool SG>
static hipError_t launch7sg(const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_kernel7<Q, TH, SG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((score_kernel7<Q, TH, SG>), dim3(blocks), dim3(threads), lds, s, a);
  return hipGetLastError();
}

template <int Q, int TH>
static hipError_t launch7(const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  // long queries: the per-row special states live in a.spec_scratch (HBM) instead of LDS
  if (a.spec_scratch) return launch7sg<Q, TH, true>(a, blocks, threads, lds, s);
  return launch7sg<Q, TH, false>(a, blocks, threads, lds, s);
}

// threads per workgroup sets the register budget of every sweep: 512 -> 256 VGPRs, 768 -> 168
template <int TH>
static hipError_t launch7_q(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  switch (Q) {
    case 4:  return launch7<4, TH>(a, blocks, threads, lds, s);
    case 8:  return launch7<8, TH>(a, blocks, threads, lds, s);
    case 12: return launch7<12, TH>(a, blocks, threads, lds, s);
    case 16: return launch7<16, TH>(a, blocks, threads, lds, s);
    case 20: return launch7<20, TH>(a, blocks, threads, lds, s);
    case 24: return launch7<24, TH>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

#endif  // WH_SWEEPS_ONLY
}  // namespace WH_K7NS

#if !defined(WH_SWEEPS_ONLY) && !defined(WH_K7B)
// four envelopes per Backward sweep (score_kernel7q): models of 16 cells per lane
hipError_t launch_score7q(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  using namespace WH_K7NS;
  if (a.spec_arrays != kSpArr || !a.spec_scratch || threads > 768) return hipErrorInvalidValue;
  if (Q == 16) return launch7q<16, 768>(a, blocks, threads, lds, s);
  return hipErrorInvalidValue;
}
#endif
#ifndef WH_SWEEPS_ONLY
hipError_t WH_K7LAUNCH(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  using namespace WH_K7NS;
  if (!a.spec_scratch && a.spec_arrays != kSpArr) return hipErrorInvalidValue;   // planner and kernel disagree about the LDS block
  if (threads <= 512) return launch7_q<512>(Q, a, blocks, threads, lds, s);
  if (threads <= 768) return launch7_q<768>(Q, a, blocks, threads, lds, s);
  return hipErrorInvalidValue;
}
#endif

}  // namespace wh
