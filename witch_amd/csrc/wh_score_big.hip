// Scoring kernel for LONG models (1536 < M <= 3072 nodes, Q = 28..48 cells per lane), and for 20/24-cell
// models whose tables + special states do not fit in LDS beside both orientations.
//
// Same algorithm as wh_score7.hip, one wavefront per (query, HMM) pair, but:
//  * only ONE transition orientation fits in LDS (8 arrays x Q x 256 B = 90 KB at Q = 44, + 45 KB of
//    emission rows).  The four waves of a workgroup therefore run the sweeps in lockstep ("pass-
//    synchronous"): Forward sweeps with the forward tables, a workgroup barrier + table swap (~100 KB from
//    L2, microseconds against a sweep of milliseconds), then the Backward sweeps.  The host hands the
//    queries over in descending LENGTH order (ScoreArgs::qorder), so the waves of a workgroup finish a
//    sweep together: the barrier wait fell from 6 % of the wave cycles to 0.3 %.
//  * the DP row of a 3072-node model is 144 VGPRs per lane: one wavefront per SIMD (256 threads), the
//    second half of the register file (AGPRs) holds what the row loops do not touch.  One wave per SIMD
//    hides no latency, so every latency is hidden by hand:
//      - LDS table reads are software-pipelined one 4-cell group ahead (pipe_groups below);
//      - the per-row special states live in HBM (no LDS left for them) and are requested ONE ROW AHEAD;
//      - the stored Forward cells the envelope Backward sweep multiplies with are requested at the top of
//        their row and consumed after the cell update (88 landing registers);
//      - the region scan fetches 64 rows per load and walks them with v_readlane.
//    With that a DP row costs ~4.6 cycles per instruction, the issue rate of a single wave.
// Measured (dna_rrna_like, 2 000 queries of 1500-2400 nt x 10 HMMs of ~2 470 nodes): 619 ms before this
// structure, 299 ms with it = 3.2e11 cells/s (the headline kernel: 3.9e11).
#include <hip/hip_runtime.h>

#include "wh_device.h"
#include "wh_launch.h"

namespace wh {

__device__ __forceinline__ float flogsum0_big(float b) {
  const float mx = b > 0.f ? b : 0.f, mn = b > 0.f ? 0.f : b;
  if (mn == -INFINITY || (mx - mn) >= 15.7f) return mx;
  const int idx = (int)((mx - mn) * 1000.0f);
  return mx + (float)log(1.0 + exp((double)-idx / 1000.0));
}

constexpr float kKeepScaleB = 9.094947e-13f;   // 2^-40, see wh_score.hip
constexpr float kMassTolB = 2e-5f;

#define BSPR(idx)  __builtin_nontemporal_load(spec + (idx))
#define BSPRI(idx) __builtin_nontemporal_load(reinterpret_cast<const int *>(spec) + (idx))
#define BSPRU(idx) __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(spec) + (idx))


// ------------------------------------------------------------------ sweeps of the long-model kernel
// With one or two waves per SIMD nothing hides an LDS round trip, and the compiler, short of registers
// at 28-48 cells per lane, sinks every table read down to its first use: ~100 exposed ds_read latencies
// per DP row were most of the kernel's time.  The cell loops below are therefore software-pipelined by
// hand: the 16-byte table pieces of 4-cell group g+1 are requested before group g is computed, and a
// scheduling barrier on both sides of the computation keeps the compiler from undoing that, so the wait
// in front of a group covers loads issued a whole group earlier (s_waitcnt lgkmcnt(N) with N = the next
// group's loads).  The sweeps also keep nothing per cell beyond the DP row itself: the emission piece of
// a group is fetched with its tables, and the Backward D row is not stored - its local chains are run
// once for the cross-lane scan and a second time, seeded with the scanned carry, inside the M/I update
// (the two FMAs per cell the carry fix-up would cost, one more table read per group).
#define WH_SB() __builtin_amdgcn_sched_barrier(0)

// scan_apply (wh_device.h) as ONE asm block: the six model-only multipliers are spilled around the cell
// loops at 256 registers, and as operands of a single block they are reloaded together, not one
// scratch round trip per step.
__device__ __forceinline__ float scan_apply_block(const ScanC &c, float B) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_fmac_f32_dpp %0, %0, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(B)
      : "v"(c.s[0]), "v"(c.s[1]), "v"(c.s[2]), "v"(c.s[3]), "v"(c.s[4]), "v"(c.s[5]));
  return B;
}

// Software pipeline over the G 4-cell groups of a lane: ld(g, buf) requests the NT table pieces of group g,
// body(g, buf) consumes them; the request runs DIST groups ahead (1 for the heavy M/I passes, whose ~36
// VALU ops per group cover an LDS round trip; 2 for the light D-chain / emission passes of 8-12 ops).
// Groups are visited in the order ord(0), ord(1), ...
#define WH_INL __attribute__((always_inline))
// Ties the results of a group to this point of the instruction stream: the arithmetic is pure, so without
// it the compiler lets a group's VALU ops drift below the following groups' loads and barriers.
__device__ __forceinline__ void pin4(float &a, float &b, float &c, float &d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void pin1(float &a) { asm volatile("" : "+v"(a)); }
template <int G, int NT, int DIST, class Ord, class Ld, class Body>
__device__ __forceinline__ void pipe_groups(Ord ord, Ld ld, Body body) {
  float4 b0[NT], b1[NT], b2[NT];
  ld(ord(0), b0);
  if (DIST >= 2 && G > 1) ld(ord(1), b1);
#pragma unroll
  for (int s = 0; s < G; s++) {
    if (DIST >= 2) { if (s + 2 < G) ld(ord(s + 2), b2); }
    else { if (s + 1 < G) ld(ord(s + 1), b1); }
    WH_SB();
    body(ord(s), b0);
    WH_SB();
#pragma unroll
    for (int t = 0; t < NT; t++) { b0[t] = b1[t]; if (DIST >= 2) b1[t] = b2[t]; }
  }
}

template <int Q, bool STORE>
__device__ __forceinline__ void forward_sweep_lean(const TransTab<Q, false> &T, const ScanC &sc, const float *emL,
                                                   const float *emG, int K, const uint8_t *seq, int L, LenCfg cfg,
                                                   float *spec, int SP, float *Fs, float keep_scale, int lane,
                                                   float &xC_out, int &ef_out) {
  constexpr int G = Q / 4;
  float Mp[Q], Ip[Q], Dp[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) { Mp[q] = 0.f; Ip[q] = 0.f; Dp[q] = 0.f; }
  float xN = 1.0f, xB = cfg.move, xJ = 0.f, xC = 0.f, xE = 0.f;
  int ef = 0;
  if (lane == 0) {
    spec[SP_N * SP] = xN; spec[SP_B * SP] = xB; spec[SP_E * SP] = 0.f; spec[SP_J * SP] = 0.f;
    spec[SP_C * SP] = 0.f; reinterpret_cast<int *>(spec)[SP_S * SP] = 0;
  }
  auto up = [](int s) WH_INL { return s; };
  auto down = [](int s) WH_INL { return G - 1 - s; };
#pragma unroll 1
  for (int i = 1; i <= L; i++) {
    asm volatile("" ::: "memory");
    const int x = __builtin_amdgcn_readfirstlane((int)seq[i - 1]);
    const float mm1 = wave_shr1(Mp[Q - 1]), im1 = wave_shr1(Ip[Q - 1]), dm1 = wave_shr1(Dp[Q - 1]);
    // M (without its emission factor) and I, in place: cell q reads the old q-1, so groups run downwards
    pipe_groups<G, 6, 1>(down,
        [&](int g, float4 (&t)[6]) WH_INL {
          t[0] = T.ld(FW_A, g); t[1] = T.ld(FW_B, g); t[2] = T.ld(FW_C, g); t[3] = T.ld(FW_E, g);
          t[4] = T.ld(FW_MI, g); t[5] = T.ld(FW_II, g);
        },
        [&](int g, const float4 (&t)[6]) WH_INL {
#pragma unroll
          for (int j = 3; j >= 0; j--) {
            const int q = 4 * g + j;
            const float pm = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mm1;
            const float pi = q > 0 ? Ip[q > 0 ? q - 1 : 0] : im1;
            const float pd = q > 0 ? Dp[q > 0 ? q - 1 : 0] : dm1;
            const float ni = fmaf(f4get(t[4], j), Mp[q], f4get(t[5], j) * Ip[q]);
            float acc = xB * f4get(t[3], j);
            acc = fmaf(f4get(t[0], j), pm, acc);
            acc = fmaf(f4get(t[1], j), pi, acc);
            acc = fmaf(f4get(t[2], j), pd, acc);
            Mp[q] = acc;
            Ip[q] = ni;
          }
          pin4(Mp[4 * g], Mp[4 * g + 1], Mp[4 * g + 2], Mp[4 * g + 3]);
          pin4(Ip[4 * g], Ip[4 * g + 1], Ip[4 * g + 2], Ip[4 * g + 3]);
        });
    // emission factor: canonical residues from the LDS copy of the table, degenerate codes (rare) from L2;
    // only this light pass is duplicated by the branch
    auto emit = [&](auto ep) WH_INL {
      pipe_groups<G, 1, 2>(up, [&](int g, float4 (&t)[1]) WH_INL { t[0] = ep[g * kWave]; },
                           [&](int g, const float4 (&t)[1]) WH_INL {
                             Mp[4 * g] *= t[0].x; Mp[4 * g + 1] *= t[0].y; Mp[4 * g + 2] *= t[0].z; Mp[4 * g + 3] *= t[0].w;
                             pin4(Mp[4 * g], Mp[4 * g + 1], Mp[4 * g + 2], Mp[4 * g + 3]);
                           });
    };
    asm volatile("" ::: "memory");       // keeps the emission reads below the M/I pass (they would be hoisted
                                         // to the top of the row and held in 4*G registers across it)
    if (x < K) emit(LdsF4(emL + (size_t)x * Q * kWave + 4 * lane));
    else emit(reinterpret_cast<const float4 *>(emG + (size_t)x * Q * kWave) + lane);
    asm volatile("" ::: "memory");
    // D row: local chains, cross-lane scan, fix-up
    const float mn1 = wave_shr1(Mp[Q - 1]);
    float dprev = 0.f;
    pipe_groups<G, 2, 2>(up, [&](int g, float4 (&t)[2]) WH_INL { t[0] = T.ld(FW_D1, g); t[1] = T.ld(FW_D2, g); },
                         [&](int g, const float4 (&t)[2]) WH_INL {
#pragma unroll
                           for (int j = 0; j < 4; j++) {
                             const int q = 4 * g + j;
                             const float src = q > 0 ? Mp[q > 0 ? q - 1 : 0] : mn1;
                             dprev = fmaf(f4get(t[1], j), dprev, f4get(t[0], j) * src);
                             Dp[q] = dprev;
                           }
                           pin1(dprev);
                         });
    float carry = wave_shr1(scan_apply_block(sc, dprev));
    float es = 0.f;
    pipe_groups<G, 1, 2>(up, [&](int g, float4 (&t)[1]) WH_INL { t[0] = T.ld(FW_D2, g); },
                         [&](int g, const float4 (&t)[1]) WH_INL {
#pragma unroll
                           for (int j = 0; j < 4; j++) {
                             const int q = 4 * g + j;
                             carry *= f4get(t[0], j);
                             Dp[q] += carry;
                             es += Mp[q] + Dp[q];
                           }
                           pin1(es);
                         });
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

// G_k = o_k(x) * B_M_k in place (reversed node order) and the partial B-state sum  sum_k E_k G_k
template <int Q>
__device__ __forceinline__ float backward_emit_lean(const TransTab<Q, false> &T, const float *emL, const float *emG,
                                                    int x, int K, int lane, float (&Mb)[Q]) {
  constexpr int G = Q / 4;
  x = __builtin_amdgcn_readfirstlane(x);
  float part = 0.f;
  auto up = [](int s) WH_INL { return s; };
  auto emit = [&](auto ep) WH_INL {
    pipe_groups<G, 2, 2>(up, [&](int g, float4 (&t)[2]) WH_INL { t[0] = T.ld(BW_E, g); t[1] = ep[(G - 1 - g) * kWave]; },
                         [&](int g, const float4 (&t)[2]) WH_INL {
                           // forward-ordered emission piece: component 3-j is position 4*g+j
                           Mb[4 * g + 0] *= t[1].w; part = fmaf(t[0].x, Mb[4 * g + 0], part);
                           Mb[4 * g + 1] *= t[1].z; part = fmaf(t[0].y, Mb[4 * g + 1], part);
                           Mb[4 * g + 2] *= t[1].y; part = fmaf(t[0].z, Mb[4 * g + 2], part);
                           Mb[4 * g + 3] *= t[1].x; part = fmaf(t[0].w, Mb[4 * g + 3], part);
                           pin1(part);
                         });
  };
  if (x < K) emit(LdsF4(emL + (size_t)x * Q * kWave + 4 * (kWave - 1 - lane)));
  else emit(reinterpret_cast<const float4 *>(emG + (size_t)x * Q * kWave) + (kWave - 1 - lane));
  return part;
}

// One Backward row (reversed node order) without a stored D row; see backward_cells in wh_device.h for
// the recurrences.  On entry Mb holds G (emission already applied), on exit rows i of M and I.
template <int Q>
__device__ __forceinline__ void backward_cells_lean(const TransTab<Q, false> &T, const ScanC &sc, float (&Mb)[Q],
                                                    float (&Ib)[Q], float xE) {
  constexpr int G = Q / 4;
  auto up = [](int s) WH_INL { return s; };
  const float gm1 = wave_shr1(Mb[Q - 1]);
  float dprev = 0.f;
  pipe_groups<G, 2, 2>(up, [&](int g, float4 (&t)[2]) WH_INL { t[0] = T.ld(BW_DM, g); t[1] = T.ld(BW_DD, g); },
                       [&](int g, const float4 (&t)[2]) WH_INL {
#pragma unroll
                         for (int j = 0; j < 4; j++) {
                           const int p = 4 * g + j;
                           const float gg = p > 0 ? Mb[p > 0 ? p - 1 : 0] : gm1;
                           dprev = fmaf(f4get(t[1], j), dprev, fmaf(f4get(t[0], j), gg, xE));
                         }
                         pin1(dprev);
                       });
  // inclusive scan value of the previous lane = its last D cell = the D this lane's chain starts from
  float dcur = wave_shr1(scan_apply_block(sc, dprev));
  float gprev = gm1;
  pipe_groups<G, 7, 1>(up,
      [&](int g, float4 (&t)[7]) WH_INL {
        t[0] = T.ld(BW_MM, g); t[1] = T.ld(BW_IM, g); t[2] = T.ld(BW_MI, g); t[3] = T.ld(BW_II, g);
        t[4] = T.ld(BW_MD, g); t[5] = T.ld(BW_DM, g); t[6] = T.ld(BW_DD, g);
      },
      [&](int g, const float4 (&t)[7]) WH_INL {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int p = 4 * g + j;
          const float gold = Mb[p];
          float nm = fmaf(f4get(t[0], j), gprev, xE);
          nm = fmaf(f4get(t[2], j), Ib[p], nm);
          nm = fmaf(f4get(t[4], j), dcur, nm);
          const float ni = fmaf(f4get(t[1], j), gprev, f4get(t[3], j) * Ib[p]);
          dcur = fmaf(f4get(t[6], j), dcur, fmaf(f4get(t[5], j), gprev, xE));
          Mb[p] = nm;
          Ib[p] = ni;
          gprev = gold;
        }
        pin4(Mb[4 * g], Mb[4 * g + 1], Mb[4 * g + 2], Mb[4 * g + 3]);
        pin4(Ib[4 * g], Ib[4 * g + 1], Ib[4 * g + 2], Ib[4 * g + 3]);
      });
}

// per-phase wave cycles (option WH_STATS): slot 4 P1, 5 P2 + region scan, 7 P3, 8 P4 + null2, 10 table swaps and
// the workgroup barriers around them (the lockstep cost), 11 everything else
#define WH_TICKB(slot) do { if (a.stats) { const long long t_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(a.stats + (slot), (unsigned long long)(t_now - t_last)); t_last = t_now; } } while (0)

template <int Q, int TH>
__global__ __launch_bounds__(TH) void score_big_kernel(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  volatile int *s_item_p = reinterpret_cast<volatile int *>(smem_raw);
  float *smem = smem_raw + 4;
  // the wave index is uniform: saying so keeps every per-wave pointer, length and loop counter in SGPRs
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  const int Klds = a.Klds;                                // emission rows staged in LDS (K, or 0: all from L2)
  float *emL = smem;
  float *trL = smem + (size_t)Klds * TBL;                 // ONE orientation: 8 arrays
  float *wbase = trL + 8 * TBL + (size_t)wave * a.wave_lds;
  float *n2tab = wbase;
  int *regs = reinterpret_cast<int *>(n2tab + 32);
  uint8_t *seq = reinterpret_cast<uint8_t *>(regs + kRegsInts);
  float *spec = a.spec_scratch + ((size_t)blockIdx.x * nwaves + wave) * a.spec_stride;
  float *Fs = a.scratch + ((size_t)blockIdx.x * nwaves + wave) * a.scratch_stride;
  const int SP = a.SP;
  const double LOG2 = 0.69314718055994529;
  int cur_h = -1, cur_orient = -1;
  long long t_last = a.stats ? __builtin_readcyclecounter() : 0;
  const DevHMM *hm = nullptr;
  const float *fwG = nullptr, *bwG = nullptr, *emG = nullptr;

  // every thread of the workgroup calls this at the same points
  auto orient = [&](int o) {
    if (cur_orient != o) {
      __syncthreads();
      const float4 *src = reinterpret_cast<const float4 *>(o ? bwG : fwG);
      float4 *dst = reinterpret_cast<float4 *>(trL);
      for (int t = threadIdx.x; t < 8 * TBL / 4; t += blockDim.x) dst[t] = src[t];
      __syncthreads();
      cur_orient = o;
    }
  };

  for (;;) {
    if (threadIdx.x == 0) *s_item_p = atomicAdd(a.counter, 1);
    __syncthreads();
    const int item = *s_item_p;
    __syncthreads();
    if (item >= a.n_items) break;
    const int h = a.hmm_list[item / a.n_qblocks];
    const int64_t q_lo = (int64_t)(item % a.n_qblocks) * a.QB;
    const int64_t q_hi = q_lo + a.QB < a.nq ? q_lo + a.QB : a.nq;
    if (h != cur_h) {
      __syncthreads();
      hm = a.hmms + h;
      fwG = a.tables + hm->fw_off; bwG = a.tables + hm->bw_off; emG = a.tables + hm->em_off;
      const float4 *src = reinterpret_cast<const float4 *>(emG);
      float4 *dst = reinterpret_cast<float4 *>(emL);
      for (int t = threadIdx.x; t < Klds * TBL / 4; t += blockDim.x) dst[t] = src[t];
      cur_h = h;
      cur_orient = -1;
      __syncthreads();
    }
    const int nround = (int)((q_hi - q_lo + nwaves - 1) / nwaves);
    for (int round = 0; round < nround; round++) {
      // the waves of a workgroup sweep in lockstep: neighbours in LENGTH order keep them busy for the same time
      const int64_t qpos = q_lo + (int64_t)round * nwaves + wave;
      const bool in_range = qpos < q_hi;
      const int64_t qi = in_range ? (a.qorder ? (int64_t)a.qorder[qpos] : qpos) : 0;
      int L = 0;
      int64_t off = 0;
      if (in_range) { off = a.offsets[qi]; L = (int)(a.offsets[qi + 1] - off); if (L > a.Lcap) L = 0; }
      const bool active = in_range && L > 0;
      const size_t out = (size_t)qi * a.H + h;
      int flags = 0, decibits = 0, nreg = 0, nenv = 0, ef_L = 0, multi_mask = 0;
      float fwd_bits_out = -INFINITY, fwdsc = 0.f, nullsc = 0.f, invZ = 0.f;
      bool ok = false;
      wh_pair_detail *dp = (a.detail && lane == 0 && active) ? a.detail + out : nullptr;
      if (dp) {
        dp->fwd_bits = -INFINITY; dp->seq_score = 0.f; dp->pre_score = 0.f; dp->seqbias_nats = 0.f;
        dp->nregions = 0; dp->nenv = 0;
      }
      for (int t = lane; t < L; t += kWave) {
        int c = a.residues[off + t];
        seq[t] = (uint8_t)(c < a.Kp ? c : a.Kp - 1);
      }
      __builtin_amdgcn_wave_barrier();
      const LenCfg cm = len_config(L > 0 ? L : 1, true);
      const LenCfg cu = len_config(L > 0 ? L : 1, false);

      // ---------------- P1: multihit Forward (forward tables)
      WH_TICKB(11);
      orient(0);
      WH_TICKB(10);
      if (active) {
        TransTab<Q, false> T;
        T.load(nullptr, trL, lane);
        const ScanC sc = scan_prepare(lane_product<Q, false>(T, FW_D2));
        float xC_L;
        forward_sweep_lean<Q, false>(T, sc, emL, emG, Klds, seq, L, cm, spec, SP, nullptr, 0.f, lane, xC_L, ef_L);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const double fwd_nats = (double)ef_L * LOG2 + log((double)(xC_L * cm.move));
        fwdsc = (float)fwd_nats;
        const float p1 = (float)L / (float)(L + 1);
        nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
        fwd_bits_out = (float)((fwd_nats - (double)nullsc) / LOG2);
        if (dp) dp->fwd_bits = fwd_bits_out;
        ok = xC_L > 0.f && isfinite(fwdsc);
        invZ = ok ? 1.0f / (xC_L * cm.move) : 0.f;
      }

      // ---------------- P2: multihit Backward + domain decoding + region scan (backward tables)
      WH_TICKB(4);
      orient(1);
      WH_TICKB(10);
      if (active && ok) {
        TransTab<Q, false> T;
        T.load(nullptr, trL, lane);
        const ScanC sc = scan_prepare(lane_product<Q, false>(T, BW_DD));
        float Mb[Q], Ib[Q];
#pragma unroll
        for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; }
        float xC = cm.move, xJ = 0.f, xN = 0.f, xB = 0.f;
        int eb = 0;
        // the special states of a row are requested one row ahead (see the envelope Backward sweep below)
        int S_i = BSPRI(SP_S * SP + L), S_m = BSPRI(SP_S * SP + L - 1);
        float E_i = BSPR(SP_E * SP + L), B_i = BSPR(SP_B * SP + L);
        float n_m = BSPR(SP_N * SP + L - 1), j_m = BSPR(SP_J * SP + L - 1), c_m = BSPR(SP_C * SP + L - 1);
#pragma unroll 1
        for (int i = L; i >= 0; i--) {
          asm volatile("" ::: "memory");
          const int ip1 = i >= 1 ? i - 1 : 0, ip2 = i >= 2 ? i - 2 : 0;
          const float E_n = BSPR(SP_E * SP + ip1), B_n = BSPR(SP_B * SP + ip1);
          const int S_m2 = BSPRI(SP_S * SP + ip2);
          const float n_m2 = BSPR(SP_N * SP + ip2), j_m2 = BSPR(SP_J * SP + ip2), c_m2 = BSPR(SP_C * SP + ip2);
          asm volatile("" ::: "memory");
          if (i < L) {
            const float part = backward_emit_lean<Q>(T, emL, emG, seq[i], Klds, lane, Mb);
            xB = wave_sum(part);
            xJ = fmaf(xJ, cm.loop, xB * cm.move);
            xC = xC * cm.loop;
            xN = fmaf(xN, cm.loop, xB * cm.move);
          }
          float xE = fmaf(xC, cm.EC, xJ * cm.EJ);
          if (i >= 1) backward_cells_lean<Q>(T, sc, Mb, Ib, xE);
          const float big = fmaxf(xB, xN);
          if (big > kRescaleHi) {
            const int e = f32_exponent(big);
            const float r = pow2f_int(-e);
#pragma unroll
            for (int p = 0; p < Q; p++) { Mb[p] *= r; Ib[p] *= r; }
            xB *= r; xJ *= r; xC *= r; xN *= r; xE *= r;
            eb += e;
          }
          const float s_i = ldexpf(invZ, S_i + eb - ef_L);
          const float pe = E_i * xE * s_i;
          const float pb = B_i * xB * s_i;
          float njc = 0.f;
          if (i >= 1) {
            const float s_p = ldexpf(invZ, S_m + eb - ef_L);
            njc = n_m * xN;
            njc = fmaf(j_m, xJ, njc);
            njc = fmaf(c_m, xC, njc);
            njc = njc * cm.loop * s_p;
          }
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) { spec[SP_E * SP + i] = pe; spec[SP_B * SP + i] = pb; spec[SP_N * SP + i] = njc; }
          __builtin_amdgcn_wave_barrier();
          S_i = S_m; S_m = S_m2; E_i = E_n; B_i = B_n; n_m = n_m2; j_m = j_m2; c_m = c_m2;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // region scan (A.4), 64 rows per fetch
        const float rt3 = 0.20f;
        region_scan_global(spec, SP, L, regs, lane, nenv, nreg, flags);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int e = 0; e < nenv; e++) {
          const int ri = regs[2 * e], rj = regs[2 * e + 1];
          float mx = -1.0f;
          const float e0 = BSPR(SP_C * SP + ri - 1), bj = BSPR(SP_J * SP + rj);
          for (int z = ri + lane; z <= rj; z += kWave) {
            const float u = BSPR(SP_C * SP + z) - e0, v = bj - BSPR(SP_J * SP + z - 1);
            mx = fmaxf(mx, fminf(u, v));
          }
          mx = wave_max(mx);
          if (mx >= rt3) { flags |= WH_FLAG_MULTI; multi_mask |= 1 << e; }
        }
        if (dp) { dp->nregions = nreg; dp->nenv = nenv; }
      }
      WH_TICKB(5);
      // a pair with a multidomain region is finished by resolve_kernel (A.4b); its single-domain regions
      // are still scored here, their results staged in LDS for the pair's queue record
      const bool queue_pair = multi_mask != 0 && a.rrecs != nullptr;
      float *envres = reinterpret_cast<float *>(regs + 3 * WH_MAX_ENVELOPES);
      if (queue_pair && lane == 0) for (int t = 0; t < 2 * WH_MAX_ENVELOPES; t++) envres[t] = 0.f;

      // ---------------- envelopes: workgroup-uniform loop over (envelope, attempt) steps
      float seqbias_sum = 0.f, sum_score = 0.f, sb2 = 0.f;
      int Ld_tot = 0, e = 0, attempt = 0;
      while (queue_pair && e < nenv && ((multi_mask >> e) & 1)) e++;       // multidomain regions are left to resolve_kernel
      bool pending = active && ok && e < nenv;
      while (__syncthreads_or(pending ? 1 : 0)) {
        int ri = 1, Ld = 0, ef_e = 0;
        float xC_e = 0.f, envsc = -INFINITY, domcorr = 0.f;
        const uint8_t *eseq = seq;
        // P3: unihit Forward of the envelope (forward tables)
        WH_TICKB(11);
        orient(0);
        WH_TICKB(10);
        if (pending) {
          ri = regs[2 * e]; Ld = regs[2 * e + 1] - ri + 1; eseq = seq + (ri - 1);
          const float keep_scale = attempt == 0 ? kKeepScaleB : -1.0f;
          TransTab<Q, false> T;
          T.load(nullptr, trL, lane);
          const ScanC sc = scan_prepare(lane_product<Q, false>(T, FW_D2));
          forward_sweep_lean<Q, true>(T, sc, emL, emG, Klds, eseq, Ld, cu, spec, SP, Fs, keep_scale, lane, xC_e, ef_e);
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          envsc = (float)((double)ef_e * LOG2 + log((double)(xC_e * cu.move)));
        }
        // P4: unihit Backward + posterior accumulation (backward tables)
        WH_TICKB(7);
        orient(1);
        WH_TICKB(10);
        if (pending) {
          bool done_env = true;
          if (xC_e > 0.f) {
            const float invZe = 1.0f / (xC_e * cu.move);
            TransTab<Q, false> T;
            T.load(nullptr, trL, lane);
            const ScanC sc = scan_prepare(lane_product<Q, false>(T, BW_DD));
            float Mb[Q], Ib[Q], fM[Q];
#pragma unroll
            for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; fM[p] = 0.f; }
            float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f, xfac = 0.f, fIs = 0.f;

            const int src = kWave - 1 - lane;
            // Everything a row needs from HBM is requested one row ahead (the special states of row i-1 and the
            // mask word of row i-1 during row i) or at the top of its own row (the stored Forward cells, consumed
            // after the cell update): with one or two waves per SIMD nothing else hides these round trips, and
            // four of them in series per row were half of the kernel's time.
            int S_i = BSPRI(SP_S * SP + Ld), S_m = BSPRI(SP_S * SP + Ld - 1), S_prev = S_i;
            unsigned mword = src < 32 ? BSPRU(SP_ML * SP + Ld) : BSPRU(SP_MH * SP + Ld);
            float n_m = BSPR(SP_N * SP + Ld - 1), j_m = BSPR(SP_J * SP + Ld - 1), c_m = BSPR(SP_C * SP + Ld - 1);
#pragma unroll 1
            for (int i = Ld; i >= 1; i--) {
              asm volatile("" ::: "memory");
              // requests for row i-1 (clamped at the envelope start; the values are not used there)
              const int im2 = i >= 2 ? i - 2 : 0, im1 = i >= 2 ? i - 1 : 1;
              const int S_m2 = BSPRI(SP_S * SP + im2);
              const unsigned mword_n = src < 32 ? BSPRU(SP_ML * SP + im1) : BSPRU(SP_MH * SP + im1);
              const float n_m2 = BSPR(SP_N * SP + im2), j_m2 = BSPR(SP_J * SP + im2), c_m2 = BSPR(SP_C * SP + im2);
              // stored Forward cells of row i (only the lanes that own a stored block)
              const bool have = (mword >> (src & 31)) & 1u;
              float4 fm[Q / 4], fi[Q / 4];
              if (have) {
                const float4 *row = reinterpret_cast<const float4 *>(Fs) + (size_t)i * (2 * (Q / 4) * kWave) + src;
#pragma unroll
                for (int p4 = 0; p4 < Q / 4; p4++) {
                  fm[p4] = nt_load4(row + (Q / 4 - 1 - p4) * kWave);
                  fi[p4] = nt_load4(row + (Q / 4 + Q / 4 - 1 - p4) * kWave);
                }
              }
              asm volatile("" ::: "memory");
              if (i < Ld) {
                mirror_scale<Q>(S_prev - S_i, Mb, Ib, xJ, xC, xN);
                const float part = backward_emit_lean<Q>(T, emL, emG, eseq[i], Klds, lane, Mb);
                xB = wave_sum(part);
                xJ = fmaf(xJ, cu.loop, xB * cu.move);
                xC = xC * cu.loop;
                xN = fmaf(xN, cu.loop, xB * cu.move);
              }
              float xE = fmaf(xC, cu.EC, xJ * cu.EJ);
              backward_cells_lean<Q>(T, sc, Mb, Ib, xE);
              clamp_backward<Q>(Mb, Ib, xB, xJ, xC, xN);
              // mirrored scaling (wh_device.h, "envelope Backward scaling")
              const float s_i = invZe;
              const float s_p = ldexpf(invZe, S_m - S_i);
              if (have) {
                float idot = 0.f;
#pragma unroll
                for (int p4 = 0; p4 < Q / 4; p4++) {
                  // reversed order: component 3-j of the forward-ordered vector is position 4*p4+j
                  fM[4 * p4 + 0] = fmaf(fm[p4].w * Mb[4 * p4 + 0], s_i, fM[4 * p4 + 0]);
                  fM[4 * p4 + 1] = fmaf(fm[p4].z * Mb[4 * p4 + 1], s_i, fM[4 * p4 + 1]);
                  fM[4 * p4 + 2] = fmaf(fm[p4].y * Mb[4 * p4 + 2], s_i, fM[4 * p4 + 2]);
                  fM[4 * p4 + 3] = fmaf(fm[p4].x * Mb[4 * p4 + 3], s_i, fM[4 * p4 + 3]);
                  idot = fmaf(fi[p4].w, Ib[4 * p4 + 0], idot); idot = fmaf(fi[p4].z, Ib[4 * p4 + 1], idot);
                  idot = fmaf(fi[p4].y, Ib[4 * p4 + 2], idot); idot = fmaf(fi[p4].x, Ib[4 * p4 + 3], idot);
                }
                fIs = fmaf(idot, s_i, fIs);
              }
              float nj = n_m * xN;
              nj = fmaf(j_m, xJ, nj);
              nj = fmaf(c_m, xC, nj);
              xfac = fmaf(nj * cu.loop, s_p, xfac);
              S_prev = S_i; S_i = S_m; S_m = S_m2; mword = mword_n; n_m = n_m2; j_m = j_m2; c_m = c_m2;
            }
            const float norm = 1.0f / (float)Ld;
            float sm = 0.f;
#pragma unroll
            for (int p = 0; p < Q; p++) sm += fM[p];
            sm = wave_sum(sm);
            const float si = wave_sum(fIs);
            const float deficit = fabsf((float)Ld - (sm + si + xfac));
            if (attempt == 0 && !(deficit <= kMassTolB * (float)Ld)) {
              attempt = 1;          // certificate failed: redo this envelope with every line stored
              done_env = false;
            } else {
              if (attempt == 1) flags |= WH_FLAG_EXACT;
              float mine = 1.0f;
              for (int x = 0; x < a.K; x++) {
                float od[Q];
                load_em_rev<Q>(od, emL, emG, x, Klds, lane);
                float s = 0.f;
#pragma unroll
                for (int p = 0; p < Q; p++) s = fmaf(fM[p], od[p], s);
                s = wave_sum(s);
                if (lane == x) mine = (s + si) * norm + xfac * norm;
              }
              __builtin_amdgcn_wave_barrier();
              if (lane < a.K) n2tab[lane] = mine;
              __builtin_amdgcn_wave_barrier();
              if (lane >= a.K && lane < a.Kp) {
                const uint32_t m = a.degen[lane];
                float s = 0.f; int n = 0;
                for (int x = 0; x < a.K; x++) if (m & (1u << x)) { s += n2tab[x]; n++; }
                mine = n > 0 ? s / (float)n : 1.0f;
              }
              __builtin_amdgcn_wave_barrier();
              if (lane < a.Kp) n2tab[lane] = logf(mine);
              __builtin_amdgcn_wave_barrier();
              float dc = 0.f;
              for (int t = lane; t < Ld; t += kWave) dc += n2tab[eseq[t]];
              domcorr = wave_sum(dc);
            }
          }
          if (done_env) {
            seqbias_sum += domcorr;
            if (envsc - domcorr > 0.0f) { sum_score += envsc; Ld_tot += Ld; sb2 += domcorr; }
            if (dp) { dp->env_i[e] = ri; dp->env_j[e] = ri + Ld - 1; dp->envsc[e] = envsc; dp->domcorr[e] = domcorr; }
            if (queue_pair && lane == 0) { envres[e] = envsc; envres[WH_MAX_ENVELOPES + e] = domcorr; }
            e++;
            while (queue_pair && e < nenv && ((multi_mask >> e) & 1)) e++;
            attempt = 0;
            pending = e < nenv;
          }
        }
      }

      WH_TICKB(8);
      // ---------------- A.6 score assembly
      if (queue_pair) {
        __builtin_amdgcn_wave_barrier();
        int slot = 0;
        if (lane == 0) slot = atomicAdd(a.rcount, 1);
        slot = __shfl(slot, 0);
        if (slot < a.rcap && lane == 0) {     // resolve_kernel writes the final score and flags of this pair
          ResolveRec *rr = a.rrecs + slot;
          rr->q = qi; rr->h = h; rr->fwdsc = fwdsc; rr->fwd_bits = fwd_bits_out; rr->nreg = nreg; rr->nenv = nenv;
          rr->multi_mask = multi_mask; rr->flags = flags;
          for (int t = 0; t < nenv; t++) { rr->ri[t] = regs[2 * t]; rr->rj[t] = regs[2 * t + 1]; rr->envsc[t] = envres[t]; rr->domcorr[t] = envres[WH_MAX_ENVELOPES + t]; }
        }
      } else if (active && ok && nenv > 0) {
        const float lomega = (float)log(1.0 / 256.0);
        const float seqbias = flogsum0_big(lomega + seqbias_sum);
        float pre_score = (float)(((double)fwdsc - (double)nullsc) / LOG2);
        float seq_score = (float)(((double)fwdsc - (double)(nullsc + seqbias)) / LOG2);
        sb2 = flogsum0_big(lomega + sb2);
        sum_score += (float)((double)(L - Ld_tot) * log((double)((float)L / (float)(L + 3))));
        const float pre2 = (float)(((double)sum_score - (double)nullsc) / LOG2);
        sum_score = (float)(((double)sum_score - (double)(nullsc + sb2)) / LOG2);
        if (Ld_tot > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2; flags |= WH_FLAG_OVERRIDE; }
        decibits = (int)rint((double)seq_score * 10.0);
        flags |= WH_FLAG_REPORTED;
        if (dp) { dp->seq_score = seq_score; dp->pre_score = pre_score; dp->seqbias_nats = seqbias; }
      }
      if (in_range && lane == 0) {
        a.decibits[out] = decibits;
        a.flags[out] = (uint8_t)flags;
        if (a.fwd_bits) a.fwd_bits[out] = fwd_bits_out;
      }
    }
  }
}

template <int Q, int TH>
static hipError_t launch_big_th(const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_big_kernel<Q, TH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((score_big_kernel<Q, TH>), dim3(blocks), dim3(threads), lds, s, a);
  return hipGetLastError();
}

// 256 threads: one wave per SIMD with the whole register file (a DP row of 3 x 48 cells, the double-buffered
// table pieces and the landing registers of the row prefetch)
#define WH_BIG_LAUNCH launch_score_big
#define WH_BIG_TH 256

hipError_t WH_BIG_LAUNCH(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  if (threads != WH_BIG_TH) return hipErrorInvalidValue;
  switch (Q) {
    case 20: return launch_big_th<20, WH_BIG_TH>(a, blocks, threads, lds, s);
    case 24: return launch_big_th<24, WH_BIG_TH>(a, blocks, threads, lds, s);
    case 28: return launch_big_th<28, WH_BIG_TH>(a, blocks, threads, lds, s);
    case 32: return launch_big_th<32, WH_BIG_TH>(a, blocks, threads, lds, s);
    case 36: return launch_big_th<36, WH_BIG_TH>(a, blocks, threads, lds, s);
    case 40: return launch_big_th<40, WH_BIG_TH>(a, blocks, threads, lds, s);
    case 44: return launch_big_th<44, WH_BIG_TH>(a, blocks, threads, lds, s);
    case 48: return launch_big_th<48, WH_BIG_TH>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace wh
