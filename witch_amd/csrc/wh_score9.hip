// Scoring kernel, TWO QUERIES PER WAVEFRONT (round 4): the phase-call kernel of wh_score7.hip with the three full-width
// sweeps of a pair - P1 multihit Forward, P2 multihit Backward + decoding, P3 unihit Forward of the envelope - run for two
// queries of one model at once (wh_pair.h): every transition-table piece read from LDS is applied to both rows.
//
// The one-query kernel is bound by the CU's LDS read port (37 KiB of tables per DP row, the port 72 % busy at twelve
// waves per CU while the vector ALUs are busy 45-50 %); with two queries per wave a row reads 20.5 KiB, and each wave
// has two independent instruction streams to issue from, so eight waves per CU (two per SIMD, 256 registers) do what
// twelve could not.  The envelope Backward sweep on its node window (P4) keeps one query per call: a window has its own
// tables in registers and nothing to share.  (hmmsearch --max per pair: witch_msa/gcmm/algorithm.py:526-532.)
//
// Everything a pair sweep does per query is the single sweep's arithmetic in the same order, so a query's result does
// not depend on which other query shared its wave (tests/test_gpu_parity.py: pair kernel == one-query kernel).
// Used for batches whose longest query fits the per-wave LDS block (two sets of per-row arrays); other batches, models
// of more than 16 cells per lane and the long-query mode stay with wh_score7.hip.
#include <hip/hip_runtime.h>

#define WH_K7NS k9
#define WH_K7LAUNCH launch_score9_unused
#define WH_SWEEPS_ONLY 1
#define WH_SLIM_SPEC 1
#include "wh_score7.hip"      // the one-query sweeps (P4 on a window / at full width, dense redo, region scan) in namespace k9
#include "wh_pair.h"

namespace wh {
namespace k9 {

// what a pair sweep needs (<= 16 dwords: travels in argument registers, see WaveCtx)
struct PairCtx {
  lds_f *emL;
  const glb_f *emG;
  lds_f *fwL, *bwL;
  lds_f *spec0;               // per-row arrays of query 0; query 1's block starts <blk> floats further on
  int blk;
  glb_f *Fs0;                 // Forward slab of query 0; query 1's <fsd> floats further on
  int fsd;
  int SP, lane;
  int alpha;
  int ret;                    // offset (floats) of a query block's result slots
};
#if defined(__HIP_DEVICE_COMPILE__)
static_assert(sizeof(PairCtx) <= 64, "PairCtx must stay register-passed");
#endif

// Arguments of a non-inlined function arrive in VECTOR registers, and the compiler keeps what it cannot prove uniform
// there (row addresses, lengths, strides: ~20 registers and, at 256, spills in the row loops).  Everything below is
// wave-uniform by construction: read lane 0's copy, the values then live in scalar registers.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
template <class T>
__device__ __forceinline__ T *uni_lds(T *p) {      // LDS pointers are 32 bits
  typedef __attribute__((address_space(3))) char lds_c;
  const int v = __builtin_amdgcn_readfirstlane((int)(size_t)(lds_c *)p);
  return (T *)(lds_c *)(size_t)(unsigned)v;
}
template <class T>
__device__ __forceinline__ T *uni_glb(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}

// ---------------------------------------------------------------- P1 / P3 for two queries
// results (C(L), scale exponent) go to the two result slots of each query's LDS block
template <int Q, bool STORE, bool MULTI>
__device__ __noinline__ void sweep_forward2(const PairCtx c, lds_u8 *seqA, lds_u8 *seqB, int LA, int LB, int LcfgA, int LcfgB, float keep_scale) {
  TransTab<Q, false> T;
  T.load(nullptr, (const float *)uni_lds(c.fwL), c.lane);
  const ScanC sc = scan_prepare(lane_product<Q, false>(T, FW_D2));
  lds_f *spec0 = uni_lds(c.spec0);
  glb_f *Fs0 = uni_glb(c.Fs0);
  const int blk = uni(c.blk);
  PairQ pq[2];
  pq[0].seq = (const uint8_t *)uni_lds(seqA); pq[0].L = uni(LA); pq[0].spec = (float *)spec0; pq[0].Fs = (float *)Fs0;
  pq[1].seq = (const uint8_t *)uni_lds(seqB); pq[1].L = uni(LB); pq[1].spec = (float *)(spec0 + blk); pq[1].Fs = (float *)(Fs0 + uni(c.fsd));
  const LenCfg cfg[2] = {len_config(uni(LcfgA), MULTI), len_config(uni(LcfgB), MULTI)};
  float xC[2];
  int ef[2];
  forward_sweep2<Q, STORE>(T, sc, (const float *)uni_lds(c.emL), (const float *)uni_glb(c.emG), uni((c.alpha >> 16) & 255), pq, cfg, uni(c.SP), unif(keep_scale), c.lane, xC, ef);
  if (c.lane == 0) {
    float *r0 = (float *)spec0 + uni(c.ret), *r1 = r0 + blk;
    r0[0] = xC[0]; reinterpret_cast<int *>(r0)[1] = ef[0];
    r1[0] = xC[1]; reinterpret_cast<int *>(r1)[1] = ef[1];
  }
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- P2 for two queries (sweep_backward_decode)
template <int Q>
__device__ __noinline__ void sweep_backward_decode2(const PairCtx c, lds_u8 *seqA, lds_u8 *seqB, int LA, int LB, float invZA, float invZB, int efA, int efB) {
  const float *emL = (const float *)uni_lds(c.emL);
  TransTab<Q, false> T;
  T.load(nullptr, (const float *)uni_lds(c.bwL), c.lane);
  const ScanC sc = scan_prepare(lane_product<Q, false>(T, BW_DD));
  const int lane = c.lane, SP = uni(c.SP), Klds = uni((c.alpha >> 16) & 255);
  const float *emG = (const float *)uni_glb(c.emG);
  const uint8_t *seq[2] = {(const uint8_t *)uni_lds(seqA), (const uint8_t *)uni_lds(seqB)};
  lds_f *spec0 = uni_lds(c.spec0);
  float *spec[2] = {(float *)spec0, (float *)(spec0 + uni(c.blk))};
  const int L[2] = {uni(LA), uni(LB)};
  const float invZ[2] = {unif(invZA), unif(invZB)};
  const int efL[2] = {uni(efA), uni(efB)};
  const LenCfg cm[2] = {len_config(L[0], true), len_config(L[1], true)};
  float Mb[2][Q], Ib[2][Q];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int p = 0; p < Q; p++) { Mb[n][p] = 0.f; Ib[n][p] = 0.f; }
  float xC[2] = {cm[0].move, cm[1].move}, xJ[2] = {0.f, 0.f}, xN[2] = {0.f, 0.f}, xB[2] = {0.f, 0.f};
  int eb[2] = {0, 0};
  const int Lmax = LA > LB ? LA : LB;
  // step t handles row i = L - t of each query; a query whose rows are used up idles on row 0 (nothing is written)
#pragma unroll 1
  for (int t = 0; t <= Lmax; t++) {
    asm volatile("" ::: "memory");
    int i[2];
    bool act[2];
#pragma unroll
    for (int n = 0; n < 2; n++) { act[n] = t <= L[n]; i[n] = act[n] ? L[n] - t : 0; }
    if (t > 0) {
      int x[2];
#pragma unroll
      for (int n = 0; n < 2; n++) x[n] = __builtin_amdgcn_readfirstlane((int)seq[n][i[n]]);
      float part[2];
      backward_emit2<Q>(T, emL, emG, x, Klds, lane, Mb, part);
      wave_sum2(part[0], part[1]);
#pragma unroll
      for (int n = 0; n < 2; n++) {
        xB[n] = part[n];
        xJ[n] = fmaf(xJ[n], cm[n].loop, xB[n] * cm[n].move);
        xC[n] = xC[n] * cm[n].loop;
        xN[n] = fmaf(xN[n], cm[n].loop, xB[n] * cm[n].move);
      }
    }
    float xE[2];
#pragma unroll
    for (int n = 0; n < 2; n++) xE[n] = fmaf(xC[n], cm[n].EC, xJ[n] * cm[n].EJ);
    // (the single sweep skips the cell update at row 0; its result is not used there, so the pair runs it regardless)
    backward_cells2<Q>(T, sc, Mb, Ib, xE);
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const float big = fmaxf(xB[n], xN[n]);
      if (big > kRescaleHi) {
        const int e = f32_exponent(big);
        const float r = pow2f_int(-e);
#pragma unroll
        for (int p = 0; p < Q; p++) { Mb[n][p] *= r; Ib[n][p] *= r; }
        xB[n] *= r; xJ[n] *= r; xC[n] *= r; xN[n] *= r; xE[n] *= r;
        eb[n] += e;
      }
    }
    float pe[2], pb[2], njc[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const float *sp = spec[n];
      const int *spi = reinterpret_cast<const int *>(sp);
      const int ii = i[n];
      const float s_i = ldexpf(invZ[n], spi[SP_S * SP + ii] + eb[n] - efL[n]);
      pe[n] = sp[SP_E * SP + ii] * xE[n] * s_i;
      pb[n] = sp[SP_B * SP + ii] * xB[n] * s_i;
      njc[n] = 0.f;
      if (ii >= 1) {
        const float s_p = ldexpf(invZ[n], spi[SP_S * SP + ii - 1] + eb[n] - efL[n]);
        float v = sp[SP_N * SP + ii - 1] * xN[n];
        v = fmaf(sp[SP_J * SP + ii - 1], xJ[n], v);
        v = fmaf(sp[SP_C * SP + ii - 1], xC[n], v);
        njc[n] = v * cm[n].loop * s_p;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
#pragma unroll
      for (int n = 0; n < 2; n++)
        if (act[n]) { spec[n][SP_E * SP + i[n]] = pe[n]; spec[n][SP_B * SP + i[n]] = pb[n]; spec[n][SP_N * SP + i[n]] = njc[n]; }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

#define WH_TICK9(slot) do { if (a.stats) { const long long t_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(a.stats + (slot), (unsigned long long)(t_now - t_last)); t_last = t_now; } } while (0)

template <int Q>
__global__ __launch_bounds__(512) void score_kernel9(ScoreArgs a) {
  constexpr int TH = 512;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  volatile int *s_item_p = reinterpret_cast<volatile int *>(smem_raw);
  float *smem = smem_raw + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  float *emL = smem;
  float *trL = smem + (size_t)a.K * TBL;
  constexpr int NARR = FW_NARR;
  const int SP = a.SP;
  const int blk = a.wave_lds / 2;                                   // floats per query block
  float *wbase = trL + 2 * NARR * TBL + (size_t)wave * a.wave_lds;
  const int o_n2 = kSpArr * SP, o_regs = o_n2 + 32, o_ret = o_regs + kRegsInts, o_seq = o_ret + 8;
  // one-query contexts of the two blocks, and the pair context
  WaveCtx c[2];
  uint32_t degen = 0;
  for (int t = 0; t < 32; t++) if (t == lane) degen = a.degen[t];
  float *Fs0 = a.scratch + ((size_t)blockIdx.x * nwaves + wave) * a.scratch_stride;
  const int fsd = (int)(a.scratch_stride / 2);
#pragma unroll
  for (int n = 0; n < 2; n++) {
    c[n].emL = (lds_f *)emL; c[n].fwL = (lds_f *)trL; c[n].bwL = (lds_f *)(trL + NARR * TBL);
    c[n].spec = (lds_f *)(wbase + n * blk); c[n].n2tab = (lds_f *)(wbase + n * blk + o_n2);
    c[n].specg = nullptr;
    c[n].degen = degen;
    c[n].Fs = (glb_f *)(Fs0 + (size_t)n * fsd);
    c[n].SP = SP; c[n].alpha = a.K | (a.Kp << 8) | (a.K << 16); c[n].lane = lane;
  }
  PairCtx pc;
  pc.emL = c[0].emL; pc.fwL = c[0].fwL; pc.bwL = c[0].bwL; pc.spec0 = c[0].spec; pc.blk = blk; pc.Fs0 = c[0].Fs; pc.fsd = fsd;
  pc.SP = SP; pc.lane = lane; pc.alpha = c[0].alpha; pc.ret = o_ret;
  const double LOG2 = 0.69314718055994529;
  int cur_h = -1;
  const DevHMM *hm = nullptr;
  unsigned n_w256 = 0, n_w512 = 0, n_wfail = 0, n_full = 0;   // this wave's envelope Backward sweeps by path (wh_last_score_paths)

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
      hm = a.hmms + h;
      const float4 *src = reinterpret_cast<const float4 *>(a.tables + hm->em_off);
      float4 *dst = reinterpret_cast<float4 *>(emL);
      for (int t = threadIdx.x; t < a.K * TBL / 4; t += blockDim.x) dst[t] = src[t];
      const float4 *s1 = reinterpret_cast<const float4 *>(a.tables + hm->fw_off);
      const float4 *s2 = reinterpret_cast<const float4 *>(a.tables + hm->bw_off);
      float4 *d1 = reinterpret_cast<float4 *>(trL);
      for (int t = threadIdx.x; t < NARR * TBL / 4; t += blockDim.x) { d1[t] = s1[t]; d1[NARR * TBL / 4 + t] = s2[t]; }
      cur_h = h;
      __syncthreads();
    }
    c[0].emG = c[1].emG = pc.emG = (const glb_f *)(a.tables + hm->em_off);

    for (int64_t qpos = q_lo + 2 * wave; qpos < q_hi; qpos += 2 * nwaves) {
      const int nqw = qpos + 1 < q_hi ? 2 : 1;
      // ---- per-query state (wave-uniform scalars)
      int64_t qi[2] = {0, 0};
      size_t outp[2] = {0, 0};
      int L[2] = {0, 0}, flags[2] = {0, 0}, decibits[2] = {0, 0};
      float fwd_bits_out[2] = {-INFINITY, -INFINITY}, fwdsc[2] = {0.f, 0.f}, nullsc[2] = {0.f, 0.f};
      bool ok[2] = {false, false};
      wh_pair_detail *dp[2] = {nullptr, nullptr};
      uint8_t *seq[2];
      int *regs[2];
      float *ret[2];
#pragma unroll
      for (int n = 0; n < 2; n++) {
        regs[n] = reinterpret_cast<int *>(wbase + n * blk + o_regs);
        ret[n] = wbase + n * blk + o_ret;
        seq[n] = reinterpret_cast<uint8_t *>(wbase + n * blk + o_seq);
        if (n < nqw) {
          qi[n] = a.qorder ? a.qorder[qpos + n] : qpos + n;
          const int64_t off = a.offsets[qi[n]];
          L[n] = (int)(a.offsets[qi[n] + 1] - off);
          outp[n] = (size_t)qi[n] * a.H + h;
          dp[n] = (a.detail && lane == 0) ? a.detail + outp[n] : nullptr;
          if (dp[n]) {
            dp[n]->fwd_bits = -INFINITY; dp[n]->seq_score = 0.f; dp[n]->pre_score = 0.f; dp[n]->seqbias_nats = 0.f;
            dp[n]->nregions = 0; dp[n]->nenv = 0;
          }
          ok[n] = L[n] > 0 && L[n] <= a.Lcap;
          if (ok[n])
            for (int t = lane; t < L[n]; t += kWave) {
              int r = a.residues[off + t];
              seq[n][t] = (uint8_t)(r < a.Kp ? r : a.Kp - 1);
            }
        }
      }
      __builtin_amdgcn_wave_barrier();
      long long t_last = a.stats ? __builtin_readcyclecounter() : 0;
      // ---------------- P1
      FwdOut f1[2] = {{0.f, 0}, {0.f, 0}};
      if (ok[0] && ok[1]) {
        sweep_forward2<Q, false, true>(pc, (lds_u8 *)seq[0], (lds_u8 *)seq[1], L[0], L[1], L[0], L[1], 0.f);
#pragma unroll
        for (int n = 0; n < 2; n++) { f1[n].xC = ret[n][0]; f1[n].ef = reinterpret_cast<int *>(ret[n])[1]; }
      } else {
#pragma unroll
        for (int n = 0; n < 2; n++)
          if (ok[n]) f1[n] = sweep_forward<Q, false, TH, false>(c[n], (lds_u8 *)seq[n], L[n], len_config(L[n], true), 0.f);
      }
      bool good[2] = {false, false};
#pragma unroll
      for (int n = 0; n < 2; n++) {
        if (!ok[n]) continue;
        const LenCfg cm = len_config(L[n], true);
        const double fwd_nats = (double)f1[n].ef * LOG2 + log((double)(f1[n].xC * cm.move));
        fwdsc[n] = (float)fwd_nats;
        const float p1 = (float)L[n] / (float)(L[n] + 1);
        nullsc[n] = (float)((double)(float)L[n] * log((double)p1) + log(1.0 - (double)p1));
        fwd_bits_out[n] = (float)((fwd_nats - (double)nullsc[n]) / LOG2);
        if (dp[n]) dp[n]->fwd_bits = fwd_bits_out[n];
        good[n] = f1[n].xC > 0.f && isfinite(fwdsc[n]);
      }
      WH_TICK9(4);
      // ---------------- P2 + region scan
      if (good[0] && good[1]) {
        sweep_backward_decode2<Q>(pc, (lds_u8 *)seq[0], (lds_u8 *)seq[1], L[0], L[1], 1.0f / (f1[0].xC * len_config(L[0], true).move),
                                  1.0f / (f1[1].xC * len_config(L[1], true).move), f1[0].ef, f1[1].ef);
      } else {
#pragma unroll
        for (int n = 0; n < 2; n++)
          if (good[n]) {
            const LenCfg cm = len_config(L[n], true);
            sweep_backward_decode<Q, TH, false>(c[n], (lds_u8 *)seq[n], L[n], cm, 1.0f / (f1[n].xC * cm.move), f1[n].ef);
          }
      }
      WH_TICK9(5);
      int nenv[2] = {0, 0}, nreg[2] = {0, 0}, multi_mask[2] = {0, 0};
#pragma unroll
      for (int n = 0; n < 2; n++) {
        if (!good[n]) continue;
        const RegOut ro = region_scan<TH, false>(c[n].spec, nullptr, SP, L[n], (lds_i *)regs[n], lane);
        nenv[n] = ro.nenv; nreg[n] = ro.nreg; multi_mask[n] = ro.flags >> 8;
        flags[n] |= ro.flags & 0xFF;
        if (dp[n]) { dp[n]->nregions = nreg[n]; dp[n]->nenv = nenv[n]; }
      }
      WH_TICK9(6);
      // ---------------- envelopes
      float seqbias_sum[2] = {0.f, 0.f}, sum_score[2] = {0.f, 0.f}, sb2[2] = {0.f, 0.f};
      int Ld_tot[2] = {0, 0};
      bool queue_pair[2];
      float *envres[2];
#pragma unroll
      for (int n = 0; n < 2; n++) {
        queue_pair[n] = multi_mask[n] != 0 && a.rrecs != nullptr;
        envres[n] = reinterpret_cast<float *>(regs[n] + 3 * WH_MAX_ENVELOPES);
      }
      const int nenv_max = nenv[0] > nenv[1] ? nenv[0] : nenv[1];
      for (int e = 0; e < nenv_max; e++) {
        // which of the two queries score their envelope e here (the others: none, or a multidomain region left to the resolver)
        bool todo[2];
        int ri[2] = {1, 1}, Ld[2] = {0, 0};
#pragma unroll
        for (int n = 0; n < 2; n++) {
          todo[n] = e < nenv[n];
          if (todo[n] && queue_pair[n] && ((multi_mask[n] >> e) & 1)) {
            if (lane == 0) { envres[n][e] = 0.f; envres[n][WH_MAX_ENVELOPES + e] = 0.f; }
            todo[n] = false;
          }
          if (todo[n]) { ri[n] = regs[n][2 * e]; Ld[n] = regs[n][2 * e + 1] - ri[n] + 1; }
        }
        const float keep0 = a.keep_scale > 0.f ? a.keep_scale : kKeepScale7;
        FwdOut f3[2] = {{0.f, 0}, {0.f, 0}};
        if (todo[0] && todo[1]) {
          sweep_forward2<Q, true, false>(pc, (lds_u8 *)(seq[0] + (ri[0] - 1)), (lds_u8 *)(seq[1] + (ri[1] - 1)), Ld[0], Ld[1], L[0], L[1], keep0);
#pragma unroll
          for (int n = 0; n < 2; n++) { f3[n].xC = ret[n][0]; f3[n].ef = reinterpret_cast<int *>(ret[n])[1]; }
        } else {
#pragma unroll
          for (int n = 0; n < 2; n++)
            if (todo[n]) f3[n] = sweep_forward<Q, true, TH, false>(c[n], (lds_u8 *)(seq[n] + (ri[n] - 1)), Ld[n], len_config(L[n], false), keep0);
        }
        WH_TICK9(7);
#pragma unroll
        for (int n = 0; n < 2; n++) {
          if (!todo[n]) continue;
          const LenCfg cu = len_config(L[n], false);
          const uint8_t *eseq = seq[n] + (ri[n] - 1);
          const int Ldn = Ld[n];
          float envsc = -INFINITY, domcorr = 0.f;
#pragma unroll 1
          for (int attempt = 0; attempt < 2; attempt++) {
            const FwdOut f3n = attempt == 0 ? f3[n] : sweep_forward<Q, true, TH, false>(c[n], (lds_u8 *)eseq, Ldn, cu, -1.0f);
            // the rows were written by other lanes of this wave: order the stores before the loads
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            envsc = (float)((double)f3n.ef * LOG2 + log((double)(f3n.xC * cu.move)));
            domcorr = 0.f;
            if (!(f3n.xC > 0.f)) break;
            const float tol = attempt == 0 ? kMassTol7 : INFINITY;
            P4Out p4;
            bool have4 = false;
            if constexpr (Q >= 8) {
              if (attempt == 0 && !a.no_window) {
                const unsigned *su = reinterpret_cast<const unsigned *>((const float *)c[n].spec);
                const unsigned long long um = ((unsigned long long)su[kSpMH * SP] << 32) | su[kSpML * SP];
                if (um != 0) {
                  int lo = __builtin_ctzll(um), hi = 63 - __builtin_clzll(um);
                  lo = lo > 1 ? lo - 2 : 0; hi = hi < 63 ? hi + 1 : 63;
                  const int nodes = (hi - lo + 1) * Q;
                  if (a.stats && lane == 0) { atomicAdd(a.stats + 12, (unsigned long long)(hi - lo + 1)); atomicAdd(a.stats + 14, 1ull); atomicAdd(a.stats + 15, (unsigned long long)__builtin_popcountll(um)); }
                  if (nodes <= 4 * kWave) {
                    const int m0 = min((63 - hi) * Q, kWave * (Q - 4));
                    p4 = sweep_backward_null2_win<4, Q, TH, false>(c[n], (lds_u8 *)eseq, Ldn, cu, 1.0f / (f3n.xC * cu.move), kWinTol7, m0);
                    have4 = fabsf((float)Ldn - p4.mass) <= kWinTol7 * (float)Ldn;
                    if (have4) n_w256++; else n_wfail++;
                    if (a.stats && lane == 0) atomicAdd(a.stats + (have4 ? 0 : 2), 1ull);
                  } else if (Q % 8 == 0 && Q > 8 && nodes <= 8 * kWave) {
                    const int m0 = min((63 - hi) * Q, kWave * (Q - 8));
                    p4 = sweep_backward_null2_win<(Q % 8 == 0 ? 8 : 4), Q, TH, false>(c[n], (lds_u8 *)eseq, Ldn, cu, 1.0f / (f3n.xC * cu.move), kWinTol7, m0);
                    have4 = fabsf((float)Ldn - p4.mass) <= kWinTol7 * (float)Ldn;
                    if (have4) n_w512++; else n_wfail++;
                    if (a.stats && lane == 0) atomicAdd(a.stats + (have4 ? 1 : 2), 1ull);
                  }
                }
              }
            }
            if (!have4) {
              p4 = sweep_backward_null2<Q, TH, false>(c[n], (lds_u8 *)eseq, Ldn, cu, 1.0f / (f3n.xC * cu.move), f3n.ef, tol);
              n_full++;
              if (a.stats && lane == 0) atomicAdd(a.stats + 3, 1ull);
            }
            domcorr = p4.domcorr;
            if (attempt == 0 && !(fabsf((float)Ldn - p4.mass) <= kMassTol7 * (float)Ldn)) continue;
            if (attempt == 1) flags[n] |= WH_FLAG_EXACT;
            break;
          }
          seqbias_sum[n] += domcorr;
          if (envsc - domcorr > 0.0f) { sum_score[n] += envsc; Ld_tot[n] += Ldn; sb2[n] += domcorr; }
          if (dp[n]) { dp[n]->env_i[e] = ri[n]; dp[n]->env_j[e] = ri[n] + Ldn - 1; dp[n]->envsc[e] = envsc; dp[n]->domcorr[e] = domcorr; }
          if (queue_pair[n] && lane == 0) { envres[n][e] = envsc; envres[n][WH_MAX_ENVELOPES + e] = domcorr; }
        }
        WH_TICK9(8);
      }
      // ---------------- queue record or A.6 score assembly, per query
#pragma unroll
      for (int n = 0; n < 2; n++) {
        if (n >= nqw) continue;
        if (nenv[n] > 0) {
          if (queue_pair[n]) {
            __builtin_amdgcn_wave_barrier();
            int slot = 0;
            if (lane == 0) slot = atomicAdd(a.rcount, 1);
            slot = __shfl(slot, 0);
            if (slot < a.rcap && lane == 0) {
              ResolveRec *rr = a.rrecs + slot;
              rr->q = qi[n]; rr->h = h; rr->fwdsc = fwdsc[n]; rr->fwd_bits = fwd_bits_out[n]; rr->nreg = nreg[n]; rr->nenv = nenv[n];
              rr->multi_mask = multi_mask[n]; rr->flags = flags[n];
              for (int e = 0; e < nenv[n]; e++) { rr->ri[e] = regs[n][2 * e]; rr->rj[e] = regs[n][2 * e + 1]; rr->envsc[e] = envres[n][e]; rr->domcorr[e] = envres[n][WH_MAX_ENVELOPES + e]; }
            }
            // provisional result: resolve_kernel writes the final score and flags of this pair
          } else {
            const float lomega = (float)log(1.0 / 256.0);
            const float seqbias = flogsum0_v7(lomega + seqbias_sum[n]);
            float pre_score = (float)(((double)fwdsc[n] - (double)nullsc[n]) / LOG2);
            float seq_score = (float)(((double)fwdsc[n] - (double)(nullsc[n] + seqbias)) / LOG2);
            const float sb2f = flogsum0_v7(lomega + sb2[n]);
            float ss = sum_score[n] + (float)((double)(L[n] - Ld_tot[n]) * log((double)((float)L[n] / (float)(L[n] + 3))));
            const float pre2 = (float)(((double)ss - (double)nullsc[n]) / LOG2);
            ss = (float)(((double)ss - (double)(nullsc[n] + sb2f)) / LOG2);
            if (Ld_tot[n] > 0 && ss > seq_score) { seq_score = ss; pre_score = pre2; flags[n] |= WH_FLAG_OVERRIDE; }
            decibits[n] = (int)rint((double)seq_score * 10.0);
            flags[n] |= WH_FLAG_REPORTED;
            if (dp[n]) { dp[n]->seq_score = seq_score; dp[n]->pre_score = pre_score; dp[n]->seqbias_nats = seqbias; }
          }
        }
        if (lane == 0) {
          a.decibits[outp[n]] = decibits[n];
          a.flags[outp[n]] = (uint8_t)flags[n];
          if (a.fwd_bits) a.fwd_bits[outp[n]] = fwd_bits_out[n];
        }
      }
    }
  }
  if (a.paths && lane == 0) {
    if (n_w256) atomicAdd(a.paths + 0, (unsigned long long)n_w256);
    if (n_w512) atomicAdd(a.paths + 1, (unsigned long long)n_w512);
    if (n_wfail) atomicAdd(a.paths + 2, (unsigned long long)n_wfail);
    if (n_full) atomicAdd(a.paths + 3, (unsigned long long)n_full);
  }
}

template <int Q>
static hipError_t launch9q(const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_kernel9<Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((score_kernel9<Q>), dim3(blocks), dim3(threads), lds, s, a);
  return hipGetLastError();
}

}  // namespace k9

// a.wave_lds: floats of LDS per WAVE = two query blocks of score9_block_floats(); a.scratch_stride: floats per wave = two Forward slabs
int score9_block_floats(int SP, int Lcap) { return k9::kSpArr * SP + 32 + kRegsInts + 8 + (Lcap + 3) / 4 + 4; }

hipError_t launch_score9(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  if (threads > 512 || a.spec_scratch || a.spec_arrays != k9::kSpArr) return hipErrorInvalidValue;
  switch (Q) {
    case 8:  return k9::launch9q<8>(a, blocks, threads, lds, s);
    case 12: return k9::launch9q<12>(a, blocks, threads, lds, s);
    case 16: return k9::launch9q<16>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace wh
