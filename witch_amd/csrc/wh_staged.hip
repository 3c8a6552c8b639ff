// Staged scoring launches (round 5): the five sweeps of hmmsearch --max per (query, HMM) pair - P1 multihit Forward,
// P2 multihit Backward + domain decoding + region scan, per envelope P3 unihit Forward and P4 unihit Backward + null2,
// score assembly - as KERNELS OF THEIR OWN over batches of pairs, instead of one fused kernel (wh_score7.hip).
// (hmmsearch per pair: witch_msa/gcmm/algorithm.py:526-532; algorithm SURVEY.md A.2-A.6.)
//
// Why: the dense sweeps (P1, P3: 16 cells per lane at full width, 168 registers) fill a SIMD at three waves; the window
// sweeps (P2 and P4 on 256 nodes: 4 cells per lane, ~70 registers) are one dependent chain per row and three waves do
// not cover it - in the fused kernel they are a fifth of the instructions and more than a third of the time.  Here every
// kind of sweep runs at the occupancy it can use:
//   p1      dense   12 waves/CU   tables: forward orientation + emission rows     -> per pair: six per-row arrays + C(L) in HBM
//   p2win   light   24 waves/CU   tables: emission rows (reversed arrays gathered from L2 once per pair)
//                                 window sweep + certified region scan             -> regions, or the pair is marked "in doubt"
//   p2full  dense   12 waves/CU   tables: reversed orientation + emission rows; the pairs in doubt at full width
//   p3      dense   12 waves/CU   one unit per envelope: Forward rows to the unit's slab, per-row arrays to HBM
//   p4win   light   20-24 waves   envelopes of the 256-node (512-node) class; a window that fails the mass certificate
//                                 moves its envelope to the full-width class
//   p4full  dense   12 waves/CU   the full-width class; a sweep that fails the spill certificate moves its envelope on
//   dense   dense                 P3 with every row stored + P4 at full width (rare)
//   assemble        one thread per pair: HMMER's float32 score assembly, or the pair's record for the multidomain resolver
// Every sweep IS the fused kernel's device function (this file compiles wh_score7.hip's sweeps into its own namespace;
// non-inlined, so the arithmetic of a sweep does not depend on the kernel around it): a pair takes the same path through
// the same instructions as in the fused kernel and the results are bit-identical (tests/test_gpu_parity.py:
// test_staged_launches_equal_the_fused_kernel).
#include <hip/hip_runtime.h>

#ifndef WH_ST_PART
#define WH_ST_PART 0          // 0: everything; 1: dense kernels only; 2: light kernels + assembly only (parallel builds)
#endif
#if WH_ST_PART == 2
#define WH_K7NS ksl
#else
#define WH_K7NS ksd
#endif
#define WH_K7LAUNCH launch_score_staged_unused
#define WH_SWEEPS_ONLY 1
#ifndef WH_SLIM_SPEC
#define WH_SLIM_SPEC 1
#endif
#include "wh_score7.hip"

namespace wh {
namespace WH_K7NS {

// WH_STATS: per kernel kind k, stats[4k] shader cycles inside the sweeps, [4k+1] the same in 100 MHz real-time ticks,
// [4k+2] wave lifetimes (shader cycles), [4k+3] sweeps.  Kinds: 0 p1, 1 p2win 256, 2 p2win 512, 3 p2full, 4 p3, 5 p4win 256,
// 6 p4win 512, 7 p4full, 8 dense
#define ST_T0() const long long st_c0 = a.stats ? (long long)__builtin_readcyclecounter() : 0; const long long st_r0 = a.stats ? (long long)wall_clock64() : 0
#define ST_T1(kind) do { if (a.stats) { const long long st_c1 = (long long)__builtin_readcyclecounter(), st_r1 = (long long)wall_clock64(); if (lane == 0) { \
    atomicAdd(a.stats + 4 * (kind), (unsigned long long)(st_c1 - st_c0)); atomicAdd(a.stats + 4 * (kind) + 1, (unsigned long long)(st_r1 - st_r0)); atomicAdd(a.stats + 4 * (kind) + 3, 1ull); } } } while (0)
#define ST_K0() const long long st_k0 = a.stats ? (long long)__builtin_readcyclecounter() : 0
#define ST_K1(kind) do { if (a.stats && lane == 0) atomicAdd(a.stats + 4 * (kind) + 2, (unsigned long long)((long long)__builtin_readcyclecounter() - st_k0)); } while (0)
[[maybe_unused]] constexpr int kStTH = 768;            // dense kernels: twelve waves, 168 registers
[[maybe_unused]] constexpr int kLightTH = 768;         // light kernels: two workgroups of up to twelve waves per CU

// pair of a batch -> query and model
struct PairPos { int h; int64_t qi; };
__device__ __forceinline__ PairPos pair_pos(const StagedArgs &g, int pl) {
  const ScoreArgs &a = g.a;
  const int gitem = g.item0 + pl / a.QB;
  PairPos p;
  p.h = a.hmm_list[gitem / a.n_qblocks];
  p.qi = (int64_t)(gitem % a.n_qblocks) * a.QB + pl % a.QB;
  return p;
}

// what a workgroup keeps in LDS: [16-byte header][emission rows K x TBL][NT transition arrays x TBL][per-wave blocks][candidates]
template <int Q>
struct StLds {
  float *em, *tr, *wbase;
  volatile int *slot;                 // [0] the group drawn, [1] candidates of the segment, [2] next candidate to hand out
  int *cand;
  __device__ __forceinline__ StLds(float *raw, int K, int ntr, int wave, int nwaves, int wave_lds) {
    slot = reinterpret_cast<volatile int *>(raw);
    em = raw + 4;
    tr = em + (size_t)K * Q * kWave;
    float *w0 = tr + (size_t)ntr * Q * kWave;
    wbase = w0 + (size_t)wave * wave_lds;
    cand = reinterpret_cast<int *>(w0 + (size_t)nwaves * wave_lds);
  }
};

__device__ __forceinline__ void copy_f4(float *dst, const float *src, int nfloats) {
  const float4 *s4 = reinterpret_cast<const float4 *>(src);
  float4 *d4 = reinterpret_cast<float4 *>(dst);
  for (int t = threadIdx.x; t < nfloats / 4; t += blockDim.x) d4[t] = s4[t];
}

// per-wave context with the block laid out as the fused kernel's: [kSpArr x SP rows][n2tab 32][regs][residues]
__device__ __forceinline__ WaveCtx make_ctx(const ScoreArgs &a, float *em, float *fw, float *bw, float *wbase, int lane) {
  WaveCtx c;
  c.emL = (lds_f *)em; c.fwL = (lds_f *)fw; c.bwL = (lds_f *)bw;
  c.spec = (lds_f *)wbase; c.n2tab = (lds_f *)(wbase + kSpArr * a.SP);
  c.specg = nullptr; c.Fs = nullptr; c.emG = nullptr;
  c.degen = 0;
  for (int t = 0; t < 32; t++) if (t == lane) c.degen = a.degen[t];
  c.SP = a.SP; c.alpha = a.K | (a.Kp << 8) | (a.K << 16); c.lane = lane;
  return c;
}
__device__ __forceinline__ int *ctx_regs(const ScoreArgs &a, float *wbase) { return reinterpret_cast<int *>(wbase + kSpArr * a.SP + 32); }
__device__ __forceinline__ uint8_t *ctx_seq(const ScoreArgs &a, float *wbase) { return reinterpret_cast<uint8_t *>(ctx_regs(a, wbase) + kRegsInts); }

__device__ __forceinline__ void load_seq(const ScoreArgs &a, uint8_t *seq, int64_t off, int L, int lane) {
  for (int t = lane; t < L; t += kWave) {
    int r = a.residues[off + t];
    seq[t] = (uint8_t)(r < a.Kp ? r : a.Kp - 1);
  }
  __builtin_amdgcn_wave_barrier();
}

// the six per-row arrays of a wave's block <-> HBM (rows 0..L of each; coalesced)
__device__ __forceinline__ void rows_out(float *dst, const float *spec, int SP, int L, int lane) {
  __builtin_amdgcn_wave_barrier();
  for (int arr = 0; arr < kSpArr; arr++)
    for (int t = lane; t <= L; t += kWave) __builtin_nontemporal_store(spec[arr * SP + t], dst + arr * SP + t);
}
__device__ __forceinline__ void rows_in(float *spec, const float *src, int SP, int L, int lane) {
  __builtin_amdgcn_wave_barrier();
  for (int arr = 0; arr < kSpArr; arr++)
    for (int t = lane; t <= L; t += kWave) spec[arr * SP + t] = __builtin_nontemporal_load(src + arr * SP + t);
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int bcast_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// the window of 64*QB nodes around the lane blocks of <um> (two blocks in front, one behind), as the fused kernel places
// it: returns QB (4 or 8) and the first reversed node, or 0 when no window fits
template <int Q>
__device__ __forceinline__ int place_window(unsigned long long um, int &m0) {
  if (um == 0) return 0;
  int lo = __builtin_ctzll(um), hi = 63 - __builtin_clzll(um);
  lo = lo > 1 ? lo - 2 : 0; hi = hi < 63 ? hi + 1 : 63;
  const int nodes = (hi - lo + 1) * Q;
  if (nodes <= 4 * kWave) { m0 = min((63 - hi) * Q, kWave * (Q - 4)); return 4; }
  if (Q % 8 == 0 && Q > 8 && nodes <= 8 * kWave) { m0 = min((63 - hi) * Q, kWave * (Q - 8)); return 8; }
  return 0;
}

// ------------------------------------------------------------------------------------------------ the work loop of every kernel
// A workgroup draws GROUPS of g.G consecutive work items of the batch (an item = a.QB queries of one model, model-major), cuts a
// group at model boundaries into SEGMENTS, stages the segment's tables when the model changed, and collects the segment's
// CANDIDATES - the pairs (or envelopes) this kernel has something to do for, found by <pred> from the per-pair records - in
// an LDS list; the waves then take candidates one by one from an LDS counter.  So a kernel that serves 6 % of the envelopes
// keeps all its waves on those, and queries of different lengths do not leave waves idle behind a fixed deal.
// <pred(pl, emit)>: called by one thread per pair, emit(code) adds a candidate; <body(code, h)>: one wave per candidate.
// A segment whose candidates exceed the list is handled in pieces of cand_cap / per_pair pairs (per_pair = the most one pair can emit).
template <class Stage, class Pred, class Body>
__device__ __forceinline__ void group_loop(const StagedArgs &g, int head_slot, volatile int *slot, int *cand, int per_pair, Stage stage, Pred pred, Body body) {
  const ScoreArgs &a = g.a;
  const int lane = threadIdx.x & 63;
  const int G = g.G;
  const int n_groups = (g.n_items_b + G - 1) / G;
  int cur_h = -1;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) slot[0] = atomicAdd(g.cnt + head_slot, 1);
    __syncthreads();
    const int gi = slot[0];
    if (gi >= n_groups) break;
    int it = gi * G;
    const int it_end = min(it + G, g.n_items_b);
    while (it < it_end) {
      const int m = (g.item0 + it) / a.n_qblocks;
      const int h = a.hmm_list[m];
      const int it2 = min(it_end, (m + 1) * a.n_qblocks - g.item0);      // first item of the next model
      if (h != cur_h) { __syncthreads(); stage(h); cur_h = h; __syncthreads(); }
      const int seg0 = it * a.QB, seg1 = it2 * a.QB;
      const int piece = (seg1 - seg0) * per_pair <= g.cand_cap ? seg1 - seg0 : max(1, g.cand_cap / per_pair);
      for (int pl0 = seg0; pl0 < seg1; pl0 += piece) {
        const int pl1 = min(pl0 + piece, seg1);
        __syncthreads();
        if (threadIdx.x == 0) { slot[1] = 0; slot[2] = 0; }
        __syncthreads();
        for (int pl = pl0 + threadIdx.x; pl < pl1; pl += blockDim.x)
          pred(pl, [&](int code) { const int k = atomicAdd(const_cast<int *>(slot) + 1, 1); if (k < g.cand_cap) cand[k] = code; });
        __syncthreads();
        const int n = min(slot[1], g.cand_cap);
        for (;;) {
          int k = 0;
          if (lane == 0) k = atomicAdd(const_cast<int *>(slot) + 2, 1);
          k = bcast_i(k);
          if (k >= n) break;
          body(cand[k], h);
        }
      }
      it = it2;
    }
  }
}

// is pair <pl> of the batch a real pair (the last query block of a model is ragged)?
__device__ __forceinline__ bool pair_valid(const StagedArgs &g, int pl) {
  const ScoreArgs &a = g.a;
  const int gitem = g.item0 + pl / a.QB;
  return (int64_t)(gitem % a.n_qblocks) * a.QB + pl % a.QB < a.nq;
}

__device__ __forceinline__ void store_regions(StPair *pp, const RegOut &ro, const int *regs, int lane, wh_pair_detail *dp) {
  if (lane == 0) {
    pp->nenv = ro.nenv; pp->nreg = ro.nreg; pp->flags = ro.flags & 0xFFFFFF;
    for (int e = 0; e < ro.nenv; e++) { pp->regs[2 * e] = regs[2 * e]; pp->regs[2 * e + 1] = regs[2 * e + 1]; pp->cls[e] = ST_CLS_NONE; }
    pp->state = 2;
    if (dp) { dp->nregions = ro.nreg; dp->nenv = ro.nenv; }
  }
}

#if WH_ST_PART != 2
// ================================================================================================ p1: multihit Forward
template <int Q>
__global__ __launch_bounds__(kStTH) void staged_p1_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  StLds<Q> S(smem_raw, a.K, FW_NARR, wave, nwaves, a.wave_lds);
  WaveCtx c = make_ctx(a, S.em, S.tr, nullptr, S.wbase, lane);
  uint8_t *seq = ctx_seq(a, S.wbase);
  float *spec = S.wbase;
  const int SP = a.SP;
  const double LOG2 = 0.69314718055994529;
  ST_K0();
  group_loop(g, ST_C_P1, S.slot, S.cand, 1,
    [&](int h) {
      const DevHMM *hm = a.hmms + h;
      copy_f4(S.em, a.tables + hm->em_off, a.K * TBL);
      copy_f4(S.tr, a.tables + hm->fw_off, FW_NARR * TBL);
      c.emG = (const glb_f *)(a.tables + hm->em_off);
    },
    [&](int pl, auto emit) { if (pair_valid(g, pl)) emit(pl); },
    [&](int pl, int h) {
      StPair *pp = g.pairs + pl;
      const PairPos P = pair_pos(g, pl);
      const int64_t qi = P.qi;
      const int64_t off = a.offsets[qi];
      const int L = (int)(a.offsets[qi + 1] - off);
      const size_t out = (size_t)qi * a.H + h;
      float fwd_bits_out = -INFINITY;
      wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + out : nullptr;
      if (dp) {
        dp->fwd_bits = -INFINITY; dp->seq_score = 0.f; dp->pre_score = 0.f; dp->seqbias_nats = 0.f;
        dp->nregions = 0; dp->nenv = 0;
      }
      int state = 0;
      if (L > 0 && L <= a.Lcap) {
        load_seq(a, seq, off, L, lane);
        const LenCfg cm = len_config(L, true);
        ST_T0();
        const FwdOut f1 = sweep_forward<Q, false, kStTH, false>(c, (lds_u8 *)seq, L, cm, 0.f);
        ST_T1(0);
        const double fwd_nats = (double)f1.ef * LOG2 + log((double)(f1.xC * cm.move));
        const float fwdsc = (float)fwd_nats;
        const float p1 = (float)L / (float)(L + 1);
        const float nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
        fwd_bits_out = (float)((fwd_nats - (double)nullsc) / LOG2);
        if (dp) dp->fwd_bits = fwd_bits_out;
        if (f1.xC > 0.f && isfinite(fwdsc)) {
          state = 1;
          rows_out(g.p1spec + (size_t)pl * g.p1stride, spec, SP, L, lane);
          if (lane == 0) {
            const unsigned *su = reinterpret_cast<const unsigned *>((const float *)c.n2tab) + kUmSlot;
            pp->xC = f1.xC; pp->ef = f1.ef;
            pp->um_lo = Q >= 8 ? su[0] : 0u; pp->um_hi = Q >= 8 ? su[1] : 0u; pp->um_steady = Q >= 8 ? su[-1] : 0u;
          }
        }
      }
      if (lane == 0) {
        pp->state = state; pp->path = 0; pp->nenv = 0; pp->nreg = 0; pp->flags = 0;
        if (state == 0) { a.decibits[out] = 0; a.flags[out] = 0; if (g.pair_paths) g.pair_paths[out] = 0; }
        if (a.fwd_bits) a.fwd_bits[out] = fwd_bits_out;
      }
    });
  ST_K1(0);
}

// ================================================================================================ p2full: pairs whose window left a doubt
template <int Q>
__global__ __launch_bounds__(kStTH) void staged_p2full_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  StLds<Q> S(smem_raw, a.K, BW_NARR, wave, nwaves, a.wave_lds);
  WaveCtx c = make_ctx(a, S.em, nullptr, S.tr, S.wbase, lane);
  uint8_t *seq = ctx_seq(a, S.wbase);
  int *regs = ctx_regs(a, S.wbase);
  ST_K0();
  group_loop(g, ST_C_P2B, S.slot, S.cand, 1,
    [&](int h) {
      const DevHMM *hm = a.hmms + h;
      copy_f4(S.em, a.tables + hm->em_off, a.K * TBL);
      copy_f4(S.tr, a.tables + hm->bw_off, BW_NARR * TBL);
      c.emG = (const glb_f *)(a.tables + hm->em_off);
    },
    [&](int pl, auto emit) { if (pair_valid(g, pl) && g.pairs[pl].state == 3) emit(pl); },
    [&](int pl, int h) {
      StPair *pp = g.pairs + pl;
      const PairPos P = pair_pos(g, pl);
      const int64_t off = a.offsets[P.qi];
      const int L = (int)(a.offsets[P.qi + 1] - off);
      load_seq(a, seq, off, L, lane);
      rows_in(S.wbase, g.p1spec + (size_t)pl * g.p1stride, a.SP, L, lane);
      const LenCfg cm = len_config(L, true);
      const float xC = pp->xC; const int ef = pp->ef;
      ST_T0();
      sweep_backward_decode<Q, kStTH, false>(c, (lds_u8 *)seq, L, cm, 1.0f / (xC * cm.move), ef);
      ST_T1(3);
      const RegOut ro = region_scan<kStTH, false>(c.spec, nullptr, a.SP, L, (lds_i *)regs, lane);
      __builtin_amdgcn_wave_barrier();
      wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + ((size_t)P.qi * a.H + h) : nullptr;
      store_regions(pp, ro, regs, lane, dp);
      if (lane == 0) pp->path |= WH_PATH_P2_FULL;
    });
  ST_K1(3);
}

// ================================================================================================ p3: unihit Forward per envelope
template <int Q>
__global__ __launch_bounds__(kStTH) void staged_p3_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  StLds<Q> S(smem_raw, a.K, FW_NARR, wave, nwaves, a.wave_lds);
  WaveCtx c = make_ctx(a, S.em, S.tr, nullptr, S.wbase, lane);
  uint8_t *seq = ctx_seq(a, S.wbase);
  float *spec = S.wbase;
  const int SP = a.SP;
  const double LOG2 = 0.69314718055994529;
  unsigned long long spill = 0;
  ST_K0();
  group_loop(g, ST_C_P3, S.slot, S.cand, WH_MAX_ENVELOPES,
    [&](int h) {
      const DevHMM *hm = a.hmms + h;
      copy_f4(S.em, a.tables + hm->em_off, a.K * TBL);
      copy_f4(S.tr, a.tables + hm->fw_off, FW_NARR * TBL);
      c.emG = (const glb_f *)(a.tables + hm->em_off);
    },
    [&](int pl, auto emit) {
      if (!pair_valid(g, pl)) return;
      StPair *pp = g.pairs + pl;
      if (pp->state != 2) return;
      const int nenv = pp->nenv, multi_mask = pp->flags >> 8;
      const bool queue_pair = multi_mask != 0 && a.rrecs != nullptr;
      for (int e = 0; e < nenv; e++) {
        // a pair with a multidomain region is finished by resolve_kernel; its single-domain regions are still scored here
        if (queue_pair && ((multi_mask >> e) & 1)) { pp->envsc[e] = 0.f; pp->domcorr[e] = 0.f; continue; }
        emit(pl * WH_MAX_ENVELOPES + e);
      }
    },
    [&](int code, int h) {
      const int pl = code / WH_MAX_ENVELOPES, e = code % WH_MAX_ENVELOPES;
      StPair *pp = g.pairs + pl;
      const PairPos P = pair_pos(g, pl);
      const int64_t off = a.offsets[P.qi];
      const int L = (int)(a.offsets[P.qi + 1] - off);
      const int ri = bcast_i(pp->regs[2 * e]), rj = bcast_i(pp->regs[2 * e + 1]);
      const int Ld = rj - ri + 1;
      int uid = 0;
      if (lane == 0) uid = atomicAdd(g.cnt + ST_N_UNITS, 1);
      uid = bcast_i(uid);
      if (uid >= g.NS) { if (lane == 0) g.cnt[ST_OVERFLOW] = 1; return; }      // the host repeats the call (wh_api.hip)
      load_seq(a, seq, off + (ri - 1), Ld, lane);
      const LenCfg cu = len_config(L, false);
      c.Fs = (glb_f *)(g.slabs + (size_t)uid * g.slab_stride);
      const float keep_scale = a.keep_scale > 0.f ? a.keep_scale : kKeepScale7;
      ST_T0();
      const P1Mask um1 = {((unsigned long long)(unsigned)bcast_i((int)pp->um_hi) << 32) | (unsigned)bcast_i((int)pp->um_lo), (unsigned)bcast_i((int)pp->um_steady)};
      const int band = spill_band<Q>(a, um1);
      const FwdOut f3 = sweep_forward<Q, true, kStTH, false>(c, (lds_u8 *)seq, Ld, cu, keep_scale, band);
      ST_T1(4);
      spill += (unsigned long long)f3.nst * (8 * Q);
      const float envsc = (float)((double)f3.ef * LOG2 + log((double)(f3.xC * cu.move)));
      int cls = ST_CLS_NONE, m0 = 0;
      if (f3.xC > 0.f) {
        rows_out(g.p3spec + (size_t)uid * g.p3stride, spec, SP, Ld, lane);
        cls = ST_CLS_FULL;
        if (Q >= 8 && !a.no_window) {
          const unsigned *su = reinterpret_cast<const unsigned *>(spec);
          const unsigned long long um = mask_in_band(((unsigned long long)su[kSpMH * SP] << 32) | su[kSpML * SP], band);
          const int w = place_window<Q>(um, m0);
          if (w == 4) cls = ST_CLS_W256; else if (w == 8) cls = ST_CLS_W512;
        }
      }
      if (lane == 0) {
        StUnit *u = g.units + uid;
        u->xC = f3.xC; u->ef = f3.ef; u->m0 = m0;
        pp->envsc[e] = envsc; pp->domcorr[e] = 0.f; pp->uid[e] = uid; pp->cls[e] = (uint8_t)cls;
      }
    });
  if (a.paths && lane == 0 && spill) atomicAdd(a.paths + 6, spill);
  ST_K1(4);
  // the row stores of this kernel are read by other workgroups in the next launch: kernel boundary = release
}

// ================================================================================================ p4full: envelopes at full width
template <int Q>
__global__ __launch_bounds__(kStTH) void staged_p4full_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  StLds<Q> S(smem_raw, a.K, BW_NARR, wave, nwaves, a.wave_lds);
  WaveCtx c = make_ctx(a, S.em, nullptr, S.tr, S.wbase, lane);
  uint8_t *seq = ctx_seq(a, S.wbase);
  unsigned n_full = 0;
  ST_K0();
  group_loop(g, ST_C_FULL, S.slot, S.cand, WH_MAX_ENVELOPES,
    [&](int h) {
      const DevHMM *hm = a.hmms + h;
      copy_f4(S.em, a.tables + hm->em_off, a.K * TBL);
      copy_f4(S.tr, a.tables + hm->bw_off, BW_NARR * TBL);
      c.emG = (const glb_f *)(a.tables + hm->em_off);
    },
    [&](int pl, auto emit) {
      if (!pair_valid(g, pl)) return;
      const StPair *pp = g.pairs + pl;
      if (pp->state != 2) return;
      for (int e = 0; e < pp->nenv; e++) if (pp->cls[e] == ST_CLS_FULL) emit(pl * WH_MAX_ENVELOPES + e);
    },
    [&](int code, int h) {
      const int pl = code / WH_MAX_ENVELOPES, e = code % WH_MAX_ENVELOPES;
      StPair *pp = g.pairs + pl;
      const int uid = bcast_i(pp->uid[e]), ri = bcast_i(pp->regs[2 * e]), rj = bcast_i(pp->regs[2 * e + 1]);
      const int Ld = rj - ri + 1;
      const StUnit *u = g.units + uid;
      const PairPos P = pair_pos(g, pl);
      const int64_t off = a.offsets[P.qi];
      const int L = (int)(a.offsets[P.qi + 1] - off);
      load_seq(a, seq, off + (ri - 1), Ld, lane);
      rows_in(S.wbase, g.p3spec + (size_t)uid * g.p3stride, a.SP, Ld, lane);
      c.Fs = (glb_f *)(g.slabs + (size_t)uid * g.slab_stride);
      const LenCfg cu = len_config(L, false);
      const float xC = u->xC; const int ef = u->ef;
      ST_T0();
      const P1Mask um1 = {((unsigned long long)(unsigned)bcast_i((int)pp->um_hi) << 32) | (unsigned)bcast_i((int)pp->um_lo), (unsigned)bcast_i((int)pp->um_steady)};
      const float tol = spill_tol(spill_band<Q>(a, um1) != kAllLanes);
      const P4Out p4 = sweep_backward_null2<Q, kStTH, false>(c, (lds_u8 *)seq, Ld, cu, 1.0f / (xC * cu.move), ef, tol);
      ST_T1(7);
      n_full++;
      const bool ok = fabsf((float)Ld - p4.mass) <= tol * (float)Ld;
      if (lane == 0) {
        atomicOr(&pp->path, WH_PATH_P4_FULL);
        if (ok) { pp->domcorr[e] = p4.domcorr; pp->cls[e] = ST_CLS_NONE; }
        else pp->cls[e] = ST_CLS_DENSE;
      }
    });
  if (a.paths && lane == 0 && n_full) atomicAdd(a.paths + 3, (unsigned long long)n_full);
  ST_K1(7);
}

// ================================================================================================ dense: every row stored, full width
template <int Q>
__global__ __launch_bounds__(kStTH) void staged_dense_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  StLds<Q> S(smem_raw, a.K, 2 * FW_NARR, wave, nwaves, a.wave_lds);
  WaveCtx c = make_ctx(a, S.em, S.tr, S.tr + FW_NARR * TBL, S.wbase, lane);
  uint8_t *seq = ctx_seq(a, S.wbase);
  EnvCounters ec = {0, 0, 0, 0, 0};
  ST_K0();
  group_loop(g, ST_C_DENSE, S.slot, S.cand, WH_MAX_ENVELOPES,
    [&](int h) {
      const DevHMM *hm = a.hmms + h;
      copy_f4(S.em, a.tables + hm->em_off, a.K * TBL);
      copy_f4(S.tr, a.tables + hm->fw_off, FW_NARR * TBL);
      copy_f4(S.tr + FW_NARR * TBL, a.tables + hm->bw_off, BW_NARR * TBL);
      c.emG = (const glb_f *)(a.tables + hm->em_off);
    },
    [&](int pl, auto emit) {
      if (!pair_valid(g, pl)) return;
      const StPair *pp = g.pairs + pl;
      if (pp->state != 2) return;
      for (int e = 0; e < pp->nenv; e++) if (pp->cls[e] == ST_CLS_DENSE) emit(pl * WH_MAX_ENVELOPES + e);
    },
    [&](int code, int h) {
      const int pl = code / WH_MAX_ENVELOPES, e = code % WH_MAX_ENVELOPES;
      StPair *pp = g.pairs + pl;
      const int uid = bcast_i(pp->uid[e]), ri = bcast_i(pp->regs[2 * e]), rj = bcast_i(pp->regs[2 * e + 1]);
      const int Ld = rj - ri + 1;
      const PairPos P = pair_pos(g, pl);
      const int64_t off = a.offsets[P.qi];
      const int L = (int)(a.offsets[P.qi + 1] - off);
      load_seq(a, seq, off + (ri - 1), Ld, lane);
      c.Fs = (glb_f *)(g.slabs + (size_t)uid * g.slab_stride);
      const LenCfg cu = len_config(L, false);
      // the band of lane blocks failed the certificate: the unbanded store next, then the dense one (envelope_attempts, wh_score7.hip)
      const P1Mask um1 = {((unsigned long long)(unsigned)bcast_i((int)pp->um_hi) << 32) | (unsigned)bcast_i((int)pp->um_lo), (unsigned)bcast_i((int)pp->um_steady)};
      const int band = spill_band<Q>(a, um1);
      int flags = 0;
      long long t_last = 0;
      float envsc;
      const float domcorr = envelope_attempts<Q, kStTH, false>(a, c, seq, Ld, cu, band, band != kAllLanes ? 1 : 2, ec, lane, flags, t_last, envsc);
      if (lane == 0) {
        pp->envsc[e] = envsc; pp->domcorr[e] = domcorr; pp->cls[e] = ST_CLS_NONE;
        if (flags & WH_FLAG_EXACT) { atomicOr(&pp->flags, WH_FLAG_EXACT); atomicOr(&pp->path, WH_PATH_DENSE); }
      }
    });
  if (a.paths && lane == 0) {
    if (ec.n_w256) atomicAdd(a.paths + 0, (unsigned long long)ec.n_w256);
    if (ec.n_w512) atomicAdd(a.paths + 1, (unsigned long long)ec.n_w512);
    if (ec.n_wfail) atomicAdd(a.paths + 2, (unsigned long long)ec.n_wfail);
    if (ec.n_full) atomicAdd(a.paths + 3, (unsigned long long)ec.n_full);
    if (ec.spill) atomicAdd(a.paths + 6, ec.spill);
  }
  ST_K1(8);
}

// ================================================================================================ env: P3 + P4 + assembly, fused per pair
// The envelope half of the fused kernel (score_envelopes, wh_score7.hip) for pairs whose regions are known: one wavefront
// per pair, its Forward slab reused from pair to pair as in the fused kernel - the spill traffic of P3 then overlaps with
// the other waves' arithmetic instead of being a launch of its own (the full split above is HBM-bound in P3 and P4).
template <int Q>
__global__ __launch_bounds__(kStTH) void staged_env_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  StLds<Q> S(smem_raw, a.K, 2 * FW_NARR, wave, nwaves, a.wave_lds);
  WaveCtx c = make_ctx(a, S.em, S.tr, S.tr + FW_NARR * TBL, S.wbase, lane);
  c.Fs = (glb_f *)(a.scratch + ((size_t)blockIdx.x * nwaves + wave) * a.scratch_stride);
  uint8_t *seq = ctx_seq(a, S.wbase);
  int *regs = ctx_regs(a, S.wbase);
  const double LOG2 = 0.69314718055994529;
  EnvCounters ec = {0, 0, 0, 0, 0};
  ST_K0();
  group_loop(g, ST_C_P3, S.slot, S.cand, 1,
    [&](int h) {
      const DevHMM *hm = a.hmms + h;
      copy_f4(S.em, a.tables + hm->em_off, a.K * TBL);
      copy_f4(S.tr, a.tables + hm->fw_off, FW_NARR * TBL);
      copy_f4(S.tr + FW_NARR * TBL, a.tables + hm->bw_off, BW_NARR * TBL);
      c.emG = (const glb_f *)(a.tables + hm->em_off);
    },
    [&](int pl, auto emit) { if (pair_valid(g, pl) && g.pairs[pl].state == 2) emit(pl); },
    [&](int pl, int h) {
      StPair *pp = g.pairs + pl;
      const PairPos P = pair_pos(g, pl);
      const int64_t qi = P.qi;
      const int64_t off = a.offsets[qi];
      const int L = (int)(a.offsets[qi + 1] - off);
      const size_t out = (size_t)qi * a.H + h;
      const int nenv = bcast_i(pp->nenv), nreg = bcast_i(pp->nreg), pflags = bcast_i(pp->flags);
      int flags = pflags & 0xFF, decibits = 0;
      const int multi_mask = pflags >> 8;
      const LenCfg cm = len_config(L, true);
      const float xC = pp->xC; const int ef = pp->ef;
      const double fwd_nats = (double)ef * LOG2 + log((double)(xC * cm.move));
      const float fwdsc = (float)fwd_nats;
      const float p1 = (float)L / (float)(L + 1);
      const float nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
      const float fwd_bits_out = (float)((fwd_nats - (double)nullsc) / LOG2);
      wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + out : nullptr;
      int path = bcast_i(pp->path);
      if (nenv > 0) {
        load_seq(a, seq, off, L, lane);
        if (lane < 2 * nenv) regs[lane] = pp->regs[lane];
        __builtin_amdgcn_wave_barrier();
        long long t_last = 0;
        ST_T0();
        const P1Mask um1 = {((unsigned long long)(unsigned)bcast_i((int)pp->um_hi) << 32) | (unsigned)bcast_i((int)pp->um_lo), (unsigned)bcast_i((int)pp->um_steady)};
        score_envelopes<Q, kStTH, false>(a, c, seq, regs, L, lane, h, qi, nenv, nreg, multi_mask, fwdsc, nullsc, fwd_bits_out, dp, flags, decibits, ec, t_last, um1);
        ST_T1(4);
        if (multi_mask != 0 && a.rrecs != nullptr) path |= WH_PATH_MULTI;
      }
      if (lane == 0) {
        if ((path & 256) && a.paths) atomicAdd(a.paths + 5, 1ull);      // wanted a window for P2, none fitted (counted as the fused kernel counts it)
        a.decibits[out] = decibits;
        a.flags[out] = (uint8_t)flags;
        if (g.pair_paths) g.pair_paths[out] = (uint8_t)path;
      }
    });
  if (a.paths && lane == 0) {
    if (ec.n_w256) atomicAdd(a.paths + 0, (unsigned long long)ec.n_w256);
    if (ec.n_w512) atomicAdd(a.paths + 1, (unsigned long long)ec.n_w512);
    if (ec.n_wfail) atomicAdd(a.paths + 2, (unsigned long long)ec.n_wfail);
    if (ec.n_full) atomicAdd(a.paths + 3, (unsigned long long)ec.n_full);
    if (ec.spill) atomicAdd(a.paths + 6, ec.spill);
  }
  ST_K1(4);
}

template <class K>
static hipError_t launch_kind(K kern, const StagedArgs &g, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, s, g);
  return hipGetLastError();
}
#endif   // dense part

#if WH_ST_PART != 1
// ================================================================================================ p2win: window sweep + certified scan
// Light kernel: the workgroup stages the emission rows only; a wave gathers its window's reversed transition arrays from
// L2 once per pair (BWG) and works on a COPY of P1's rows (INPL).  The 256-node launch sees every pair after P1 (state 1)
// and passes on what it cannot serve: to the 512-node launch (state 4) or to the full-width launch (state 3).
template <int Q, int QB>
__global__ __launch_bounds__(kLightTH) __attribute__((amdgpu_waves_per_eu(QB == 4 ? 6 : 4, QB == 4 ? 6 : 4))) void staged_p2win_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  StLds<Q> S(smem_raw, a.K, 0, wave, nwaves, a.wave_lds);
  WaveCtx c = make_ctx(a, S.em, nullptr, nullptr, S.wbase, lane);
  uint8_t *seq = ctx_seq(a, S.wbase);
  int *regs = ctx_regs(a, S.wbase);
  unsigned n_p2w = 0, n_p2rej = 0;
  ST_K0();
  group_loop(g, QB == 8 ? ST_C_P2W8 : ST_C_P2, S.slot, S.cand, 1,
    [&](int h) {
      const DevHMM *hm = a.hmms + h;
      copy_f4(S.em, a.tables + hm->em_off, a.K * TBL);
      c.emG = (const glb_f *)(a.tables + hm->em_off);
      c.specg = (glb_f *)const_cast<float *>(a.tables + hm->bw_off);
    },
    [&](int pl, auto emit) {
      if (!pair_valid(g, pl)) return;
      StPair *pp = g.pairs + pl;
      if (pp->state != (QB == 4 ? 1 : 4)) return;
      if (QB == 4) {
        // pairs no window serves go on without a wave's time: 512 nodes -> state 4, none -> state 3
        const unsigned long long um = ((unsigned long long)pp->um_hi << 32) | pp->um_lo;
        int m0 = 0;
        const int w = (Q >= 8 && !a.no_window) ? place_window<Q>(um, m0) : 0;
        if (w == 8) { pp->state = 4; return; }
        if (w == 0) { pp->state = 3; if (um != 0) pp->path |= 256; return; }     // (bit 8: a window was wanted - counted below)
      }
      emit(pl);
    },
    [&](int pl, int h) {
      StPair *pp = g.pairs + pl;
      const PairPos P = pair_pos(g, pl);
      const int64_t off = a.offsets[P.qi];
      const int L = (int)(a.offsets[P.qi + 1] - off);
      const unsigned long long um = ((unsigned long long)(unsigned)bcast_i((int)pp->um_hi) << 32) | (unsigned)bcast_i((int)pp->um_lo);
      int m0 = 0;
      place_window<Q>(um, m0);
      load_seq(a, seq, off, L, lane);
      rows_in(S.wbase, g.p1spec + (size_t)pl * g.p1stride, a.SP, L, lane);
      const LenCfg cm = len_config(L, true);
      const float xC = pp->xC; const int ef = pp->ef;
      ST_T0();
      const WinDec wd = sweep_backward_decode_win<QB, Q, kLightTH, true, true>(c, (lds_u8 *)seq, L, cm, 1.0f / (xC * cm.move), ef, m0);
      ST_T1((QB == 4 ? 1 : 2));
      bool have_ro = false;
      if (wd.eps > -1e-4f && wd.eps < 0.01f) {
        RegOut ro = region_scan_cert<kLightTH, true>(c.spec, a.SP, L, (lds_i *)regs, lane, fmaxf(wd.eps, 0.f));
        have_ro = ((ro.flags >> 24) & 3) == 0;
        if (have_ro) {
          ro.flags &= 0xFFFFFF;
          __builtin_amdgcn_wave_barrier();
          wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + ((size_t)P.qi * a.H + h) : nullptr;
          store_regions(pp, ro, regs, lane, dp);
          if (lane == 0) pp->path |= WH_PATH_P2_WIN;
        }
      }
      if (have_ro) n_p2w++; else { n_p2rej++; if (lane == 0) pp->state = 3; }
    });
  if (a.paths && lane == 0) {
    if (n_p2w) atomicAdd(a.paths + 4, (unsigned long long)n_p2w);
    if (n_p2rej) atomicAdd(a.paths + 5, (unsigned long long)n_p2rej);
  }
  ST_K1((QB == 4 ? 1 : 2));
}

// ================================================================================================ p4win: envelope Backward on a node window
template <int Q, int QB>
__global__ __launch_bounds__(kLightTH) __attribute__((amdgpu_waves_per_eu(QB == 4 ? 5 : 3, QB == 4 ? 5 : 3))) void staged_p4win_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  StLds<Q> S(smem_raw, a.K, 0, wave, nwaves, a.wave_lds);
  WaveCtx c = make_ctx(a, S.em, nullptr, nullptr, S.wbase, lane);
  uint8_t *seq = ctx_seq(a, S.wbase);
  unsigned n_ok = 0, n_fail = 0;
  ST_K0();
  group_loop(g, QB == 4 ? ST_C_256 : ST_C_512, S.slot, S.cand, WH_MAX_ENVELOPES,
    [&](int h) {
      const DevHMM *hm = a.hmms + h;
      copy_f4(S.em, a.tables + hm->em_off, a.K * TBL);
      c.emG = (const glb_f *)(a.tables + hm->em_off);
      c.specg = (glb_f *)const_cast<float *>(a.tables + hm->bw_off);
    },
    [&](int pl, auto emit) {
      if (!pair_valid(g, pl)) return;
      const StPair *pp = g.pairs + pl;
      if (pp->state != 2) return;
      for (int e = 0; e < pp->nenv; e++) if (pp->cls[e] == (QB == 4 ? ST_CLS_W256 : ST_CLS_W512)) emit(pl * WH_MAX_ENVELOPES + e);
    },
    [&](int code, int h) {
      const int pl = code / WH_MAX_ENVELOPES, e = code % WH_MAX_ENVELOPES;
      StPair *pp = g.pairs + pl;
      const int uid = bcast_i(pp->uid[e]), ri = bcast_i(pp->regs[2 * e]), rj = bcast_i(pp->regs[2 * e + 1]);
      const int Ld = rj - ri + 1;
      const StUnit *u = g.units + uid;
      const int m0 = bcast_i(u->m0);
      const PairPos P = pair_pos(g, pl);
      const int64_t off = a.offsets[P.qi];
      const int L = (int)(a.offsets[P.qi + 1] - off);
      load_seq(a, seq, off + (ri - 1), Ld, lane);
      rows_in(S.wbase, g.p3spec + (size_t)uid * g.p3stride, a.SP, Ld, lane);
      c.Fs = (glb_f *)(g.slabs + (size_t)uid * g.slab_stride);
      const LenCfg cu = len_config(L, false);
      const float xC = u->xC;
      ST_T0();
      const P4Out p4 = sweep_backward_null2_win<QB, Q, kLightTH, false, true>(c, (lds_u8 *)seq, Ld, cu, 1.0f / (xC * cu.move), kWinTol7, m0);
      ST_T1((QB == 4 ? 5 : 6));
      const bool ok = fabsf((float)Ld - p4.mass) <= kWinTol7 * (float)Ld;
      if (ok) n_ok++; else n_fail++;
      if (lane == 0) {
        if (ok) { pp->domcorr[e] = p4.domcorr; pp->cls[e] = ST_CLS_NONE; atomicOr(&pp->path, QB == 4 ? WH_PATH_P4_W256 : WH_PATH_P4_W512); }
        else { pp->cls[e] = ST_CLS_FULL; atomicOr(&pp->path, WH_PATH_P4_WFAIL); }
      }
    });
  if (a.paths && lane == 0) {
    if (n_ok) atomicAdd(a.paths + (QB == 4 ? 0 : 1), (unsigned long long)n_ok);
    if (n_fail) atomicAdd(a.paths + 2, (unsigned long long)n_fail);
  }
  ST_K1((QB == 4 ? 5 : 6));
}

// ================================================================================================ assemble: A.6, one thread per pair
__global__ __launch_bounds__(256) void staged_assemble_kernel(StagedArgs g) {
  const ScoreArgs &a = g.a;
  const int pl = blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= g.n_items_b * a.QB || !pair_valid(g, pl)) return;
  const PairPos P = pair_pos(g, pl);
  const int h = P.h;
  const int64_t qi = P.qi;
  StPair *pp = g.pairs + pl;
  if (pp->state == 0) return;                       // P1 wrote the result
  const size_t out = (size_t)qi * a.H + h;
  const int L = (int)(a.offsets[qi + 1] - a.offsets[qi]);
  const double LOG2 = 0.69314718055994529;
  const LenCfg cm = len_config(L, true);
  const double fwd_nats = (double)pp->ef * LOG2 + log((double)(pp->xC * cm.move));
  const float fwdsc = (float)fwd_nats;
  const float p1 = (float)L / (float)(L + 1);
  const float nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
  const float fwd_bits_out = (float)((fwd_nats - (double)nullsc) / LOG2);
  int flags = pp->flags & 0xFF, decibits = 0;
  const int nenv = pp->nenv, nreg = pp->nreg, multi_mask = (pp->flags >> 8) & 0xFFFF;
  wh_pair_detail *dp = a.detail ? a.detail + out : nullptr;
  int path = pp->path;
  if ((path & 256) && a.paths) atomicAdd(a.paths + 5, 1ull);     // wanted a window for P2, none fitted (the fused kernel counts these as rejected)
  path &= 255;
  if (nenv > 0) {
    const bool queue_pair = multi_mask != 0 && a.rrecs != nullptr;
    float seqbias_sum = 0.f, sum_score = 0.f, sb2 = 0.f;
    int Ld_tot = 0;
    for (int e = 0; e < nenv; e++) {
      if (queue_pair && ((multi_mask >> e) & 1)) continue;
      const int ri = pp->regs[2 * e], rj = pp->regs[2 * e + 1];
      const int Ld = rj - ri + 1;
      const float envsc = pp->envsc[e], domcorr = pp->domcorr[e];
      seqbias_sum += domcorr;
      if (envsc - domcorr > 0.0f) { sum_score += envsc; Ld_tot += Ld; sb2 += domcorr; }
      if (dp) { dp->env_i[e] = ri; dp->env_j[e] = rj; dp->envsc[e] = envsc; dp->domcorr[e] = domcorr; }
    }
    if (queue_pair) {
      path |= WH_PATH_MULTI;
      const int slot = atomicAdd(a.rcount, 1);
      if (slot < a.rcap) {
        ResolveRec *rr = a.rrecs + slot;
        rr->q = qi; rr->h = h; rr->fwdsc = fwdsc; rr->fwd_bits = fwd_bits_out; rr->nreg = nreg; rr->nenv = nenv;
        rr->multi_mask = multi_mask; rr->flags = flags;
        for (int e = 0; e < nenv; e++) {
          const bool md = (multi_mask >> e) & 1;
          rr->ri[e] = pp->regs[2 * e]; rr->rj[e] = pp->regs[2 * e + 1];
          rr->envsc[e] = md ? 0.f : pp->envsc[e]; rr->domcorr[e] = md ? 0.f : pp->domcorr[e];
        }
      }
    } else {
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
  }
  a.decibits[out] = decibits;
  a.flags[out] = (uint8_t)flags;
  if (g.pair_paths) g.pair_paths[out] = (uint8_t)path;
}

template <class K>
static hipError_t launch_kind(K kern, const StagedArgs &g, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, s, g);
  return hipGetLastError();
}
#endif   // light part

}  // namespace WH_K7NS

#if WH_ST_PART != 2
#define WH_ST_DENSE_LAUNCHER(NAME, KERNEL)                                                                                  \
  hipError_t NAME(int Q, const StagedArgs &g, int blocks, int threads, size_t lds, hipStream_t s) {                         \
    using namespace WH_K7NS;                                                                                                \
    if (g.a.spec_arrays != kSpArr || threads > kStTH) return hipErrorInvalidValue;                                          \
    switch (Q) {                                                                                                            \
      case 8:  return launch_kind(&KERNEL<8>, g, blocks, threads, lds, s);                                                  \
      case 12: return launch_kind(&KERNEL<12>, g, blocks, threads, lds, s);                                                 \
      case 16: return launch_kind(&KERNEL<16>, g, blocks, threads, lds, s);                                                 \
      case 20: return launch_kind(&KERNEL<20>, g, blocks, threads, lds, s);                                                 \
      case 24: return launch_kind(&KERNEL<24>, g, blocks, threads, lds, s);                                                 \
      default: return hipErrorInvalidValue;                                                                                 \
    }                                                                                                                       \
  }
WH_ST_DENSE_LAUNCHER(launch_staged_p1, staged_p1_kernel)
WH_ST_DENSE_LAUNCHER(launch_staged_p2full, staged_p2full_kernel)
WH_ST_DENSE_LAUNCHER(launch_staged_p3, staged_p3_kernel)
WH_ST_DENSE_LAUNCHER(launch_staged_p4full, staged_p4full_kernel)
WH_ST_DENSE_LAUNCHER(launch_staged_dense, staged_dense_kernel)
WH_ST_DENSE_LAUNCHER(launch_staged_env, staged_env_kernel)
#endif

#if WH_ST_PART != 1
#define WH_ST_LIGHT_LAUNCHER(NAME, KERNEL)                                                                                  \
  hipError_t NAME(int Q, int QB, const StagedArgs &g, int blocks, int threads, size_t lds, hipStream_t s) {                 \
    using namespace WH_K7NS;                                                                                                \
    if (g.a.spec_arrays != kSpArr || threads > kLightTH) return hipErrorInvalidValue;                                       \
    if (QB == 4) {                                                                                                          \
      switch (Q) {                                                                                                          \
        case 8:  return launch_kind(&KERNEL<8, 4>, g, blocks, threads, lds, s);                                             \
        case 12: return launch_kind(&KERNEL<12, 4>, g, blocks, threads, lds, s);                                            \
        case 16: return launch_kind(&KERNEL<16, 4>, g, blocks, threads, lds, s);                                            \
        case 20: return launch_kind(&KERNEL<20, 4>, g, blocks, threads, lds, s);                                            \
        case 24: return launch_kind(&KERNEL<24, 4>, g, blocks, threads, lds, s);                                            \
        default: return hipErrorInvalidValue;                                                                               \
      }                                                                                                                     \
    }                                                                                                                       \
    if (QB == 8 && Q == 16) return launch_kind(&KERNEL<16, 8>, g, blocks, threads, lds, s);                                 \
    if (QB == 8 && Q == 24) return launch_kind(&KERNEL<24, 8>, g, blocks, threads, lds, s);                                 \
    return hipErrorInvalidValue;                                                                                            \
  }
WH_ST_LIGHT_LAUNCHER(launch_staged_p2win, staged_p2win_kernel)
WH_ST_LIGHT_LAUNCHER(launch_staged_p4win, staged_p4win_kernel)
hipError_t launch_staged_assemble(const StagedArgs &g, hipStream_t s) {
  using namespace WH_K7NS;
  const int blocks = (g.n_items_b * g.a.QB + 255) / 256;
  hipLaunchKernelGGL(staged_assemble_kernel, dim3(blocks), dim3(256), 0, s, g);
  return hipGetLastError();
}
#endif

}  // namespace wh
