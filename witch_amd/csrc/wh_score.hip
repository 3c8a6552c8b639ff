// Scoring kernel: what "hmmsearch --cpu 1 --noali -E 99999999 --max" computes for one
// (query, HMM) pair (witch_msa/gcmm/algorithm.py:526-532; algorithm: SURVEY.md A.2-A.6),
// fused into one launch per model-size class:
//   P1 multihit-local Forward (special states per row kept in LDS)
//   P2 multihit-local Backward fused with domain decoding (btot/etot/mocc) and region scan
//   per envelope: P3 unihit Forward (M/I rows spilled to a per-wave HBM slab),
//                 P4 unihit Backward fused with posterior accumulation -> null2 -> bias
//   score assembly, "%6.1f" rounding to deci-bits.
// One wavefront per pair; a workgroup shares one model's emission table in LDS and pulls
// (model, query-block) items from a global counter until the list is drained.
//
// This fused body (two waves per SIMD) was the default until wh_score7.hip (the same sweeps as
// non-inlined functions, three waves per SIMD) replaced it; it stays selectable
// (WH_SCORE_KERNEL=1, and 3 for the split-phase experiment) as the A/B reference the design
// notes quote, and wh_score_big.hip's long-model variant shares its structure.
#include <hip/hip_runtime.h>

#include "wh_device.h"
#include "wh_launch.h"

namespace wh {

// float32 table value of p7_FLogsum (A.6): table[i] = log(1 + exp(-i/1000)), 16000 entries
__device__ __forceinline__ float flogsum0_v1(float b) {
  // FLogsum(0, b)
  const float mx = b > 0.f ? b : 0.f, mn = b > 0.f ? 0.f : b;
  if (mn == -INFINITY || (mx - mn) >= 15.7f) return mx;
  const int idx = (int)((mx - mn) * 1000.0f);
  return mx + (float)log(1.0 + exp((double)-idx / 1000.0));
}

// lanes whose Forward cells are all below kKeepScale * E(row) are not spilled (attempt 0).
// 2^-24 keeps ~19 of 64 lane blocks per row on the headline workload (2^-40: ~29) with no
// certificate failure and bit-identical deci-bit scores on 1.6M pairs (tools/ab_score.py).
constexpr float kKeepScale1 = 5.9604645e-08f;   // 2^-24
// tolerated |Ld - posterior mass| / Ld of the certificate (float32 accumulation noise is ~1e-6)
constexpr float kMassTol1 = 2e-5f;

#ifndef WH_SCORE1_THREADS
#define WH_SCORE1_THREADS 512
#endif
#ifndef WH_SCOREA_THREADS
#define WH_SCOREA_THREADS 1024
#endif
#ifndef WH_SCOREB_THREADS
#define WH_SCOREB_THREADS 768
#endif
// PHASE 0: everything in one launch.  PHASE 1: P1 + P2 + region scan, result to a PairRec.
// PHASE 2: envelopes (P3/P4) + score assembly from the PairRec.  Splitting gives each half its
// own register allocation (the fused kernel spills) and its own occupancy.
// Special-state rows live in the per-wave LDS block, or (SPECG, long queries) in a per-wave
// global scratch region read with L1-bypassing loads (the slots are rewritten for every pair).
#define SPR(idx)  (SPECG ? __builtin_nontemporal_load(spec + (idx)) : spec[idx])
#define SPRI(idx) (SPECG ? __builtin_nontemporal_load(reinterpret_cast<const int *>(spec) + (idx)) : reinterpret_cast<const int *>(spec)[idx])
#define SPRU(idx) (SPECG ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(spec) + (idx)) : reinterpret_cast<const unsigned *>(spec)[idx])

#define WH_TICK(slot) do { if (a.stats) { const long long t_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(a.stats + (slot), (unsigned long long)(t_now - t_last)); t_last = t_now; } } while (0)

template <int Q, bool TREG, int PHASE, bool SPECG, int TRM = 0>
__global__ __launch_bounds__(PHASE == 1 ? WH_SCOREA_THREADS : (PHASE == 2 ? WH_SCOREB_THREADS : WH_SCORE1_THREADS)) void score_kernel(ScoreArgs a) {
  // all LDS in ONE 16-byte aligned dynamic array: a static __shared__ object in front of it
  // would shift the base by 4 bytes and split every ds_read_b128 (measured: 13x LDS time)
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  volatile int *s_item_p = reinterpret_cast<volatile int *>(smem_raw);
  float *smem = smem_raw + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;   // floats per table array
  float *emL = smem;
  float *trL = smem + (size_t)a.K * TBL;                                   // !TREG: fw[8] then bw[8]
  float *wbase = trL + (TREG ? 0 : 2 * FW_NARR * TBL) + (size_t)wave * a.wave_lds;  // per-wave block
  float *spec = SPECG ? a.spec_scratch + ((size_t)blockIdx.x * nwaves + wave) * a.spec_stride : wbase;   // SP_NARR * SP floats
  float *n2tab = wbase + (SPECG ? 0 : SP_NARR * a.SP);                                         // 32 floats
  int *regs = reinterpret_cast<int *>(n2tab + 32);                         // 3 * WH_MAX_ENVELOPES ints
  uint8_t *seq = reinterpret_cast<uint8_t *>(regs + 3 * WH_MAX_ENVELOPES);
  float *Fs = a.scratch + ((size_t)blockIdx.x * nwaves + wave) * a.scratch_stride;
  const int SP = a.SP;
  int cur_h = -1;
  const DevHMM *hm = nullptr;

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
      if (!TREG) {
        const float4 *s1 = reinterpret_cast<const float4 *>(a.tables + hm->fw_off);
        const float4 *s2 = reinterpret_cast<const float4 *>(a.tables + hm->bw_off);
        float4 *d1 = reinterpret_cast<float4 *>(trL);
        for (int t = threadIdx.x; t < FW_NARR * TBL / 4; t += blockDim.x) { d1[t] = s1[t]; d1[FW_NARR * TBL / 4 + t] = s2[t]; }
      }
      cur_h = h;
      __syncthreads();
    }
    const float *emG = a.tables + hm->em_off;
    const float *fwG = a.tables + hm->fw_off, *bwG = a.tables + hm->bw_off;

    for (int64_t qi = q_lo + wave; qi < q_hi; qi += nwaves) {
      const int64_t off = a.offsets[qi];
      const int L = (int)(a.offsets[qi + 1] - off);
      const size_t out = (size_t)qi * a.H + h;
      int flags = 0, decibits = 0;
      float fwd_bits_out = -INFINITY;
      wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + out : nullptr;
      if (dp && PHASE != 2) {
        dp->fwd_bits = -INFINITY; dp->seq_score = 0.f; dp->pre_score = 0.f; dp->seqbias_nats = 0.f;
        dp->nregions = 0; dp->nenv = 0;
      }
      if (L > 0 && L <= a.Lcap) {
        for (int t = lane; t < L; t += kWave) {
          int c = a.residues[off + t];
          seq[t] = (uint8_t)(c < a.Kp ? c : a.Kp - 1);
        }
        __builtin_amdgcn_wave_barrier();

        float fwdsc = 0.f, nullsc = 0.f, invZ = 0.f;
        long long t_last = a.stats ? __builtin_readcyclecounter() : 0;
        int ef_L = 0, nreg = 0, nenv = 0;
        bool ok = false;
        PairRec *rec = a.recs ? a.recs + out : nullptr;
        if constexpr (PHASE != 2) {
        // ---------------- P1: multihit Forward
        const LenCfg cm = len_config(L, true);
        float xC_L;
        {
          constexpr bool TR1 = TREG || (TRM & 1);
          TransTab<Q, TR1> T;
          T.load(fwG, trL, lane);
          const ScanC sc = scan_prepare(lane_product<Q, TR1>(T, FW_D2));
          forward_sweep<Q, TR1, false>(T, sc, emL, emG, a.K, seq, L, cm, spec, SP, nullptr, 0.f, lane, xC_L, ef_L);
        }
        if (SPECG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const double fwd_nats = (double)ef_L * 0.69314718055994529 + log((double)(xC_L * cm.move));
        fwdsc = (float)fwd_nats;
        // A.3 null1 in float32 as p7_bg_SetLength / p7_bg_NullOne do
        const float p1 = (float)L / (float)(L + 1);
        nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
        fwd_bits_out = (float)((fwd_nats - (double)nullsc) / 0.69314718055994529);
        if (dp) dp->fwd_bits = fwd_bits_out;
        ok = xC_L > 0.f && isfinite(fwdsc);
        invZ = ok ? 1.0f / (xC_L * cm.move) : 0.f;
        if (ok) {
          WH_TICK(4);
          // ---------------- P2: multihit Backward + domain decoding
          {
            constexpr bool TR2 = TREG || (TRM & 2);
            TransTab<Q, TR2> T;
            T.load(bwG, trL + FW_NARR * TBL, lane);
            const ScanC sc = scan_prepare(lane_product<Q, TR2>(T, BW_DD));
            float Mb[Q], Ib[Q];
#pragma unroll
            for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; }
            float xC = cm.move, xJ = 0.f, xN = 0.f, xB = 0.f;
            int eb = 0;
#pragma unroll 1
            for (int i = L; i >= 0; i--) {
              asm volatile("" ::: "memory");
              if (i < L) {
                float od[Q];
                load_em_rev<Q>(od, emL, emG, seq[i], a.K, lane);
                float part = 0.f;
#pragma unroll
                for (int p4 = 0; p4 < Q / 4; p4++) {
                  const float4 E = T.ld(BW_E, p4);
#pragma unroll
                  for (int j = 0; j < 4; j++) {
                    const int p = 4 * p4 + j;
                    Mb[p] *= od[p];
                    part = fmaf(f4get(E, j), Mb[p], part);
                  }
                }
                xB = wave_sum(part);
                xJ = fmaf(xJ, cm.loop, xB * cm.move);
                xC = xC * cm.loop;
                xN = fmaf(xN, cm.loop, xB * cm.move);
              }
              float xE = fmaf(xC, cm.EC, xJ * cm.EJ);
              if (i >= 1) backward_cells<Q, TR2>(T, sc, Mb, Ib, xE);
              const float big = fmaxf(xB, xN);
              if (big > kRescaleHi) {
                const int e = f32_exponent(big);
                const float r = pow2f_int(-e);
#pragma unroll
                for (int p = 0; p < Q; p++) { Mb[p] *= r; Ib[p] *= r; }
                xB *= r; xJ *= r; xC *= r; xN *= r; xE *= r;
                eb += e;
              }
              // domain decoding for row i (A.4); results overwrite row i's forward slots
              const float s_i = ldexpf(invZ, SPRI(SP_S * SP + i) + eb - ef_L);
              const float pe = SPR(SP_E * SP + i) * xE * s_i;
              const float pb = SPR(SP_B * SP + i) * xB * s_i;
              float njc = 0.f;
              if (i >= 1) {
                const float s_p = ldexpf(invZ, SPRI(SP_S * SP + i - 1) + eb - ef_L);
                njc = SPR(SP_N * SP + i - 1) * xN;
                njc = fmaf(SPR(SP_J * SP + i - 1), xJ, njc);
                njc = fmaf(SPR(SP_C * SP + i - 1), xC, njc);
                njc = njc * cm.loop * s_p;
              }
              __builtin_amdgcn_wave_barrier();
              if (lane == 0) { spec[SP_E * SP + i] = pe; spec[SP_B * SP + i] = pb; spec[SP_N * SP + i] = njc; }
              __builtin_amdgcn_wave_barrier();
            }
          }

          if (SPECG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          WH_TICK(5);
          // ---------------- region scan (A.4), uniform over the wave
          const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
          {
            float btot = 0.f, etot = 0.f;
            int i0 = -1;
            bool trig = false;
            if (lane == 0) { spec[SP_J * SP] = 0.f; spec[SP_C * SP] = 0.f; }
            for (int j = 1; j <= L; j++) {
              const float mocc = 1.0f - SPR(SP_N * SP + j);
              const float bold = btot, eold = etot;
              btot += SPR(SP_B * SP + j - 1);
              etot += SPR(SP_E * SP + j);
              if (lane == 0) { spec[SP_J * SP + j] = btot; spec[SP_C * SP + j] = etot; }
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
            __builtin_amdgcn_wave_barrier();
            if (SPECG) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            // multidomain test: max_z min(etot[z]-etot[i-1], btot[j]-btot[z-1]) >= rt3
            for (int e = 0; e < nenv; e++) {
              const int ri = regs[2 * e], rj = regs[2 * e + 1];
              float mx = -1.0f;
              const float e0 = SPR(SP_C * SP + ri - 1), bj = SPR(SP_J * SP + rj);
              for (int z = ri + lane; z <= rj; z += kWave) {
                const float u = SPR(SP_C * SP + z) - e0, v = bj - SPR(SP_J * SP + z - 1);
                mx = fmaxf(mx, fminf(u, v));
              }
              mx = wave_max(mx);
              if (mx >= rt3) flags |= WH_FLAG_MULTI;
            }
          }
          if (dp) { dp->nregions = nreg; dp->nenv = nenv; }
        }   // ok (P2 + region scan)
        }   // PHASE != 2
        if constexpr (PHASE == 1) {
          if (lane == 0) {
            rec->fwdsc = fwdsc; rec->nullsc = nullsc; rec->fwd_bits = fwd_bits_out;
            rec->nenv = ok ? nenv : 0; rec->flags = flags;
            for (int t = 0; t < 2 * nenv && t < 2 * WH_MAX_ENVELOPES; t++) rec->regs[t] = (int16_t)regs[t];
          }
        }
        if constexpr (PHASE == 2) {
          fwdsc = rec->fwdsc; nullsc = rec->nullsc; fwd_bits_out = rec->fwd_bits;
          nenv = rec->nenv; flags = rec->flags; ok = nenv > 0;
          if (lane < 2 * nenv) regs[lane] = rec->regs[lane];
          __builtin_amdgcn_wave_barrier();
        }
        if constexpr (PHASE != 1) {
          if (ok && nenv > 0) {
            WH_TICK(6);
            // ---------------- envelopes: unihit Forward/Backward, null2 by expectation (A.5)
            const LenCfg cu = len_config(L, false);
            float seqbias_sum = 0.f, sum_score = 0.f, sb2 = 0.f;
            int Ld_tot = 0;
            for (int e = 0; e < nenv; e++) {
              const int ri = regs[2 * e], rj = regs[2 * e + 1];
              const int Ld = rj - ri + 1;
              const uint8_t *eseq = seq + (ri - 1);
              float envsc = -INFINITY, domcorr = 0.f;
              // Attempt 0 spills only the lanes whose Forward cells exceed 2^-24 of the row total;
              // the posterior mass that reached the accumulators must then add up to Ld residues
              // (every residue is emitted by exactly one state).  If the certificate fails the
              // envelope is redone with every line stored.
#pragma unroll 1
              for (int attempt = 0; attempt < 2; attempt++) {
                const float keep_scale = (a.dbg & 1) ? INFINITY : (attempt == 0 ? (a.keep_scale > 0.f ? a.keep_scale : kKeepScale1) : -1.0f);
                float xC_e; int ef_e;
                {
                  constexpr bool TR3 = TREG || (TRM & 4);
                  TransTab<Q, TR3> T;
                  T.load(fwG, trL, lane);
                  const ScanC sc = scan_prepare(lane_product<Q, TR3>(T, FW_D2));
                  forward_sweep<Q, TR3, true>(T, sc, emL, emG, a.K, eseq, Ld, cu, spec, SP, Fs, keep_scale, lane, xC_e, ef_e);
                }
                // the rows were written by other lanes of this wave: order the stores before the loads
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (a.stats) {
                  unsigned kept = 0;
                  for (int t = 1 + lane; t <= Ld; t += kWave) kept += __popc(SPRU(SP_ML * SP + t)) + __popc(SPRU(SP_MH * SP + t));
                  kept = (unsigned)wave_sum((float)kept);
                  if (lane == 0) {
                    atomicAdd(a.stats + 0, (unsigned long long)Ld); atomicAdd(a.stats + 1, (unsigned long long)kept);
                    atomicAdd(a.stats + (attempt == 0 ? 2 : 3), 1ull);
                  }
                }
                envsc = (float)((double)ef_e * 0.69314718055994529 + log((double)(xC_e * cu.move)));
                domcorr = 0.f;
                if (!(xC_e > 0.f)) break;
                WH_TICK(7);
                const float invZe = 1.0f / (xC_e * cu.move);
                TransTab<Q, TREG> T;
                T.load(bwG, trL + FW_NARR * TBL, lane);
                const ScanC sc = scan_prepare(lane_product<Q, TREG>(T, BW_DD));
                float Mb[Q], Ib[Q], fM[Q], fI[Q];
#pragma unroll
                for (int p = 0; p < Q; p++) { Mb[p] = 0.f; Ib[p] = 0.f; fM[p] = 0.f; fI[p] = 0.f; }
                float xC = cu.move, xJ = 0.f, xN = 0.f, xB = 0.f, xfac = 0.f;

                  const int src = kWave - 1 - lane;   // the forward-order lane that owns my (reversed) cells
#pragma unroll 1
                for (int i = Ld; i >= 1; i--) {
                  asm volatile("" ::: "memory");
                  // issue the loads of Forward row i early; they are consumed after the cell update
                  const unsigned mword = src < 32 ? SPRU(SP_ML * SP + i) : SPRU(SP_MH * SP + i);
                  const bool have = ((mword >> (src & 31)) & 1u) && !(a.dbg & 2);
                  const float4 *row = reinterpret_cast<const float4 *>(Fs) + (size_t)i * (2 * (Q / 4) * kWave) + src;
                  float4 fm4[Q / 4], fi4[Q / 4];
                  if (have) {
#pragma unroll
                    for (int p4 = 0; p4 < Q / 4; p4++) {
                      fm4[p4] = nt_load4(row + (Q / 4 - 1 - p4) * kWave);
                      fi4[p4] = nt_load4(row + (Q / 4 + Q / 4 - 1 - p4) * kWave);
                    }
                  } else {
#pragma unroll
                    for (int p4 = 0; p4 < Q / 4; p4++) { fm4[p4] = make_float4(0.f, 0.f, 0.f, 0.f); fi4[p4] = fm4[p4]; }
                  }
                  if (i < Ld) {
                    mirror_scale<Q>(SPRI(SP_S * SP + i + 1) - SPRI(SP_S * SP + i), Mb, Ib, xJ, xC, xN);
                    float od[Q];
                    load_em_rev<Q>(od, emL, emG, eseq[i], a.K, lane);
                    float part = 0.f;
#pragma unroll
                    for (int p4 = 0; p4 < Q / 4; p4++) {
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
                  clamp_backward<Q>(Mb, Ib, xB, xJ, xC, xN);
                  // mirrored scaling (wh_device.h, "envelope Backward scaling")
                  const float s_i = invZe;
                  const float s_p = ldexpf(invZe, SPRI(SP_S * SP + i - 1) - SPRI(SP_S * SP + i));
#pragma unroll
                  for (int p4 = 0; p4 < Q / 4; p4++) {
                    // reversed order: component 3-j of the forward-ordered vector is position 4*p4+j
                    fM[4 * p4 + 0] = fmaf(fm4[p4].w * Mb[4 * p4 + 0], s_i, fM[4 * p4 + 0]);
                    fM[4 * p4 + 1] = fmaf(fm4[p4].z * Mb[4 * p4 + 1], s_i, fM[4 * p4 + 1]);
                    fM[4 * p4 + 2] = fmaf(fm4[p4].y * Mb[4 * p4 + 2], s_i, fM[4 * p4 + 2]);
                    fM[4 * p4 + 3] = fmaf(fm4[p4].x * Mb[4 * p4 + 3], s_i, fM[4 * p4 + 3]);
                    fI[4 * p4 + 0] = fmaf(fi4[p4].w * Ib[4 * p4 + 0], s_i, fI[4 * p4 + 0]);
                    fI[4 * p4 + 1] = fmaf(fi4[p4].z * Ib[4 * p4 + 1], s_i, fI[4 * p4 + 1]);
                    fI[4 * p4 + 2] = fmaf(fi4[p4].y * Ib[4 * p4 + 2], s_i, fI[4 * p4 + 2]);
                    fI[4 * p4 + 3] = fmaf(fi4[p4].x * Ib[4 * p4 + 3], s_i, fI[4 * p4 + 3]);
                  }
                  float nj = SPR(SP_N * SP + i - 1) * xN;
                  nj = fmaf(SPR(SP_J * SP + i - 1), xJ, nj);
                  nj = fmaf(SPR(SP_C * SP + i - 1), xC, nj);
                  xfac = fmaf(nj * cu.loop, s_p, xfac);
                }
                WH_TICK(8);
                // null2[a] = sum_k fM_k o_k(a) + sum_k fI_k + f_NJC, all / Ld
                const float norm = 1.0f / (float)Ld;
                float si = 0.f, sm = 0.f;
#pragma unroll
                for (int p = 0; p < Q; p++) { si += fI[p]; sm += fM[p]; }
                si = wave_sum(si);
                sm = wave_sum(sm);
                // certificate: posterior mass over all emitting states = number of residues
                const float deficit = fabsf((float)Ld - (sm + si + xfac));
                if (a.stats && lane == 0) atomicMax(a.stats + 10 + attempt, (unsigned long long)__float_as_uint(deficit / (float)Ld));
                if (attempt == 0 && !a.dbg && !(deficit <= kMassTol1 * (float)Ld)) continue;
                if (attempt == 1) flags |= WH_FLAG_EXACT;
                float mine = 1.0f;
                for (int x = 0; x < a.K; x++) {
                  float od[Q];
                  load_em_rev<Q>(od, emL, emG, x, a.K, lane);
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
                  // degenerate codes: unweighted mean of the canonical ratios; gap/*/~ -> 1
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
                break;
              }
              seqbias_sum += domcorr;
              if (envsc - domcorr > 0.0f) { sum_score += envsc; Ld_tot += Ld; sb2 += domcorr; }
              if (dp) { dp->env_i[e] = ri; dp->env_j[e] = rj; dp->envsc[e] = envsc; dp->domcorr[e] = domcorr; }
            }
            WH_TICK(9);
            // ---------------- A.6 score assembly (float32 where HMMER is float32)
            const double LOG2 = 0.69314718055994529;
            const float lomega = (float)log(1.0 / 256.0);
            const float seqbias = flogsum0_v1(lomega + seqbias_sum);
            float pre_score = (float)(((double)fwdsc - (double)nullsc) / LOG2);
            float seq_score = (float)(((double)fwdsc - (double)(nullsc + seqbias)) / LOG2);
            sb2 = flogsum0_v1(lomega + sb2);
            sum_score += (float)((double)(L - Ld_tot) * log((double)((float)L / (float)(L + 3))));
            const float pre2 = (float)(((double)sum_score - (double)nullsc) / LOG2);
            sum_score = (float)(((double)sum_score - (double)(nullsc + sb2)) / LOG2);
            if (Ld_tot > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2; flags |= WH_FLAG_OVERRIDE; }
            decibits = (int)rint((double)seq_score * 10.0);
            flags |= WH_FLAG_REPORTED;
            if (dp) { dp->seq_score = seq_score; dp->pre_score = pre_score; dp->seqbias_nats = seqbias; }
          }
        }   // PHASE != 1
      }
      if (PHASE != 1 && lane == 0) {
        a.decibits[out] = decibits;
        a.flags[out] = (uint8_t)flags;
        if (a.fwd_bits) a.fwd_bits[out] = fwd_bits_out;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
template <int Q, bool TREG, int PHASE, bool SPECG, int TRM = 0>
static hipError_t launch_one(const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_kernel<Q, TREG, PHASE, SPECG, TRM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((score_kernel<Q, TREG, PHASE, SPECG, TRM>), dim3(blocks), dim3(threads), lds, s, a);
  return hipGetLastError();
}

template <int PHASE, bool SPECG>
static hipError_t launch_phase(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  switch (Q) {
    case 4:  return launch_one<4, false, PHASE, SPECG>(a, blocks, threads, lds, s);
    case 8:  return launch_one<8, false, PHASE, SPECG>(a, blocks, threads, lds, s);
    case 12: return launch_one<12, false, PHASE, SPECG>(a, blocks, threads, lds, s);
    case 16: return launch_one<16, false, PHASE, SPECG>(a, blocks, threads, lds, s);
    case 20: return launch_one<20, false, PHASE, SPECG>(a, blocks, threads, lds, s);
    case 24: return launch_one<24, false, PHASE, SPECG>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

// transition tables in VGPRs (one orientation at a time, reloaded from L2 per sweep)
hipError_t launch_score_treg(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  switch (Q) {
    case 4:  return launch_one<4, true, 0, false>(a, blocks, threads, lds, s);
    case 8:  return launch_one<8, true, 0, false>(a, blocks, threads, lds, s);
    case 12: return launch_one<12, true, 0, false>(a, blocks, threads, lds, s);
    case 16: return launch_one<16, true, 0, false>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

// selected sweeps with register-resident transition tables (mask: 1 = P1, 2 = P2, 4 = P3)
hipError_t launch_score_tr12(int Q, int mask, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  if (Q == 16 && mask == 1) return launch_one<16, false, 0, false, 1>(a, blocks, threads, lds, s);
  if (Q == 16 && mask == 5) return launch_one<16, false, 0, false, 5>(a, blocks, threads, lds, s);
  if (Q == 16 && mask == 3) return launch_one<16, false, 0, false, 3>(a, blocks, threads, lds, s);
  if (Q == 16 && mask == 7) return launch_one<16, false, 0, false, 7>(a, blocks, threads, lds, s);
  return launch_one<16, false, 0, false, 0>(a, blocks, threads, lds, s);
}

hipError_t launch_score(int Q, int phase, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  if (a.spec_scratch) return launch_phase<0, true>(Q, a, blocks, threads, lds, s);   // long queries: fused only
  if (phase == 1) return launch_phase<1, false>(Q, a, blocks, threads, lds, s);
  if (phase == 2) return launch_phase<2, false>(Q, a, blocks, threads, lds, s);
  return launch_phase<0, false>(Q, a, blocks, threads, lds, s);
}

}  // namespace wh
