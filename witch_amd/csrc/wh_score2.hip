// Scoring kernel, two (query, HMM) problems per wavefront (packed float32 math).
// EXPERIMENT kept for A/B (WH_SCORE_KERNEL=2): measured 968 ms against 719 ms for the fused
// single-problem kernel and 555 ms for the phase-call kernel on 8192 queries x 200 HMMs - on gfx950 a
// packed fp32 op holds the SIMD for two passes, so packing saves issue slots, not VALU time.
//
// What "hmmsearch --cpu 1 --noali -E 99999999 --max" computes for a (query, HMM) pair
// (witch_msa/gcmm/algorithm.py:526-532; algorithm: SURVEY.md A.2-A.6):
//   P1 multihit-local Forward (special states per row kept in LDS)
//   P2 multihit-local Backward fused with domain decoding (btot/etot/mocc) and region scan
//   per envelope: P3 unihit Forward (M/I rows spilled sparsely to a per-wave HBM slab),
//                 P4 unihit Backward fused with posterior accumulation -> null2 -> bias
//   score assembly, "%6.1f" rounding to deci-bits.
// A wavefront advances two queries of the same model together (wh_device2.h); a workgroup
// shares the model's tables in LDS and pulls (model, query-block) items from a global counter.
#include <hip/hip_runtime.h>

#include "wh_device2.h"
#include "wh_launch.h"

namespace wh {

// float32 table value of p7_FLogsum (A.6): table[i] = log(1 + exp(-i/1000)), 16000 entries
__device__ __forceinline__ float flogsum0(float b) {
  const float mx = b > 0.f ? b : 0.f, mn = b > 0.f ? 0.f : b;
  if (mn == -INFINITY || (mx - mn) >= 15.7f) return mx;
  const int idx = (int)((mx - mn) * 1000.0f);
  return mx + (float)log(1.0 + exp((double)-idx / 1000.0));
}

// lanes whose Forward cells are all below kKeepScale * E(row) are not spilled (attempt 0)
constexpr float kKeepScale = 9.094947e-13f;   // 2^-40
// tolerated |Ld - posterior mass| / Ld of the certificate (float32 accumulation noise is ~1e-6)
constexpr float kMassTol = 2e-5f;

#define COMP(v, c) ((c) ? (v).y : (v).x)

#ifndef WH_SCORE2_THREADS
#define WH_SCORE2_THREADS 512
#endif
template <int Q>
__global__ __launch_bounds__(WH_SCORE2_THREADS) void score_kernel2(ScoreArgs a) {
  // all LDS in ONE 16-byte aligned dynamic array (a static __shared__ object in front of it
  // would shift the base by 4 bytes and split every ds_read_b128)
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  volatile int *s_item_p = reinterpret_cast<volatile int *>(smem_raw);
  float *smem = smem_raw + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  float *emL = smem;
  float *trL = smem + (size_t)a.K * TBL;                    // fw[8] then bw[8]
  float *wbase = trL + 2 * FW_NARR * TBL + (size_t)wave * a.wave_lds;
  const int SP = a.SP;
  // per-wave block: for each of the two problems: spec[SP_NARR*SP], n2tab[32], regs[3*MAXENV], seq[Lcap pad 4]
  const int prob_lds = a.wave_lds / 2;
  float *spec_[2] = {wbase, wbase + prob_lds};
  float *n2tab_[2] = {spec_[0] + SP_NARR * SP, spec_[1] + SP_NARR * SP};
  int *regs_[2] = {reinterpret_cast<int *>(n2tab_[0] + 32), reinterpret_cast<int *>(n2tab_[1] + 32)};
  uint8_t *seq_[2] = {reinterpret_cast<uint8_t *>(regs_[0] + 3 * WH_MAX_ENVELOPES),
                      reinterpret_cast<uint8_t *>(regs_[1] + 3 * WH_MAX_ENVELOPES)};
  float *Fs_[2];
  Fs_[0] = a.scratch + ((size_t)blockIdx.x * nwaves + wave) * a.scratch_stride;
  Fs_[1] = Fs_[0] + a.scratch_stride / 2;
  int cur_h = -1;
  const DevHMM *hm = nullptr;
  const double LOG2 = 0.69314718055994529;
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;

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
      for (int t = threadIdx.x; t < FW_NARR * TBL / 4; t += blockDim.x) { d1[t] = s1[t]; d1[FW_NARR * TBL / 4 + t] = s2[t]; }
      cur_h = h;
      __syncthreads();
    }
    const float *emG = a.tables + hm->em_off;

    for (int64_t qbase = q_lo + 2 * wave; qbase < q_hi; qbase += 2 * nwaves) {
      // the wave's two problems; an odd tail duplicates the last query (its second copy is discarded)
      int64_t qi_[2] = {qbase, qbase + 1 < q_hi ? qbase + 1 : qbase};
      const bool dup = qi_[1] == qi_[0];
      int L_[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int64_t off = a.offsets[qi_[c]];
        int L = (int)(a.offsets[qi_[c] + 1] - off);
        if (L > a.Lcap) L = 0;
        L_[c] = L;
        for (int t = lane; t < a.Lcap; t += kWave) {
          int ch = t < L ? a.residues[off + t] : 0;
          seq_[c][t] = (uint8_t)(ch < a.Kp ? ch : a.Kp - 1);
        }
      }
      __builtin_amdgcn_wave_barrier();
      int flags_[2] = {0, 0}, deci_[2] = {0, 0};
      float fwdbits_[2] = {-INFINITY, -INFINITY};
      wh_pair_detail *dp_[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        dp_[c] = (a.detail && lane == 0 && !(c == 1 && dup)) ? a.detail + ((size_t)qi_[c] * a.H + h) : nullptr;
        if (dp_[c]) {
          dp_[c]->fwd_bits = -INFINITY; dp_[c]->seq_score = 0.f; dp_[c]->pre_score = 0.f; dp_[c]->seqbias_nats = 0.f;
          dp_[c]->nregions = 0; dp_[c]->nenv = 0;
        }
      }
      if (L_[0] > 0 || L_[1] > 0) {
        TransTab<Q, false> Tf, Tb;
        Tf.load(nullptr, trL, lane);
        Tb.load(nullptr, trL + FW_NARR * TBL, lane);
        const ScanC scf = scan_prepare(lane_product<Q, false>(Tf, FW_D2));
        const ScanC scb = scan_prepare(lane_product<Q, false>(Tb, BW_DD));
        Prob p0 = {seq_[0], L_[0], spec_[0], Fs_[0]}, p1 = {seq_[1], L_[1], spec_[1], Fs_[1]};
        const int Lmax = L_[0] > L_[1] ? L_[0] : L_[1];

        // ---------------- P1: multihit Forward
        const LenCfg2 cm = len_config2(L_[0] > 0 ? L_[0] : 1, L_[1] > 0 ? L_[1] : 1, true);
        v2f xC_L; v2i ef_L;
        forward_sweep2<Q, false>(Tf, scf, emL, emG, a.K, p0, p1, cm, SP, 0.f, lane, xC_L, ef_L);
        float fwdsc_[2], nullsc_[2], invZ_[2];
        bool ok_[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const int L = L_[c];
          const float xc = COMP(xC_L, c), mv = COMP(cm.move, c);
          const double fwd_nats = (double)COMP(ef_L, c) * LOG2 + log((double)(xc * mv));
          fwdsc_[c] = (float)fwd_nats;
          // A.3 null1 in float32 as p7_bg_SetLength / p7_bg_NullOne do
          const float p1f = (float)L / (float)(L + 1);
          nullsc_[c] = (float)((double)(float)L * log((double)p1f) + log(1.0 - (double)p1f));
          ok_[c] = L > 0 && xc > 0.f && isfinite(fwdsc_[c]);
          if (L > 0) fwdbits_[c] = (float)((fwd_nats - (double)nullsc_[c]) / LOG2);
          if (dp_[c]) dp_[c]->fwd_bits = fwdbits_[c];
          invZ_[c] = ok_[c] ? 1.0f / (xc * mv) : 0.f;
        }

        if (ok_[0] || ok_[1]) {
          // ---------------- P2: multihit Backward + domain decoding (A.4)
          {
            v2f Mb[Q], Ib[Q];
#pragma unroll
            for (int p = 0; p < Q; p++) { Mb[p] = splat(0.f); Ib[p] = splat(0.f); }
            Bck2 st;
            st.xC = cm.move; st.xJ = splat(0.f); st.xN = splat(0.f); st.xB = splat(0.f); st.eb = (v2i){0, 0};
#pragma unroll 1
            for (int i = Lmax; i >= 0; i--) {
              asm volatile("" ::: "memory");
              const v2f xE = backward_row2<Q>(Tb, scb, emL, emG, a.K, p0, p1, cm, i, lane, Mb, Ib, st, i >= 1);
              // decoding of row i; lane c handles problem c; results overwrite row i's forward slots
              __builtin_amdgcn_wave_barrier();
              if (lane < 2) {
                const bool z = lane == 0;
                const int Lc = z ? L_[0] : L_[1];
                if (i <= Lc && (z ? ok_[0] : ok_[1])) {
                  float *spec = z ? spec_[0] : spec_[1];
                  const int *specI = reinterpret_cast<const int *>(spec);
                  const float invZ = z ? invZ_[0] : invZ_[1];
                  const int efL = z ? ef_L.x : ef_L.y, eb = z ? st.eb.x : st.eb.y;
                  const float xe = z ? xE.x : xE.y, xb = z ? st.xB.x : st.xB.y, xn = z ? st.xN.x : st.xN.y;
                  const float xj = z ? st.xJ.x : st.xJ.y, xc = z ? st.xC.x : st.xC.y, lp = z ? cm.loop.x : cm.loop.y;
                  const float s_i = ldexpf(invZ, specI[SP_S * SP + i] + eb - efL);
                  const float pe = spec[SP_E * SP + i] * xe * s_i;
                  const float pb = spec[SP_B * SP + i] * xb * s_i;
                  float njc = 0.f;
                  if (i >= 1) {
                    const float s_p = ldexpf(invZ, specI[SP_S * SP + i - 1] + eb - efL);
                    njc = spec[SP_N * SP + i - 1] * xn;
                    njc = fmaf(spec[SP_J * SP + i - 1], xj, njc);
                    njc = fmaf(spec[SP_C * SP + i - 1], xc, njc);
                    njc = njc * lp * s_p;
                  }
                  spec[SP_E * SP + i] = pe; spec[SP_B * SP + i] = pb; spec[SP_N * SP + i] = njc;
                }
              }
              __builtin_amdgcn_wave_barrier();
            }
          }

          // ---------------- region scan (A.4); uniform over the wave, one problem after the other
          int nenv_[2] = {0, 0};
#pragma unroll
          for (int c = 0; c < 2; c++) {
            if (!ok_[c]) continue;
            float *spec = spec_[c];
            int *regs = regs_[c];
            const int L = L_[c];
            int nreg = 0, nenv = 0;
            float btot = 0.f, etot = 0.f;
            int i0 = -1;
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
                if (nenv < WH_MAX_ENVELOPES) {
                  if (lane == 0) { regs[2 * nenv] = i0; regs[2 * nenv + 1] = j; }
                  nenv++;
                } else flags_[c] |= WH_FLAG_TRUNC;
                nreg++;
                i0 = -1;
                trig = false;
              }
            }
            __builtin_amdgcn_wave_barrier();
            for (int e = 0; e < nenv; e++) {
              const int ri = regs[2 * e], rj = regs[2 * e + 1];
              float mx = -1.0f;
              const float e0 = spec[SP_C * SP + ri - 1], bj = spec[SP_J * SP + rj];
              for (int z = ri + lane; z <= rj; z += kWave) {
                const float u = spec[SP_C * SP + z] - e0, v = bj - spec[SP_J * SP + z - 1];
                mx = fmaxf(mx, fminf(u, v));
              }
              mx = wave_max(mx);
              if (mx >= rt3) flags_[c] |= WH_FLAG_MULTI;
            }
            nenv_[c] = nenv;
            if (dp_[c]) { dp_[c]->nregions = nreg; dp_[c]->nenv = nenv; }
          }

          // ---------------- envelopes: unihit Forward/Backward, null2 by expectation (A.5)
          float seqbias_sum_[2] = {0.f, 0.f}, sum_score_[2] = {0.f, 0.f}, sb2_[2] = {0.f, 0.f};
          int Ldtot_[2] = {0, 0};
          const int nround = nenv_[0] > nenv_[1] ? nenv_[0] : nenv_[1];
          const LenCfg2 cu = len_config2(L_[0] > 0 ? L_[0] : 1, L_[1] > 0 ? L_[1] : 1, false);
#pragma unroll 1
          for (int e = 0; e < nround; e++) {
            int ri_[2], Ld_[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
              const bool has = e < nenv_[c];
              ri_[c] = has ? regs_[c][2 * e] : 1;
              Ld_[c] = has ? regs_[c][2 * e + 1] - ri_[c] + 1 : 0;
            }
            Prob e0 = {seq_[0] + (ri_[0] - 1), Ld_[0], spec_[0], Fs_[0]}, e1 = {seq_[1] + (ri_[1] - 1), Ld_[1], spec_[1], Fs_[1]};
            const int Ldmax = Ld_[0] > Ld_[1] ? Ld_[0] : Ld_[1];
            float envsc_[2] = {-INFINITY, -INFINITY}, domcorr_[2] = {0.f, 0.f};
            // Attempt 0 spills only the lanes whose Forward cells exceed 2^-40 of the row total; the
            // posterior mass that reached the accumulators must then add up to Ld residues (every
            // residue is emitted by exactly one state).  If a certificate fails both envelopes are
            // redone with every line stored.
#pragma unroll 1
            for (int attempt = 0; attempt < 2; attempt++) {
              const float keep_scale = attempt == 0 ? kKeepScale : -1.0f;
              v2f xC_e; v2i ef_e;
              forward_sweep2<Q, true>(Tf, scf, emL, emG, a.K, e0, e1, cu, SP, keep_scale, lane, xC_e, ef_e);
              // the rows were written by other lanes of this wave: order the stores before the loads
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
              float invZe_[2];
              bool eok_[2];
#pragma unroll
              for (int c = 0; c < 2; c++) {
                const float xc = COMP(xC_e, c), mv = COMP(cu.move, c);
                eok_[c] = Ld_[c] > 0 && xc > 0.f;
                envsc_[c] = Ld_[c] > 0 ? (float)((double)COMP(ef_e, c) * LOG2 + log((double)(xc * mv))) : -INFINITY;
                invZe_[c] = eok_[c] ? 1.0f / (xc * mv) : 0.f;
                domcorr_[c] = 0.f;
              }
              if (!eok_[0] && !eok_[1]) break;
              v2f Mb[Q], Ib[Q], fM[Q];
#pragma unroll
              for (int p = 0; p < Q; p++) { Mb[p] = splat(0.f); Ib[p] = splat(0.f); fM[p] = splat(0.f); }
              v2f fIs = splat(0.f), xfac = splat(0.f);
              Bck2 st;
              st.xC = cu.move; st.xJ = splat(0.f); st.xN = splat(0.f); st.xB = splat(0.f); st.eb = (v2i){0, 0};
              const int src = kWave - 1 - lane;   // the forward-order lane that owns my (reversed) cells
#pragma unroll 1
              for (int i = Ldmax; i >= 1; i--) {
                asm volatile("" ::: "memory");
                const v2f xE = backward_row2<Q>(Tb, scb, emL, emG, a.K, e0, e1, cu, i, lane, Mb, Ib, st, true);
                (void)xE;
                // posterior accumulation for row i, per problem (its Forward row comes from its own slab)
                v2f s_i, s_p, njv;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                  const float *spec = spec_[c];
                  const int *specI = reinterpret_cast<const int *>(spec);
                  const bool live = i <= Ld_[c] && eok_[c];
                  const int ii = live ? i : 1;
                  const float si = live ? ldexpf(invZe_[c], specI[SP_S * SP + ii] + COMP(st.eb, c) - COMP(ef_e, c)) : 0.f;
                  const float sp = live ? ldexpf(invZe_[c], specI[SP_S * SP + ii - 1] + COMP(st.eb, c) - COMP(ef_e, c)) : 0.f;
                  float nj = spec[SP_N * SP + ii - 1] * COMP(st.xN, c);
                  nj = fmaf(spec[SP_J * SP + ii - 1], COMP(st.xJ, c), nj);
                  nj = fmaf(spec[SP_C * SP + ii - 1], COMP(st.xC, c), nj);
                  if (c == 0) { s_i.x = si; s_p.x = sp; njv.x = live ? nj : 0.f; } else { s_i.y = si; s_p.y = sp; njv.y = live ? nj : 0.f; }
                }
                xfac = fma2(njv * cu.loop, s_p, xfac);
                v2f idot = splat(0.f);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                  const unsigned *specU = reinterpret_cast<const unsigned *>(spec_[c]);
                  const bool live = i <= Ld_[c] && eok_[c];
                  const unsigned mword = live ? (src < 32 ? specU[SP_ML * SP + i] : specU[SP_MH * SP + i]) : 0u;
                  if ((mword >> (src & 31)) & 1u) {
                    const float4 *row = reinterpret_cast<const float4 *>(Fs_[c]) + (size_t)i * (2 * (Q / 4) * kWave) + src;
                    const float sc_i = COMP(s_i, c);
#pragma unroll
                    for (int p4 = 0; p4 < Q / 4; p4++) {
                      // reversed order: component 3-j of the forward-ordered vector is position 4*p4+j
                      const float4 fm = nt_load4(row + (Q / 4 - 1 - p4) * kWave);
                      const float4 fi = nt_load4(row + (Q / 4 + Q / 4 - 1 - p4) * kWave);
                      if (c == 0) {
                        fM[4 * p4 + 0].x = fmaf(fm.w * Mb[4 * p4 + 0].x, sc_i, fM[4 * p4 + 0].x);
                        fM[4 * p4 + 1].x = fmaf(fm.z * Mb[4 * p4 + 1].x, sc_i, fM[4 * p4 + 1].x);
                        fM[4 * p4 + 2].x = fmaf(fm.y * Mb[4 * p4 + 2].x, sc_i, fM[4 * p4 + 2].x);
                        fM[4 * p4 + 3].x = fmaf(fm.x * Mb[4 * p4 + 3].x, sc_i, fM[4 * p4 + 3].x);
                        idot.x = fmaf(fi.w, Ib[4 * p4 + 0].x, idot.x); idot.x = fmaf(fi.z, Ib[4 * p4 + 1].x, idot.x);
                        idot.x = fmaf(fi.y, Ib[4 * p4 + 2].x, idot.x); idot.x = fmaf(fi.x, Ib[4 * p4 + 3].x, idot.x);
                      } else {
                        fM[4 * p4 + 0].y = fmaf(fm.w * Mb[4 * p4 + 0].y, sc_i, fM[4 * p4 + 0].y);
                        fM[4 * p4 + 1].y = fmaf(fm.z * Mb[4 * p4 + 1].y, sc_i, fM[4 * p4 + 1].y);
                        fM[4 * p4 + 2].y = fmaf(fm.y * Mb[4 * p4 + 2].y, sc_i, fM[4 * p4 + 2].y);
                        fM[4 * p4 + 3].y = fmaf(fm.x * Mb[4 * p4 + 3].y, sc_i, fM[4 * p4 + 3].y);
                        idot.y = fmaf(fi.w, Ib[4 * p4 + 0].y, idot.y); idot.y = fmaf(fi.z, Ib[4 * p4 + 1].y, idot.y);
                        idot.y = fmaf(fi.y, Ib[4 * p4 + 2].y, idot.y); idot.y = fmaf(fi.x, Ib[4 * p4 + 3].y, idot.y);
                      }
                    }
                  }
                }
                fIs = fma2(idot, s_i, fIs);
              }
              // null2[a] = sum_k fM_k o_k(a) + sum_k fI_k + f_NJC, all / Ld; certificate on the mass
              v2f sm = splat(0.f);
#pragma unroll
              for (int p = 0; p < Q; p++) sm += fM[p];
              sm = wave_sum(sm);
              const v2f si = wave_sum(fIs);
              bool redo = false;
#pragma unroll
              for (int c = 0; c < 2; c++) {
                if (!eok_[c]) continue;
                const float deficit = fabsf((float)Ld_[c] - (COMP(sm, c) + COMP(si, c) + COMP(xfac, c)));
                if (!(deficit <= kMassTol * (float)Ld_[c])) redo = true;
              }
              if (attempt == 0 && redo) continue;
              v2f mine = splat(1.0f);
              for (int x = 0; x < a.K; x++) {
                float od[Q];
                load_em_rev<Q>(od, emL, emG, x, a.K, lane);
                v2f s = splat(0.f);
#pragma unroll
                for (int p = 0; p < Q; p++) s = fma2(od[p], fM[p], s);
                s = wave_sum(s);
                if (lane == x) {
                  mine.x = (s.x + si.x) / (float)(Ld_[0] > 0 ? Ld_[0] : 1) + xfac.x / (float)(Ld_[0] > 0 ? Ld_[0] : 1);
                  mine.y = (s.y + si.y) / (float)(Ld_[1] > 0 ? Ld_[1] : 1) + xfac.y / (float)(Ld_[1] > 0 ? Ld_[1] : 1);
                }
              }
#pragma unroll
              for (int c = 0; c < 2; c++) {
                if (!eok_[c]) continue;
                if (attempt == 1) flags_[c] |= WH_FLAG_EXACT;
                float *n2tab = n2tab_[c];
                float mn = COMP(mine, c);
                __builtin_amdgcn_wave_barrier();
                if (lane < a.K) n2tab[lane] = mn;
                __builtin_amdgcn_wave_barrier();
                if (lane >= a.K && lane < a.Kp) {
                  // degenerate codes: unweighted mean of the canonical ratios; gap/*/~ -> 1
                  const uint32_t m = a.degen[lane];
                  float s = 0.f; int n = 0;
                  for (int x = 0; x < a.K; x++) if (m & (1u << x)) { s += n2tab[x]; n++; }
                  mn = n > 0 ? s / (float)n : 1.0f;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < a.Kp) n2tab[lane] = logf(mn);
                __builtin_amdgcn_wave_barrier();
                float dc = 0.f;
                const uint8_t *eseq = seq_[c] + (ri_[c] - 1);
                for (int t = lane; t < Ld_[c]; t += kWave) dc += n2tab[eseq[t]];
                domcorr_[c] = wave_sum(dc);
              }
              break;
            }
#pragma unroll
            for (int c = 0; c < 2; c++) {
              if (e >= nenv_[c]) continue;
              seqbias_sum_[c] += domcorr_[c];
              if (envsc_[c] - domcorr_[c] > 0.0f) { sum_score_[c] += envsc_[c]; Ldtot_[c] += Ld_[c]; sb2_[c] += domcorr_[c]; }
              if (dp_[c]) {
                dp_[c]->env_i[e] = ri_[c]; dp_[c]->env_j[e] = ri_[c] + Ld_[c] - 1;
                dp_[c]->envsc[e] = envsc_[c]; dp_[c]->domcorr[e] = domcorr_[c];
              }
            }
          }

          // ---------------- A.6 score assembly (float32 where HMMER is float32)
#pragma unroll
          for (int c = 0; c < 2; c++) {
            if (!ok_[c] || nenv_[c] == 0) continue;
            const int L = L_[c];
            const float lomega = (float)log(1.0 / 256.0);
            const float seqbias = flogsum0(lomega + seqbias_sum_[c]);
            float pre_score = (float)(((double)fwdsc_[c] - (double)nullsc_[c]) / LOG2);
            float seq_score = (float)(((double)fwdsc_[c] - (double)(nullsc_[c] + seqbias)) / LOG2);
            const float sb2 = flogsum0(lomega + sb2_[c]);
            float sum_score = sum_score_[c] + (float)((double)(L - Ldtot_[c]) * log((double)((float)L / (float)(L + 3))));
            const float pre2 = (float)(((double)sum_score - (double)nullsc_[c]) / LOG2);
            sum_score = (float)(((double)sum_score - (double)(nullsc_[c] + sb2)) / LOG2);
            if (Ldtot_[c] > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2; flags_[c] |= WH_FLAG_OVERRIDE; }
            deci_[c] = (int)rint((double)seq_score * 10.0);
            flags_[c] |= WH_FLAG_REPORTED;
            if (dp_[c]) { dp_[c]->seq_score = seq_score; dp_[c]->pre_score = pre_score; dp_[c]->seqbias_nats = seqbias; }
          }
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
          if (c == 1 && dup) continue;
          const size_t out = (size_t)qi_[c] * a.H + h;
          const bool rep = (flags_[c] & WH_FLAG_REPORTED) != 0;
          a.decibits[out] = rep ? deci_[c] : 0;
          a.flags[out] = (uint8_t)(rep ? flags_[c] : (flags_[c] & (WH_FLAG_MULTI | WH_FLAG_TRUNC)));
          if (a.fwd_bits) a.fwd_bits[out] = fwdbits_[c];
        }
      }
    }
  }
}

template <int Q>
static hipError_t launch_one2(const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&score_kernel2<Q>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((score_kernel2<Q>), dim3(blocks), dim3(threads), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_score2(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  switch (Q) {
    case 4:  return launch_one2<4>(a, blocks, threads, lds, s);
    case 8:  return launch_one2<8>(a, blocks, threads, lds, s);
    case 12: return launch_one2<12>(a, blocks, threads, lds, s);
    case 16: return launch_one2<16>(a, blocks, threads, lds, s);
    case 20: return launch_one2<20>(a, blocks, threads, lds, s);
    case 24: return launch_one2<24>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace wh
