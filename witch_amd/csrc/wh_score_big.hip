// Scoring kernel for LONG models (1536 < M <= 3072 nodes, Q = 28..48 cells per lane).
//
// Same algorithm and device functions as wh_score.hip, but both transition orientations no
// longer fit in LDS together.  The workgroup therefore keeps ONE orientation resident and its
// waves run the sweeps in lockstep ("pass-synchronous"): Forward sweeps with the forward
// tables, then a workgroup barrier + table swap (~100 KB from L2, a few microseconds against
// a sweep of hundreds), then the Backward sweeps.  The per-row special states live in HBM
// (the SPECG layout of wh_score.hip), so LDS holds only tables + residues.  The DP row of a
// 3072-node model is 144 VGPRs per lane: one wavefront per SIMD (__launch_bounds__(256)),
// spilling into the AGPR half of the register file.
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

template <int Q, int TH>
__global__ __launch_bounds__(TH) void score_big_kernel(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem_raw[];
  volatile int *s_item_p = reinterpret_cast<volatile int *>(smem_raw);
  float *smem = smem_raw + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
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
      const int64_t qi = q_lo + (int64_t)round * nwaves + wave;
      int L = 0;
      int64_t off = 0;
      if (qi < q_hi) { off = a.offsets[qi]; L = (int)(a.offsets[qi + 1] - off); if (L > a.Lcap) L = 0; }
      const bool active = qi < q_hi && L > 0;
      const size_t out = (size_t)(qi < q_hi ? qi : q_lo) * a.H + h;
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
      orient(0);
      if (active) {
        TransTab<Q, false> T;
        T.load(nullptr, trL, lane);
        const ScanC sc = scan_prepare(lane_product<Q, false>(T, FW_D2));
        float xC_L;
        forward_sweep<Q, false, false>(T, sc, emL, emG, Klds, seq, L, cm, spec, SP, nullptr, 0.f, lane, xC_L, ef_L);
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
      orient(1);
      if (active && ok) {
        TransTab<Q, false> T;
        T.load(nullptr, trL, lane);
        const ScanC sc = scan_prepare(lane_product<Q, false>(T, BW_DD));
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
            load_em_rev<Q>(od, emL, emG, seq[i], Klds, lane);
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
          if (i >= 1) backward_cells<Q, false>(T, sc, Mb, Ib, xE);
          const float big = fmaxf(xB, xN);
          if (big > kRescaleHi) {
            const int e = f32_exponent(big);
            const float r = pow2f_int(-e);
#pragma unroll
            for (int p = 0; p < Q; p++) { Mb[p] *= r; Ib[p] *= r; }
            xB *= r; xJ *= r; xC *= r; xN *= r; xE *= r;
            eb += e;
          }
          const float s_i = ldexpf(invZ, BSPRI(SP_S * SP + i) + eb - ef_L);
          const float pe = BSPR(SP_E * SP + i) * xE * s_i;
          const float pb = BSPR(SP_B * SP + i) * xB * s_i;
          float njc = 0.f;
          if (i >= 1) {
            const float s_p = ldexpf(invZ, BSPRI(SP_S * SP + i - 1) + eb - ef_L);
            njc = BSPR(SP_N * SP + i - 1) * xN;
            njc = fmaf(BSPR(SP_J * SP + i - 1), xJ, njc);
            njc = fmaf(BSPR(SP_C * SP + i - 1), xC, njc);
            njc = njc * cm.loop * s_p;
          }
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) { spec[SP_E * SP + i] = pe; spec[SP_B * SP + i] = pb; spec[SP_N * SP + i] = njc; }
          __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // region scan (A.4)
        const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
        float btot = 0.f, etot = 0.f;
        int i0 = -1;
        bool trig = false;
        if (lane == 0) { spec[SP_J * SP] = 0.f; spec[SP_C * SP] = 0.f; }
        for (int j = 1; j <= L; j++) {
          const float mocc = 1.0f - BSPR(SP_N * SP + j);
          const float bold = btot, eold = etot;
          btot += BSPR(SP_B * SP + j - 1);
          etot += BSPR(SP_E * SP + j);
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
        orient(0);
        if (pending) {
          ri = regs[2 * e]; Ld = regs[2 * e + 1] - ri + 1; eseq = seq + (ri - 1);
          const float keep_scale = attempt == 0 ? kKeepScaleB : -1.0f;
          TransTab<Q, false> T;
          T.load(nullptr, trL, lane);
          const ScanC sc = scan_prepare(lane_product<Q, false>(T, FW_D2));
          forward_sweep<Q, false, true>(T, sc, emL, emG, Klds, eseq, Ld, cu, spec, SP, Fs, keep_scale, lane, xC_e, ef_e);
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          envsc = (float)((double)ef_e * LOG2 + log((double)(xC_e * cu.move)));
        }
        // P4: unihit Backward + posterior accumulation (backward tables)
        orient(1);
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
#pragma unroll 1
            for (int i = Ld; i >= 1; i--) {
              asm volatile("" ::: "memory");
              if (i < Ld) {
                mirror_scale<Q>(BSPRI(SP_S * SP + i + 1) - BSPRI(SP_S * SP + i), Mb, Ib, xJ, xC, xN);
                float od[Q];
                load_em_rev<Q>(od, emL, emG, eseq[i], Klds, lane);
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
              backward_cells<Q, false>(T, sc, Mb, Ib, xE);
              clamp_backward<Q>(Mb, Ib, xB, xJ, xC, xN);
              // mirrored scaling (wh_device.h, "envelope Backward scaling")
              const float s_i = invZe;
              const float s_p = ldexpf(invZe, BSPRI(SP_S * SP + i - 1) - BSPRI(SP_S * SP + i));
              const unsigned mword = src < 32 ? BSPRU(SP_ML * SP + i) : BSPRU(SP_MH * SP + i);
              if ((mword >> (src & 31)) & 1u) {
                const float4 *row = reinterpret_cast<const float4 *>(Fs) + (size_t)i * (2 * (Q / 4) * kWave) + src;
                float idot = 0.f;
#pragma unroll
                for (int p4 = 0; p4 < Q / 4; p4++) {
                  // reversed order: component 3-j of the forward-ordered vector is position 4*p4+j
                  const float4 fm = nt_load4(row + (Q / 4 - 1 - p4) * kWave);
                  const float4 fi = nt_load4(row + (Q / 4 + Q / 4 - 1 - p4) * kWave);
                  fM[4 * p4 + 0] = fmaf(fm.w * Mb[4 * p4 + 0], s_i, fM[4 * p4 + 0]);
                  fM[4 * p4 + 1] = fmaf(fm.z * Mb[4 * p4 + 1], s_i, fM[4 * p4 + 1]);
                  fM[4 * p4 + 2] = fmaf(fm.y * Mb[4 * p4 + 2], s_i, fM[4 * p4 + 2]);
                  fM[4 * p4 + 3] = fmaf(fm.x * Mb[4 * p4 + 3], s_i, fM[4 * p4 + 3]);
                  idot = fmaf(fi.w, Ib[4 * p4 + 0], idot); idot = fmaf(fi.z, Ib[4 * p4 + 1], idot);
                  idot = fmaf(fi.y, Ib[4 * p4 + 2], idot); idot = fmaf(fi.x, Ib[4 * p4 + 3], idot);
                }
                fIs = fmaf(idot, s_i, fIs);
              }
              float nj = BSPR(SP_N * SP + i - 1) * xN;
              nj = fmaf(BSPR(SP_J * SP + i - 1), xJ, nj);
              nj = fmaf(BSPR(SP_C * SP + i - 1), xC, nj);
              xfac = fmaf(nj * cu.loop, s_p, xfac);
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
      if (qi < q_hi && lane == 0) {
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

// 256 threads: one wave per SIMD with the whole register file; 512 threads: two waves per SIMD at 256
// registers each (the DP row of the longest models then spills to scratch)
template <int Q>
static hipError_t launch_big(const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  if (threads > 256) return launch_big_th<Q, 512>(a, blocks, threads, lds, s);
  return launch_big_th<Q, 256>(a, blocks, threads, lds, s);
}

hipError_t launch_score_big(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  switch (Q) {
    case 20: return launch_big<20>(a, blocks, threads, lds, s);
    case 24: return launch_big<24>(a, blocks, threads, lds, s);
    case 28: return launch_big<28>(a, blocks, threads, lds, s);
    case 32: return launch_big<32>(a, blocks, threads, lds, s);
    case 36: return launch_big<36>(a, blocks, threads, lds, s);
    case 40: return launch_big<40>(a, blocks, threads, lds, s);
    case 44: return launch_big<44>(a, blocks, threads, lds, s);
    case 48: return launch_big<48>(a, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace wh
