// Alignment kernel: what "hmmalign -o OUT HMM QUERY" + the reference's Stockholm decode
// produce for one (query, HMM) pair (witch_msa/gcmm/aligner.py:96-142; algorithm:
// SURVEY.md A.7): unihit-local Forward, Backward, posterior decoding, optimal-accuracy
// (MEA) fill and traceback.  Output per residue: 0-based match column, or -1 when the
// residue sits in an insert state or in the N/C flanks.
//
// One wavefront per pair.  The Forward rows, then the posteriors (in place), then the
// OA rows live in per-wave HBM slabs - this kernel is the HBM-bound one:
//   F write 8 + read 8, posterior write 8 + read 8, OA write 12 bytes per DP cell.
#include <hip/hip_runtime.h>

#include "wh_device.h"
#include "wh_align_log.h"
#include "wh_launch.h"

namespace wh {

enum { AL_PN = 0, AL_B, AL_E, AL_PJ, AL_PC, AL_S, AL_ML, AL_MH, AL_ON, AL_OB, AL_OE, AL_OJ, AL_OC, AL_NARR };

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

      // ---------------- unihit Forward, rows spilled to slab A
      float xC_L = 0.f, lZ = -INFINITY; int ef_L = 0;
      orient(0);
      if (active) {
        TransTab<Q, TREG> T;
        T.load(fwG, trF, lane);
        if constexpr (LOGSP) {
          lZ = forward_sweep_log<Q>(T, emL, emG, Klds, seq, L, cu, spec, SP, slabA, lane);
          xC_L = lZ > -INFINITY ? 1.f : 0.f;
        } else {
          const ScanC sc = scan_prepare(lane_product<Q, TREG>(T, FW_D2));
          // forward_sweep uses spec slots 0..5 = N,B,E,J,C,S with stride SP (AL_PN..AL_S coincide)
          forward_sweep<Q, TREG, true>(T, sc, emL, emG, Klds, seq, L, cu, spec, SP, slabA, -1.0f, lane, xC_L, ef_L);   // dense
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      if (!(xC_L > 0.f)) active = false;   // no alignment has non-zero probability: all residues stay -1

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

      // ---------------- traceback: first maximum wins, candidate orders as in SURVEY.md A.7
      if (active) {
        enum { stS, stN, stB, stM, stI, stD, stE, stJ, stC };
        int s0 = stC, s1 = stS, i = L, k = 0;
        int guard = 4 * (L + M) + 16;
        const int Qh = (M - 1) / 4 + 1 < 2 ? 2 : (M - 1) / 4 + 1;   // HMMER's SSE stripe count
        // row 0 of the OA matrix is -inf (never stored): handled by the i == 1 tests below
        while (s0 != stS && guard-- > 0) {
          switch (s0) {
            case stC: {
              const float av = tNl * (SPR(AL_OC * SP + i - 1) + SPR(AL_PC * SP + i)), bv = tEC * SPR(AL_OE * SP + i);
              s1 = bv > av ? stE : stC;
              break;
            }
            case stJ: {
              const float av = tNl * (SPR(AL_OJ * SP + i - 1) + SPR(AL_PJ * SP + i)), bv = tEJ * SPR(AL_OE * SP + i);
              s1 = bv > av ? stE : stJ;
              break;
            }
            case stE: {
              // argmax over M (">=": the later cell in HMMER's striped scan wins) and D (">")
              const float4 *orow = reinterpret_cast<const float4 *>(slabB) + (size_t)i * (3 * Q4 * kWave) + lane;
              float vmax = -INFINITY;
              float om[Q], odd[Q];
#pragma unroll
              for (int q4 = 0; q4 < Q4; q4++) {
                const float4 m4 = nt_load4(orow + q4 * kWave), d4 = nt_load4(orow + (2 * Q4 + q4) * kWave);
                om[4 * q4] = m4.x; om[4 * q4 + 1] = m4.y; om[4 * q4 + 2] = m4.z; om[4 * q4 + 3] = m4.w;
                odd[4 * q4] = d4.x; odd[4 * q4 + 1] = d4.y; odd[4 * q4 + 2] = d4.z; odd[4 * q4 + 3] = d4.w;
              }
#pragma unroll
              for (int q = 0; q < Q; q++)
                if (lane * Q + q < M) vmax = fmaxf(vmax, fmaxf(om[q], odd[q]));
              vmax = wave_max(vmax);
              int bestM = -1, bestD = -1;
#pragma unroll
              for (int q = 0; q < Q; q++) {
                const int kk = lane * Q + q + 1;
                if (kk <= M) {
                  const int qh = (kk - 1) % Qh, rh = (kk - 1) / Qh;
                  if (om[q] == vmax) { const int pos = qh * 8 + rh; bestM = pos > bestM ? pos : bestM; }
                  if (odd[q] == vmax) { const int pos = 0x3FFFFFFF - (qh * 8 + 4 + rh); bestD = pos > bestD ? pos : bestD; }
                }
              }
              bestM = wave_max_i32(bestM);
              bestD = wave_max_i32(bestD);
              // a D cell wins only if it comes before every tied M cell AND no tied M follows it;
              // M uses ">=", so any tied M scanned after the D takes over: M wins whenever one exists
              // after the first tied D, or when the first tied cell is an M.
              int pos;
              if (bestM >= 0) { pos = bestM; s1 = stM; }
              else { pos = 0x3FFFFFFF - bestD; s1 = stD; }
              k = (pos % 8 % 4) * Qh + pos / 8 + 1;
              break;
            }
            case stM: {
              float path[4];
              path[0] = gate(tab_load<Q>(fwG, FW_E, k), SPR(AL_OB * SP + i - 1));
              if (i > 1 && k > 1) {
                path[1] = gate(tab_load<Q>(fwG, FW_A, k), cell_load<Q>(slabB, i - 1, 3, 0, k - 1));
                path[2] = gate(tab_load<Q>(fwG, FW_B, k), cell_load<Q>(slabB, i - 1, 3, 1, k - 1));
                path[3] = gate(tab_load<Q>(fwG, FW_C, k), cell_load<Q>(slabB, i - 1, 3, 2, k - 1));
              } else if (k > 1) {   // previous row is row 0: -inf behind an open gate, 0 behind a closed one
                path[1] = gate(tab_load<Q>(fwG, FW_A, k), -INFINITY);
                path[2] = gate(tab_load<Q>(fwG, FW_B, k), -INFINITY);
                path[3] = gate(tab_load<Q>(fwG, FW_C, k), -INFINITY);
              } else { path[1] = 0.f; path[2] = 0.f; path[3] = 0.f; }
              int best = 0;
              if (path[1] > path[best]) best = 1;
              if (path[2] > path[best]) best = 2;
              if (path[3] > path[best]) best = 3;
              s1 = best == 0 ? stB : best == 1 ? stM : best == 2 ? stI : stD;
              if (lane == 0) cols[i - 1] = k - 1;
              k--; i--;
              break;
            }
            case stD: {
              const float av = k > 1 ? gate(tab_load<Q>(fwG, FW_D1, k), cell_load<Q>(slabB, i, 3, 0, k - 1)) : 0.f;
              const float bv = k > 1 ? gate(tab_load<Q>(fwG, FW_D2, k), cell_load<Q>(slabB, i, 3, 2, k - 1)) : 0.f;
              s1 = bv > av ? stD : stM;
              k--;
              break;
            }
            case stI: {
              const float pmv = i > 1 ? cell_load<Q>(slabB, i - 1, 3, 0, k) : -INFINITY;
              const float piv = i > 1 ? cell_load<Q>(slabB, i - 1, 3, 1, k) : -INFINITY;
              const float av = gate(tab_load<Q>(fwG, FW_MI, k), pmv), bv = gate(tab_load<Q>(fwG, FW_II, k), piv);
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
