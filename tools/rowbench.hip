// Row-throughput harness for the DP sweeps of wh_device.h: every wave runs the Forward sweep
// (optionally with the sparse spill) or the Backward row core over synthetic tables, so that
// variants (tables in LDS / VGPRs, waves per SIMD, compiler flags) can be compared in isolation.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iwitch_amd/csrc -o tools/rowbench tools/rowbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wh_device6.h"
#include "wh_device2.h"
using namespace wh;

struct RArgs { const float *tables; float *scratch; float *out; int L, reps, K, SP, wave_lds; size_t scratch_stride; };

// MODE 0: forward no store, 1: forward with sparse store, 2: backward cells only
template <int Q, bool TREG, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void rb(RArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  constexpr int TBL = Q * kWave;
  float *emL = smem;
  float *trL = smem + (size_t)a.K * TBL;
  float *wbase = trL + (TREG ? 0 : FW_NARR * TBL) + (size_t)wave * a.wave_lds;
  float *spec = wbase;
  uint8_t *seq = reinterpret_cast<uint8_t *>(wbase + SP_NARR * a.SP);
  const float *emG = a.tables, *fwG = a.tables + (size_t)a.K * TBL;
  for (int t = threadIdx.x; t < a.K * TBL; t += blockDim.x) emL[t] = emG[t];
  if (!TREG) for (int t = threadIdx.x; t < FW_NARR * TBL; t += blockDim.x) trL[t] = fwG[t];
  for (int t = lane; t < a.L; t += kWave) seq[t] = (uint8_t)((t * 7 + wave + blockIdx.x) % a.K);
  __syncthreads();
  float *Fs = a.scratch + ((size_t)blockIdx.x * nwaves + wave) * a.scratch_stride;
  float acc = 0.f;
  const LenCfg cm = len_config(a.L, true);
  for (int r = 0; r < a.reps; r++) {
    TransTab<Q, TREG> T;
    T.load(fwG, trL, lane);
    if constexpr (MODE == 5) {
      const ScanC sc = scan_prepare(lane_product<Q, TREG>(T, FW_D2));
      Prob p0, p1;
      p0.seq = seq; p0.L = a.L; p0.spec = spec; p0.Fs = Fs;
      p1.seq = seq + 1; p1.L = a.L - 1; p1.spec = spec + SP_NARR * a.SP + (a.L + 3) / 4 + 4; p1.Fs = Fs;
      v2f xC; v2i ef;
      forward_sweep2<Q, false>(T, sc, emL, emG, a.K, p0, p1, len_config2(a.L, a.L - 1, true), a.SP, 9.094947e-13f, lane, xC, ef);
      acc += xC.x + xC.y + ef.x + ef.y;
    } else if constexpr (MODE == 3 || MODE == 4) {
      const ScanC sc = scan_prepare(lane_product<Q, TREG>(T, FW_D2));
      float xC; int ef;
      forward_sweep_r<Q, MODE == 4>(T, sc, emL, emG, a.K, seq, a.L, cm, spec, a.SP, Fs, 9.094947e-13f, lane, xC, ef);
      acc += xC + ef;
    } else if constexpr (MODE <= 1) {
      const ScanC sc = scan_prepare(lane_product<Q, TREG>(T, FW_D2));
      float xC; int ef;
      forward_sweep<Q, TREG, MODE == 1>(T, sc, emL, emG, a.K, seq, a.L, cm, spec, a.SP, Fs, 9.094947e-13f, lane, xC, ef);
      acc += xC + ef;
    } else {
      const ScanC sc = scan_prepare(lane_product<Q, TREG>(T, BW_DD));
      float Mb[Q], Ib[Q];
#pragma unroll
      for (int p = 0; p < Q; p++) { Mb[p] = 1e-3f * (lane + p); Ib[p] = 1e-3f; }
#pragma unroll 1
      for (int i = a.L; i >= 1; i--) {
        asm volatile("" ::: "memory");
        float od[Q];
        load_em_rev<Q>(od, emL, emG, seq[i - 1], a.K, lane);
        float part = 0.f;
#pragma unroll
        for (int p4 = 0; p4 < Q / 4; p4++) {
          const float4 E = T.ld(BW_E, p4);
#pragma unroll
          for (int j = 0; j < 4; j++) { const int p = 4 * p4 + j; Mb[p] *= od[p]; part = fmaf(f4get(E, j), Mb[p], part); }
        }
        const float xB = wave_sum(part);
        backward_cells<Q, TREG>(T, sc, Mb, Ib, xB * 0.01f);
        if (xB > kRescaleHi) {
          const float rr = pow2f_int(-f32_exponent(xB));
#pragma unroll
          for (int p = 0; p < Q; p++) { Mb[p] *= rr; Ib[p] *= rr; }
        }
      }
      acc += Mb[0] + Ib[Q - 1];
    }
  }
  a.out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int Q, bool TREG, int MODE, int THREADS>
void run(const char *name, int L, int reps) {
  const int K = 4, blocks = 256;
  constexpr int TBL = Q * kWave;
  std::vector<float> h((size_t)(K + FW_NARR) * TBL);
  srand(1);
  for (int x = 0; x < K; x++) for (int t = 0; t < TBL; t++) h[(size_t)x * TBL + t] = 0.5f + (rand() % 1000) * 1e-3f;
  const float base[FW_NARR] = {0.9f, 0.4f, 0.4f, 1e-3f, 0.05f, 0.5f, 0.05f, 0.5f};
  for (int a = 0; a < FW_NARR; a++) for (int t = 0; t < TBL; t++) h[(size_t)(K + a) * TBL + t] = base[a] * (0.9f + (rand() % 100) * 1e-3f);
  float *d_t, *d_s, *d_o;
  hipMalloc(&d_t, h.size() * 4); hipMemcpy(d_t, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  RArgs a; a.tables = d_t; a.L = L; a.reps = reps; a.K = K; a.SP = (L + 1 + 3) / 4 * 4;
  a.wave_lds = (MODE == 5 ? 2 : 1) * (SP_NARR * a.SP + (L + 3) / 4 + 4);
  const int waves = THREADS / 64;
  a.scratch_stride = (size_t)(L + 1) * 2 * Q * kWave;
  hipMalloc(&d_s, (size_t)blocks * waves * a.scratch_stride * 4);
  hipMalloc(&d_o, (size_t)blocks * THREADS * 4);
  a.scratch = d_s; a.out = d_o;
  const size_t lds = ((size_t)K * TBL + (TREG ? 0 : FW_NARR * TBL) + (size_t)waves * a.wave_lds) * 4;
  if (lds > 160 * 1024) { printf("%-28s Q=%d waves=%d: LDS %zu too large\n", name, Q, waves, lds); return; }
  hipFuncSetAttribute(reinterpret_cast<const void *>(&rb<Q, TREG, MODE, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  a.reps = 1;
  hipLaunchKernelGGL((rb<Q, TREG, MODE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, a);
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) { printf("%-28s launch failed: %s\n", name, hipGetErrorString(err)); return; }
  a.reps = reps;
  hipEventRecord(e0);
  hipLaunchKernelGGL((rb<Q, TREG, MODE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, a);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double rows = (double)blocks * waves * reps * L * (MODE == 5 ? 2 : 1);
  printf("%-28s Q=%2d treg=%d waves/CU=%2d  %8.2f ms  %7.1f ns per wave-row  %.3f Grow/s chip   (per-CU row time %.1f ns)\n", name, Q, (int)TREG, waves,
         ms, ms * 1e6 / ((double)reps * L), rows / (ms * 1e-3) * 1e-9, ms * 1e6 / ((double)reps * L) / waves / (MODE == 5 ? 2 : 1));
  hipFree(d_t); hipFree(d_s); hipFree(d_o);
}

int main() {
  const int L = 150, reps = 40;
  run<16, false, 0, 256>("fwd", L, reps);
  run<16, false, 0, 512>("fwd", L, reps);
  run<16, false, 0, 768>("fwd", L, reps);
  run<16, false, 0, 1024>("fwd", L, reps);
  run<16, true, 0, 256>("fwd", L, reps);
  run<16, true, 0, 512>("fwd", L, reps);
  run<16, false, 5, 256>("fwd2 (2 problems/wave)", L, reps);
  run<16, false, 5, 512>("fwd2 (2 problems/wave)", L, reps);
  run<12, false, 5, 512>("fwd2 (2 problems/wave)", L, reps);
  run<12, false, 5, 768>("fwd2 (2 problems/wave)", L, reps);
  run<16, true, 3, 256>("fwd_r", L, reps);
  run<16, true, 3, 512>("fwd_r", L, reps);
  run<12, true, 3, 512>("fwd_r", L, reps);
  run<12, true, 0, 512>("fwd", L, reps);
  run<12, false, 0, 512>("fwd", L, reps);
  run<16, false, 1, 512>("fwd+store", L, reps);
  run<16, false, 1, 768>("fwd+store", L, reps);
  run<16, true, 1, 512>("fwd+store", L, reps);
  run<16, false, 2, 512>("bwd cells", L, reps);
  run<16, false, 2, 768>("bwd cells", L, reps);
  run<16, false, 2, 1024>("bwd cells", L, reps);
  run<16, true, 2, 512>("bwd cells", L, reps);
  return 0;
}
