// Weights, deterministic top-k and the 0.999 prefix - one wavefront per query.
//
// Replaces (reference paths relative to the WITCH checkout):
//   readAndRankBitscoreMP  witch_msa/gcmm/loader.py:299-332   (rank reported HMMs by score)
//   calculateWeights       witch_msa/gcmm/weighting.py:58-74  (w_i = 1 / sum_j 2^((s_j-s_i)+log2(n_j/n_i)))
//   adaptive cut           witch_msa/gcmm/aligner.py:58-63    (prefix until cumulative weight >= 0.999)
// The reference's order among equal keys is the arrival order of pool futures; here ties are
// broken deterministically by (-weight, -decibits, +hmm_index).
// Exact weight ties (n_i 2^(d_i/10) == n_j 2^(d_j/10)) are detected in integers: write
// n = odd * 2^t, then w is a function of (odd, d + 10 t) only.
#include <hip/hip_runtime.h>

#include "wh_launch.h"

namespace wh {

struct Key {
  unsigned long long k1;   // order-preserving image of 10*log2(odd) + (d + 10 t)
  unsigned long long k2;   // (decibits + 2^31) << 32 | (0xFFFFFFFF - hmm_index)
};

__device__ __forceinline__ bool key_less(const Key &a, const Key &b) {
  return a.k1 < b.k1 || (a.k1 == b.k1 && a.k2 < b.k2);
}

__device__ __forceinline__ unsigned long long sortable(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  int lo = __shfl_xor((int)(v & 0xFFFFFFFFu), m), hi = __shfl_xor((int)(v >> 32), m);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

__device__ __forceinline__ double wave_sum_f64(double x) {
  for (int m = 32; m >= 1; m >>= 1) x += __longlong_as_double((long long)shfl_xor_u64((unsigned long long)__double_as_longlong(x), m));
  return x;
}

constexpr int kMaxPerLane = 16;   // H <= 1024

__global__ __launch_bounds__(256) void topk_kernel(TopkArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= a.nq) return;
  const int H = a.H;
  Key key[kMaxPerLane];
  bool alive[kMaxPerLane];
  int nrep = 0;
#pragma unroll
  for (int t = 0; t < kMaxPerLane; t++) {
    const int j = lane + t * 64;
    alive[t] = false;
    key[t].k1 = 0; key[t].k2 = 0;
    if (j < H && (a.flags[q * H + j] & WH_FLAG_REPORTED)) {
      const int d = a.decibits[q * H + j];
      unsigned n = (unsigned)a.nseq[j];
      if (n == 0) n = 1;
      const int tz = __ffs(n) - 1;
      const unsigned odd = n >> tz;
      const double v = 10.0 * log2((double)odd) + (double)(d + 10 * tz);
      key[t].k1 = sortable(v);
      key[t].k2 = ((unsigned long long)(unsigned)(d + 0x40000000) << 32) | (0xFFFFFFFFull - (unsigned)a.hmm_index[j]);
      alive[t] = true;
      nrep++;
    }
  }
  for (int m = 32; m >= 1; m >>= 1) nrep += __shfl_xor(nrep, m);
  const int nkeep = nrep < a.k ? nrep : a.k;
  double cum = 0.0;
  int nused = 0;
  for (int r = 0; r < a.k; r++) {
    int sel_j = -1;
    double w = 0.0;
    if (r < nkeep) {
      // wave-wide argmax of the composite key among the remaining candidates
      Key best; best.k1 = 0; best.k2 = 0;
      int bj = -1;
#pragma unroll
      for (int t = 0; t < kMaxPerLane; t++)
        if (alive[t] && (bj < 0 || key_less(best, key[t]))) { best = key[t]; bj = lane + t * 64; }
      for (int m = 32; m >= 1; m >>= 1) {
        Key o; o.k1 = shfl_xor_u64(best.k1, m); o.k2 = shfl_xor_u64(best.k2, m);
        const int oj = __shfl_xor(bj, m);
        if (oj >= 0 && (bj < 0 || key_less(best, o))) { best = o; bj = oj; }
      }
      sel_j = bj;
#pragma unroll
      for (int t = 0; t < kMaxPerLane; t++)
        if (lane + t * 64 == sel_j) alive[t] = false;
      // weight of the selected model with the reference's formula (weighting.py:64-69), float64
      const double s_i = (double)a.decibits[q * H + sel_j] / 10.0;
      const double n_i = (double)a.nseq[sel_j];
      double part = 0.0;
      for (int j = lane; j < H; j += 64) {
        if (a.flags[q * H + j] & WH_FLAG_REPORTED) {
          const double s_j = (double)a.decibits[q * H + j] / 10.0;
          const double ex = (s_j - s_i) + log2((double)a.nseq[j] / n_i);
          part += exp2(ex);
        }
      }
      w = 1.0 / wave_sum_f64(part);
      if (cum < 0.999) { cum += w; nused++; }   // aligner.py:58-63
    }
    if (lane == 0) {
      a.idx[q * a.k + r] = sel_j >= 0 ? a.hmm_index[sel_j] : -1;
      a.w[q * a.k + r] = w;
    }
  }
  if (lane == 0) {
    a.n_kept[q] = nkeep;
    a.n_used[q] = nused;
  }
}

hipError_t launch_topk(const TopkArgs &a, hipStream_t s) {
  if (a.H > 64 * kMaxPerLane) return hipErrorInvalidValue;
  const int waves = 4;
  const int blocks = (int)((a.nq + waves - 1) / waves);
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(topk_kernel, dim3(blocks), dim3(waves * 64), 0, s, a);
  return hipGetLastError();
}

}  // namespace wh
