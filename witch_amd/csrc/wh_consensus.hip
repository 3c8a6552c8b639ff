// Weighted consensus of one query's per-HMM alignments (witch-ng merge DP).
//
// Replaces the pure-Python loops of alignSubQueriesNew (witch_msa/gcmm/aligner.py:376-473):
//   * edge weights   combined[(i, j)] += nongaps[h][c] * w_h, accumulated in top-k order   (:399-418)
//   * max-weight trace DP over (len(seq)+1) x (max_col-min_col+2) cells, float64          (:426-448)
//       value  = max(0, diag + cw [only if cw > 0], up, left)   first maximum wins in that order
//   * traceback from (len, max_col+1)                                                      (:452-473)
// All arithmetic is IEEE float64 in the reference's operation order, so results are
// bit-identical to the reference's Python floats.  One wavefront per query: lanes run over
// backbone columns; the "left" dependency is an exact prefix-max scan; the DP row lives in LDS (in HBM
// for backbones wider than ~19 000 columns), the 2-bit back-pointers in a per-wave HBM slab.
// Output per residue: backbone column (>= 0) for a match, -1 - nc for an insertion that sits
// before backbone column nc; the host rebuilds the reference's string from it.
#include <hip/hip_runtime.h>

#include "wh_launch.h"

namespace wh {

__device__ __forceinline__ double shfl_up_f64(double v, int d) {
  int lo = __shfl_up((int)(__double_as_longlong(v) & 0xFFFFFFFFll), d);
  int hi = __shfl_up((int)(__double_as_longlong(v) >> 32), d);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) & 0xFFFFFFFFll), l);
  int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), l);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int wave_min_i32(int x) { for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(x, m); x = o < x ? o : x; } return x; }
__device__ __forceinline__ int wave_max_i32c(int x) { for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(x, m); x = o > x ? o : x; } return x; }

__global__ __launch_bounds__(256) void consensus_kernel(ConsArgs a) {
  extern __shared__ __attribute__((aligned(16))) double rows_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const size_t wid = (size_t)blockIdx.x * nwaves + wave;
  // the DP row: in LDS while a backbone row fits there (up to ~19 000 columns), else in the wave's HBM region - a lane
  // re-reads only what it wrote itself (column jj stays with lane (jj - 1) % 64), so no ordering beyond program order is needed
  double *row = a.rowg ? a.rowg + wid * (size_t)(a.Wcap + 2) : rows_raw + (size_t)wave * (a.Wcap + 2);
  uint8_t *back = a.back + wid * (size_t)(a.Lcap + 1) * (a.Wcap + 2);
  int32_t *cwj = a.cwj + wid * (size_t)a.Lcap * a.KMAX;
  double *cwv = a.cwv + wid * (size_t)a.Lcap * a.KMAX;
  int32_t *cwn = a.cwn + wid * (size_t)a.Lcap;

  for (;;) {
    long long q = 0;
    if (lane == 0) q = (long long)atomicAdd(a.counter, 1);
    q = __shfl((int)q, 0);
    if (q >= a.nq) break;
    const int64_t off = a.offsets[q];
    const int L = (int)(a.offsets[q + 1] - off);
    const int64_t p_lo = a.qpair_off[q], p_hi = a.qpair_off[q + 1];
    int32_t *out = a.out + off;
    // ---- edge weights (lanes over residues), min/max touched backbone column
    int mn = a.backbone_length + 1, mx = -1;
    for (int i = lane; i < L; i += 64) {
      int cnt = 0;
      for (int64_t p = p_lo; p < p_hi; p++) {
        const int c = a.cols[a.col_offsets[p] + i];
        if (c < 0) continue;
        const int h = a.pair_h[p];
        const int64_t ro = a.ret_off[h] + c;
        const int j = a.retained[ro];
        const double add = (double)a.nongaps[ro] * a.pair_w[p];
        int e = 0;
        for (; e < cnt; e++) if (cwj[(size_t)i * a.KMAX + e] == j) break;
        if (e < cnt) cwv[(size_t)i * a.KMAX + e] += add;
        else if (cnt < a.KMAX) { cwj[(size_t)i * a.KMAX + cnt] = j; cwv[(size_t)i * a.KMAX + cnt] = 0.0 + add; cnt++; }
        mn = j < mn ? j : mn;
        mx = j > mx ? j : mx;
      }
      cwn[i] = cnt;
    }
    mn = wave_min_i32(mn);
    mx = wave_max_i32c(mx);
    if (lane == 0) { a.minmax[2 * q] = mn; a.minmax[2 * q + 1] = mx; }
    if (mx < 0 || L <= 0) {   // nothing aligned: every residue is an insertion before column 0
      for (int i = lane; i < L; i += 64) out[i] = -1;
      continue;
    }
    const int W = mx - mn + 1;            // DP columns jj = 1..W  <->  j = mn + jj
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (int jj = lane; jj <= W; jj += 64) row[jj] = 0.0;
    __builtin_amdgcn_wave_barrier();
    // ---- DP rows
    for (int i = 1; i <= L; i++) {
      const int cnt = __builtin_nontemporal_load(cwn + (i - 1));
      double carry_left = 0.0;     // value of this row at jj-1 (row[0] = 0: the j == min_col boundary)
      double carry_diag = 0.0;     // previous row at jj-1 for the first lane of the chunk (row[0] = 0)
      for (int c0 = 1; c0 <= W; c0 += 64) {
        const int jj = c0 + lane;
        const bool valid = jj <= W;
        const double up = valid ? row[jj] : 0.0;
        double diag = shfl_up_f64(up, 1);
        if (lane == 0) diag = carry_diag;
        carry_diag = readlane_f64(up, 63);
        double cw = 0.0;
        for (int e = 0; e < cnt; e++) {
          const int je = __builtin_nontemporal_load(cwj + (size_t)(i - 1) * a.KMAX + e);
          const double ve = __builtin_nontemporal_load(cwv + (size_t)(i - 1) * a.KMAX + e);
          if (je + 1 - mn == jj) cw = ve;
        }
        double cur = 0.0;
        int bt = 0;
        if (cw <= 0.0) bt = 1;
        else { const double v = diag + cw; if (v > cur) { cur = v; bt = 0; } }
        if (up > cur) { cur = up; bt = 1; }
        if (!valid) cur = 0.0;
        // exclusive prefix max of <cur> over the lanes, seeded with the row value left of the chunk
        double inc = cur;
        for (int d = 1; d < 64; d <<= 1) { const double o = shfl_up_f64(inc, d); if (lane >= d && o > inc) inc = o; }
        double left = shfl_up_f64(inc, 1);
        if (lane == 0) left = carry_left; else if (carry_left > left) left = carry_left;
        double val = cur;
        if (left > cur) { val = left; bt = 2; }
        if (valid) { row[jj] = val; back[(size_t)i * (a.Wcap + 2) + jj] = (uint8_t)bt; }
        const double last = readlane_f64(val, 63);
        carry_left = last;   // lanes beyond W hold the running maximum too, never read again
        __builtin_amdgcn_wave_barrier();
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // ---- traceback (uniform control flow; lane 0 writes)
    int i = L, jj = W;
    while (i > 0 && jj > 0) {
      const int bt = __builtin_nontemporal_load(back + (size_t)i * (a.Wcap + 2) + jj);
      if (bt == 0) { if (lane == 0) out[i - 1] = mn + jj - 1; i--; jj--; }
      else if (bt == 1) { if (lane == 0) out[i - 1] = -1 - (mn + jj); i--; }
      else jj--;
    }
    while (i > 0) { if (lane == 0) out[i - 1] = -1 - (mn + jj); i--; }
  }
}

hipError_t launch_consensus(const ConsArgs &a, int blocks, int threads, size_t lds, hipStream_t s) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&consensus_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(consensus_kernel, dim3(blocks), dim3(threads), lds, s, a);
  return hipGetLastError();
}

}  // namespace wh
