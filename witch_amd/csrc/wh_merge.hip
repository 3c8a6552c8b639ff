// Final transitive merge on the device (reference: witch_msa/gcmm/merger.py:40-131,
// helpers/alignment_tools.py:1183-1316 merge_in, 1356-1384 compressInsertions, 1140-1156 masked writer).
//
// The reference merges the per-query alignments into the backbone one at a time, splicing every new run of
// insertion columns into every row collected so far.  What that loop computes has a closed form
// (witch_amd/gcmm/merger.py states and tests it against the reference's own output): gap g in front of
// backbone column g (g = B: after the last) is as wide as the LONGEST insertion run any query has there, every
// query's run is left-justified in it, everything else is '-'.  Here it is computed from the consensus
// kernel's per-residue codes (code >= 0: backbone column, uppercase; code = -1 - g: insertion in gap g,
// lowercase) without building the per-query strings on the host:
//   merge_runs_kernel    one thread per query: compressInsertions (insertions in front of the first / after
//                        the last aligned residue belong to gap 0 / gap B), run lengths -> atomicMax into W[g],
//                        per residue its gap and its position inside the run;
//   merge_layout_kernel  one workgroup: exclusive scan of W -> first column of every gap, column of every
//                        backbone column, total width;
//   merge_render_kernel  one workgroup per output row: the backbone rows, then the queries' rows, into the
//                        full matrix and the masked matrix (backbone columns only).
// HBM-bound byte work: (rows x width) bytes written once; nothing to do with MFMA.
#include <hip/hip_runtime.h>

#include "wh_launch.h"

namespace wh {

__global__ void merge_runs_kernel(MergeArgs a) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.nq) return;
  if (a.q_row[q] == -2) return;                        // no alignment for this query (skipped / no weights)
  const int64_t lo = a.q_off[q], hi = a.q_off[q + 1];
  int64_t first_up = -1, last_up = -1;
  for (int64_t r = lo; r < hi; r++)
    if (a.codes[r] >= 0) { if (first_up < 0) first_up = r; last_up = r; }
  int prev_gap = -1, k = 0;
  for (int64_t r = lo; r < hi; r++) {
    const int c = a.codes[r];
    if (c >= 0) { a.res_gap[r] = -1; a.res_k[r] = 0; continue; }
    int g = -1 - c;
    if (first_up >= 0) { if (r < first_up) g = 0; else if (r > last_up) g = a.B; }
    if (g < 0) g = 0;
    if (g > a.B) g = a.B;
    if (g == prev_gap) k++;
    else {
      if (prev_gap >= 0) atomicMax(a.W + prev_gap, k);
      prev_gap = g; k = 1;
    }
    a.res_gap[r] = g; a.res_k[r] = k - 1;
  }
  if (prev_gap >= 0) atomicMax(a.W + prev_gap, k);
}

// gap_start[g] = sum_{g' < g} W[g'] + g ; col_pos[c] = gap_start[c] + W[c] ; layout[B+1 .. ] ; width
__global__ void merge_layout_kernel(MergeArgs a) {
  __shared__ long long part[256];
  const int t = threadIdx.x, n = a.B + 1;
  const int per = (n + 255) / 256, lo = t * per, hi = lo + per < n ? lo + per : n;
  long long s = 0;
  for (int g = lo; g < hi; g++) s += a.W[g];
  part[t] = s;
  __syncthreads();
  if (t == 0) { long long run = 0; for (int u = 0; u < 256; u++) { const long long v = part[u]; part[u] = run; run += v; } a.width[0] = run + a.B; }
  __syncthreads();
  long long run = part[t];
  for (int g = lo; g < hi; g++) {
    a.gap_start[g] = run + g;
    if (g < a.B) a.col_pos[g] = run + g + a.W[g];
    run += a.W[g];
  }
}

__global__ void merge_render_kernel(MergeArgs a) {
  const int64_t row = blockIdx.x;
  const int64_t width = a.width[0];
  uint8_t *full = a.out_full + (size_t)row * (size_t)width;
  uint8_t *mask = a.out_masked + (size_t)row * (size_t)a.B;
  for (int64_t c = threadIdx.x; c < width; c += blockDim.x) full[c] = '-';
  if (row >= a.nb) for (int c = threadIdx.x; c < a.B; c += blockDim.x) mask[c] = '-';
  __syncthreads();
  if (row < a.nb) {
    const uint8_t *src = a.bb + (size_t)row * (size_t)a.B;
    for (int c = threadIdx.x; c < a.B; c += blockDim.x) { const uint8_t ch = src[c]; full[a.col_pos[c]] = ch; mask[c] = ch; }
    return;
  }
  const int64_t q = a.row_q[row - a.nb];
  const int64_t lo = a.q_off[q], hi = a.q_off[q + 1];
  for (int64_t r = lo + threadIdx.x; r < hi; r += blockDim.x) {
    uint8_t ch = a.q_text[r];
    const bool alpha = (ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z');
    const int c = a.codes[r];
    if (c >= 0) {
      if (alpha) ch &= 0xDF;
      if (c < a.B) { full[a.col_pos[c]] = ch; mask[c] = ch; }
    } else {
      if (alpha) ch |= 32;
      full[a.gap_start[a.res_gap[r]] + a.res_k[r]] = ch;
    }
  }
}

hipError_t launch_merge_runs(const MergeArgs &a, hipStream_t s) {
  if (a.nq > 0) hipLaunchKernelGGL(merge_runs_kernel, dim3((unsigned)((a.nq + 255) / 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(merge_layout_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_merge_layout(const MergeArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(merge_layout_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_merge_render(const MergeArgs &a, int64_t nrows, hipStream_t s) {
  if (nrows > 0) hipLaunchKernelGGL(merge_render_kernel, dim3((unsigned)nrows), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace wh
