// VALU operand-bank / instruction-mix microbenchmark for gfx950 (hard-coded registers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define REP8(x) x x x x x x x x

// MODE 0: fmac, all three operands in the same bank (reg % 4 equal)
// MODE 1: fmac, operands in three different banks
// MODE 2: fma VOP3 (8-byte encoding), different banks
// MODE 3: MODE 1 + one ds_read_b128 per 8 VALU
// MODE 4: mul with 2 operands same bank
// MODE 5: mul different banks
// MODE 6: fmac, src0 varies over many registers (distinct), different banks
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
  __shared__ float4 lds[1024];
  lds[threadIdx.x] = make_float4(1.f, 1.f, 1.f, 1.f);
  __syncthreads();
  const unsigned addr = (threadIdx.x & 63) * 16;
  float r = 0.f;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
      asm volatile(REP8("v_fmac_f32 v8, v12, v16\n v_fmac_f32 v20, v24, v28\n v_fmac_f32 v32, v36, v40\n v_fmac_f32 v44, v48, v52\n"
                        "v_fmac_f32 v9, v13, v17\n v_fmac_f32 v21, v25, v29\n v_fmac_f32 v33, v37, v41\n v_fmac_f32 v45, v49, v53\n")
                   ::: "v8","v9","v20","v21","v32","v33","v44","v45","v12","v13","v16","v17","v24","v25","v28","v29","v36","v37","v40","v41","v48","v49","v52","v53");
    } else if (MODE == 1) {
      asm volatile(REP8("v_fmac_f32 v8, v13, v18\n v_fmac_f32 v20, v25, v30\n v_fmac_f32 v32, v37, v42\n v_fmac_f32 v44, v49, v54\n"
                        "v_fmac_f32 v9, v14, v19\n v_fmac_f32 v21, v26, v31\n v_fmac_f32 v33, v38, v43\n v_fmac_f32 v45, v50, v55\n")
                   ::: "v8","v9","v20","v21","v32","v33","v44","v45","v13","v14","v18","v19","v25","v26","v30","v31","v37","v38","v42","v43","v49","v50","v54","v55");
    } else if (MODE == 2) {
      asm volatile(REP8("v_fma_f32 v8, v13, v18, v8\n v_fma_f32 v20, v25, v30, v20\n v_fma_f32 v32, v37, v42, v32\n v_fma_f32 v44, v49, v54, v44\n"
                        "v_fma_f32 v9, v14, v19, v9\n v_fma_f32 v21, v26, v31, v21\n v_fma_f32 v33, v38, v43, v33\n v_fma_f32 v45, v50, v55, v45\n")
                   ::: "v8","v9","v20","v21","v32","v33","v44","v45","v13","v14","v18","v19","v25","v26","v30","v31","v37","v38","v42","v43","v49","v50","v54","v55");
    } else if (MODE == 3) {
      asm volatile(REP8("v_fmac_f32 v8, v13, v18\n v_fmac_f32 v20, v25, v30\n v_fmac_f32 v32, v37, v42\n v_fmac_f32 v44, v49, v54\n"
                        "ds_read_b128 v[60:63], %0\n"
                        "v_fmac_f32 v9, v14, v19\n v_fmac_f32 v21, v26, v31\n v_fmac_f32 v33, v38, v43\n v_fmac_f32 v45, v50, v55\n")
                   "s_waitcnt lgkmcnt(0)\n"
                   :: "v"(addr) : "v8","v9","v20","v21","v32","v33","v44","v45","v13","v14","v18","v19","v25","v26","v30","v31","v37","v38","v42","v43","v49","v50","v54","v55","v60","v61","v62","v63");
    } else if (MODE == 4) {
      asm volatile(REP8("v_mul_f32 v8, v12, v16\n v_mul_f32 v20, v24, v28\n v_mul_f32 v32, v36, v40\n v_mul_f32 v44, v48, v52\n"
                        "v_mul_f32 v9, v13, v17\n v_mul_f32 v21, v25, v29\n v_mul_f32 v33, v37, v41\n v_mul_f32 v45, v49, v53\n")
                   ::: "v8","v9","v20","v21","v32","v33","v44","v45","v12","v13","v16","v17","v24","v25","v28","v29","v36","v37","v40","v41","v48","v49","v52","v53");
    } else if (MODE == 5) {
      asm volatile(REP8("v_mul_f32 v8, v13, v18\n v_mul_f32 v20, v25, v30\n v_mul_f32 v32, v37, v42\n v_mul_f32 v44, v49, v54\n"
                        "v_mul_f32 v9, v14, v19\n v_mul_f32 v21, v26, v31\n v_mul_f32 v33, v38, v43\n v_mul_f32 v45, v50, v55\n")
                   ::: "v8","v9","v20","v21","v32","v33","v44","v45","v13","v14","v18","v19","v25","v26","v30","v31","v37","v38","v42","v43","v49","v50","v54","v55");
    } else if (MODE == 6) {
      // dependent chain inside one wave with a far-apart register set (like the DP row)
      asm volatile(REP8("v_fmac_f32 v8, v61, v114\n v_fmac_f32 v20, v65, v118\n v_fmac_f32 v32, v69, v122\n v_fmac_f32 v44, v73, v126\n"
                        "v_fmac_f32 v9, v62, v115\n v_fmac_f32 v21, v66, v119\n v_fmac_f32 v33, v70, v123\n v_fmac_f32 v45, v74, v127\n")
                   ::: "v8","v9","v20","v21","v32","v33","v44","v45","v61","v62","v65","v66","v69","v70","v73","v74","v114","v115","v118","v119","v122","v123","v126","v127");
    }
  }
  asm volatile("v_mov_b32 %0, v8" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int threads, int blocks) {
  float *out;
  hipMalloc(&out, 4 << 20);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double valu_per_wave = (double)iters * 64;
  const double waves = (double)blocks * threads / 64;
  printf("%-34s waves/SIMD=%d  %8.3f ms  %.3f G VALU instr/s/SIMD\n", name, threads / 256, ms, valu_per_wave * waves / (ms * 1e-3) / 1024.0 * 1e-9);
  hipFree(out);
}

int main() {
  for (int t : {256, 512, 1024}) {
    run<0>("fmac same-bank", t, 256);
    run<1>("fmac 3 banks", t, 256);
    run<2>("fma(VOP3) 3 banks", t, 256);
    run<3>("fmac 3 banks + ds_read_b128/8", t, 256);
    run<4>("mul same-bank", t, 256);
    run<5>("mul 2 banks", t, 256);
    run<6>("fmac far registers", t, 256);
  }
  return 0;
}
