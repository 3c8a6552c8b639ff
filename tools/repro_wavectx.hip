// Compile-only reproducer of the calling-convention cliff behind the phase-call scoring kernel's WaveCtx rule
// (witch_amd/csrc/wh_score7.hip: "The context travels to the non-inlined sweeps in argument registers while it
// flattens to at most 16 dwords").  hipcc (ROCm 7.2, gfx950) passes a struct argument of a __noinline__ device
// function in argument registers up to 16 dwords; one dword more and the CALLER spills the whole struct to scratch
// and passes a pointer, the CALLEE reads every member back with scratch_load.  In the real kernel (address-space
// typed LDS/HBM pointers inside the struct, 168-VGPR budget, five call sites) the by-reference variant died at run
// time with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION; this file only shows the codegen difference, which is what
// tests/test_abi_host.py::test_sweep_context_is_register_passed guards.  It deliberately does NOT try to fault: a
// faulting kernel can take a shared 8-GPU host down.
//
//   tools/repro_wavectx.sh          # compiles both variants to ISA and counts scratch accesses in the callee
//
// Observed (ROCm 7.2.0):  callee<0> (16 dwords): 0 scratch_load    callee<1> (17 dwords): 6 scratch_load (wide loads of the byval copy)
#include <hip/hip_runtime.h>

typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(1))) float glb_f;

template <int EXTRA>
struct Ctx {
  lds_f *a, *b, *c, *d, *e;      // 5 dwords (LDS pointers are 32-bit)
  const glb_f *g;                // 2
  glb_f *h, *s;                  // 4
  int sp, lane, alpha;           // 3
  unsigned degen;                // 1
  int pad0;                      // 1  -> 16 dwords
  int extra[EXTRA];              // the 17th dword (EXTRA = 1)
};
template <>
struct Ctx<0> {
  lds_f *a, *b, *c, *d, *e;
  const glb_f *g;
  glb_f *h, *s;
  int sp, lane, alpha;
  unsigned degen;
  int pad0;
};

template <int EXTRA>
__device__ __noinline__ float callee(const Ctx<EXTRA> c, int n) {
  float acc = 0.f;
  for (int i = 0; i < n; i++) acc += c.a[c.lane + i * c.sp] * c.g[c.lane + i] + c.b[i] + c.c[i] + c.d[i] + c.e[i] + c.h[i] + c.s[i] + (float)(c.alpha + (int)c.degen + c.pad0);
  return acc;
}

template <int EXTRA>
__global__ void kern(const float *g, float *h, float *out, int n) {
  extern __shared__ float sm[];
  Ctx<EXTRA> c;
  c.a = (lds_f *)sm; c.b = c.a + 64; c.c = c.b + 64; c.d = c.c + 64; c.e = c.d + 64;
  c.g = (const glb_f *)g; c.h = (glb_f *)h; c.s = (glb_f *)h + 64;
  c.sp = 64; c.lane = threadIdx.x; c.alpha = n; c.degen = 3u; c.pad0 = 1;
  if constexpr (EXTRA > 0) c.extra[0] = n;
  out[threadIdx.x] = callee<EXTRA>(c, n) + callee<EXTRA>(c, n / 2);
}

template __global__ void kern<0>(const float *, float *, float *, int);
template __global__ void kern<1>(const float *, float *, float *, int);
