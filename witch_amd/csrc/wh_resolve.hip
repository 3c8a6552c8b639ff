// Multidomain regions: HMMER's stochastic resolver on the device.
//
// `hmmsearch` (called at witch_msa/gcmm/algorithm.py:526-532) resolves a region whose posterior
// decoding suggests more than one domain by: multihit Forward on the region's sub-sequence with
// the full matrix; the pipeline's RNG re-seeded (Easel's "fast" generator, seed 42:
// x <- 69069 x + 1 mod 2^32 started from Jenkins' mix3); 200 stochastic tracebacks; every
// sampled domain becomes a segment pair (seq from/to, model from/to) and bumps per-residue null2
// accumulators (null2 by trace); single-linkage clustering of the segments (overlap >= 0.8 of the
// smaller in sequence and model, start OR end diagonals within 4), clusters present in >= 25 % of
// the traces survive, their end points are the outermost ones reached by >= 2 % of the cluster;
// clusters dominated by an overlapping more probable one are dropped; each survivor is an
// envelope rescored by unihit Forward, its null2 correction being the trace-derived one.
//
// The scoring kernels queue every pair that has such a region (ResolveRec: regions, the results of
// its single-domain regions, the Forward score); this kernel finishes those pairs, ONE WAVEFRONT
// PER PAIR: float64 Forward sweeps (generic in the model size: lane r owns nodes r*Q+1..r*Q+Q, the
// D->D chain is a cross-lane affine scan in double; the dense matrix lives in a per-wave HBM slab,
// double precision keeps the sampled choices identical to the float64 oracle's), the traces run
// wave-uniform (one random number per choice, strictly serial by construction of the generator),
// the E-state choice and the per-trace null2 / segment bookkeeping use the 64 lanes, the
// clustering reproduces Easel's vertex order exactly (parallel link tests, serial stack updates).
// -DWH_RESOLVE_DEBUG compiles the device printf dumps in (option WH_RDBG; tests/dbg/dbg_resolve.py):
// off by default, device printf alone costs the kernel half of its register budget.
#include <hip/hip_runtime.h>

#include "wh_launch.h"
#include "wh_f64.h"

namespace wh {

namespace {

constexpr int kSamples = 200;
constexpr int kDomMax = 32;          // domains per sampled trace kept (more: TRUNC)
constexpr int kSegCap = 8192;         // sampled segments per region kept, in HBM (200 traces x up to kDomMax domains = 6 400: never short)
constexpr int kSegLds = 2048;         // ... of them, the clustering keeps its two vertex stacks in LDS up to this many (beyond: in HBM)
constexpr int kClusMax = 64;          // significant clusters of ONE region kept (more: TRUNC)
constexpr int kHist = 64;            // decision fetches of a trace whose keys are remembered for the next trace
constexpr int kEnvMax = 16;          // envelopes per pair kept internally (detail reports WH_MAX_ENVELOPES)
enum { stM = 1, stD, stI, stN, stC, stJ, stE, stB, stS };

struct Rng { uint32_t x; };
__device__ __forceinline__ uint32_t mix3(uint32_t a, uint32_t b, uint32_t c) {
  a -= b; a -= c; a ^= (c >> 13);
  b -= c; b -= a; b ^= (a << 8);
  c -= a; c -= b; c ^= (b >> 13);
  a -= b; a -= c; a ^= (c >> 12);
  b -= c; b -= a; b ^= (a << 16);
  c -= a; c -= b; c ^= (b >> 5);
  a -= b; a -= c; a ^= (c >> 3);
  b -= c; b -= a; b ^= (a << 10);
  c -= a; c -= b; c ^= (b >> 15);
  return c;
}
__device__ __forceinline__ double rng_next(Rng &r) { r.x = r.x * 69069u + 1u; return (double)r.x / 4294967296.0; }
// esl_vec_FNorm + esl_rnd_FChoose: float probabilities, double running sum, first t with sum/norm > roll
__device__ __forceinline__ int rng_choose(Rng &r, const double *pd, int n) {
  float p[4];
  double tot = 0.0;
  for (int t = 0; t < n; t++) tot += pd[t];
  for (int t = 0; t < n; t++) p[t] = tot > 0.0 ? (float)(pd[t] / tot) : 1.0f / (float)n;
  const double roll = rng_next(r);
  double norm = 0.0, sum = 0.0;
  for (int t = 0; t < n; t++) norm += (double)p[t];
  for (int t = 0; t < n; t++) { sum += (double)p[t]; if (sum / norm > roll) return t; }
  return n - 1;
}

// seq overlap, model overlap, diagonal test of link_spsamples (p7_spensemble.c), see file header
__device__ __forceinline__ bool seg_linked(int i1, int j1, int k1, int m1, int i2, int j2, int k2, int m2) {
  int nov = min(j1, j2) - max(i1, i2) + 1;
  int n = min(j1 - i1 + 1, j2 - i2 + 1);
  if ((float)nov / (float)n < 0.8f) return false;
  nov = min(m1, m2) - max(k1, k2);
  n = min(m1 - k1 + 1, m2 - k2 + 1);
  if ((float)nov / (float)n < 0.8f) return false;
  if (abs((i1 - k1) - (i2 - k2)) <= 4) return true;
  if (abs((j1 - m1) - (j2 - m2)) <= 4) return true;
  return false;
}

}  // namespace

// LDS per wave (4-byte units).  What the traces use - the domains of the current trace, their null2 vectors, the emitting
// state per residue (int16) - and what the clustering uses afterwards - Easel's two vertex stacks (uint16) - are never
// alive together and share one block (round 4: 19 -> 11 KB per wave at 2 000-residue queries, which makes room for a
// model's float64 transition arrays beside eight waves).  The end-point histograms of the cluster statistics live in the
// wave's HBM slab.
__host__ __device__ inline size_t resolve_uni_ints(int Lcap) {
  const int Lp = (Lcap + 4) & ~1;
  const size_t traces = (size_t)kDomMax * (4 + 32) + Lp / 2 + 2, clustering = kSegLds;
  return ((traces > clustering ? traces : clustering) + 1) & ~(size_t)1;
}
__host__ __device__ inline size_t resolve_lds_ints(int Lcap, int Mmax) {
  (void)Mmax;
  return (((size_t)(Lcap + 8) / 4 + 2 + 1) & ~(size_t)1) /*seq*/ + resolve_uni_ints(Lcap) + 128 /*64 float64 bins of the E-state row pass*/ + 4 * kEnvMax + 3 * kClusMax + 16
         + 2 * 2 * kHist /*WH_STATS: fetch keys of the previous and the current trace*/;
}
// waves per SIMD the kernel is compiled for (registers per lane = 512 / WH_RES_OCC)
#ifndef WH_RES_OCC
#define WH_RES_OCC 2
#endif
size_t resolve_lds_bytes(int Lcap, int Mmax) { return resolve_lds_ints(Lcap, Mmax) * 4; }       // per WAVE
// per workgroup in front of the wave blocks: a 16-byte header (chunk number, chunk cursor) and, when the launch stages
// them, the eight float64 transition arrays of ONE model of up to <Qt> cells per lane
size_t resolve_lds_header_bytes(int Qt) { return 16 + (size_t)gNARR * Qt * 64 * sizeof(double); }
// per wave in HBM: the segment arrays, the end-point histogram, and the two per-residue float arrays (null2 scores of
// the pair, accumulators of the region): in LDS they cost 8 bytes per residue of the LONGEST query of the batch and
// halved the resident waves for 2 000-residue proteins
size_t resolve_seg_ints(int Lcap, int Mmax) { return (size_t)7 * kSegCap + (size_t)(Lcap > Mmax ? Lcap : Mmax) + 8 + 2 * ((size_t)Lcap + 8); }
int resolve_seg_cap() { return kSegCap; }
// the walk's cache of threshold lines: 2^kDcBits lines of 64 x 16 bytes + their 8-byte tags, in doubles
constexpr int kDcBits = 13;
// doubles per sequence row behind the threshold-line cache: 65 of the E-state row cache + 6 compact special-state arrays
constexpr int kTailRow = 72;
size_t resolve_tail_row_doubles() { return kTailRow; }
size_t resolve_dcache_doubles() { return (size_t)129 * (1 << kDcBits); }
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
int resolve_waves_per_cu() { return 4 * WH_RES_OCC; }
constexpr int kResMaxWaves = 4 * WH_RES_OCC;       // waves per workgroup the kernel is compiled for (512 threads, 256 registers at WH_RES_OCC = 2)

// 29 validation bits of a threshold line per lane (two different mixes in even and odd lanes: 58 bits per line), over the
// line's key and the epoch of the region: a line left by another region, pair or launch in the wave's slab never matches,
// so nothing is cleared between regions, and a line torn by a concurrent rewrite fails in some lane
__device__ __forceinline__ unsigned line_check_bits(unsigned long long key, unsigned epoch, int lane) {
  unsigned long long z = key ^ (((unsigned long long)epoch << 32) | ((lane & 1) ? 0x5bd1e995u : 0x27d4eb2fu));
  z ^= z >> 33; z *= 0xff51afd7ed558ccdULL; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ULL; z ^= z >> 33;
  return (unsigned)(z >> 35);
}
#define RTICK(slot) do { if (a.stats) { const long long t_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(a.stats + (slot), (unsigned long long)(t_now - tk0)); tk0 = t_now; } } while (0)

// Workgroups of <W> waves (one per CU).  The queue is ordered model by model; a workgroup draws SLOTS - each the right to
// work on one model's segment - and its waves then pull that segment's pairs one by one from a cursor in global memory,
// together with the waves of every other workgroup that holds a slot of the same model (a model gets slots in
// proportion to its share of the work, so its segment drains from several CUs at once and the only idle time is a
// wave's wait for its neighbours' last pair when the segment runs out).  Per segment the model's eight float64
// transition arrays are staged in LDS once for all waves (models of up to 16 cells per lane; a.lds_tables), which is
// what the two Forward sweeps of a pair read from: they were 34-45 % of the kernel and bound by streaming those arrays
// from L2 once per row and wave.
__global__ __launch_bounds__(64 * kResMaxWaves) void resolve_kernel(ResolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds_all[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  volatile int *s_hdr = lds_all;                                  // [0] chunk of the workgroup, [1] cursor inside it
  double *tabL = reinterpret_cast<double *>(lds_all + 4);         // staged transition arrays (a.lds_tables > 0)
  int *lds_raw = lds_all + 4 + (a.lds_tables > 0 ? gNARR * a.lds_tables * 64 * 2 : 0) + (size_t)wave * a.wave_lds_ints;
  const int Lp = (a.Lcap + 4) & ~1;
  uint8_t *seq = reinterpret_cast<uint8_t *>(lds_raw);
  int *uni = lds_raw + ((((a.Lcap + 8) / 4 + 2) + 1) & ~1);
  int *dom = uni;                                                 // traces: kDomMax x (sqfrom, sqto, hmmfrom, hmmto)
  float *dnull = reinterpret_cast<float *>(dom + 4 * kDomMax);   // kDomMax x 32
  short *stk = reinterpret_cast<short *>(dnull + 32 * kDomMax);  // emitting state of each residue: +k match, -k insert
  const int SEGCAP = a.seg_cap;
  unsigned short *s_a = reinterpret_cast<unsigned short *>(uni); // clustering (after the traces): Easel's vertex stacks, same block
  unsigned short *s_b = s_a + kSegLds;
  double *bins = reinterpret_cast<double *>(uni + resolve_uni_ints(a.Lcap));   // 64 float64 bins of the E-state row pass (8-byte aligned)
  int *misc = reinterpret_cast<int *>(bins + 64);                // 4 x kEnvMax + 3 x kClusMax ints: envelope list of the pair (detail), cluster list of a region
  unsigned long long *hprev = reinterpret_cast<unsigned long long *>(misc + 4 * kEnvMax + 3 * kClusMax + 16), *hcur = hprev + kHist;
  (void)Lp;
  const size_t wslot = (size_t)blockIdx.x * nwaves + wave;        // this wave's slab / segment arrays
  int32_t *sg = a.segs + wslot * a.seg_stride;      // per wave in HBM: 6 arrays of SEGCAP ints, the vertex stacks of a large region, the histogram
  int32_t *s_idx = sg, *s_i = sg + SEGCAP, *s_j = sg + 2 * SEGCAP, *s_k = sg + 3 * SEGCAP, *s_m = sg + 4 * SEGCAP;
  int32_t *s_as = sg + 5 * SEGCAP;
  unsigned short *h_a = reinterpret_cast<unsigned short *>(sg + 6 * SEGCAP), *h_b = h_a + SEGCAP;   // vertex stacks of a region of more than kSegLds segments
  int32_t *epc = sg + 7 * SEGCAP;                                // end-point histogram of one cluster
  float *n2sc = reinterpret_cast<float *>(epc + (a.Lcap > a.Mmax ? a.Lcap : a.Mmax) + 8);   // per residue, HBM (read with L1 bypass)
  float *acc = n2sc + a.Lcap + 8;
  // sum of n2sc[lo..hi] in position order (float32, as HMMER adds them): 64 values per fetch, walked with v_readlane
  auto n2sum = [&](int lo, int hi) -> float {
    float sum = 0.f;
    for (int p0 = lo; p0 <= hi; p0 += 64) {
      const int pp = p0 + lane;
      const float v = pp <= hi ? __builtin_nontemporal_load(n2sc + pp) : 0.f;
      const int cnt = hi - p0 + 1 < 64 ? hi - p0 + 1 : 64;
      for (int t = 0; t < cnt; t++) sum += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
    }
    return sum;
  };
  const double LOG2 = 0.69314718055994529;
  int cur_h = -1;
  bool use_tl = false;
  unsigned regions_done = 0;          // regions this wave has walked in this launch (epoch of its threshold-line cache)
  if (a.stats && threadIdx.x == 0) atomicMin(a.stats + 16, (unsigned long long)__builtin_amdgcn_s_memrealtime());

  for (;;) {
    // ---- next slot of the workgroup: the right to work on one model's segment of the queue
    const long long t_idle0 = a.stats ? __builtin_readcyclecounter() : 0;
    __syncthreads();                                   // every wave is done with the previous segment (and its tables)
    if (wave == 0) {
      // wave 0 picks the segment: the next slot while there are slots; afterwards ANY segment that still has pairs (the
      // workgroup joins whoever is still working - without this the launch ended on the last slots' workgroups alone:
      // mean wave lifetime 714 ms of a 919 ms launch), -1 when every segment is drained
      int seg_pick = -1;
      int slot = 0;
      if (lane == 0) slot = atomicAdd(a.counter, 1);
      slot = __shfl(slot, 0);
      if (slot < a.n_slots) seg_pick = a.slots[slot];
      else {
        for (int base = 0; base < a.n_chunks && seg_pick < 0; base += 64) {
          const int sgi = (int)((blockIdx.x * 7u + (unsigned)(base + lane)) % (unsigned)a.n_chunks);
          const bool in = base + lane < a.n_chunks;
          const int left = in ? a.chunks[4 * sgi + 1] - __hip_atomic_load(a.cursors + sgi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
          const unsigned long long have = __ballot(left > 0);
          if (have) seg_pick = __shfl(sgi, __ffsll((long long)have) - 1);
        }
      }
      if (lane == 0) s_hdr[0] = seg_pick;
    }
    __syncthreads();
    if (a.stats && lane == 0) atomicAdd(a.stats + 13, (unsigned long long)(__builtin_readcyclecounter() - t_idle0));
    const int seg = __builtin_amdgcn_readfirstlane(s_hdr[0]);
    if (seg < 0) break;                                // the same for every wave of the workgroup
    const int c_start = a.chunks[4 * seg], c_count = a.chunks[4 * seg + 1], c_h = a.chunks[4 * seg + 2], c_Q = a.chunks[4 * seg + 3];
    // (a segment that other workgroups have drained already costs this one a table staging and one atomic per wave)
    if (c_h != cur_h) {
      // (c_h < 0: a segment of mixed models - the whole of a small queue in descending cost - reads its tables from L2)
      use_tl = c_h >= 0 && a.lds_tables > 0 && c_Q <= a.lds_tables && (c_Q == 4 || c_Q == 8 || c_Q == 12 || c_Q == 16);
      if (use_tl) {
        const d2_t *src = reinterpret_cast<const d2_t *>(a.gtab + a.hmms[c_h].gfw_off);
        d2_t *dst = reinterpret_cast<d2_t *>(tabL);
        for (int t = threadIdx.x; t < gNARR * c_Q * 32; t += blockDim.x) dst[t] = src[t];
      }
      cur_h = c_h;
      __syncthreads();
    }
  for (;;) {
    // the segment's next pair, for whichever wave of whichever workgroup asks first (pairs are in descending cost)
    int item = 0;
    if (lane == 0) item = atomicAdd(a.cursors + seg, 1);
    item = __shfl(item, 0);
    if (item >= c_count) break;
    // By value: with a reference into the queue AND the LDS lists below, this toolchain produced a kernel that
    // read a garbage record (out-of-slab writes); either alone was fine.  Record fields are range-checked below
    // and the sampling loop is bounded, so a bad record can no longer run the wave out of its slab.
    const long long t_pair0 = a.stats ? __builtin_readcyclecounter() : 0;
    const unsigned long long r_pair0 = a.stats ? __builtin_amdgcn_s_memrealtime() : 0;
    const int ridx = a.order ? a.order[c_start + item] : c_start + item;
    const ResolveRec rec = a.recs[ridx];
    // long-list pass: the pair's regions are in HBM (any number of them), the record carries the pair alone
    const int32_t *xl = a.rext ? a.rext + (size_t)ridx * (size_t)a.rext_stride : nullptr;
    if (c_h >= 0 && rec.h != c_h) {                    // never true for a well-formed segment list (the staged tables are c_h's):
      if (lane == 0 && a.err) atomicAdd(a.err, 1);     // counted, and the next scoring call on the handle fails with WH_EHIP
      continue;
    }
    const DevHMM hm = a.hmms[rec.h];
    GModel m;
    m.tf = a.gtab + hm.gfw_off; m.te = a.gtab + hm.gem_off;
    const float *emn = a.ftab + hm.emn_off;
    m.Q = __builtin_amdgcn_readfirstlane(hm.Q); m.M = __builtin_amdgcn_readfirstlane(hm.M);
    GMx mx;
    mx.Q = m.Q; mx.rowlen = (size_t)3 * m.Q * 64 + xNSPEC;
    mx.p = a.mx + wslot * a.mx_stride;
    const int64_t off = a.offsets[rec.q];
    const int L = (int)(a.offsets[rec.q + 1] - off);
    const size_t out = (size_t)rec.q * a.H + rec.h;
    for (int t = lane; t < L; t += 64) { const int r = a.residues[off + t]; seq[t] = (uint8_t)(r < a.Kp ? r : a.Kp - 1); }
    for (int t = lane; t <= L + 1; t += 64) n2sc[t] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const GLen cm = glen_config(L, true), cu = glen_config(L, false);
    int flags = rec.flags | (rec.multi_mask ? WH_FLAG_MULTI : 0);     // (the any-size front end queues every pair with a region)
    // envelope list of the pair
    int nenv = 0;
    // small per-pair lists live in LDS (every lane writes the same value): as arrays in registers they pushed the
    // kernel to one wave per SIMD
    int *env_i = misc, *env_j = misc + kEnvMax;
    float *env_sc = reinterpret_cast<float *>(misc + 2 * kEnvMax), *env_dc = reinterpret_cast<float *>(misc + 3 * kEnvMax);
    float seqbias_sum = 0.f;
    // A.6's sums over the envelopes, taken as the envelopes arrive (the order HMMER adds them in): the lists above keep the
    // first kEnvMax for the detail record only, the score has no limit
    float sum_score = 0.f, sb2 = 0.f;
    int Ld_tot = 0;
    auto add_envelope = [&](int ei, int ej, float esc, float edc) {
      if (nenv < kEnvMax) { env_i[nenv] = ei; env_j[nenv] = ej; env_sc[nenv] = esc; env_dc[nenv] = edc; }
      nenv++;
      if (esc - edc > 0.0f) { sum_score += esc; Ld_tot += ej - ei + 1; sb2 += edc; }
    };
    const int nrec = rec.nenv < 0 ? 0 : xl ? rec.nenv : rec.nenv > WH_MAX_ENVELOPES ? WH_MAX_ENVELOPES : rec.nenv;
    for (int e = 0; e < nrec; e++) {
      // (values that steer the wave-uniform walk are made provably uniform: the walk then compiles to SALU code)
      const int ireg = __builtin_amdgcn_readfirstlane(xl ? xl[kRextInts * e] : rec.ri[e]);
      const int jreg = __builtin_amdgcn_readfirstlane(xl ? xl[kRextInts * e + 1] : rec.rj[e]), Lr = jreg - ireg + 1;
      if (ireg < 1 || jreg > L || Lr < 1 || L > a.Lcap) { flags |= WH_FLAG_TRUNC; continue; }     // never true for a well-formed record
      const bool multi_reg = xl ? __builtin_amdgcn_readfirstlane(xl[kRextInts * e + 4]) != 0 : ((rec.multi_mask >> (e & 31)) & 1) != 0;
      if (!multi_reg) {
        // single-domain region: envelope = region, scored by the scoring kernel (A.5)
        const float r_sc = xl ? __builtin_bit_cast(float, xl[kRextInts * e + 2]) : rec.envsc[e];
        const float r_dc = xl ? __builtin_bit_cast(float, xl[kRextInts * e + 3]) : rec.domcorr[e];
        seqbias_sum += r_dc;
        add_envelope(ireg, jreg, r_sc, r_dc);
        continue;
      }
      // ---------------- A.4b
      const uint8_t *rs = seq + (ireg - 1);       // rs[pos-1] = residue at region position pos
      long long tk0 = a.stats ? __builtin_readcyclecounter() : 0;
      const double regfwd = gforward_any<true>(m, rs, Lr, cm, mx, lane, use_tl ? (const ldbl *)tabL : nullptr);
      RTICK(0);
#ifdef WH_RESOLVE_DEBUG
      if (a.dbg && lane == 0) printf("[resolve] region forward %.12f\n", regfwd);
#endif
      for (int t = lane; t <= Lr + 1; t += 64) acc[t] = 0.f;
      // E-state choice: per row of the region a lazily filled line of 64 chunk prefix sums + a valid flag, at the
      // end of the wave's slab
      double *ecache = mx.p + a.mx_stride - (size_t)(a.Lcap + 2) * kTailRow;
      // The walk reads special states of 64 CONSECUTIVE rows per decision fetch (B of rows i-1-t for an M run, C / J / E of
      // a flank run): in the row tails that is 64 cache lines, in compact per-state arrays four.  Copied once per region.
      double *cN = ecache + (size_t)(a.Lcap + 2) * 65, *cB = cN + (a.Lcap + 2), *cE = cB + (a.Lcap + 2), *cJ = cE + (a.Lcap + 2);
      double *cC = cJ + (a.Lcap + 2), *cLS = cC + (a.Lcap + 2);
      for (int t = lane; t <= Lr; t += 64) {
        cN[t] = mx.spec(t, xN); cB[t] = mx.spec(t, xB); cE[t] = mx.spec(t, xE); cJ[t] = mx.spec(t, xJ); cC[t] = mx.spec(t, xC); cLS[t] = mx.spec(t, xLS);
      }
      u4_t *dlines = reinterpret_cast<u4_t *>(mx.p + a.dc_off);
      for (int t = lane; t <= Lr; t += 64) ecache[(size_t)t * 65 + 64] = 0.0;
      const unsigned epoch = a.launch_id * 0x9E3779B1u + (++regions_done) * 0x85EBCA77u;     // (the threshold-line cache needs no clearing: see line_check_bits)
      wave_mem_sync();
      int nseg = 0;
      Rng rng;
      rng.x = mix3(42u, 87654321u, 12345678u);
      if (rng.x == 0) rng.x = 42;
      const int Qs = ((m.M - 1) / 4 + 1) > 2 ? ((m.M - 1) / 4 + 1) : 2;     // HMMER's striping: vectors of 4 floats
      long long c_build = 0, c_e = 0, c_post = 0, c_load = 0;
      unsigned n_bm = 0, n_bd = 0, n_bf = 0, n_i = 0, n_hit = 0;     // WH_STATS: fetches of M / D / flank (C, J) runs, scalar I steps
      // lane t's jump of esl_random's LCG by t+1 steps: x_{n+t+1} = lcgA * x_n + lcgC (mod 2^32)
      unsigned lcgA = 1u, lcgC = 0u;
      for (int u = 0; u < 64; u++) if (u <= lane) { lcgA *= 69069u; lcgC = lcgC * 69069u + 1u; }
      int n_prev = 0;                            // fetches of the previous trace (keys in hprev)
      unsigned n_pred = 0, n_resync = 0;
      for (int t = 0; t < kSamples; t++) {
        int i = Lr, k = 0, s0 = stC, ndom = 0, sqto = 0, hmmto = 0, sqfrom = 0, hmmfrom = 0;
        int jf = 0, jp = 0;                      // fetch number of this trace; the fetch of the previous trace expected next
        int run_state = 0, run_j = 0;            // decision cache of the current run (see below)
        // thresholds as integers: (sum / norm > x / 2^32) <=> x < ceil(2^32 sum / norm), exactly (both sides are exact
        // in double); 33-bit values: a low word and an 'always true' bit (bits 0..2 of run_hi)
        unsigned run_r1 = 0, run_r2 = 0, run_r3 = 0, run_hi = 0;
        // ... and, since the random numbers of those decisions are known too (one LCG step each), their OUTCOMES:
        // lane t holds the number (run_x) and the choice (run_c) of the run's t-th decision, run_cont has bit t set
        // where that choice continues the run.  A run is then consumed in one step up to its first exit.
        unsigned run_x = 0;
        int run_c = 0;
        unsigned long long run_cont = 0;
        int guard = 4 * (Lr + m.M) + 64;         // a sampled path has at most Lr + M + a few states
        while (s0 != stS && --guard > 0 && i >= 0 && k >= 0 && k <= m.M) {
          double path[4] = {0.0, 0.0, 0.0, 0.0};
          int s1;
          if (s0 == stN) { i = 0; s0 = stS; continue; }     // N(i) <- N(i-1) ... <- S: no random number is drawn on the way
          if (s0 == stM || s0 == stD || s0 == stC || s0 == stJ) {
            // A sampled path mostly RUNS: M along a diagonal, D along a row, C / J along a flank.  One random
            // number per decision keeps the walk serial, but the inputs of the next 64 decisions of a run are
            // independent of the outcomes: lane t fetches those of the run's t-th state in one round of loads
            // (instead of one memory round trip per step) and turns them into esl_rnd_FChoose's thresholds.
            if (run_state != s0 || run_j >= 64) {
              const long long tb0 = a.stats ? __builtin_readcyclecounter() : 0;
              const int it = (s0 == stD) ? i : i - lane, kt = (s0 == stM || s0 == stD) ? k - lane : 0;
              // The thresholds depend on the run's first cell only, and the 200 paths of a region mostly re-enter the
              // same cells (they leave and rejoin the diagonals at the same insertions): a direct-mapped cache of
              // threshold lines in the wave's slab, keyed by (state, i, k).  A hit is ONE coalesced 1 KB read instead
              // of three scattered matrix reads per lane (192 cache lines; at 2 048 waves the walk was bound by HBM
              // line traffic).  The line and its tag are requested together.
              const int sidx = s0 == stM ? 0 : s0 == stD ? 1 : s0 == stC ? 2 : 3;
              const unsigned kkey = (s0 == stM || s0 == stD) ? (unsigned)k : 0u;
              const unsigned long long key = (1ull << 63) | ((unsigned long long)sidx << 60) | ((unsigned long long)(unsigned)i << 30) | kkey;
              const unsigned slot = (((unsigned)i * 0x9E3779B1u) ^ (kkey * 0x85EBCA77u) ^ ((unsigned)sidx * 0xC2B2AE3Du)) >> (32 - kDcBits);
              u4_t *dline = dlines + (size_t)slot * 64 + lane;
              const unsigned chk = line_check_bits(key, epoch, lane);
              // (Round 4, measured and dropped: 56 % of a trace's fetches are the fetch that followed the last matched one in the
              // previous trace of the region, so the lines can be requested ahead - by LDS-DMA loads without a register - one
              // fetch ahead, several ahead, or all of the previous trace's lines at the top of a trace.  None of it shortens a
              // fetch: a line warm in L2 still takes 1 200 of its 1 350 cycles; the time is queueing behind the float64 matrices
              // the neighbouring waves write - 594 cycles with one wave per CU, 751 / 948 / 1 347 with two / four / eight.)
              const long long tl0 = a.stats ? __builtin_readcyclecounter() : 0;
              const u4_t ent = *dline;
              const bool valid = __ballot((ent.w >> 3) == chk) == ~0ull;
              if (a.stats) c_load += __builtin_readcyclecounter() - tl0;
              if (a.stats) {
                // (how far the fetches repeat from trace to trace: the fetch that followed the last matched one in the previous trace?)
                const bool pred = jp < n_prev && hprev[jp] == key;
                if (pred) { n_pred++; jp++; }
                else {
                  const unsigned long long hit = __ballot(lane < n_prev && hprev[lane] == key);
                  if (hit) { jp = __ffsll((long long)hit); n_resync++; }
                }
                if (jf < kHist && lane == 0) hcur[jf] = key;
                jf++;
              }
              if (valid) {
                run_r1 = ent.x; run_r2 = ent.y; run_r3 = ent.z; run_hi = ent.w & 7u;
                if (a.stats) n_hit++;
              } else {
              double pd[4] = {0.0, 0.0, 0.0, 0.0};
              if (s0 == stM) {
                if (it >= 1 && kt >= 1) {
                  pd[0] = cB[it - 1] * m.t(gE, kt);
                  pd[1] = mx.cellc(it - 1, kt - 1, 0) * m.t(gA, kt);
                  pd[2] = mx.cellc(it - 1, kt - 1, 1) * m.t(gB, kt);
                  pd[3] = mx.cellc(it - 1, kt - 1, 2) * m.t(gC, kt);
                }
              } else if (s0 == stD) {
                if (kt >= 1) {
                  pd[0] = mx.cellc(it, kt - 1, 0) * m.t(gD1, kt);
                  pd[1] = mx.cellc(it, kt - 1, 2) * m.t(gD2, kt);
                }
              } else if (it >= 1) {
                pd[0] = (s0 == stC ? cC : cJ)[it - 1] * cm.loop;
                pd[1] = cE[it] * (s0 == stC ? cm.EC : cm.EJ) * exp(cLS[it]);
              }
              const int nch = s0 == stM ? 4 : 2;
              double tot = 0.0;
              for (int u = 0; u < nch; u++) tot += pd[u];
              float pf[4] = {0.f, 0.f, 0.f, 0.f};
              for (int u = 0; u < nch; u++) pf[u] = tot > 0.0 ? (float)(pd[u] / tot) : 1.0f / (float)nch;
              double norm = 0.0;
              for (int u = 0; u < nch; u++) norm += (double)pf[u];
              const double c1 = (double)pf[0], c2 = c1 + (double)pf[1], c3 = c2 + (double)pf[2];
              auto as_int = [](double thr, unsigned &lo, unsigned &hi_bit) {      // x < ceil(2^32 thr)
                const double T = ceil(thr * 4294967296.0);
                if (!(T > 0.0)) { lo = 0; hi_bit = 0; }
                else if (T >= 4294967296.0) { lo = 0; hi_bit = 1; }
                else { lo = (unsigned)T; hi_bit = 0; }
              };
              unsigned hb;
              run_hi = 0;
              as_int(c1 / norm, run_r1, hb); run_hi |= hb;
              as_int(c2 / norm, run_r2, hb); run_hi |= hb << 1;
              as_int(c3 / norm, run_r3, hb); run_hi |= hb << 2;
              u4_t wr; wr.x = run_r1; wr.y = run_r2; wr.z = run_r3; wr.w = run_hi | (chk << 3);
              *dline = wr;
              wave_mem_sync();
              }
              run_state = s0; run_j = 0;
              run_x = lcgA * rng.x + lcgC;
              bool cont;
              if (s0 == stM) {
                run_c = ((run_hi & 1) || run_x < run_r1) ? 0 : ((run_hi & 2) || run_x < run_r2) ? 1 : ((run_hi & 4) || run_x < run_r3) ? 2 : 3;
                cont = run_c == 1 && it >= 1 && kt >= 1;
              } else {
                run_c = ((run_hi & 1) || run_x < run_r1) ? 0 : 1;
                cont = s0 == stD ? (run_c == 1 && kt >= 1) : (run_c == 0 && it >= 1);
              }
              run_cont = __ballot(cont);
              if (a.stats) { c_build += __builtin_readcyclecounter() - tb0; if (s0 == stM) n_bm++; else if (s0 == stD) n_bd++; else n_bf++; }
            }
            {
              // the leading decisions of the run that stay in its state, all at once
              const unsigned long long rest = ~(run_cont >> run_j);
              int n = rest ? __builtin_ctzll(rest) : 64;
              if (n > 64 - run_j) n = 64 - run_j;
              if (n > guard - 1) n = guard - 1 > 0 ? guard - 1 : 0;
              if (n > 0) {
                rng.x = (unsigned)__builtin_amdgcn_readlane((int)run_x, run_j + n - 1);
                if (s0 == stM) {
                  if (sqto == 0) { sqto = i - 1; hmmto = k - 1; }
                  if (lane < n) stk[i - 1 - lane] = (short)(k - 1 - lane);
                  i -= n; k -= n;
                  sqfrom = i; hmmfrom = k;
                } else if (s0 == stD) {
                  k -= n;
                } else {
                  i -= n;
                }
                run_j += n;
                guard -= n;
                continue;
              }
            }
            // the run's next decision, one random number (esl_random): its outcome was formed with the cache
            rng.x = (unsigned)__builtin_amdgcn_readlane((int)run_x, run_j);
            const int c4 = __builtin_amdgcn_readlane(run_c, run_j);
            if (s0 == stM) {
              s1 = c4 == 0 ? stB : c4 == 1 ? stM : c4 == 2 ? stI : stD;
              k--; i--;
            } else if (s0 == stD) {
              s1 = c4 == 0 ? stM : stD;
              k--;
            } else {
              s1 = c4 == 0 ? s0 : stE;
            }
            run_j++;
            if (s1 != s0) run_state = 0;
          } else
          switch (s0) {
            case stI:
              n_i++;
              path[0] = mx.cellc(i - 1, k, 0) * m.t(gMI, k);
              path[1] = mx.cellc(i - 1, k, 1) * m.t(gII, k);
              s1 = __builtin_amdgcn_readfirstlane(rng_choose(rng, path, 2)) == 0 ? stM : stI;
              i--;
              break;
            case stE: {
              const long long te0 = a.stats ? __builtin_readcyclecounter() : 0;
              // FChoose over M(i,*) and D(i,*) in HMMER's striped order: position p = q*8 + state*4 + r
              // holds node r*Qs + q + 1.  Lanes take contiguous chunks, an exclusive scan finds the chunk.
              const double roll = rng_next(rng), norm = 1.0 / cE[i];
              const int total = 8 * Qs, chunk = (total + 63) / 64;
              const int p0 = lane * chunk, p1 = min(total, p0 + chunk);
              auto term = [&](int p) -> double {
                const int q = p >> 3, st = (p >> 2) & 1, r = p & 3, kk = r * Qs + q + 1;
                return kk <= m.M ? (double)(float)(mx.cellc(i, kk, st ? 2 : 0) * norm) : 0.0;
              };
              // The chunk prefix sums of a row are the same for every trace that ends a domain there: computed once per
              // row and kept in the slab, a hit replaces the pass over the 2M cells by one coalesced 512-byte read.
              // (the line is requested together with its valid flag: one round trip on a hit, a harmless read on a miss)
              double incl = __builtin_nontemporal_load(ecache + (size_t)i * 65 + lane);
              if (__builtin_nontemporal_load(ecache + (size_t)i * 65 + 64) == 0.0) {
                // The chunk sums from ONE coalesced pass over the row: every lane reads its own nodes (16-byte pairs, 1 KB
                // per wave instruction) and adds each term into the bin of the chunk that holds the node's striped position -
                // 64 float64 bins in LDS (the clustering stacks' block, idle during the traces; one wave per workgroup, so
                // the order of the LDS adds is the program's).  Reading the terms in striped order instead took 81 scattered
                // 8-byte loads per lane: ~330 KB of cache lines for 45 KB of cells, and this step was bound by exactly that.
                bins[lane] = 0.0;
                __builtin_amdgcn_wave_barrier();
                {
                  const double *rowp = mx.row(i);
                  const size_t SQm = (size_t)m.Q * 64;
                  int kn = lane * m.Q + 1;
                  int rr = (kn - 1) / Qs, qs = (kn - 1) % Qs;          // node kn sits at striped position qs*8 + state*4 + rr
                  int cb = (qs * 8 + rr) / chunk, rem = (qs * 8 + rr) % chunk;
                  for (int q0 = 0; q0 < m.Q; q0 += 4) {
                    const d2_t m0 = ld_d2(rowp + ofs2(q0, lane)), m1 = ld_d2(rowp + ofs2(q0 + 2, lane));
                    const d2_t d0 = ld_d2(rowp + 2 * SQm + ofs2(q0, lane)), d1 = ld_d2(rowp + 2 * SQm + ofs2(q0 + 2, lane));
                    const double vM[4] = {m0.x, m0.y, m1.x, m1.y}, vD[4] = {d0.x, d0.y, d1.x, d1.y};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                      if (kn <= m.M) {
                        atomicAdd(bins + cb, (double)(float)(vM[u] * norm));
                        int cd = cb, remd = rem + 4;
                        while (remd >= chunk) { remd -= chunk; cd++; }
                        atomicAdd(bins + cd, (double)(float)(vD[u] * norm));
                      }
                      kn++; qs++; rem += 8;
                      if (qs == Qs) { qs = 0; rr++; cb = rr / chunk; rem = rr % chunk; }
                      else while (rem >= chunk) { rem -= chunk; cb++; }
                    }
                  }
                }
                __builtin_amdgcn_wave_barrier();
                incl = bins[lane];
                for (int d = 1; d < 64; d <<= 1) { const double o = shfl_up_d(incl, d); if (lane >= d) incl += o; }
                ecache[(size_t)i * 65 + lane] = incl;
                if (lane == 0) ecache[(size_t)i * 65 + 64] = 1.0;
                wave_mem_sync();
              }
              const double up1 = shfl_up_d(incl, 1);
              const double excl = lane > 0 ? up1 : 0.0;
              const unsigned long long hit = __ballot(roll < incl);
              int found_p = -1;
              if (hit) {
                // the chunk that holds the crossing is re-walked by the whole wave: lane u takes its u-th term
                const int src = __ffsll((long long)hit) - 1;
                const int c0p = __shfl(p0, src), c1p = __shfl(p1, src);
                double base = shfl_d(excl, src);
                // (a chunk holds at most 2 x 64 terms for models of up to 4096 nodes: both halves are requested at once)
                const double tvA = c0p + lane < c1p ? term(c0p + lane) : 0.0;
                const double tvB = c0p + 64 + lane < c1p ? term(c0p + 64 + lane) : 0.0;
                for (int pb = c0p; pb < c1p && found_p < 0; pb += 64) {
                  const int p = pb + lane;
                  const double tv = pb == c0p ? tvA : pb == c0p + 64 ? tvB : (p < c1p ? term(p) : 0.0);
                  double run = tv;
                  for (int d = 1; d < 64; d <<= 1) { const double o = shfl_up_d(run, d); if (lane >= d) run += o; }
                  const unsigned long long h2 = __ballot(p < c1p && roll < base + run);
                  if (h2) found_p = pb + __ffsll((long long)h2) - 1;
                  base += shfl_d(run, 63);
                }
                if (found_p < 0) found_p = c1p - 1;
              }
              found_p = __builtin_amdgcn_readfirstlane(found_p);
              if (found_p < 0) { k = 1; s1 = stM; }        // rounding left the sum below the roll: HMMER rescans, first non-zero cell wins
              else { k = (found_p & 3) * Qs + (found_p >> 3) + 1; s1 = ((found_p >> 2) & 1) ? stD : stM; }
              if (a.stats) c_e += __builtin_readcyclecounter() - te0;
              break;
            }
            case stB:
              path[0] = cN[i] * cm.move;
              path[1] = cJ[i] * cm.move;
              s1 = __builtin_amdgcn_readfirstlane(rng_choose(rng, path, 2)) == 0 ? stN : stJ;
              break;
            default: s1 = stS; break;
          }
#ifdef WH_RESOLVE_DEBUG
          if (a.dbg >= 1000 && t == a.dbg - 1000 && lane == 0)
            printf("   [trace %d] s0 %d -> s1 %d at i %d k %d  path %.9g %.9g %.9g %.9g rng %u\n", t, s0, s1, i, k, path[0], path[1], path[2], path[3], rng.x);
#endif
          // the state just chosen sits at (k, i)
          if (s1 == stE) { sqto = 0; hmmto = 0; }
          else if (s1 == stM) {
            if (sqto == 0) { sqto = i; hmmto = k; }
            sqfrom = i; hmmfrom = k;
            if (lane == 0) stk[i] = (short)k;
          } else if (s1 == stI) { if (lane == 0) stk[i] = (short)-k; }
          else if (s1 == stB) {
            if (ndom < kDomMax) {
              if (lane == 0) { dom[4 * ndom] = sqfrom; dom[4 * ndom + 1] = sqto; dom[4 * ndom + 2] = hmmfrom; dom[4 * ndom + 3] = hmmto; }
              ndom++;
            } else flags |= WH_FLAG_TRUNC;
          }
          if ((s1 == stN || s1 == stJ || s1 == stC) && s1 == s0) i--;
          s0 = s1;
        }
        __builtin_amdgcn_wave_barrier();
        if (a.stats) { n_prev = jf < kHist ? jf : kHist; unsigned long long *tsw = hprev; hprev = hcur; hcur = tsw; }
        const long long tp0 = a.stats ? __builtin_readcyclecounter() : 0;
        // null2 by trace of every sampled domain (A.4b / p7_Null2_ByTrace): mean emission odds of the
        // M/I states that emitted the domain's residues
        for (int d = 0; d < ndom; d++) {
          const int df = dom[4 * d], dt = dom[4 * d + 1], Ld = dt - df + 1;
          float mine = 1.0f;
          // Round 4: the sum of the emission odds over the domain's match states from PREFIX SUMS over the nodes (float64,
          // DevHMM::esum_off): a sampled domain is a dozen runs of consecutive nodes, so its sum is a dozen differences of
          // table rows - found from the per-residue states in LDS, fetched in one or two rounds of loads - instead of one
          // table row per residue in three to five dependent rounds (this loop was a third of the trace time).  The sum is
          // exact to 1e-16 before it is rounded to float (HMMER adds the same terms in float32, in striped node order).
          bool by_runs = false;
          {
            int *blist = reinterpret_cast<int *>(bins);             // signed node numbers: +k adds row k, -k subtracts it (128 entries)
            int nb = 0, nI = 0;
            for (int p0 = df; p0 <= dt; p0 += 64) {
              const int p = p0 + lane;
              const bool valid = p <= dt;
              const int kk = valid ? (int)stk[p] : 0;
              const int km = (valid && p > df) ? (int)stk[p - 1] : -32768, kp = (valid && p < dt) ? (int)stk[p + 1] : -32768;
              const bool isM = valid && kk > 0;
              nI += __builtin_popcountll(__ballot(valid && kk <= 0));
              const bool st = isM && km != kk - 1 && kk > 1, en = isM && kp != kk + 1;
              const unsigned long long bs = __ballot(st), be = __ballot(en);
              const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
              const int is_ = nb + __builtin_popcountll(bs & below), ie = nb + __builtin_popcountll(bs) + __builtin_popcountll(be & below);
              if (st && is_ < 128) blist[is_] = -(kk - 1);
              if (en && ie < 128) blist[ie] = kk;
              nb += __builtin_popcountll(bs) + __builtin_popcountll(be);
            }
            __builtin_amdgcn_wave_barrier();
            if (nb <= 128 && !a.null2_gather) {
              by_runs = true;
              const double *esum = a.gtab + hm.esum_off;
              const int G = 64 / a.K, g = lane / a.K, x = lane - g * a.K;
              double sd = 0.0;
              if (g < G) {
#pragma unroll 4
                for (int b = g; b < nb; b += G) {
                  const int e = blist[b];
                  const double v = esum[(size_t)(e < 0 ? -e : e) * a.K + x];
                  sd += e < 0 ? -v : v;
                }
              }
              __builtin_amdgcn_wave_barrier();
              bins[lane] = sd;
              __builtin_amdgcn_wave_barrier();
              if (lane < a.K) {
                double tot = 0.0;
                for (int gg = 0; gg < G; gg++) tot += bins[gg * a.K + lane];
                mine = (float)(tot + (double)nI) / (float)Ld;
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
          if (!by_runs) {
          // one lane per sampled position: the K odds of its emitting node are contiguous (node-major float
          // copy, DevHMM::emn_off), summed per residue in registers, then K wave sums
          float part[20];
#pragma unroll
          for (int x = 0; x < 20; x++) part[x] = 0.f;
          // (two positions per lane and step: the reads of both are requested before either is used; a lane's
          // positions are still added in ascending order)
          for (int pos = df + lane; pos <= dt; pos += 128) {
            const int kk0 = stk[pos], kk1 = pos + 64 <= dt ? stk[pos + 64] : 0;
            const bool has1 = pos + 64 <= dt;
            float4 v0[5], v1[5];
            const int nq4 = a.K == 4 ? 1 : 5;
#pragma unroll
            for (int x4 = 0; x4 < 5; x4++) {
              v0[x4] = make_float4(1.f, 1.f, 1.f, 1.f); v1[x4] = make_float4(1.f, 1.f, 1.f, 1.f);
              if (x4 < nq4) {
                if (kk0 > 0) v0[x4] = *reinterpret_cast<const float4 *>(emn + (size_t)kk0 * a.K + 4 * x4);
                if (kk1 > 0) v1[x4] = *reinterpret_cast<const float4 *>(emn + (size_t)kk1 * a.K + 4 * x4);
              }
            }
#pragma unroll
            for (int x4 = 0; x4 < 5; x4++)
              if (x4 < nq4) { part[4 * x4] += v0[x4].x; part[4 * x4 + 1] += v0[x4].y; part[4 * x4 + 2] += v0[x4].z; part[4 * x4 + 3] += v0[x4].w; }
            if (has1) {
#pragma unroll
              for (int x4 = 0; x4 < 5; x4++)
                if (x4 < nq4) { part[4 * x4] += v1[x4].x; part[4 * x4 + 1] += v1[x4].y; part[4 * x4 + 2] += v1[x4].z; part[4 * x4 + 3] += v1[x4].w; }
            }
          }
#pragma unroll
          for (int x = 0; x < 20; x++) {
            if (x < a.K) {
              const float s = wave_sum_f(part[x]);
              if (lane == x) mine = s / (float)Ld;
            }
          }
          }
          __builtin_amdgcn_wave_barrier();
          if (lane < a.K) dnull[32 * d + lane] = mine;
          __builtin_amdgcn_wave_barrier();
          if (lane >= a.K && lane < a.Kp) {
            const uint32_t msk = a.degen[lane];
            float s = 0.f; int n = 0;
            for (int x = 0; x < a.K; x++) if (msk & (1u << x)) { s += dnull[32 * d + x]; n++; }
            dnull[32 * d + lane] = n > 0 ? s / (float)n : 1.0f;
          }
          __builtin_amdgcn_wave_barrier();
        }
        // per-residue accumulators: +1 outside sampled domains AND at a domain's first residue (sic),
        // + null2[x] at the domain's other residues
        // (the accumulators live in the wave's HBM slab: eight positions per lane are requested at once)
        // (the domains' bounds once into a register, lane d holding domain d, read back as scalars: the inner loop was a chain of
        // dependent LDS reads - two bounds per domain and position)
        const int dlo_v = lane < ndom ? dom[4 * lane] : 0, dhi_v = lane < ndom ? dom[4 * lane + 1] : -1;
        for (int p0 = 1 + lane; p0 <= Lr; p0 += 512) {
          float old[8];
          int rsv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const bool in = p0 + 64 * u <= Lr;
            old[u] = in ? __builtin_nontemporal_load(acc + p0 + 64 * u) : 0.f;
            rsv[u] = in ? (int)rs[p0 + 64 * u - 1] : 0;
          }
          float add[8];
#pragma unroll
          for (int u = 0; u < 8; u++) add[u] = 1.0f;
          for (int d = 0; d < ndom; d++) {
            const int lo = __builtin_amdgcn_readlane(dlo_v, d), hi = __builtin_amdgcn_readlane(dhi_v, d);
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const int pos = p0 + 64 * u;
              if (pos > lo && pos <= hi) add[u] = dnull[32 * d + rsv[u]];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int pos = p0 + 64 * u;
            if (pos <= Lr) acc[pos] = old[u] + add[u];
          }
        }
        // the ensemble takes the domains left to right (they were found right to left)
        if (lane == 0)
          for (int d = ndom - 1; d >= 0; d--)
            if (nseg + (ndom - 1 - d) < SEGCAP) {
              const int z = nseg + (ndom - 1 - d);
              s_idx[z] = t; s_i[z] = dom[4 * d] + ireg - 1; s_j[z] = dom[4 * d + 1] + ireg - 1; s_k[z] = dom[4 * d + 2]; s_m[z] = dom[4 * d + 3];
            }
        if (nseg + ndom > SEGCAP) flags |= WH_FLAG_TRUNC;
        nseg = min(SEGCAP, nseg + ndom);
        __builtin_amdgcn_wave_barrier();
        if (a.stats) c_post += __builtin_readcyclecounter() - tp0;
      }
      if (a.stats && lane == 0) {
        atomicAdd(a.stats + 5, (unsigned long long)c_build); atomicAdd(a.stats + 6, (unsigned long long)c_e); atomicAdd(a.stats + 7, (unsigned long long)c_post);
        atomicAdd(a.stats + 20, (unsigned long long)n_pred); atomicAdd(a.stats + 21, (unsigned long long)n_resync); atomicAdd(a.stats + 23, (unsigned long long)c_load);
        atomicAdd(a.stats + 8, (unsigned long long)n_bm); atomicAdd(a.stats + 9, (unsigned long long)n_bd); atomicAdd(a.stats + 10, (unsigned long long)n_bf); atomicAdd(a.stats + 11, (unsigned long long)n_i); atomicAdd(a.stats + 12, (unsigned long long)n_hit);
      }
      RTICK(1);
      for (int pos = 1 + lane; pos <= Lr; pos += 64) n2sc[ireg + pos - 1] = logf(__builtin_nontemporal_load(acc + pos) / (float)kSamples);
      wave_mem_sync();
      // ---------------- single-linkage clustering in Easel's vertex order (esl_cluster_SingleLinkage)
      // (the two vertex stacks: in LDS, ordered by the LDS pipeline itself - or, for a region of more than kSegLds segments,
      // in the wave's HBM block with every hand-over between lanes ordered by hand; same code, <sync> is what differs)
      auto single_linkage = [&](unsigned short *s_a, unsigned short *s_b, auto sync) -> int {
        int nc = 0;
        for (int v = lane; v < nseg; v += 64) s_a[v] = (unsigned short)(nseg - v - 1);
        sync();
        int na = nseg;
        while (na > 0) {
          int v = s_a[na - 1];
          na--;
          int nb = 1;
          sync();
          if (lane == 0) s_b[0] = (unsigned short)v;
          sync();
          while (nb > 0) {
            v = s_b[nb - 1];
            nb--;
            sync();
            if (lane == 0) s_as[v] = nc;
            const int vi = __builtin_nontemporal_load(s_i + v), vj = __builtin_nontemporal_load(s_j + v), vk = __builtin_nontemporal_load(s_k + v), vm = __builtin_nontemporal_load(s_m + v);
            // scan a[na-1 .. 0]: link tests in parallel, deletions (swap with the last entry) in scan order
            for (int hi = na - 1; hi >= 0; hi -= 64) {
              const int tpos = hi - lane;
              bool link = false;
              int w = 0;
              if (tpos >= 0) {
                w = s_a[tpos];
                link = seg_linked(vi, vj, vk, vm, __builtin_nontemporal_load(s_i + w), __builtin_nontemporal_load(s_j + w),
                                  __builtin_nontemporal_load(s_k + w), __builtin_nontemporal_load(s_m + w));
              }
              unsigned long long hits = __ballot(link);
              // entries moved in by a swap come from positions above the scan point of this chunk: they were
              // tested in this very scan already (the scan runs downwards), so the flags stay valid
              while (hits) {
                const int l = __ffsll((long long)hits) - 1;
                hits &= hits - 1;
                const int tp = hi - l;
                const int wv = __shfl(w, l);
                sync();
                if (lane == 0) {
                  s_a[tp] = s_a[na - 1];
                  s_b[nb] = (unsigned short)wv;
                }
                na--; nb++;
                sync();
              }
            }
          }
          nc++;
        }
        return nc;
      };
      const int nc = nseg <= kSegLds ? single_linkage(s_a, s_b, []() { __builtin_amdgcn_wave_barrier(); })
                                     : single_linkage(h_a, h_b, []() { wave_mem_sync(); });
      wave_mem_sync();
#ifdef WH_RESOLVE_DEBUG
      if (a.dbg && lane == 0) {
        printf("[resolve q=%lld h=%d region %d..%d] nseg %d nc %d rng %u\n", (long long)rec.q, rec.h, ireg, jreg, nseg, nc, rng.x);
        for (int z = 0; z < nseg && z < a.dbg; z++) printf("   seg %d: t %d i %d j %d k %d m %d cluster %d\n", z, s_idx[z], s_i[z], s_j[z], s_k[z], s_m[z], s_as[z]);
      }
#endif
      RTICK(2);
      // ---------------- clusters -> envelopes (p7_spensemble_Cluster)
      int nsig = 0;
      int *g_i = misc + 4 * kEnvMax, *g_j = g_i + kClusMax;
      float *g_p = reinterpret_cast<float *>(g_j + kClusMax);
      for (int c = 0; c < nc; c++) {
        // posterior of the cluster: traces that contribute (segments are in trace order)
        int ninc = 0;
        {
          int last = -1;      // idx of the previous member in the serial order ("idx_of_last")
          for (int h0 = 0; h0 < nseg; h0 += 64) {
            const int h = h0 + lane;
            const bool mem = h < nseg && __builtin_nontemporal_load(s_as + h) == c;
            const int idx = mem ? __builtin_nontemporal_load(s_idx + h) : -1;
            const unsigned long long mm = __ballot(mem);
            const unsigned long long below = mm & ((1ull << lane) - 1ull);
            const int src = below ? 63 - __clzll((long long)below) : 0;
            const int pv = __shfl(idx, src);
            const int prev = below ? pv : last;
            ninc += wave_sum_i((mem && idx != prev) ? 1 : 0);
            if (mm) last = __shfl(idx, 63 - __clzll((long long)mm));
          }
        }
        if ((float)ninc / (float)kSamples < 0.25f) continue;
        int imin = 1 << 30, imax = -1, jmin = 1 << 30, jmax = -1, kmin = 1 << 30, kmax = -1, mmin = 1 << 30, mmax = -1;
        for (int h = lane; h < nseg; h += 64)
          if (__builtin_nontemporal_load(s_as + h) == c) {
            const int si = __builtin_nontemporal_load(s_i + h), sj = __builtin_nontemporal_load(s_j + h), sk = __builtin_nontemporal_load(s_k + h), sm = __builtin_nontemporal_load(s_m + h);
            imin = min(imin, si); imax = max(imax, si); jmin = min(jmin, sj); jmax = max(jmax, sj);
            kmin = min(kmin, sk); kmax = max(kmax, sk); mmin = min(mmin, sm); mmax = max(mmax, sm);
          }
        imin = wave_min_i(imin); imax = wave_max_i(imax); jmin = wave_min_i(jmin); jmax = wave_max_i(jmax);
        kmin = wave_min_i(kmin); kmax = wave_max_i(kmax); mmin = wave_min_i(mmin); mmax = wave_max_i(mmax);
        const int thr = (int)ceilf((float)ninc * 0.02f);
        // end-point histograms; which = 0 i (leftmost), 1 k (leftmost), 2 j (rightmost), 3 m (rightmost)
        int best[4];
        for (int which = 0; which < 4; which++) {
          const int lo = which == 0 ? imin : which == 1 ? kmin : which == 2 ? jmin : mmin;
          const int hi = which == 0 ? imax : which == 1 ? kmax : which == 2 ? jmax : mmax;
          const int32_t *src = which == 0 ? s_i : which == 1 ? s_k : which == 2 ? s_j : s_m;
          const int n = hi - lo + 1;
          for (int t = lane; t < n; t += 64) epc[t] = 0;
          wave_mem_sync();
          for (int h = lane; h < nseg; h += 64)
            if (__builtin_nontemporal_load(s_as + h) == c) atomicAdd(&epc[__builtin_nontemporal_load(src + h) - lo], 1);
          wave_mem_sync();
          int b = -1;
          if (which < 2) {          // leftmost position with enough end points, else the (first) most frequent one
            int cand = 1 << 30;
            for (int t = lane; t < n; t += 64) if (__builtin_nontemporal_load(epc + t) >= thr) cand = min(cand, t);
            cand = wave_min_i(cand);
            if (cand < (1 << 30)) b = cand;
          } else {
            int cand = -1;
            for (int t = lane; t < n; t += 64) if (__builtin_nontemporal_load(epc + t) >= thr) cand = max(cand, t);
            cand = wave_max_i(cand);
            if (cand >= 0) b = cand;
          }
          if (b < 0) {              // esl_vec_IArgMax: first maximum
            int bv = -1, bt = 1 << 30;
            for (int t = lane; t < n; t += 64) { const int e2 = __builtin_nontemporal_load(epc + t); if (e2 > bv) { bv = e2; bt = t; } }
            const int gmax = wave_max_i(bv);
            b = wave_min_i(bv == gmax ? bt : (1 << 30));
          }
          best[which] = lo + b;
          __builtin_amdgcn_wave_barrier();
        }
#ifdef WH_RESOLVE_DEBUG
        if (a.dbg && lane == 0) printf("[resolve q=%lld h=%d region %d..%d] cluster %d: ninc %d thr %d i %d..%d j %d..%d k %d..%d m %d..%d best %d %d %d %d\n", (long long)rec.q, rec.h, ireg, jreg, c, ninc, thr, imin, imax, jmin, jmax, kmin, kmax, mmin, mmax, best[0], best[2], best[1], best[3]);
#endif
        if (best[0] > best[2] || best[1] > best[3]) continue;
        if (nsig < kClusMax) { g_i[nsig] = best[0]; g_j[nsig] = best[2]; g_p[nsig] = (float)ninc / (float)kSamples; nsig++; }
        else flags |= WH_FLAG_TRUNC;
      }
      RTICK(3);
      // order by start (stable), drop dominated clusters (region_trace_ensemble)
      for (int d = 1; d < nsig; d++) {
        const int ti = g_i[d], tj = g_j[d]; const float tp = g_p[d];
        int d2 = d - 1;
        for (; d2 >= 0 && g_i[d2] > ti; d2--) { g_i[d2 + 1] = g_i[d2]; g_j[d2 + 1] = g_j[d2]; g_p[d2 + 1] = g_p[d2]; }
        g_i[d2 + 1] = ti; g_j[d2 + 1] = tj; g_p[d2 + 1] = tp;
      }
      unsigned long long dominated = 0;
      for (int d = 0; d < nsig; d++)
        for (int d2 = d + 1; d2 < nsig; d2++) {
          const int nov = min(g_j[d], g_j[d2]) - max(g_i[d], g_i[d2]) + 1;
          if (nov == 0) break;
          const int nn = min(g_j[d] - g_i[d] + 1, g_j[d2] - g_i[d2] + 1);
          if ((float)nov / (float)nn >= 0.8f) { if (g_p[d] > g_p[d2]) dominated |= 1ull << d2; else dominated |= 1ull << d; }
        }
      // ---------------- every surviving cluster is an envelope: unihit Forward score, trace-derived null2
      // HMMER sums n2sc over the whole sequence in position order; per region here (float32 either way)
      const float regsum = n2sum(ireg, jreg);
      seqbias_sum += regsum;
#ifdef WH_RESOLVE_DEBUG
      if (a.dbg && lane == 0) { printf("[resolve] region n2sc sum %.6f; n2sc:", regsum); for (int pos = ireg; pos <= jreg; pos++) printf(" %.3f", __builtin_nontemporal_load(n2sc + pos)); printf("\n"); }
#endif
      for (int d = 0; d < nsig; d++) {
        if (dominated & (1ull << d)) continue;
        const int i2 = g_i[d], j2 = g_j[d], Ld = j2 - i2 + 1;
        const double envsc = gforward_any<false>(m, seq + (i2 - 1), Ld, cu, mx, lane, use_tl ? (const ldbl *)tabL : nullptr);
        const float dc = n2sum(i2, j2);
        add_envelope(i2, j2, (float)envsc, dc);
      }
      RTICK(4);
    }
    // ---------------- A.6 score assembly (float32 where HMMER is float32), as in the scoring kernels
    int decibits = 0;
    wh_pair_detail *dp = (a.detail && lane == 0) ? a.detail + out : nullptr;
    if (nenv > 0) {
      const float fwdsc = rec.fwdsc;
      const float p1 = (float)L / (float)(L + 1);
      const float nullsc = (float)((double)(float)L * log((double)p1) + log(1.0 - (double)p1));
      const float lomega = (float)log(1.0 / 256.0);
      auto flogsum0 = [](float b) -> float {
        const float mxv = b > 0.f ? b : 0.f, mn = b > 0.f ? 0.f : b;
        if (mn == -INFINITY || (mxv - mn) >= 15.7f) return mxv;
        const int idx = (int)((mxv - mn) * 1000.0f);
        return mxv + (float)log(1.0 + exp((double)-idx / 1000.0));
      };
      const float seqbias = flogsum0(lomega + seqbias_sum);
      float pre_score = (float)(((double)fwdsc - (double)nullsc) / LOG2);
      float seq_score = (float)(((double)fwdsc - (double)(nullsc + seqbias)) / LOG2);
      sb2 = flogsum0(lomega + sb2);
      sum_score += (float)((double)(L - Ld_tot) * log((double)((float)L / (float)(L + 3))));
      const float pre2 = (float)(((double)sum_score - (double)nullsc) / LOG2);
      sum_score = (float)(((double)sum_score - (double)(nullsc + sb2)) / LOG2);
      if (Ld_tot > 0 && sum_score > seq_score) { seq_score = sum_score; pre_score = pre2; flags |= WH_FLAG_OVERRIDE; }
      decibits = (int)rint((double)seq_score * 10.0);
      flags |= WH_FLAG_REPORTED;
      if (dp) { dp->seq_score = seq_score; dp->pre_score = pre_score; dp->seqbias_nats = seqbias; }
    }
    if (dp) {
      dp->nregions = rec.nreg;
      dp->nenv = nenv < WH_MAX_ENVELOPES ? nenv : WH_MAX_ENVELOPES;       // (the first ones; the score above is over all of them)
      for (int e = 0; e < dp->nenv; e++) { dp->env_i[e] = env_i[e]; dp->env_j[e] = env_j[e]; dp->envsc[e] = env_sc[e]; dp->domcorr[e] = env_dc[e]; }
    }
    if (lane == 0) { a.decibits[out] = decibits; a.flags[out] = (uint8_t)flags; }
    if (a.stats && lane == 0) {      // shader cycles and 100 MHz ticks of this pair: their ratio is the clock the kernel ran at
      atomicAdd(a.stats + 14, (unsigned long long)(__builtin_readcyclecounter() - t_pair0));
      atomicAdd(a.stats + 15, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - r_pair0));
    }
  }
  }
  if (a.stats && lane == 0) {        // this wave's lifetime since the first workgroup started (100 MHz ticks): sum, maximum, count
    const unsigned long long life = __builtin_amdgcn_s_memrealtime() - *(volatile unsigned long long *)(a.stats + 16);
    atomicAdd(a.stats + 17, life); atomicMax(a.stats + 18, life); atomicAdd(a.stats + 19, 1ull);
  }
}

// One thread per queued pair: the cells of its multidomain regions (region length x model length), the
// quantity the Forward fill, the walk and the envelope rescoring all scale with; and its model.
__global__ void resolve_keys_kernel(const ResolveRec *recs, int n, const DevHMM *hmms, float *keys, int32_t *models, const int32_t *rext, int64_t rext_stride) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const ResolveRec &r = recs[t];
  const int ne = r.nenv < 0 ? 0 : rext ? r.nenv : r.nenv > WH_MAX_ENVELOPES ? WH_MAX_ENVELOPES : r.nenv;
  float cost = 0.f;
  if (rext) {
    const int32_t *xl = rext + (size_t)t * (size_t)rext_stride;
    for (int e = 0; e < ne; e++) if (xl[kRextInts * e + 4]) cost += (float)(xl[kRextInts * e + 1] - xl[kRextInts * e] + 1);
  } else
  for (int e = 0; e < ne; e++)
    if ((r.multi_mask >> e) & 1) cost += (float)(r.rj[e] - r.ri[e] + 1);
  keys[t] = cost * (float)hmms[r.h].M;
  models[t] = r.h;
}

hipError_t launch_resolve_keys(const ResolveRec *recs, int n, const DevHMM *hmms, float *keys, int32_t *models, hipStream_t s, const int32_t *rext, int64_t rext_stride) {
  hipLaunchKernelGGL(resolve_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, recs, n, hmms, keys, models, rext, rext_stride);
  return hipGetLastError();
}

hipError_t launch_resolve(const ResolveArgs &a, int blocks, int waves, size_t lds, hipStream_t s) {
  if (waves < 1 || waves > kResMaxWaves) return hipErrorInvalidValue;
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(64 * waves), lds, s, a);
  return hipGetLastError();
}

}  // namespace wh
