// C ABI of libwitch_hip.so (declared in include/witch_hip.h): handle management, device
// workspace, kernel launches and optional HIP-event timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

#include "wh_launch.h"

namespace wh {
const char *last_error();
}

using namespace wh;

#define HIPCHK(expr)                                                                  \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return WH_EHIP;                                                                 \
    }                                                                                 \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return WH_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) {
      set_error("hipMalloc of %zu bytes failed", want);
      return WH_ENOMEM;
    }
    cap = want;
    return WH_OK;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
};

struct KernelTimer {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double ms = 0.0;
  int launches = 0;
  bool pending = false;
};

// Development knobs.  Read from the environment ONCE, at wh_ehmm_load;
// wh_set_option changes them on a live handle (tools/ab_score.py).  None is needed in production.
struct Knobs {
  int kernel = 7;            // 7 phase-call scoring kernel (one query per wavefront); 8 its second compilation (A/B slot);
                             // 9 two queries per wavefront where a batch fits (wh_score9.hip; measured slower, kept for A/B: DESIGN.md);
                             // 10 staged launches (wh_staged.hip) for the size classes and batches they serve, 7 for the rest
  float keep_scale = 0.f;    // Forward-row spill threshold relative to E(row); 0 = the kernel's default
  int spill_band = 1;        // 0: envelope Forward rows stored at every lane block that passes keep_scale (A/B; the two-query kernel has no band)
  int max_waves = 0;         // cap on waves per workgroup (0 = planner's choice)
  bool force_specg = false;  // force the HBM special-state mode
  bool no_logspace = false;  // skip the log-space alignment pass
  bool no_wide_align = false; // models beyond 3 072 nodes are aligned by the float64 kernel only (A/B and debugging)
  bool no_window = false;    // envelope Backward sweeps run full width (no node window; A/B and debugging)
  bool no_p2win = false;     // the multihit Backward sweep runs full width only (A/B and debugging)
  bool no_long_list = false; // WH_NO_LONG_LIST: pairs with more than WH_MAX_ENVELOPES regions keep WH_FLAG_TRUNC (no second pass; tests)
  bool no_resolve = false;   // multidomain regions stay ONE envelope (round-1 behaviour) instead of HMMER's stochastic resolver
  int rqueue_cap = 0;        // test hook: size the resolver's queue for this many pairs instead of the estimate (forces the overflow re-run)
  int item_g = 0;            // queries per wave in a work item of the phase-call kernels (0 = 32; A/B)
  int st_units = 0;          // staged launches: envelope units (Forward slabs) per batch (0 = sized from the free HBM)
  bool stats = false, trace = false;
  int dbg = 0;
  int rdbg = 0;              // resolver: print the first <n> sampled segments and the cluster statistics of every region
};
static const int kScorePathSlot = 128;   // d_counter[128..143]: eight 64-bit counters of the last scoring call: six paths (wh_last_score_paths), bytes of Forward rows stored, spare (wh_last_score_counters); [96..123] belong to wh_align_dev
static const int kLongListSlot = 148;    // d_counter[148]: pairs flagged WH_FLAG_TRUNC after the resolver (long-list pass)
static const int kResolveErrSlot = 146;  // d_counter[140]: queue records the resolver found in a segment of another model (never, for a well-formed segment list)
static const int kStagedMaxBatches = 1 << 15;   // staged launches: batches per scoring call (32 counters each: 4 MB)
static const int kMaxLaunches = 60;   // work-queue heads in d_counter (slot 63 belongs to the consensus kernel)

struct wh_ehmm {
  Knobs knobs;
  int device = 0;
  int alphabet = 0, K = 0, Kp = 0;
  int cu_count = 256;
  std::vector<HostHMM> hmms;
  std::vector<DevHMM> dev;          // host copy of the descriptors
  std::map<int, std::vector<int32_t>> by_q;   // Q class -> model positions
  std::vector<int32_t> generic;               // models beyond the register-resident classes (wh_generic.hip)
  std::vector<int32_t> generic_front;         // ... of them, those SCORED by the float64 front end (the others: wide_by_w)
  std::map<int, std::vector<int32_t>> wide_by_w;   // cells per lane * 16 + waves per pair -> models scored by wh_score_wide.hip (3 073 - 12 288 nodes)
  unsigned resolver_launches = 0;             // see ResolveArgs::launch_id
  int force_wide_q = 0;                       // WH_FORCE_WIDE=<4|12|24>: cells per lane of every model's wide tables (tests)
  bool force_wide = false;                    // WH_FORCE_WIDE: EVERY model is scored by the wide kernel (test hook)
  DevBuf d_wscratch;                          // Forward slabs of the wide kernel's workgroups
  DevBuf d_hmms, d_tables, d_nseq, d_index, d_lists, d_counter, d_scratch;
  DevBuf d_ascratch;                        // per-wave slabs of the alignment kernels (allocated while the scoring kernels run)
  DevBuf d_gtab, d_rrecs, d_rmx, d_rsegs;   // multidomain resolver: float64 tables, pair queue, matrix slabs, segment arrays
  int last_resolved = 0;                    // pairs the resolver finished in the last wh_score call
  int64_t last_long_list = 0;               // ... of them, pairs of the long-list pass (more than WH_MAX_ENVELOPES regions)
  DevBuf d_tlist, d_rext;                   // long-list pass: pair positions, their region lists
  int64_t rq_cap = 0;                       // records the queue of the current scoring call holds
  double rq_rate = 0.0;                     // largest share of queued pairs any call on this handle has seen (sizes the next queue)
  int64_t rq_floor = 0;                     // ... at least this many (set when a call overflowed its estimate; the call then runs again)
  int last_queue_reruns = 0;                // scoring passes the last wh_score call repeated because its queue overflowed (0 or 1)
  // staged launches (wh_staged.hip): per-batch state in HBM
  DevBuf d_p2bak;                           // 20- / 24-cell classes: P1's per-row arrays of every resident wave while its P2 window sweep works in place
  DevBuf d_st_pairs, d_st_p1spec, d_st_units, d_st_p3spec, d_st_slabs, d_st_cnt;
  double st_upp = 1.25;                     // envelope units per pair the next call's batches are sized for (learned: 1.25 x the largest seen)
  int st_last_NB = 0;                       // pairs per batch of the last full-split class launch
  bool st_off = false;                      // a batch of the current call ran out of units: the call is repeated with the fused kernel
  int last_staged_batches = 0;              // batches the staged launches of the last scoring call went through
  std::vector<int> st_cnt_host;             // the batches' counters of the last call (read back once, at the end of the scoring pass)
  uint8_t *path_buf = nullptr;              // wh_set_path_buffer: device array [nq x H] the next scoring calls fill with WH_PATH_* bits
  // staging for the host-pointer entry points
  DevBuf s_res, s_off, s_deci, s_flags, s_fwd, s_det, s_idx, s_w, s_nk, s_nu, s_pq, s_ph, s_co, s_cols, s_pos;
  DevBuf d_rkeys, d_rorder, d_rchunks, d_qorder, d_order, d_items, d_recs, d_spec, d_back, d_cwj, d_cwv, d_cwn, d_crow, c_buf[10];
  uint32_t degen[32];
  bool timing = false;
  KernelTimer timers[5];
  int max_M = 0;
  int max_Q = 4;                              // largest cells-per-lane of any model (sizes the float64 slabs)
  int last_align_redo = 0;          // pairs of the last wh_align call that went through the log-space pass
  int last_align_unaligned = 0;     // ... that the any-size kernel could not align (float64 range)
  int64_t last_align_paths[4] = {0, 0, 0, 0};   // pairs of the last wh_align call: 256-node window, window rejected, no window, 512-node window
  std::vector<int64_t> last_unaligned_pairs;   // their pair numbers (wh_last_align_status)
  // timing only: one event in front of every scoring launch of the last call (+ one behind the last), its cells-per-lane class
  // and kernel family (0 phase-call, 1 pass-synchronous, 2 any-size front end): wh_last_score_launches
  std::vector<hipEvent_t> cls_ev;
  std::vector<int> cls_q, cls_kind;
  int cls_n = 0;
};

static int g_device = -1;
static inline bool kn_trace(const wh_ehmm *e) { return e->knobs.trace; }

extern "C" {

const char *wh_version(void) { return "witch_hip 0.1.0 (gfx950)"; }
const char *wh_last_error(void) { return wh::last_error(); }

int wh_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device visible");
    return WH_ENODEV;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range (0..%d)", device, n - 1);
    return WH_EINVAL;
  }
  HIPCHK(hipSetDevice(device));
  (void)hipFree(nullptr);        // creates the device context now (otherwise the first hipMalloc of wh_ehmm_load pays for it)
  {                              // ... and the host-to-device copy path (its first use costs ~0.1 s)
    void *tmp = nullptr;
    int word = 0;
    if (hipMalloc(&tmp, 256) == hipSuccess) { (void)hipMemcpy(tmp, &word, sizeof word, hipMemcpyHostToDevice); (void)hipFree(tmp); }
  }
  g_device = device;
  return WH_OK;
}

int wh_device_info(char *name, int name_len, int *cu_count, int64_t *hbm_bytes) {
  if (g_device < 0) { int rc = wh_init(0); if (rc) return rc; }
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, g_device));
  if (name && name_len > 0) { strncpy(name, p.gcnArchName, (size_t)name_len - 1); name[name_len - 1] = 0; }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return WH_OK;
}

int wh_digitize(int alphabet, const char *text, int64_t n, uint8_t *out) {
  int K, Kp;
  if (alphabet_sizes(alphabet, &K, &Kp) != 0 || !text || !out) { set_error("wh_digitize: bad argument"); return WH_EINVAL; }
  return digitize(alphabet, text, n, out);
}

void wh_ehmm_free(wh_ehmm *e) {
  if (!e) return;
  for (DevBuf *b : {&e->d_tlist, &e->d_rext, &e->d_gtab, &e->d_rrecs, &e->d_rmx, &e->d_rsegs, &e->d_hmms, &e->d_tables, &e->d_nseq, &e->d_index, &e->d_lists, &e->d_counter, &e->d_scratch, &e->d_ascratch, &e->d_wscratch,
                    &e->s_res, &e->s_off, &e->s_deci, &e->s_flags, &e->s_fwd, &e->s_det, &e->s_idx, &e->s_w,
                    &e->s_nk, &e->s_nu, &e->s_pq, &e->s_ph, &e->s_co, &e->s_cols, &e->s_pos, &e->d_rkeys, &e->d_rorder, &e->d_rchunks, &e->d_qorder, &e->d_order, &e->d_items, &e->d_recs, &e->d_spec, &e->d_back, &e->d_cwj, &e->d_cwv, &e->d_cwn, &e->d_crow,
                    &e->c_buf[0], &e->c_buf[1], &e->c_buf[2], &e->c_buf[3], &e->c_buf[4], &e->c_buf[5], &e->c_buf[6],
                    &e->c_buf[7], &e->c_buf[8], &e->c_buf[9],
                    &e->d_p2bak, &e->d_st_pairs, &e->d_st_p1spec, &e->d_st_units, &e->d_st_p3spec, &e->d_st_slabs, &e->d_st_cnt})
    b->release();
  for (hipEvent_t ev : e->cls_ev) (void)hipEventDestroy(ev);
  for (auto &t : e->timers) {
    if (t.e0) (void)hipEventDestroy(t.e0);
    if (t.e1) (void)hipEventDestroy(t.e1);
  }
  delete e;
}

static void knobs_from_env(wh_ehmm *e);

wh_ehmm *wh_ehmm_load(const char *const *hmm_paths, const int32_t *hmm_index, const int32_t *nseq, int n) {
  if (!hmm_paths || n <= 0) { set_error("wh_ehmm_load: no models"); return nullptr; }
  if (g_device < 0 && wh_init(0) != WH_OK) return nullptr;
  std::unique_ptr<wh_ehmm, void (*)(wh_ehmm *)> e(new wh_ehmm, wh_ehmm_free);
  e->device = g_device;
  const bool trace_load = getenv("WH_TRACE") != nullptr;
  auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_l0 = now_ms();
  e->hmms.resize((size_t)n);
  std::vector<float> tables;
  std::vector<double> gtab;
  e->dev.resize((size_t)n);
  // Parsing the text files and laying out the tables is host work per model (a few ms per 1 500-node model):
  // done on a small thread pool, then concatenated in model order so that the buffers do not depend on timing.
  struct Built { int rc = WH_OK; std::string err; int Q = -1, wideW = 0; std::vector<float> fw, bw, em, emn, wfw, wbw, wem; std::vector<double> gfw, gem, gsum; };
  // WH_FORCE_WIDE=<4|24>: every model that fits 8 waves of that many cells per lane ALSO gets wide tables and is scored by the
  // several-waves-per-pair kernel (tests run the golden cases through it; production: models beyond 3 072 nodes only)
  const int force_wide_q = getenv("WH_FORCE_WIDE") ? atoi(getenv("WH_FORCE_WIDE")) : 0;
  e->force_wide = force_wide_q == 4 || force_wide_q == kWideQ || force_wide_q == kWideQReg || force_wide_q == kWideQReg2;
  e->force_wide_q = e->force_wide ? force_wide_q : 0;
  const bool force_wide = e->force_wide;
  // cells per lane of a model's wide tables: 12 (transition tables in registers) up to 6 144 nodes, 24 beyond
  // (12 up to 6 144 nodes, 16 up to 8 192: transition tables in registers; 24 beyond: tables from L2)
  auto wide_q_of = [force_wide, force_wide_q](int M) {
    if (force_wide) return force_wide_q;
    return M <= kWideQReg * kWave * kWideWavesMax ? kWideQReg : M <= kWideQReg2 * kWave * kWideWavesMax ? kWideQReg2 : kWideQ;
  };
  std::vector<Built> built((size_t)n);
  {
    std::atomic<int> next{0};
    auto work = [&]() {
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= n) break;
        HostHMM &h = e->hmms[(size_t)i];
        Built &b = built[(size_t)i];
        if (parse_hmm_file(hmm_paths[i], h) != WH_OK) { b.rc = WH_EIO; b.err = last_error(); continue; }
        h.index = hmm_index ? hmm_index[i] : i;
        if (nseq) h.nseq = nseq[i];
        b.Q = choose_Q(h.M);
        if (b.Q < 0) continue;
        if (b.Q <= kMaxQ) build_tables(h, b.Q, b.fw, b.bw, b.em);     // (the any-size kernels read the float64 tables only)
        if (b.Q > kMaxQ || force_wide) {
          const int wide_q = wide_q_of(h.M);
          const int ww = (h.M + kWave * wide_q - 1) / (kWave * wide_q);
          if (ww <= kWideWavesMax) { b.wideW = ww; build_tables(h, wide_q, b.wfw, b.wbw, b.wem, ww * kWave); }
        }
        // node-major float32 odds of the canonical residues: the resolver's null2-by-trace reads the K values of ONE
        // node together (one lane per sampled position), not K lane-blocked arrays
        b.emn.assign((size_t)(h.M + 1) * h.K, 0.f);
        for (int k = 1; k <= h.M; k++)
          for (int x = 0; x < h.K; x++) b.emn[(size_t)k * h.K + x] = (float)h.odds[(size_t)x * (h.M + 1) + k];
        build_tables_f64(h, b.Q, b.gfw, b.gem);
        // prefix sums over the nodes of those float32 odds, in double: the null2 vector of a sampled domain is a handful of
        // differences of these rows (one per run of match states) instead of one table row per residue (wh_resolve.hip)
        b.gsum.assign((size_t)(h.M + 1) * h.K, 0.0);
        for (int k = 1; k <= h.M; k++)
          for (int x = 0; x < h.K; x++) b.gsum[(size_t)k * h.K + x] = b.gsum[(size_t)(k - 1) * h.K + x] + (double)b.emn[(size_t)k * h.K + x];
      }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = std::max(1, std::min<int>(n, (int)std::min<unsigned>(16u, hw ? hw : 4u)));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
  }
  const double t_l1 = now_ms();
  for (int i = 0; i < n; i++) {
    HostHMM &h = e->hmms[(size_t)i];
    Built &b = built[(size_t)i];
    if (b.rc != WH_OK) { set_error("%s", b.err.c_str()); return nullptr; }
    if (i == 0) { e->alphabet = h.alphabet; e->K = h.K; e->Kp = h.Kp; }
    else if ((h.alphabet == WH_ALPH_AMINO) != (e->alphabet == WH_ALPH_AMINO)) {
      set_error("%s: alphabet differs from the first model", hmm_paths[i]);
      return nullptr;
    }
    const int Q = b.Q;
    if (Q < 0) {
      set_error("%s: model length %d exceeds this build's limit of %d nodes", hmm_paths[i], h.M, kMaxQGen * kWave);
      return nullptr;
    }
    DevHMM &d = e->dev[(size_t)i];
    d.M = h.M; d.Q = Q; d.Mpad = Q * kWave; d.K = h.K; d.Kp = h.Kp; d.nseq = h.nseq; d.index = h.index; d.qclass = Q;
    d.fw_off = (int64_t)tables.size(); tables.insert(tables.end(), b.fw.begin(), b.fw.end());
    d.bw_off = (int64_t)tables.size(); tables.insert(tables.end(), b.bw.begin(), b.bw.end());
    d.em_off = (int64_t)tables.size(); tables.insert(tables.end(), b.em.begin(), b.em.end());
    tables.resize((tables.size() + 3) / 4 * 4, 0.f);          // 16-byte aligned: the node-major rows are read as float4
    d.emn_off = (int64_t)tables.size(); tables.insert(tables.end(), b.emn.begin(), b.emn.end());
    d.wideQ = 0; d.wideW = 0;
    if (b.wideW > 0) {
      tables.resize((tables.size() + 3) / 4 * 4, 0.f);
      d.wideQ = wide_q_of(h.M); d.wideW = b.wideW;
      d.wfw_off = (int64_t)tables.size(); tables.insert(tables.end(), b.wfw.begin(), b.wfw.end());
      d.wbw_off = (int64_t)tables.size(); tables.insert(tables.end(), b.wbw.begin(), b.wbw.end());
      d.wem_off = (int64_t)tables.size(); tables.insert(tables.end(), b.wem.begin(), b.wem.end());
      e->wide_by_w[d.wideQ * 16 + b.wideW].push_back(i);
    }
    d.gfw_off = (int64_t)gtab.size(); gtab.insert(gtab.end(), b.gfw.begin(), b.gfw.end());
    d.gem_off = (int64_t)gtab.size(); gtab.insert(gtab.end(), b.gem.begin(), b.gem.end());
    gtab.resize((gtab.size() + 1) & ~(size_t)1, 0.0);
    d.esum_off = (int64_t)gtab.size(); gtab.insert(gtab.end(), b.gsum.begin(), b.gsum.end());
    gtab.resize((gtab.size() + 1) & ~(size_t)1, 0.0);
    b = Built();                                              // release the per-model copies as we go
    if (Q <= kMaxQ) e->by_q[Q].push_back(i);
    else { e->generic.push_back(i); if (d.wideW == 0) e->generic_front.push_back(i); }
    e->max_Q = std::max(e->max_Q, Q);
    e->max_M = std::max(e->max_M, h.M);
  }
  degen_masks(e->alphabet, e->degen);
  knobs_from_env(e.get());
  const double t_l2 = now_ms();
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, e->device) == hipSuccess) e->cu_count = p.multiProcessorCount;
  std::vector<int32_t> ns((size_t)n), ix((size_t)n);
  for (int i = 0; i < n; i++) { ns[(size_t)i] = e->hmms[(size_t)i].nseq; ix[(size_t)i] = e->hmms[(size_t)i].index; }
  if (e->d_hmms.ensure(sizeof(DevHMM) * (size_t)n) || e->d_tables.ensure(sizeof(float) * tables.size()) ||
      e->d_nseq.ensure(sizeof(int32_t) * (size_t)n) || e->d_index.ensure(sizeof(int32_t) * (size_t)n) ||
      e->d_lists.ensure(sizeof(int32_t) * (size_t)(2 * n + 4)) || e->d_counter.ensure(1024) || e->d_gtab.ensure(sizeof(double) * gtab.size()))
    return nullptr;
  if (hipMemset(e->d_counter.p, 0, 1024) != hipSuccess) { set_error("hipMemset of the counter block failed"); return nullptr; }
  const double t_l3 = now_ms();
  auto up = [&](void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess; };
  std::vector<int32_t> lists;
  for (auto &kv : e->by_q) lists.insert(lists.end(), kv.second.begin(), kv.second.end());
  lists.insert(lists.end(), e->generic_front.begin(), e->generic_front.end());      // after the size classes
  for (auto &kv : e->wide_by_w) lists.insert(lists.end(), kv.second.begin(), kv.second.end());   // ... and the wide classes
  if (!up(e->d_hmms.p, e->dev.data(), sizeof(DevHMM) * (size_t)n) || !up(e->d_tables.p, tables.data(), sizeof(float) * tables.size()) ||
      !up(e->d_nseq.p, ns.data(), sizeof(int32_t) * (size_t)n) || !up(e->d_index.p, ix.data(), sizeof(int32_t) * (size_t)n) ||
      !up(e->d_lists.p, lists.data(), sizeof(int32_t) * lists.size()) || !up(e->d_gtab.p, gtab.data(), sizeof(double) * gtab.size())) {
    set_error("upload of the eHMM tables failed");
    return nullptr;
  }
  if (trace_load) fprintf(stderr, "[wh] eHMM load: %d models parsed + tables built in %.1f ms, concatenated in %.1f ms, device buffers allocated in %.1f ms, uploaded (%.1f MB float + %.1f MB float64) in %.1f ms\n", n,
                          t_l1 - t_l0, t_l2 - t_l1, t_l3 - t_l2, tables.size() * 4e-6, gtab.size() * 8e-6, now_ms() - t_l3);
  return e.release();
}

int wh_ehmm_count(const wh_ehmm *e) { return e ? (int)e->hmms.size() : WH_EINVAL; }
int wh_ehmm_alphabet(const wh_ehmm *e) { return e ? e->alphabet : WH_EINVAL; }

int wh_ehmm_info(const wh_ehmm *e, int32_t *M, int32_t *nseq, int32_t *hmm_index) {
  if (!e) { set_error("null handle"); return WH_EINVAL; }
  for (size_t i = 0; i < e->hmms.size(); i++) {
    if (M) M[i] = e->hmms[i].M;
    if (nseq) nseq[i] = e->hmms[i].nseq;
    if (hmm_index) hmm_index[i] = e->hmms[i].index;
  }
  return WH_OK;
}

int wh_ehmm_map(const wh_ehmm *e, int h, int32_t *map_cols) {
  if (!e || h < 0 || h >= (int)e->hmms.size() || !map_cols) { set_error("wh_ehmm_map: bad argument"); return WH_EINVAL; }
  const HostHMM &m = e->hmms[(size_t)h];
  for (int k = 1; k <= m.M; k++) map_cols[k - 1] = m.map[(size_t)k];
  return WH_OK;
}

int wh_set_option(wh_ehmm *e, const char *name, const char *value) {
  if (!e || !name) { set_error("wh_set_option: bad argument"); return WH_EINVAL; }
  const char *v = value ? value : "";
  const bool on = *v && strcmp(v, "0") != 0;
  Knobs &k = e->knobs;
  if (!strcmp(name, "WH_SCORE_KERNEL")) {
    const int kv = *v ? atoi(v) : 7;
    if (kv < 7 || kv > 12) { set_error("WH_SCORE_KERNEL=%s: this build has kernels 7 to 12", v); return WH_EINVAL; }
    k.kernel = kv;
  } else if (!strcmp(name, "WH_KEEP_LOG2")) k.keep_scale = *v ? ldexpf(1.0f, atoi(v)) : 0.f;
  else if (!strcmp(name, "WH_SPILL_BAND")) k.spill_band = *v ? atoi(v) : 1;
  else if (!strcmp(name, "WH_MAX_WAVES")) k.max_waves = *v ? std::max(1, std::min(16, atoi(v))) : 0;
  else if (!strcmp(name, "WH_FORCE_SPECG")) k.force_specg = on;
  else if (!strcmp(name, "WH_NO_LOGSPACE")) k.no_logspace = on;
  else if (!strcmp(name, "WH_NO_RESOLVE")) k.no_resolve = on;
  else if (!strcmp(name, "WH_NO_LONG_LIST")) k.no_long_list = on;
  else if (!strcmp(name, "WH_NO_WINDOW")) k.no_window = on;
  else if (!strcmp(name, "WH_NO_P2WIN")) k.no_p2win = on;
  else if (!strcmp(name, "WH_RQUEUE_CAP")) k.rqueue_cap = *v ? std::max(1, atoi(v)) : 0;
  else if (!strcmp(name, "WH_ITEM_G")) k.item_g = *v ? std::max(1, std::min(1024, atoi(v))) : 0;
  else if (!strcmp(name, "WH_ST_UNITS")) k.st_units = *v ? std::max(16, atoi(v)) : 0;
  else if (!strcmp(name, "WH_NO_WIDE_ALIGN")) k.no_wide_align = on;
  else if (!strcmp(name, "WH_STATS")) k.stats = on;
  else if (!strcmp(name, "WH_TRACE")) k.trace = on;
  else if (!strcmp(name, "WH_DBG")) k.dbg = atoi(v);
  else if (!strcmp(name, "WH_RDBG")) k.rdbg = atoi(v);
  else { set_error("wh_set_option: unknown option %s", name); return WH_EINVAL; }
  return WH_OK;
}

static void knobs_from_env(wh_ehmm *e) {
  for (const char *name : {"WH_SCORE_KERNEL", "WH_ITEM_G", "WH_ST_UNITS", "WH_KEEP_LOG2", "WH_MAX_WAVES", "WH_FORCE_SPECG", "WH_NO_LOGSPACE", "WH_NO_RESOLVE", "WH_NO_LONG_LIST", "WH_NO_WINDOW", "WH_NO_P2WIN", "WH_RQUEUE_CAP", "WH_NO_WIDE_ALIGN", "WH_STATS", "WH_TRACE", "WH_DBG", "WH_RDBG"})
    if (const char *v = getenv(name)) (void)wh_set_option(e, name, v);
}

int wh_set_timing(wh_ehmm *e, int enabled) {
  if (!e) return WH_EINVAL;
  e->timing = enabled != 0;
  return WH_OK;
}

// one event per scoring launch (timing mode only); events are created once and reused
static int class_mark(wh_ehmm *e, hipStream_t s, int Q, int kind) {
  if (!e->timing) return WH_OK;
  if ((int)e->cls_ev.size() <= e->cls_n) { hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); e->cls_ev.push_back(ev); e->cls_q.push_back(0); e->cls_kind.push_back(0); }
  HIPCHK(hipEventRecord(e->cls_ev[(size_t)e->cls_n], s));
  e->cls_q[(size_t)e->cls_n] = Q; e->cls_kind[(size_t)e->cls_n] = kind;
  e->cls_n++;
  return WH_OK;
}

static int timer_begin(wh_ehmm *e, int which, hipStream_t s) {
  KernelTimer &t = e->timers[which];
  t.pending = false; t.ms = 0.0; t.launches = 0;
  if (!e->timing) return WH_OK;
  if (!t.e0) { HIPCHK(hipEventCreate(&t.e0)); HIPCHK(hipEventCreate(&t.e1)); }
  HIPCHK(hipEventRecord(t.e0, s));
  return WH_OK;
}
static int timer_end(wh_ehmm *e, int which, hipStream_t s, int launches) {
  KernelTimer &t = e->timers[which];
  t.launches = launches;
  if (!e->timing) return WH_OK;
  HIPCHK(hipEventRecord(t.e1, s));
  t.pending = true;
  return WH_OK;
}

int wh_last_align_status(wh_ehmm *e, int64_t *n_logspace, int64_t *n_unaligned, int64_t *unaligned_pairs, int64_t cap) {
  if (!e || cap < 0 || (cap > 0 && !unaligned_pairs)) { set_error("wh_last_align_status: bad argument"); return WH_EINVAL; }
  if (n_logspace) *n_logspace = e->last_align_redo;
  if (n_unaligned) *n_unaligned = (int64_t)e->last_unaligned_pairs.size();
  for (int64_t t = 0; t < cap && t < (int64_t)e->last_unaligned_pairs.size(); t++) unaligned_pairs[t] = e->last_unaligned_pairs[(size_t)t];
  return WH_OK;
}

int wh_last_align_paths(wh_ehmm *e, int64_t *paths4) {
  if (!e || !paths4) { set_error("wh_last_align_paths: bad argument"); return WH_EINVAL; }
  for (int t = 0; t < 4; t++) paths4[t] = e->last_align_paths[t];
  return WH_OK;
}

int wh_last_score_paths(wh_ehmm *e, int64_t *paths6) {
  int64_t *paths4 = paths6;
  if (!e || !paths4) { set_error("wh_last_score_paths: bad argument"); return WH_EINVAL; }
  HIPCHK(hipSetDevice(e->device));
  unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
  // (the counters stay on the device until the next scoring call resets them; this copy waits for the device)
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(v, (int *)e->d_counter.p + kScorePathSlot, sizeof v, hipMemcpyDeviceToHost));
  for (int t = 0; t < 6; t++) paths4[t] = (int64_t)v[t];
  return WH_OK;
}

int wh_last_score_counters(wh_ehmm *e, int64_t *out8) {
  if (!e || !out8) { set_error("wh_last_score_counters: bad argument"); return WH_EINVAL; }
  HIPCHK(hipSetDevice(e->device));
  unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(v, (int *)e->d_counter.p + kScorePathSlot, sizeof v, hipMemcpyDeviceToHost));
  for (int t = 0; t < 8; t++) out8[t] = (int64_t)v[t];
  out8[7] = e->last_long_list;
  return WH_OK;
}

int wh_set_path_buffer(wh_ehmm *e, uint8_t *d_paths) {
  if (!e) { set_error("wh_set_path_buffer: null handle"); return WH_EINVAL; }
  e->path_buf = d_paths;
  return WH_OK;
}

int wh_last_queue_reruns(wh_ehmm *e) {
  if (!e) { set_error("wh_last_queue_reruns: null handle"); return WH_EINVAL; }
  return e->last_queue_reruns;
}

int wh_last_score_launches(wh_ehmm *e, int32_t *cells_per_lane, int32_t *kind, double *ms, int cap) {
  if (!e || cap < 0) { set_error("wh_last_score_launches: bad argument"); return WH_EINVAL; }
  const int n = e->cls_n > 0 ? e->cls_n - 1 : 0;
  for (int t = 0; t < n && t < cap; t++) {
    HIPCHK(hipEventSynchronize(e->cls_ev[(size_t)t + 1]));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, e->cls_ev[(size_t)t], e->cls_ev[(size_t)t + 1]));
    if (cells_per_lane) cells_per_lane[t] = e->cls_q[(size_t)t];
    if (kind) kind[t] = e->cls_kind[(size_t)t];
    if (ms) ms[t] = f;
  }
  return n;
}

int wh_last_kernel_ms(wh_ehmm *e, int which, double *ms, int *launches) {
  if (!e || which < 0 || which > 4) return WH_EINVAL;
  KernelTimer &t = e->timers[which];
  if (t.pending) {
    HIPCHK(hipEventSynchronize(t.e1));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, t.e0, t.e1));
    t.ms = f;
    t.pending = false;
  }
  if (ms) *ms = t.ms;
  if (launches) *launches = t.launches;
  return WH_OK;
}

// ------------------------------------------------------------------------------------ score
static const size_t kLdsBudget = 160 * 1024 - 512;
static const size_t kLdsHeader = 16;   // work-item slot in front of the tables (keeps them 16-byte aligned)

// per-row special-state arrays of a wave's LDS block in the phase-call scoring kernel (wh_score7.hip is built with
// WH_SLIM_SPEC: N, B, E, J, C, scale; an envelope's mask words share the B / E slots)
static const int kScoreSpecArrays = 6;
// LDS plan of the phase-call scoring kernel: tables (K emission rows + both transition
// orientations) + per wave one block (special-state arrays, null2 table, region list, residues).
static int plan_block1(const wh_ehmm *e, int Q, int K, int Lcap, int wmax, int *waves, int *SP, int *wave_lds, size_t *lds, int extra_arrays = 0) {
  const int sp = (Lcap + 1 + 3) / 4 * 4;
  const int wl = (kScoreSpecArrays + extra_arrays) * sp + 32 + kRegsInts + (Lcap + 3) / 4 + 4;
  const size_t table = (size_t)(K + 2 * FW_NARR) * Q * kWave * sizeof(float);
  int w = wmax;
  if (e->knobs.max_waves > 0) w = std::max(1, std::min(wmax, e->knobs.max_waves));
  while (w >= 1 && kLdsHeader + table + (size_t)w * wl * sizeof(float) > kLdsBudget) w--;
  if (w < 1) return WH_ERANGE;
  *waves = w; *SP = sp; *wave_lds = wl; *lds = kLdsHeader + table + (size_t)w * wl * sizeof(float);
  return WH_OK;
}

// Resident workgroups are capped so that <per_block> bytes of per-wave workspace each fit in
// about 70 % of the free HBM (the work-item counter loops tolerate fewer workgroups than CUs).
static int clamp_blocks(int blocks, size_t per_block, const DevBuf &have) {
  if (blocks <= 1 || per_block == 0) return blocks;
  if ((size_t)blocks * per_block <= have.cap) return blocks;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return blocks;
  const size_t budget = (size_t)((double)(free_b + have.cap) * 0.7);
  const size_t fit = budget / per_block;
  return (int)std::max<size_t>(1, std::min<size_t>((size_t)blocks, fit));
}

static int score_dev_pass(wh_ehmm *e, const uint8_t *d_residues, const int64_t *d_offsets, int64_t nq,
                          int64_t total_residues, int32_t max_len, int32_t *d_decibits, uint8_t *d_flags,
                          float *d_fwd_bits, wh_pair_detail *d_detail, void *stream, bool *overflow);

// One size class through the staged launches (wh_staged.hip).  <a> arrives with the class's model list, the queries, the
// outputs and the resolver's queue filled in; this function plans the LDS blocks of the three kinds of kernel, sizes the
// batches from the free HBM, and enqueues eight to ten launches per batch - nothing is read back in between: every
// kernel takes its work from device-side lists and counters.  Batches are ranges of the class's work items (model-major,
// a.QB queries each), so a batch holds one or two models' tables worth of pairs.
static int score_staged_class(wh_ehmm *e, ScoreArgs a, int Q, int Lc, hipStream_t s, int *launches, size_t *need_scratch, bool *served) {
  *served = false;                 // (a class whose batch does not fit the staged kernels' LDS plans is left to the fused kernel)
  const Knobs &kn = e->knobs;
  const bool split = kn.kernel == 11;           // 11: P3 and P4 as launches of their own too (one Forward slab per envelope of a batch)
  const int sp = (Lc + 1 + 3) / 4 * 4;
  const int wl = kScoreSpecArrays * sp + 32 + kRegsInts + (Lc + 3) / 4 + 4;           // floats per wave block (plan_block1's, no extra rows)
  const size_t tbl = (size_t)Q * kWave * sizeof(float);
  auto lds_of = [&](int arrays, int waves, int cand) { return kLdsHeader + (size_t)arrays * tbl + (size_t)waves * wl * sizeof(float) + (size_t)cand * sizeof(int); };
  // work items of QB queries; a workgroup draws G of them at a time and deals their candidates to its waves one by one
  const int QB = 48;
  const int G_all = 1, G_most = 4, G_few = 8, G_rare = 32;        // kernels that serve every pair / most / a few per cent / next to none
  auto cap_of = [&](int G, int per_pair) { return std::min(G * QB * per_pair, 2048); };
  // dense kernels: twelve waves beside one orientation (+ the emission rows); the rare dense redo needs both
  int w_one = 12, w_both = 12;
  while (w_one >= 1 && lds_of(e->K + FW_NARR, w_one, cap_of(G_few, WH_MAX_ENVELOPES)) > kLdsBudget) w_one--;
  while (w_both >= 1 && lds_of(e->K + 2 * FW_NARR, w_both, cap_of(G_rare, WH_MAX_ENVELOPES)) > kLdsBudget) w_both--;
  // light kernels: two workgroups per CU, the emission rows only
  int w_p2 = 12, w_p4 = 10;
  while (w_p2 >= 1 && 2 * lds_of(e->K, w_p2, cap_of(G_most, 1)) > kLdsBudget) w_p2--;
  while (w_p4 >= 1 && 2 * lds_of(e->K, w_p4, cap_of(G_most, WH_MAX_ENVELOPES)) > kLdsBudget) w_p4--;
  if (w_one < 4 || w_both < 4 || w_p2 < 4 || w_p4 < 4) return WH_OK;
  *served = true;
  a.SP = sp; a.wave_lds = wl; a.spec_arrays = kScoreSpecArrays;
  a.paths = reinterpret_cast<unsigned long long *>((int *)e->d_counter.p + kScorePathSlot);
  a.p2win = 0; a.qorder = nullptr;
  a.QB = QB;
  a.scratch_stride = (size_t)(Lc + 1) * 2 * Q * kWave;          // the envelope kernel's per-wave Forward slab (as the fused kernel's)
  if (need_scratch) {                                           // planning pass: the caller allocates once for all classes
    if (!split) *need_scratch = std::max(*need_scratch, (size_t)e->cu_count * w_both * a.scratch_stride * sizeof(float));
    return WH_OK;
  }
  a.scratch = (float *)e->d_scratch.p;
  a.n_qblocks = (int)((a.nq + a.QB - 1) / a.QB);
  a.n_items = a.n_list * a.n_qblocks;
  StagedArgs g;
  memset(&g, 0, sizeof g);
  g.slab_stride = (size_t)(Lc + 1) * 2 * Q * kWave;
  g.p1stride = (size_t)kScoreSpecArrays * sp;
  g.p3stride = g.p1stride;
  // ---- batch size.  Full split: units (Forward slabs) from the free HBM, at most sixteen per resident dense wave; pairs =
  // units / (units per pair).  Otherwise a batch is bounded by its per-pair rows alone (3.6 KB per pair at L = 150).
  const int resident = e->cu_count * w_one;
  int64_t NS = (int64_t)resident * 16;
  if (!split) NS = (int64_t)1 << 20;
  if (split) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t have = e->d_st_slabs.cap + e->d_st_p3spec.cap;
      const size_t budget = (size_t)((double)(free_b + have) * 0.45);
      NS = std::min<int64_t>(NS, (int64_t)(budget / ((g.slab_stride + g.p3stride) * sizeof(float))));
    }
  }
  if (kn.st_units > 0) NS = kn.st_units;
  const int64_t total_pairs = (int64_t)a.n_items * a.QB;
  NS = std::min<int64_t>(NS, (int64_t)((double)total_pairs * e->st_upp) + a.QB * WH_MAX_ENVELOPES);
  if (NS < 16) { set_error("staged launches: no HBM for the Forward slabs (Q=%d, L=%d)", Q, Lc); return WH_ENOMEM; }
  const double upp = split ? e->st_upp : 1.0;
  int items_b = (int)std::max<int64_t>(1, (int64_t)((double)NS / upp) / a.QB);
  items_b = std::min(items_b, a.n_items);
  // (batches of equal size, each a multiple of the workgroup count where the class is large enough for that)
  {
    const int nb = (a.n_items + items_b - 1) / items_b;
    items_b = (a.n_items + nb - 1) / nb;
    if (items_b > 2 * e->cu_count) items_b = std::min((items_b + e->cu_count - 1) / e->cu_count * e->cu_count, (int)std::max<int64_t>(1, (int64_t)((double)NS / upp) / a.QB));
  }
  const int NB = items_b * a.QB;
  const int n_batches = (a.n_items + items_b - 1) / items_b;
  if (e->d_st_pairs.ensure(sizeof(StPair) * (size_t)NB) || e->d_st_p1spec.ensure(sizeof(float) * g.p1stride * (size_t)NB) ||
      e->d_st_cnt.ensure(sizeof(int) * 32 * (size_t)kStagedMaxBatches))
    return WH_ENOMEM;
  // (the counters of EVERY batch of the call are read back once, at its end: the block is allocated at its full size the
  // first time - growing it between two size classes of a call would drop the first class's counters)
  if (e->last_staged_batches + n_batches > kStagedMaxBatches) { set_error("staged launches: more than %d batches in one call", kStagedMaxBatches); return WH_ERANGE; }
  if (split && (e->d_st_units.ensure(sizeof(StUnit) * (size_t)NS) || e->d_st_p3spec.ensure(sizeof(float) * g.p3stride * (size_t)NS) ||
                e->d_st_slabs.ensure(sizeof(float) * g.slab_stride * (size_t)NS)))
    return WH_ENOMEM;
  g.NB = NB; g.NS = (int)NS;
  if (split) e->st_last_NB = NB;
  g.pairs = (StPair *)e->d_st_pairs.p; g.p1spec = (float *)e->d_st_p1spec.p;
  g.units = (StUnit *)e->d_st_units.p; g.p3spec = (float *)e->d_st_p3spec.p; g.slabs = (float *)e->d_st_slabs.p;
  g.pair_paths = e->path_buf;
  int *cnt0 = (int *)e->d_st_cnt.p + 32 * (size_t)e->last_staged_batches;
  HIPCHK(hipMemsetAsync(cnt0, 0, sizeof(int) * 32 * (size_t)n_batches, s));
  const bool w512 = Q == 16 || Q == 24;
  if (kn.trace) fprintf(stderr, "[wh] staged Q=%d: %d items of %d queries in %d batches of %d pairs, %lld units (%.1f GB of slabs), waves dense %d / both %d / p2win %d / p4win %d\n",
                        Q, a.n_items, a.QB, n_batches, NB, (long long)NS, (double)NS * g.slab_stride * 4e-9, w_one, w_both, w_p2, w_p4);
  if (class_mark(e, s, Q, 4)) return WH_EHIP;
  if (kn.stats) {
    if (e->d_recs.ensure(512)) return WH_ENOMEM;
    HIPCHK(hipMemsetAsync(e->d_recs.p, 0, 512, s));
    a.stats = (unsigned long long *)e->d_recs.p;
  }
  const int cu = e->cu_count;
  for (int b = 0; b < n_batches; b++) {
    g.a = a;
    g.item0 = b * items_b;
    g.n_items_b = std::min(items_b, a.n_items - g.item0);
    g.cnt = cnt0 + 32 * (size_t)b;
    auto groups = [&](int G) { return (g.n_items_b + G - 1) / G; };
    hipError_t err = hipSuccess;
    auto go = [&](int G, int per_pair) { g.G = G; g.cand_cap = cap_of(G, per_pair); return err == hipSuccess; };
    if (go(G_all, 1)) err = launch_staged_p1(Q, g, std::min(groups(g.G), cu), w_one * kWave, lds_of(e->K + FW_NARR, w_one, g.cand_cap), s);
    if (go(G_most, 1)) err = launch_staged_p2win(Q, 4, g, std::min(groups(g.G), 2 * cu), w_p2 * kWave, lds_of(e->K, w_p2, g.cand_cap), s);
    if (w512 && go(G_few, 1)) err = launch_staged_p2win(Q, 8, g, std::min(groups(g.G), 2 * cu), w_p2 * kWave, lds_of(e->K, w_p2, g.cand_cap), s);
    if (go(G_few, 1)) err = launch_staged_p2full(Q, g, std::min(groups(g.G), cu), w_one * kWave, lds_of(e->K + BW_NARR, w_one, g.cand_cap), s);
    if (!split) {
      if (go(G_all, 1)) err = launch_staged_env(Q, g, std::min(groups(g.G), cu), w_both * kWave, lds_of(e->K + 2 * FW_NARR, w_both, g.cand_cap), s);
      if (err != hipSuccess) { set_error("staged launch (Q=%d, batch %d) failed: %s", Q, b, hipGetErrorString(err)); return WH_EHIP; }
      continue;
    }
    if (go(G_all, WH_MAX_ENVELOPES)) err = launch_staged_p3(Q, g, std::min(groups(g.G), cu), w_one * kWave, lds_of(e->K + FW_NARR, w_one, g.cand_cap), s);
    if (go(G_most, WH_MAX_ENVELOPES)) err = launch_staged_p4win(Q, 4, g, std::min(groups(g.G), 2 * cu), w_p4 * kWave, lds_of(e->K, w_p4, g.cand_cap), s);
    if (w512 && go(G_few, WH_MAX_ENVELOPES)) err = launch_staged_p4win(Q, 8, g, std::min(groups(g.G), 2 * cu), w_p4 * kWave, lds_of(e->K, w_p4, g.cand_cap), s);
    if (go(G_few, WH_MAX_ENVELOPES)) err = launch_staged_p4full(Q, g, std::min(groups(g.G), cu), w_one * kWave, lds_of(e->K + BW_NARR, w_one, g.cand_cap), s);
    if (go(G_rare, WH_MAX_ENVELOPES)) err = launch_staged_dense(Q, g, std::min(groups(g.G), cu), w_both * kWave, lds_of(e->K + 2 * FW_NARR, w_both, g.cand_cap), s);
    if (err == hipSuccess) err = launch_staged_assemble(g, s);
    if (err != hipSuccess) { set_error("staged launch (Q=%d, batch %d) failed: %s", Q, b, hipGetErrorString(err)); return WH_EHIP; }
  }
  e->last_staged_batches += n_batches;
  (*launches)++;
  if (a.stats) {
    unsigned long long st[64];
    HIPCHK(hipMemcpyAsync(st, a.stats, sizeof st, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    static const char *kind[9] = {"p1", "p2win 256", "p2win 512", "p2full", "p3", "p4win 256", "p4win 512", "p4full", "dense"};
    for (int k = 0; k < 9; k++)
      fprintf(stderr, "[wh] staged Q=%d %-10s sweeps %9llu  shader cycles per sweep %10.0f  real time per sweep %8.1f us  (clock %.2f GHz)  wave lifetimes %.3g cycles, in sweeps %.1f%%\n", Q, kind[k], st[4 * k + 3],
              st[4 * k + 3] ? (double)st[4 * k] / (double)st[4 * k + 3] : 0.0, st[4 * k + 3] ? 0.01 * (double)st[4 * k + 1] / (double)st[4 * k + 3] : 0.0,
              st[4 * k + 1] ? 0.1 * (double)st[4 * k] / (double)st[4 * k + 1] : 0.0, (double)st[4 * k + 2], st[4 * k + 2] ? 100.0 * (double)st[4 * k] / (double)st[4 * k + 2] : 0.0);
  }
  return WH_OK;
}

int wh_score_dev(wh_ehmm *e, const uint8_t *d_residues, const int64_t *d_offsets, int64_t nq,
                 int64_t total_residues, int32_t max_len, int32_t *d_decibits, uint8_t *d_flags,
                 float *d_fwd_bits, wh_pair_detail *d_detail, void *stream) {
  if (!e || !d_residues || !d_offsets || !d_decibits || !d_flags || nq < 0 || max_len < 0) {
    set_error("wh_score_dev: bad argument");
    return WH_EINVAL;
  }
  // One call serves fewer than 2^31 pairs (pair numbers and the resolver's queue are 32-bit).  Until round 4 a larger call
  // ran WITHOUT the multidomain resolver and said nothing - a different reported set; now it is refused: the caller feeds
  // the queries in chunks (QueryAlignmentEngine.run: 20 000 at a time, the reference's own hmmsearch chunk).
  if (nq * (int64_t)e->hmms.size() >= 0x7FFFFFFF) {
    set_error("wh_score_dev: %lld queries x %zu models is 2^31 pairs or more; score the queries in chunks", (long long)nq, e->hmms.size());
    return WH_ERANGE;
  }
  // The queue of pairs with a multidomain region is sized by ESTIMATE (a per-pair record is 296 bytes; the worst case,
  // one record per pair, was 3.4 GB at the headline for a class that is 0.005 % of its pairs).  The kernels count every
  // pair that wants a slot; when the count exceeds the capacity, the queue is grown to the count and the scoring pass
  // runs once more (every pair is scored again, so the queue then holds exactly what the first pass counted).
  e->last_queue_reruns = 0;
  e->rq_floor = 0;
  e->st_off = false;
  // (diagnostics only: a pair the kernels leave early - an empty query, one beyond the length cap - has a record of zeros)
  if (d_detail && nq > 0) HIPCHK(hipMemsetAsync(d_detail, 0, sizeof(wh_pair_detail) * (size_t)nq * e->hmms.size(), (hipStream_t)stream));
  bool overflow = false;
  int rc = score_dev_pass(e, d_residues, d_offsets, nq, total_residues, max_len, d_decibits, d_flags, d_fwd_bits, d_detail, stream, &overflow);
  // (two independent reasons to repeat a pass: the resolver's queue, and a staged batch that ran out of envelope units)
  for (int again = 0; rc == WH_OK && overflow; again++) {
    if (again == 2) { set_error("wh_score_dev: the resolver's queue overflowed twice"); rc = WH_ERANGE; break; }
    e->last_queue_reruns++;
    overflow = false;
    rc = score_dev_pass(e, d_residues, d_offsets, nq, total_residues, max_len, d_decibits, d_flags, d_fwd_bits, d_detail, stream, &overflow);
  }
  e->rq_floor = 0;
  e->st_off = false;
  return rc;
}

static int score_dev_pass(wh_ehmm *e, const uint8_t *d_residues, const int64_t *d_offsets, int64_t nq,
                          int64_t total_residues, int32_t max_len, int32_t *d_decibits, uint8_t *d_flags,
                          float *d_fwd_bits, wh_pair_detail *d_detail, void *stream, bool *overflow) {
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(e->device));
  if (timer_begin(e, 0, s)) return WH_EHIP;
  e->cls_n = 0;
  e->last_staged_batches = 0;
  int launches = 0;
  const int32_t *qorder_all = nullptr;     // queries in descending length order, when the call formed it
  bool wide_done = false;
  if (nq > 0) {
    if ((int)e->by_q.size() > kMaxLaunches) { set_error("too many model size classes (%zu)", e->by_q.size()); return WH_ERANGE; }
    const int H = (int)e->hmms.size();
    const Knobs &kn = e->knobs;
    const int Lc = std::max(max_len, 1);
    int list_off = 0;
    // queue of pairs with a multidomain region (finished by resolve_kernel after the scoring launches);
    // one record per pair in the worst case
    const size_t rlds = resolve_lds_bytes(Lc, e->max_M);
    const int64_t npairs_all = nq * (int64_t)H;
    const bool resolve = !kn.no_resolve && rlds <= kLdsBudget && npairs_all < 0x7FFFFFFF;
    int *d_rcount = (int *)e->d_counter.p + 64;      // [64] queue length, [65] work-queue head of the resolver
    e->last_resolved = 0;
    if (resolve) {
      // estimate: 5 % of the pairs (at least 65 536) or 1.25 x the largest share an earlier call on this handle queued,
      // plus every pair of the any-size float64 front end, which hands each pair with a region to the resolver; never
      // more than one record per pair.  (Synthetic family fragments queue 0.005 % of their pairs, the reference's rRNA
      // fragments 28 %: a first call on such data repeats its scoring pass once, later calls are sized by what it saw.)
      int64_t cap = std::max<int64_t>(65536, std::max<int64_t>(npairs_all / 20, (int64_t)(1.25 * e->rq_rate * (double)npairs_all) + 1024)) +
                    nq * (int64_t)e->generic_front.size();
      cap = std::max<int64_t>(cap, (int64_t)(e->d_rrecs.cap / sizeof(ResolveRec)));   // what an earlier call allocated is free to use
      if (kn.rqueue_cap > 0) cap = kn.rqueue_cap;                                      // test hook
      cap = std::min<int64_t>(std::max(cap, e->rq_floor), npairs_all);
      if (e->d_rrecs.ensure(sizeof(ResolveRec) * (size_t)cap)) return WH_ENOMEM;
      e->rq_cap = cap;
      HIPCHK(hipMemsetAsync(d_rcount, 0, 2 * sizeof(int), s));
    }
    HIPCHK(hipMemsetAsync((int *)e->d_counter.p + kScorePathSlot, 0, 8 * sizeof(unsigned long long), s));
    // Long models run four waves in lockstep per workgroup (wh_score_big.hip): hand them the queries in
    // descending length order, so that the waves of a workgroup finish their sweeps together and the longest
    // pairs start first.  (One D2H copy of the offsets and a host sort; only when such a class exists.)
    bool any_long = false;
    for (auto &kv : e->by_q) any_long = any_long || kv.first >= 20;
    any_long = any_long || !e->wide_by_w.empty();
    // ... and the phase-call kernel deals the queries of a work item to its waves in fixed turns: with lengths of
    // 50-2 000 residues in one batch a wave that drew long queries keeps the eleven others waiting at the item's
    // end (about 30 % of the launch on the protein workload) - same cure.  Batches of near-equal lengths (the
    // headline: all 150 nt) skip the copy and the sort.
    const bool mixed = total_residues > 0 && (double)max_len > 1.25 * (double)total_residues / (double)nq;
    const int32_t *d_qorder = nullptr;
    if ((any_long || mixed) && nq > 4 && nq < 0x7FFFFFFF) {
      std::vector<int64_t> offs((size_t)nq + 1);
      HIPCHK(hipMemcpyAsync(offs.data(), d_offsets, sizeof(int64_t) * offs.size(), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      std::vector<int32_t> ord((size_t)nq);
      for (int64_t q = 0; q < nq; q++) ord[(size_t)q] = (int32_t)q;
      std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return offs[x + 1] - offs[x] > offs[y + 1] - offs[y]; });
      if (e->d_qorder.ensure(sizeof(int32_t) * ord.size())) return WH_ENOMEM;
      HIPCHK(hipMemcpyAsync(e->d_qorder.p, ord.data(), sizeof(int32_t) * ord.size(), hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));   // ord is a local
      d_qorder = (const int32_t *)e->d_qorder.p;
      qorder_all = d_qorder;
    }
    // pass 0 sizes the per-wave workspace of every class and allocates ONCE (growing a DevBuf class by class
    // meant a hipFree + hipMalloc of tens of GB per class: ~25 ms per GB); pass 1 launches
    size_t need_scratch = 0, need_spec = 0;
    for (int pass = 0; pass < 2; pass++) {
    list_off = 0;
    if (pass == 1 && (e->d_scratch.ensure(need_scratch) || (need_spec && e->d_spec.ensure(need_spec)))) return WH_ENOMEM;
    for (auto &kv : e->by_q) {
      if (e->force_wide && e->dev[(size_t)kv.second[0]].wideW > 0) {   // test hook: these models go through the wide kernel below
        list_off += (int)kv.second.size();
        continue;
      }
      const int Q = kv.first;
      ScoreArgs a;
      memset(&a, 0, sizeof a);
      a.hmms = (const DevHMM *)e->d_hmms.p;
      a.tables = (const float *)e->d_tables.p;
      a.hmm_list = (const int32_t *)e->d_lists.p + list_off;
      a.n_list = (int)kv.second.size();
      list_off += a.n_list;
      a.residues = d_residues; a.offsets = d_offsets; a.nq = nq;
      a.counter = (int *)e->d_counter.p + launches;
      a.Lcap = Lc;
      a.decibits = d_decibits; a.flags = d_flags; a.fwd_bits = d_fwd_bits; a.detail = d_detail;
      a.H = H; a.K = e->K; a.Kp = e->Kp;
      a.dbg = kn.dbg;
      a.no_window = kn.no_window ? 1 : 0;
      a.keep_scale = kn.keep_scale;
      a.spill_band = kn.kernel != 9 ? kn.spill_band : 0;
      if (resolve) { a.rrecs = (ResolveRec *)e->d_rrecs.p; a.rcount = d_rcount; a.rcap = (int)e->rq_cap; }
      memcpy(a.degen, e->degen, sizeof a.degen);
      int waves = 0, SP = 0, wave_lds = 0;
      size_t lds = 0;
      // Three kernels serve a size class:
      //  * phase-call kernel, special states in LDS: models of up to 24 cells per lane, short queries
      //  * the same kernel with the special-state rows in HBM ("SG"): long queries
      //  * pass-synchronous kernel (wh_score_big.hip): 28+ cells per lane, and 20/24-cell models whose
      //    emission rows do not fit in LDS beside both orientations (protein)
      bool big = Q > kMaxQFast, specg = false, pairk = false, p2win = false, p2inpl = false;
      if (!big && kn.kernel == 9 && !kn.force_specg && (Q == 8 || Q == 12 || Q == 16)) {
        // two queries per wavefront (wh_score9.hip): eight waves, each with two blocks of per-row arrays
        const int sp9 = (Lc + 1 + 3) / 4 * 4;
        const int wl9 = 2 * score9_block_floats(sp9, Lc);
        const size_t table = (size_t)(e->K + 2 * FW_NARR) * Q * kWave * sizeof(float);
        int w9 = 8;
        if (kn.max_waves > 0) w9 = std::max(1, std::min(8, kn.max_waves));
        if (kLdsHeader + table + (size_t)w9 * wl9 * sizeof(float) <= kLdsBudget) {
          pairk = true; waves = w9; SP = sp9; wave_lds = wl9;
          lds = kLdsHeader + table + (size_t)w9 * wl9 * sizeof(float);
        }
      }
      if (!big && !pairk) {
        // (twelve waves = three per SIMD at 168 registers; 20-cell models keep that since the six-array block, 24-cell
        // models get the nine or ten waves that fit beside their 120 KB of tables)
        int rc_plan = plan_block1(e, Q, e->K, Lc, 12, &waves, &SP, &wave_lds, &lds);
        // ... and, where the waves still fit with them, three more per-row arrays per wave: the multihit Backward sweep then
        // tries a node window first (wh_score7.hip, "P2 on a node window")
        if (rc_plan == WH_OK && waves >= 4 && !kn.force_specg && !kn.no_window && !kn.no_p2win && Q >= 8) {
          int w2 = 0, sp2 = 0, wl2 = 0;
          size_t lds2 = 0;
          if (plan_block1(e, Q, e->K, Lc, 12, &w2, &sp2, &wl2, &lds2, 3) == WH_OK && (w2 >= waves || (getenv("WH_P2WIN_FORCE") && w2 >= 8))) { p2win = true; waves = w2; SP = sp2; wave_lds = wl2; lds = lds2; }
          else if (Q >= 20 && kn.kernel != 9) p2inpl = true;      // round 5: the window sweep in place, P1's rows backed up in HBM (ScoreArgs::p2win == 2)
        }
        if (rc_plan != WH_OK || waves < 4 || kn.force_specg) {
          specg = true;
          SP = (Lc + 1 + 3) / 4 * 4;
          wave_lds = 32 + kRegsInts + (Lc + 3) / 4 + 4;
          const size_t table = (size_t)(e->K + 2 * FW_NARR) * Q * kWave * sizeof(float);
          waves = Q <= 16 ? 12 : 8;
          if (kn.max_waves > 0) waves = std::max(1, std::min(waves, kn.max_waves));
          while (waves >= 1 && kLdsHeader + table + (size_t)waves * wave_lds * sizeof(float) > kLdsBudget) waves--;
          rc_plan = waves >= 1 ? WH_OK : WH_ERANGE;
          lds = kLdsHeader + table + (size_t)waves * wave_lds * sizeof(float);
          if (Q >= 20 && (rc_plan != WH_OK || waves < 4)) big = true;
        }
        if (!big && rc_plan != WH_OK) {
          set_error("query length %d with model class Q=%d does not fit in LDS", max_len, Q);
          return WH_ERANGE;
        }
      }
      if (big) {
        wave_lds = 32 + kRegsInts + (Lc + 3) / 4 + 4;
        waves = 4;            // one per SIMD: the long-model kernel uses the whole register file
        a.Klds = e->K;
        size_t table = (size_t)(a.Klds + 8) * Q * kWave * sizeof(float);
        if (kLdsHeader + table + (size_t)waves * wave_lds * sizeof(float) > kLdsBudget) { a.Klds = 0; table = (size_t)8 * Q * kWave * sizeof(float); }
        lds = kLdsHeader + table + (size_t)waves * wave_lds * sizeof(float);
        if (lds > kLdsBudget) { set_error("query length %d with model class Q=%d does not fit in LDS", max_len, Q); return WH_ERANGE; }
        SP = (Lc + 1 + 3) / 4 * 4;
        specg = true;
      }
      // ---- staged launches (wh_staged.hip): short-query batches of the one-wave classes, special states in LDS
      const bool staged = (kn.kernel == 10 || kn.kernel == 11) && !e->st_off && !big && !pairk && !specg && !kn.dbg &&
                          (Q == 8 || Q == 12 || Q == 16 || Q == 20 || Q == 24);
      if (staged) {
        bool served = false;
        int rc_st = score_staged_class(e, a, Q, Lc, s, &launches, pass == 0 ? &need_scratch : nullptr, &served);
        if (rc_st != WH_OK) return rc_st;
        if (served) continue;
      }
      // ---- four envelopes per Backward sweep (score_kernel7q, WH_SCORE_KERNEL=12): 16-cell models, special states in LDS
      bool quadk = false;
      if (kn.kernel == 12 && !big && !pairk && !specg && Q == 16) {
        const int spq = (Lc + 1 + 3) / 4 * 4, seqw = (Lc + 3) / 4 + 4;
        const int wlq = kScoreSpecArrays * spq + 128 + kRegsInts + 4 * 16 + 4 * 16 + 4 * seqw;
        const size_t table = (size_t)(e->K + 2 * FW_NARR) * Q * kWave * sizeof(float);
        int wq = 12;
        if (kn.max_waves > 0) wq = std::max(1, std::min(12, kn.max_waves));
        while (wq >= 1 && kLdsHeader + table + (size_t)wq * wlq * sizeof(float) > kLdsBudget) wq--;
        if (wq >= 8) { quadk = true; waves = wq; SP = spq; wave_lds = wlq; lds = kLdsHeader + table + (size_t)wq * wlq * sizeof(float); }
      }
      a.SP = SP; a.wave_lds = wave_lds; a.spec_arrays = kScoreSpecArrays;
      a.paths = reinterpret_cast<unsigned long long *>((int *)e->d_counter.p + kScorePathSlot);
      a.p2win = (p2win && !specg && !big && !pairk) ? 1 : (p2inpl && !specg && !big && !pairk) ? 2 : 0;
      a.qorder = (big || mixed) ? d_qorder : nullptr;
      // (the phase-call kernels deal an item's queries to the waves one by one, so an item can be large - the wait at its
      // end is one pair's time whatever its size: 32 queries per wave; long models: a pair is milliseconds, smaller items
      // shorten the tail of the launch)
      a.QB = big ? waves * 2 : pairk ? waves * 4 : waves * (kn.item_g > 0 ? kn.item_g : 32);
      const int per_turn = pairk ? 2 : 1;   // queries a wave takes per turn
      // small batches (the reference's example as shipped: 500 fragments x 15 models): with the default item size there
      // are fewer than a handful of items per workgroup and the launch ends on its stragglers - one query per wave and
      // item then (the tables of a model are re-staged more often, which a small batch can afford)
      {
        const int max_blocks = big ? e->cu_count : e->cu_count * std::max(1, 8 / waves);
        const int64_t items_default = (int64_t)a.n_list * ((nq + a.QB - 1) / a.QB);
        if (items_default < 4 * (int64_t)max_blocks) {
          // fewer items than that: smaller ones, down to one query per wave
          a.QB = waves * per_turn;
          if (!big && !pairk) for (int g_ = 16; g_ > 1; g_ /= 2)
            if ((int64_t)a.n_list * ((nq + waves * g_ - 1) / (waves * g_)) >= 4 * (int64_t)max_blocks) { a.QB = waves * g_; break; }
        }
      }
      a.n_qblocks = (int)((nq + a.QB - 1) / a.QB);
      a.n_items = a.n_list * a.n_qblocks;
      a.scratch_stride = (size_t)(quadk ? 5 : per_turn) * (size_t)(a.Lcap + 1) * 2 * Q * kWave;   // Forward slab(s) per wave
      a.spec_stride = specg ? (size_t)8 * a.SP : quadk ? (size_t)(5 * kScoreSpecArrays + 1) * a.SP : 0;
      if (quadk) { specg = true; a.p2win = 0; a.QB = std::max(a.QB, waves * 8); a.n_qblocks = (int)((nq + a.QB - 1) / a.QB); a.n_items = a.n_list * a.n_qblocks; }   // (HBM region per wave; items of two quads per wave)
      int blocks = std::min(a.n_items, big ? e->cu_count : e->cu_count * std::max(1, 8 / waves));
      blocks = clamp_blocks(blocks, (size_t)waves * (a.scratch_stride + a.spec_stride) * sizeof(float), e->d_scratch);
      if (pass == 0) {
        need_scratch = std::max(need_scratch, (size_t)blocks * waves * a.scratch_stride * sizeof(float));
        if (specg) need_spec = std::max(need_spec, (size_t)blocks * waves * a.spec_stride * sizeof(float));
        continue;
      }
      // never more resident workgroups than the workspace allocated after pass 0 holds
      blocks = (int)std::min<size_t>((size_t)blocks, e->d_scratch.cap / ((size_t)waves * a.scratch_stride * sizeof(float)));
      if (specg) blocks = (int)std::min<size_t>((size_t)blocks, e->d_spec.cap / ((size_t)waves * a.spec_stride * sizeof(float)));
      if (blocks < 1) { set_error("workspace planning failed (Q=%d)", Q); return WH_ENOMEM; }
      a.scratch = (float *)e->d_scratch.p;
      if (specg) a.spec_scratch = (float *)e->d_spec.p;
      if (a.p2win == 2) {
        a.p2_backup_stride = (size_t)kScoreSpecArrays * a.SP;
        if (e->d_p2bak.ensure((size_t)blocks * waves * a.p2_backup_stride * sizeof(float))) return WH_ENOMEM;
        a.p2_backup = (float *)e->d_p2bak.p;
      }
      if (kn.stats) {
        if (e->d_recs.ensure(320)) return WH_ENOMEM;
        HIPCHK(hipMemsetAsync(e->d_recs.p, 0, 320, s));
        { unsigned long long bigv = ~0ull; HIPCHK(hipMemcpyAsync((char *)e->d_recs.p + 13 * 8, &bigv, 8, hipMemcpyHostToDevice, s)); }
        a.stats = (unsigned long long *)e->d_recs.p;
      }
      if (kn.trace) fprintf(stderr, "[wh] score Q=%d kernel=%s specg=%d waves=%d blocks=%d lds=%zu SP=%d wave_lds=%d items=%d Lcap=%d\n", Q,
                            big ? "pass-synchronous" : pairk ? "two-queries-per-wave" : kn.kernel == 8 ? "phase-call(B)" : "phase-call", (int)specg, waves, blocks, lds, SP, wave_lds, a.n_items, a.Lcap);
      HIPCHK(hipMemsetAsync(a.counter, 0, sizeof(int), s));
      if (class_mark(e, s, Q, big ? 1 : 0)) return WH_EHIP;
      hipError_t err = big ? launch_score_big(Q, a, blocks, waves * kWave, lds, s)
                       : pairk ? launch_score9(Q, a, blocks, waves * kWave, lds, s)
                       : quadk ? launch_score7q(Q, a, blocks, waves * kWave, lds, s)
                       : kn.kernel == 8 ? launch_score7b(Q, a, blocks, waves * kWave, lds, s)
                                        : launch_score7(Q, a, blocks, waves * kWave, lds, s);
      if (err != hipSuccess) { set_error("score kernel launch (Q=%d) failed: %s", Q, hipGetErrorString(err)); return WH_EHIP; }
      launches++;
      if (a.stats) {
        unsigned long long st[40];
        HIPCHK(hipMemcpyAsync(st, a.stats, sizeof st, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const double tot = (double)(st[4] + st[5] + st[6] + st[7] + st[8] + st[9] + st[10] + st[11]);
        fprintf(stderr, "[wh] Q=%d wave cycles: P1 %.1f%%  P2 %.1f%%  regions %.1f%%  P3 %.1f%%  P4 %.1f%%  null2 %.1f%%  swaps+barriers %.1f%%  other %.1f%%  (total %.3g ticks)\n", Q, 100.0 * st[4] / tot,
                100.0 * st[5] / tot, 100.0 * st[6] / tot, 100.0 * st[7] / tot, 100.0 * st[8] / tot, 100.0 * st[9] / tot, 100.0 * st[10] / tot, 100.0 * st[11] / tot, tot);
        if (st[38]) fprintf(stderr, "[wh] Q=%d wave lifetimes %.3g cycles: %.1f%% in the phases above, %.1f%% fetching an item (two barriers)\n", Q, (double)st[38], 100.0 * tot / (double)st[38], 100.0 * (double)st[39] / (double)st[38]);
        if (st[37]) fprintf(stderr, "[wh] Q=%d four-envelope sweeps: %llu envelopes, %.0f wave cycles per envelope (slot 'null2' above)\n", Q, st[37], (double)st[9] / (double)st[37]);
        fprintf(stderr, "[wh] Q=%d envelope Backward sweeps: %llu on a 256-node window, %llu on a 512-node window, %llu windows failed the mass certificate, %llu full width; union of the stored lane blocks: span %.1f blocks (with margin), %.1f blocks set, of %llu envelopes\n", Q, st[0], st[1], st[2], st[3], (double)st[12] / (double)std::max(1ull, st[14]), (double)st[15] / (double)std::max(1ull, st[14]), st[14]);
        fprintf(stderr, "[wh] Q=%d multihit Backward on a window: %llu scans, of them in doubt at a threshold %llu, at the multidomain bound %llu; window loss out of range %llu; mean eps %.3g\n", Q,
                st[35], st[32], st[33], st[36], st[35] ? 1e-9 * (double)st[34] / (double)st[35] : 0.0);
        fprintf(stderr, "[wh] Q=%d |Ld - mass| / Ld  (<3e-7, <1e-6, <3e-6, <1e-5, <2e-5, more): window sweeps %llu %llu %llu %llu %llu %llu; full-width sweeps %llu %llu %llu %llu %llu %llu\n", Q,
                st[16], st[17], st[18], st[19], st[20], st[21], st[22], st[23], st[24], st[25], st[26], st[27]);
      }
    }
    }
  }
  if (nq > 0 && !e->wide_by_w.empty()) {
    // ---- models of 3 073 - 12 288 nodes: several wavefronts per pair, float32 (wh_score_wide.hip); one launch per
    // waves-per-pair class.  A query batch too long for the kernel's LDS block falls back to the float64 front end below.
    const int Lc = std::max(max_len, 1);
    const size_t wlds0 = wide_lds_bytes(Lc);
    const int64_t npairs_all = nq * (int64_t)e->hmms.size();
    const bool resolve = !e->knobs.no_resolve && resolve_lds_bytes(Lc, e->max_M) <= kLdsBudget && npairs_all < 0x7FFFFFFF;
    if (wlds0 <= kLdsBudget) {
      size_t woff = 0;            // (the queue of the resolver was sized and reset with the one-wave launches above)
      for (auto &kv : e->by_q) woff += kv.second.size();
      woff += e->generic_front.size();
      int wclass = 0;
      for (auto &kv : e->wide_by_w) {
        const int W = kv.first & 15, wq = kv.first >> 4;
        // 12-cell classes: the emission rows of the canonical residues go to LDS where they fit behind the block
        const size_t em_floats = (size_t)e->K * wq * W * kWave;
        const bool em_lds = (wq == kWideQReg || wq == kWideQReg2) && !getenv("WH_WIDE_NO_EM_LDS") && wide_lds_bytes(Lc, em_floats) <= kLdsBudget;
        const size_t wlds = em_lds ? wide_lds_bytes(Lc, em_floats) : wlds0;
        WideArgs a;
        memset(&a, 0, sizeof a);
        a.hmms = (const DevHMM *)e->d_hmms.p; a.tables = (const float *)e->d_tables.p;
        a.hmm_list = (const int32_t *)e->d_lists.p + woff; a.n_list = (int)kv.second.size();
        woff += kv.second.size();
        a.residues = d_residues; a.offsets = d_offsets; a.nq = nq;
        if (wclass > 8) { set_error("too many classes of long models"); return WH_ERANGE; }
        a.counter = (int *)e->d_counter.p + 68 + wclass++;
        a.em_lds = em_lds ? 1 : 0;
        a.Lcap = Lc; a.SP = (Lc + 1 + 3) / 4 * 4;
        a.decibits = d_decibits; a.flags = d_flags; a.fwd_bits = d_fwd_bits; a.detail = d_detail;
        a.H = (int)e->hmms.size(); a.K = e->K; a.Kp = e->Kp;
        memcpy(a.degen, e->degen, sizeof a.degen);
        if (resolve) { a.rrecs = (ResolveRec *)e->d_rrecs.p; a.rcount = (int *)e->d_counter.p + 64; a.rcap = (int)e->rq_cap; }
        a.qorder = qorder_all;
        a.sparse = getenv("WH_WIDE_DENSE") ? 0 : 1;
        a.scratch_stride = (size_t)(Lc + 1) * 2 * wq * W * kWave + (a.sparse ? (size_t)(Lc + 1) * W * 2 + 4 : 0);
        a.scratch_stride = (a.scratch_stride + 3) & ~(size_t)3;
        const int64_t n_items = nq * (int64_t)a.n_list;
        const int per_cu = (W <= 4 && 2 * wlds <= kLdsBudget) ? 2 : 1;
        int blocks = (int)std::min<int64_t>(n_items, (int64_t)e->cu_count * per_cu);
        blocks = clamp_blocks(blocks, a.scratch_stride * sizeof(float), e->d_wscratch);
        if (e->d_wscratch.ensure((size_t)blocks * a.scratch_stride * sizeof(float))) return WH_ENOMEM;
        a.scratch = (float *)e->d_wscratch.p;
        HIPCHK(hipMemsetAsync(a.counter, 0, sizeof(int), s));
        if (e->knobs.trace) fprintf(stderr, "[wh] wide scoring: %lld pairs on %d models, %d waves per pair x %d cells per lane, %d workgroups, lds %zu, slab %zu MB per workgroup\n",
                                    (long long)n_items, a.n_list, W, wq, blocks, wlds, a.scratch_stride * 4 >> 20);
        if (class_mark(e, s, wq * W, 3)) return WH_EHIP;
        if (e->knobs.stats) {
          if (e->d_recs.ensure(320)) return WH_ENOMEM;
          HIPCHK(hipMemsetAsync(e->d_recs.p, 0, 320, s));
          a.stats = (unsigned long long *)e->d_recs.p;
        }
        hipError_t werr = launch_score_wide(wq, a, blocks, W, wlds, s);
        if (werr != hipSuccess) { set_error("wide score kernel launch failed: %s", hipGetErrorString(werr)); return WH_EHIP; }
        if (a.stats) {
          unsigned long long st[24];
          HIPCHK(hipMemcpyAsync(st, a.stats, sizeof st, hipMemcpyDeviceToHost, s));
          HIPCHK(hipStreamSynchronize(s));
          for (int wv = 0; wv < 2; wv++) {
            const unsigned long long *g = st + 8 + 8 * wv;
            double rt = 0; for (int k = 0; k < 7; k++) rt += (double)g[k];
            if (rt > 0) fprintf(stderr, "[wh] wide P1 row, %s wave: cells %.1f%%  barrier0 %.1f%%  local D %.1f%%  barrier1 %.1f%%  fix-up+sum %.1f%%  barrier2 %.1f%%  specials+tail %.1f%%\n", wv ? "last" : "first",
                                100 * g[0] / rt, 100 * g[1] / rt, 100 * g[2] / rt, 100 * g[3] / rt, 100 * g[4] / rt, 100 * g[5] / rt, 100 * g[6] / rt);
          }
          const double tot = (double)st[5] > 0 ? (double)st[5] : 1.0;
          fprintf(stderr, "[wh] wide %d x %d cells per lane, cycles of the first wave: P1 %.1f%%  P2 %.1f%%  regions %.1f%%  P3 %.1f%%  P4 %.1f%%  (of %.3g)\n", W, wq,
                  100.0 * st[0] / tot, 100.0 * st[1] / tot, 100.0 * st[2] / tot, 100.0 * st[3] / tot, 100.0 * st[4] / tot, tot);
        }
        launches++;
      }
    }
    wide_done = wlds0 <= kLdsBudget;
    if (!wide_done && e->force_wide) { set_error("WH_FORCE_WIDE: query length %d does not fit the wide kernel's LDS block", max_len); return WH_ERANGE; }
  }
  if (nq > 0 && (!e->generic_front.empty() || (!wide_done && !e->wide_by_w.empty() && !e->force_wide))) {
    // ---- models of more than 3072 nodes: the any-size float64 front end (wh_generic.hip), one wavefront per pair;
    // every pair with a region goes through the resolver's queue, which also assembles its score
    const int Lc = std::max(max_len, 1);
    const int64_t npairs_all = nq * (int64_t)e->hmms.size();
    if (e->knobs.no_resolve || resolve_lds_bytes(Lc, e->max_M) > kLdsBudget || npairs_all >= 0x7FFFFFFF || nq * (int64_t)e->generic.size() >= 0x7FFFFFFF) {
      set_error("models of more than %d nodes need the resolver stage (query length %d, %lld pairs)", kMaxQ * kWave, max_len, (long long)npairs_all);
      return WH_ERANGE;
    }
    GenericArgs g;
    memset(&g, 0, sizeof g);
    g.hmms = (const DevHMM *)e->d_hmms.p; g.gtab = (const double *)e->d_gtab.p;
    size_t goff = 0;
    for (auto &kv : e->by_q) goff += kv.second.size();
    const size_t n_gen = wide_done ? e->generic_front.size() : e->generic.size();     // (front list and wide lists are adjacent)
    g.hmm_list = (const int32_t *)e->d_lists.p + goff; g.n_list = (int)n_gen;
    g.residues = d_residues; g.offsets = d_offsets; g.nq = nq;
    g.counter = (int *)e->d_counter.p + 66;
    g.Lcap = Lc; g.Qmax = e->max_Q;
    g.slab_stride = (generic_front_doubles(Lc, e->max_Q) + 1) & ~(size_t)1;
    g.decibits = d_decibits; g.flags = d_flags; g.fwd_bits = d_fwd_bits; g.detail = d_detail;
    g.H = (int)e->hmms.size(); g.K = e->K; g.Kp = e->Kp;
    memcpy(g.degen, e->degen, sizeof g.degen);
    g.rrecs = (ResolveRec *)e->d_rrecs.p; g.rcount = (int *)e->d_counter.p + 64; g.rcap = (int)e->rq_cap;
    const size_t glds = generic_lds_bytes(Lc);
    if (glds > kLdsBudget) { set_error("query length %d does not fit the any-size kernel's LDS", max_len); return WH_ERANGE; }
    const int64_t n_items = nq * (int64_t)n_gen;
    int blocks = (int)std::min<int64_t>(n_items, (int64_t)e->cu_count * std::min<size_t>(12, kLdsBudget / glds));
    blocks = clamp_blocks(blocks, g.slab_stride * sizeof(double), e->d_rmx);
    if (e->d_rmx.ensure((size_t)blocks * g.slab_stride * sizeof(double))) return WH_ENOMEM;
    g.slab = (double *)e->d_rmx.p;
    HIPCHK(hipMemsetAsync(g.counter, 0, sizeof(int), s));
    if (e->knobs.trace) fprintf(stderr, "[wh] any-size front end: %lld pairs on %zu models (up to %d nodes), %d wavefronts, slab %zu MB per wave\n",
                                (long long)n_items, n_gen, e->max_M, blocks, g.slab_stride * 8 >> 20);
    if (class_mark(e, s, e->max_Q, 2)) return WH_EHIP;
    hipError_t gerr = launch_generic_front(g, blocks, glds, s);
    if (gerr != hipSuccess) { set_error("any-size front kernel launch failed: %s", hipGetErrorString(gerr)); return WH_EHIP; }
    launches++;
  }
  if (class_mark(e, s, 0, -1)) return WH_EHIP;        // closes the last launch's interval
  if (timer_end(e, 0, s, launches)) return WH_EHIP;
  if (e->last_staged_batches > 0) {
    // the staged batches' counters, once per call: units per pair (sizes the next call's batches) and the overflow flag
    e->st_cnt_host.resize((size_t)32 * e->last_staged_batches);
    HIPCHK(hipMemcpyAsync(e->st_cnt_host.data(), e->d_st_cnt.p, sizeof(int) * e->st_cnt_host.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    bool over = false;
    int most_units = 0;
    for (int b = 0; b < e->last_staged_batches; b++) {
      over = over || e->st_cnt_host[(size_t)32 * b + ST_OVERFLOW] != 0;
      most_units = std::max(most_units, e->st_cnt_host[(size_t)32 * b + ST_N_UNITS]);
    }
    // (full split: the next call's batches are sized for 1.25 x the densest batch seen, never below 1.05 units per pair)
    if (!over && most_units > 0 && e->st_last_NB > 0) e->st_upp = std::max(1.05, 1.25 * (double)most_units / (double)e->st_last_NB);
    if (over) {
      // a batch held more envelopes than it had slabs for (sixteen regions per pair are possible, batches are sized for the
      // rate seen so far): this call runs again with the fused kernel, the next ones with batches sized for what was seen
      if (kn_trace(e)) fprintf(stderr, "[wh] staged launches: a batch ran out of envelope units, the scoring pass is repeated with the fused kernel\n");
      e->st_upp = std::min<double>(WH_MAX_ENVELOPES, e->st_upp * 2.0);
      e->st_off = true;
      *overflow = true;
      if (timer_begin(e, 4, s) || timer_end(e, 4, s, 0)) return WH_EHIP;
      return WH_OK;
    }
  }
  if (nq > 0 && !e->by_q.empty()) {
    // While the scoring kernels run, the host sets up what the NEXT stage needs: the alignment kernels' per-wave slabs
    // ((L+1) x 5 x Q x 64 floats per resident wave: 6 GB at L = 150, Q = 16 - a first-call hipMalloc of 0.3 s that used to
    // sit between the two stages).  Bounded: skipped when it would take more than a tenth of the free HBM.
    const size_t need = (size_t)8 * (size_t)e->cu_count * (size_t)(std::max(max_len, 1) + 1) * 5 * (size_t)e->by_q.rbegin()->first * kWave * sizeof(float);
    size_t free_b = 0, total_b = 0;
    if (need > e->d_ascratch.cap && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need < free_b / 10) (void)e->d_ascratch.ensure(need);
  }
  if (timer_begin(e, 4, s)) return WH_EHIP;
  int rlaunches = 0;
  if (nq > 0 && !e->knobs.no_resolve && e->d_rrecs.p) {
    // ---- multidomain regions: HMMER's stochastic resolver (wh_resolve.hip), one wavefront per queued pair
    const int Lc = std::max(max_len, 1);
    const size_t rlds = resolve_lds_bytes(Lc, e->max_M);
    int *d_rcount = (int *)e->d_counter.p + 64, *d_rwork = (int *)e->d_counter.p + 65;
    int n_multi = 0;
    if (rlds <= kLdsBudget && nq * (int64_t)e->hmms.size() < 0x7FFFFFFF) {
      int n_bad = 0;          // (the resolver launches of EARLIER calls: counted on the device, read at this call's first synchronisation)
      HIPCHK(hipMemcpyAsync(&n_multi, d_rcount, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(hipMemcpyAsync(&n_bad, (int *)e->d_counter.p + kResolveErrSlot, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      if (n_bad != 0) {
        HIPCHK(hipMemsetAsync((int *)e->d_counter.p + kResolveErrSlot, 0, sizeof(int), s));
        set_error("resolver: %d queued pair(s) sat in a segment of another model and were NOT scored (internal error)", n_bad);
        return WH_EHIP;
      }
      if (nq > 0) e->rq_rate = std::max(e->rq_rate, (double)n_multi / (double)(nq * (int64_t)e->hmms.size()));
      if ((int64_t)n_multi > e->rq_cap) {
        // more pairs asked for a slot than the estimate allowed: the caller repeats the scoring pass with room for all
        if (e->knobs.trace) fprintf(stderr, "[wh] resolver queue: %d pairs for %lld slots, scoring pass repeated\n", n_multi, (long long)e->rq_cap);
        e->rq_floor = n_multi;
        *overflow = true;
        if (timer_end(e, 4, s, 0)) return WH_EHIP;
        return WH_OK;
      }
    }
    // one resolver launch over the first <n_multi> records of the queue; <rext>: the long-list pass below, whose records
    // keep their regions in HBM
    auto resolve_queue = [&](int n_multi, const int32_t *rext, int64_t rext_stride) -> int {
      const int Qmax = e->max_Q;
      ResolveArgs r;
      memset(&r, 0, sizeof r);
      r.rext = rext; r.rext_stride = rext_stride;
      r.hmms = (const DevHMM *)e->d_hmms.p; r.gtab = (const double *)e->d_gtab.p; r.ftab = (const float *)e->d_tables.p;
      r.residues = d_residues; r.offsets = d_offsets;
      r.recs = (const ResolveRec *)e->d_rrecs.p; r.count = d_rcount; r.rec_cap = (int)e->rq_cap;
      r.counter = d_rwork;
      r.Lcap = Lc; r.Mmax = e->max_M;
      // a wave's slab: matrix rows | threshold-line cache of the walk | E-state row cache (at the end)
      r.dc_off = ((size_t)(Lc + 2) * ((size_t)3 * Qmax * kWave + 8) + 1) & ~(size_t)1;
      r.mx_stride = r.dc_off + resolve_dcache_doubles() + (size_t)(Lc + 2) * resolve_tail_row_doubles();
      r.mx_stride = (r.mx_stride + 1) & ~(size_t)1;      // every wave's slab 16-byte aligned: the Forward sweep moves node pairs
      r.seg_cap = resolve_seg_cap();
      r.seg_stride = resolve_seg_ints(Lc, e->max_M);
      r.decibits = d_decibits; r.flags = d_flags; r.detail = d_detail;
      r.H = (int)e->hmms.size(); r.K = e->K; r.Kp = e->Kp;
      memcpy(r.degen, e->degen, sizeof r.degen);
      r.dbg = e->knobs.rdbg;
      r.launch_id = ++e->resolver_launches;
      r.err = (int *)e->d_counter.p + kResolveErrSlot;
      r.null2_gather = getenv("WH_RES_NULL2_GATHER") ? 1 : 0;
      if (e->knobs.stats) {
        if (e->d_recs.ensure(256)) return WH_ENOMEM;
        HIPCHK(hipMemsetAsync(e->d_recs.p, 0, 256, s));
        { unsigned long long bigv = ~0ull; HIPCHK(hipMemcpyAsync((char *)e->d_recs.p + 16 * 8, &bigv, 8, hipMemcpyHostToDevice, s)); HIPCHK(hipStreamSynchronize(s)); }
        r.stats = (unsigned long long *)e->d_recs.p;
      }
      // ---- launch geometry: ONE workgroup of up to eight waves per CU.  Models of up to 16 cells per lane get their
      // eight float64 transition arrays staged in the workgroup's LDS (49 KB at 12 cells per lane) when that fits beside
      // the waves' blocks; the Forward sweeps of their pairs then read one array per cell from L2 instead of nine.
      int Qt = 0;
      for (auto &kv : e->by_q) if (kv.first <= 16 && (kv.first == 4 || kv.first == 8 || kv.first == 12 || kv.first == 16)) Qt = std::max(Qt, kv.first);
      const bool small_queue = n_multi < 64 * e->cu_count;        // fewer than eight pairs per wave (see below)
      if (getenv("WH_RES_NO_LDS_TABLES") || small_queue) Qt = 0;
      int waves = resolve_waves_per_cu();
      if (const char *wv = getenv("WH_RES_WAVES")) waves = std::max(1, std::min(resolve_waves_per_cu(), atoi(wv)));      // experiments: waves per workgroup (= per CU)
      if (Qt > 0 && resolve_lds_header_bytes(Qt) + (size_t)waves * rlds > kLdsBudget) {
        // fewer waves WITH the tables only while at least six fit; otherwise the tables stay in L2
        int w2 = waves;
        while (w2 > 0 && resolve_lds_header_bytes(Qt) + (size_t)w2 * rlds > kLdsBudget) w2--;
        if (w2 >= 6) waves = w2; else Qt = 0;
      }
      while (waves > 1 && resolve_lds_header_bytes(Qt) + (size_t)waves * rlds > kLdsBudget) waves--;
      const size_t lds_total = resolve_lds_header_bytes(Qt) + (size_t)waves * rlds;
      r.lds_tables = Qt;
      r.wave_lds_ints = (int)(rlds / 4);
      // ---- the order of the queue.  Pairs are grouped model by model (longest pair first inside a model): the waves of a
      // workgroup work on ONE model at a time, so they share the staged tables and, for the models whose tables stay in
      // L2, stream the same arrays (wh_resolve.hip: slots and segments).
      if (e->d_rkeys.ensure(2 * sizeof(float) * (size_t)n_multi) || e->d_rorder.ensure(sizeof(int32_t) * (size_t)n_multi)) return WH_ENOMEM;
      std::vector<int32_t> chunk_list;
      {
        int32_t *d_models = (int32_t *)e->d_rkeys.p + n_multi;
        hipError_t kerr = launch_resolve_keys(r.recs, n_multi, r.hmms, (float *)e->d_rkeys.p, d_models, s, rext, rext_stride);
        if (kerr != hipSuccess) { set_error("resolve key kernel launch failed: %s", hipGetErrorString(kerr)); return WH_EHIP; }
        std::vector<float> keys((size_t)n_multi);
        std::vector<int32_t> models((size_t)n_multi);
        HIPCHK(hipMemcpyAsync(keys.data(), e->d_rkeys.p, sizeof(float) * keys.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(models.data(), d_models, sizeof(int32_t) * models.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::vector<int32_t> ord((size_t)n_multi);
        for (int t = 0; t < n_multi; t++) ord[(size_t)t] = t;
        // A small queue (fewer than eight pairs per wave: the reference's example data as shipped, 2 103 pairs) is ONE
        // segment in descending cost, models mixed, tables from L2: there the order decides the tail of the launch and
        // nothing else matters.  Otherwise: model by model.
        if (small_queue)
          std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return keys[(size_t)x] > keys[(size_t)y]; });
        else
          std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) {
            return models[(size_t)x] != models[(size_t)y] ? models[(size_t)x] < models[(size_t)y] : keys[(size_t)x] > keys[(size_t)y];
          });
        // one segment per model; slots in proportion to the segments' cost (four per workgroup in all, at least one per model)
        struct Seg { int start, count, h; double cost; };
        std::vector<Seg> segs;
        double total_cost = 0.0;
        if (small_queue) { segs.push_back({0, n_multi, -1, 1.0}); total_cost = 1.0; }
        for (int t = small_queue ? n_multi : 0; t < n_multi;) {
          const int h = models[(size_t)ord[(size_t)t]];
          int u = t;
          double cost = 0.0;
          while (u < n_multi && models[(size_t)ord[(size_t)u]] == h) { cost += std::max(1.0f, keys[(size_t)ord[(size_t)u]]); u++; }
          segs.push_back({t, u - t, h, cost});
          total_cost += cost;
          t = u;
        }
        std::vector<int> order_s(segs.size());
        for (size_t t = 0; t < segs.size(); t++) order_s[t] = (int)t;
        std::stable_sort(order_s.begin(), order_s.end(), [&](int x, int y) { return segs[(size_t)x].cost > segs[(size_t)y].cost; });
        const double per_slot = total_cost / (4.0 * (double)e->cu_count);
        std::vector<int32_t> slot_list;
        for (int sidx : order_s) {
          const Seg &g = segs[(size_t)sidx];
          int ns = (int)std::ceil(g.cost / std::max(per_slot, 1e-30));
          ns = std::max(1, std::min(ns, std::max(1, (g.count + 7) / 8)));       // never more slots than groups of eight pairs
          if (small_queue) ns = std::max(1, std::min(e->cu_count, (g.count + waves - 1) / waves));
          for (int v = 0; v < ns; v++) slot_list.push_back(sidx);
        }
        chunk_list.reserve(segs.size() * 4);
        for (const Seg &g : segs) { chunk_list.push_back(g.start); chunk_list.push_back(g.count); chunk_list.push_back(g.h); chunk_list.push_back(g.h >= 0 ? e->dev[(size_t)g.h].Q : 0); }
        // one buffer: segments | slots | cursors
        const size_t n_seg = segs.size(), n_slot = slot_list.size();
        if (e->d_rchunks.ensure(sizeof(int32_t) * (4 * n_seg + n_slot + n_seg))) return WH_ENOMEM;
        int32_t *d_chunks = (int32_t *)e->d_rchunks.p, *d_slots = d_chunks + 4 * n_seg, *d_cursors = d_slots + n_slot;
        HIPCHK(hipMemcpyAsync(e->d_rorder.p, ord.data(), sizeof(int32_t) * ord.size(), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_chunks, chunk_list.data(), sizeof(int32_t) * chunk_list.size(), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_slots, slot_list.data(), sizeof(int32_t) * slot_list.size(), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(d_cursors, 0, sizeof(int32_t) * n_seg, s));
        HIPCHK(hipStreamSynchronize(s));     // ord, chunk_list and slot_list are locals
        r.chunks = d_chunks; r.n_chunks = (int)n_seg;
        r.slots = d_slots; r.n_slots = (int)n_slot;
        r.cursors = d_cursors;
        r.order = (const int32_t *)e->d_rorder.p;
        r.chunks = (const int32_t *)e->d_rchunks.p;
        r.n_chunks = (int)(chunk_list.size() / 4);
      }
      int blocks = std::min(r.n_slots, e->cu_count);
      {
        // Every resident wavefront brings a slab of tens of MB, and hipMalloc costs ~40 ms per GB: a queue of a few thousand
        // pairs (the reference's example data: 8 412) spent 2.4 s allocating 55 GB for 0.12 s of work.  Unless the slabs exist
        // already, a wave gets at least four pairs.  (The cost is the driver scrubbing VRAM that another process used
        // before: on a fresh device the same allocation takes milliseconds.)
        const size_t have = std::min(e->d_rmx.cap / (r.mx_stride * sizeof(double)), e->d_rsegs.cap / std::max<size_t>(1, r.seg_stride * sizeof(int32_t))) / (size_t)waves;
        const int economy = std::max(32, n_multi / (4 * waves));
        if ((size_t)blocks > have) blocks = std::max((int)std::min<size_t>(have, (size_t)blocks), std::min(blocks, economy));
      }
      blocks = clamp_blocks(blocks, (size_t)waves * (r.mx_stride * sizeof(double) + r.seg_stride * sizeof(int32_t)), e->d_rmx);
      if (e->d_rmx.ensure((size_t)blocks * waves * r.mx_stride * sizeof(double)) || e->d_rsegs.ensure((size_t)blocks * waves * r.seg_stride * sizeof(int32_t)))
        return WH_ENOMEM;
      r.mx = (double *)e->d_rmx.p; r.segs = (int32_t *)e->d_rsegs.p;
      if (e->knobs.trace) fprintf(stderr, "[wh] resolve: %d pairs with a multidomain region on %d models, %d workgroups of %d waves, lds %zu (float64 tables of up to %d cells per lane staged: %s), slab %zu MB per wave\n",
                                  n_multi, r.n_chunks, blocks, waves, lds_total, Qt, Qt ? "yes" : "no", r.mx_stride * 8 >> 20);
      const auto t_rl0 = std::chrono::steady_clock::now();
      hipError_t err = launch_resolve(r, blocks, waves, lds_total, s);
      if (err != hipSuccess) { set_error("resolve kernel launch failed: %s", hipGetErrorString(err)); return WH_EHIP; }
      if (e->knobs.trace) {
        HIPCHK(hipStreamSynchronize(s));
        fprintf(stderr, "[wh] resolve kernel alone: %.1f ms (host clock around launch + synchronize)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rl0).count());
      }
      rlaunches++;
      e->last_resolved += n_multi;
      if (r.stats) {
        unsigned long long st[24];
        HIPCHK(hipMemcpyAsync(st, r.stats, sizeof st, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const double tot = (double)(st[0] + st[1] + st[2] + st[3] + st[4]);
        fprintf(stderr, "[wh] resolver wave cycles: region Forward %.1f%%  traces %.1f%%  clustering %.1f%%  cluster statistics %.1f%%  envelope Forward %.1f%%  (%.3g cycles per pair)\n",
                100.0 * st[0] / tot, 100.0 * st[1] / tot, 100.0 * st[2] / tot, 100.0 * st[3] / tot, 100.0 * st[4] / tot, tot / n_multi);
        fprintf(stderr, "[wh]   inside the traces: decision fetches %.1f%%  E-state choice %.1f%%  null2/accumulators/segments %.1f%%  (of the trace cycles)\n",
                100.0 * st[5] / (double)st[1], 100.0 * st[6] / (double)st[1], 100.0 * st[7] / (double)st[1]);
        fprintf(stderr, "[wh]   per multidomain region and trace: %.1f fetches of M runs, %.1f of D runs, %.1f of flank (C/J) runs, %.1f single I steps; %.0f cycles per fetch\n",
                st[8] / (200.0 * n_multi), st[9] / (200.0 * n_multi), st[10] / (200.0 * n_multi), st[11] / (200.0 * n_multi),
                (double)st[5] / (double)std::max<unsigned long long>(1, st[8] + st[9] + st[10]));
        fprintf(stderr, "[wh]   threshold-line cache: %.1f%% of the fetches hit\n", 100.0 * st[12] / (double)std::max<unsigned long long>(1, st[8] + st[9] + st[10]));
        fprintf(stderr, "[wh]   fetch order: %.1f%% of the fetches are the one that followed the last matched fetch in the previous trace, %.1f%% re-synchronise elsewhere in it\n",
                100.0 * st[20] / (double)std::max<unsigned long long>(1, st[8] + st[9] + st[10]), 100.0 * st[21] / (double)std::max<unsigned long long>(1, st[8] + st[9] + st[10]));
        fprintf(stderr, "[wh]   the line's load alone (issue -> validated): %.0f cycles per fetch\n", (double)st[23] / (double)std::max<unsigned long long>(1, st[8] + st[9] + st[10]));
        fprintf(stderr, "[wh]   shader clock while a pair is resolved: %.2f GHz (cycle counter / 100 MHz real-time counter); pair cycles %.3g\n", st[15] ? 0.1 * (double)st[14] / (double)st[15] : 0.0, (double)st[14]);
        fprintf(stderr, "[wh]   wave lifetimes: %llu waves, mean %.1f ms, longest %.1f ms (a wave leaves when no slot is left)\n", st[19], st[19] ? 1e-5 * (double)st[17] / (double)st[19] : 0.0, 1e-5 * (double)st[18]);
        fprintf(stderr, "[wh]   waiting at the workgroup's slot barriers: %.1f%% on top of the pair cycles (%d slots on %d models, %d workgroups of %d waves)\n", 100.0 * st[13] / tot, r.n_slots, r.n_chunks, blocks, waves);
      }
      return WH_OK;
    };
    if (n_multi > 0) { const int rc = resolve_queue(n_multi, nullptr, 0); if (rc != WH_OK) return rc; }
    // ---- the long-list pass.  The scoring kernels keep the regions of a pair in a list of WH_MAX_ENVELOPES entries in LDS;
    // HMMER has no such limit.  A pair with more regions comes out of them flagged WH_FLAG_TRUNC - and is scored
    // AGAIN here: the any-size float64 front end (wh_generic.hip) with a region list in HBM that holds every region a
    // sequence of this length can have, then a resolver launch of its own that reads the regions from that list and sums
    // over all envelopes.  Costs one pass over the flags (a byte per pair) and one 4-byte read-back per call; the float64
    // kernels run only when a pair needs them.
    e->last_long_list = 0;
    if (rlds <= kLdsBudget && nq * (int64_t)e->hmms.size() < 0x7FFFFFFF && generic_lds_bytes(Lc) <= kLdsBudget && !e->knobs.no_long_list) {
      const int64_t npairs_all = nq * (int64_t)e->hmms.size();
      int *d_tcount = (int *)e->d_counter.p + kLongListSlot;
      const int list_cap = (int)std::min<int64_t>(npairs_all, (int64_t)1 << 22);
      if (e->d_tlist.ensure(sizeof(int64_t) * (size_t)list_cap)) return WH_ENOMEM;
      HIPCHK(hipMemsetAsync(d_tcount, 0, sizeof(int), s));
      hipError_t terr = launch_trunc_list(d_flags, npairs_all, d_tcount, (int64_t *)e->d_tlist.p, list_cap, s);
      if (terr != hipSuccess) { set_error("flag scan launch failed: %s", hipGetErrorString(terr)); return WH_EHIP; }
      int n_trunc = 0;
      HIPCHK(hipMemcpyAsync(&n_trunc, d_tcount, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      n_trunc = std::min(n_trunc, list_cap);          // (beyond four million such pairs in one call the rest stay flagged)
      if (n_trunc > 0) {
        // a region is at least two rows long (the row that triggers it and a later one that ends it)
        const int ext_cap = Lc / 2 + 2;
        const int64_t rext_stride = (int64_t)kRextInts * ext_cap;
        // rounds of as many pairs as 256 MB of region lists hold
        const int per_round = (int)std::max<int64_t>(1, std::min<int64_t>(n_trunc, ((int64_t)64 << 20) / rext_stride));
        if (e->d_rext.ensure(sizeof(int32_t) * (size_t)per_round * (size_t)rext_stride)) return WH_ENOMEM;
        if (e->d_rrecs.cap < sizeof(ResolveRec) * (size_t)per_round) {
          HIPCHK(hipStreamSynchronize(s));
          if (e->d_rrecs.ensure(sizeof(ResolveRec) * (size_t)per_round)) return WH_ENOMEM;
        }
        e->rq_cap = std::max<int64_t>(e->rq_cap, per_round);
        if (e->knobs.trace) fprintf(stderr, "[wh] long-list pass: %d pairs with more than %d regions, %d per round, up to %d regions each\n", n_trunc, WH_MAX_ENVELOPES, per_round, ext_cap);
        for (int t0 = 0; t0 < n_trunc; t0 += per_round) {
          const int n_round = std::min(per_round, n_trunc - t0);
          GenericArgs g;
          memset(&g, 0, sizeof g);
          g.hmms = (const DevHMM *)e->d_hmms.p; g.gtab = (const double *)e->d_gtab.p;
          g.residues = d_residues; g.offsets = d_offsets; g.nq = nq;
          g.counter = (int *)e->d_counter.p + 66;
          g.Lcap = Lc; g.Qmax = e->max_Q;
          g.slab_stride = (generic_front_doubles(Lc, e->max_Q) + 1) & ~(size_t)1;
          g.decibits = d_decibits; g.flags = d_flags; g.fwd_bits = nullptr; g.detail = d_detail;
          g.H = (int)e->hmms.size(); g.K = e->K; g.Kp = e->Kp;
          memcpy(g.degen, e->degen, sizeof g.degen);
          g.rrecs = (ResolveRec *)e->d_rrecs.p; g.rcount = d_rcount; g.rcap = n_round;
          g.pair_list = (const int64_t *)e->d_tlist.p + t0; g.n_pairs = n_round;
          g.rext = (int32_t *)e->d_rext.p; g.rext_stride = rext_stride; g.ext_cap = ext_cap;
          const size_t glds = generic_lds_bytes(Lc);
          int gblocks = (int)std::min<int64_t>(n_round, (int64_t)e->cu_count * std::min<size_t>(12, kLdsBudget / glds));
          gblocks = clamp_blocks(gblocks, g.slab_stride * sizeof(double), e->d_rmx);
          if (e->d_rmx.ensure((size_t)gblocks * g.slab_stride * sizeof(double))) return WH_ENOMEM;
          g.slab = (double *)e->d_rmx.p;
          HIPCHK(hipMemsetAsync(g.counter, 0, sizeof(int), s));
          hipError_t gerr = launch_generic_front(g, gblocks, glds, s);
          if (gerr != hipSuccess) { set_error("long-list front kernel launch failed: %s", hipGetErrorString(gerr)); return WH_EHIP; }
          // the resolver's queue is now this round's records: length and work-queue head
          const int two[2] = {n_round, 0};
          HIPCHK(hipMemcpyAsync(d_rcount, two, sizeof two, hipMemcpyHostToDevice, s));
          HIPCHK(hipStreamSynchronize(s));
          const int rc = resolve_queue(n_round, (const int32_t *)e->d_rext.p, rext_stride);
          if (rc != WH_OK) return rc;
          e->last_long_list += n_round;
        }
      }
    }
  }
  if (timer_end(e, 4, s, rlaunches)) return WH_EHIP;
  return WH_OK;
}

// ---------------------------------------------------------------------------------------------- final merge
int wh_merge(int device, const uint8_t *q_text, const int64_t *q_off, int64_t nq, const int32_t *codes, const int32_t *q_row,
             const uint8_t *backbone, int32_t nb, int32_t B, uint8_t **out_full, uint8_t **out_masked, int64_t *out_rows,
             int64_t *out_width) {
  return wh_merge_sharded(device, q_text, q_off, nq, codes, q_row, backbone, nb, B, nullptr, nullptr, out_full, out_masked, out_rows, out_width);
}

int wh_merge_sharded(int device, const uint8_t *q_text, const int64_t *q_off, int64_t nq, const int32_t *codes, const int32_t *q_row,
                     const uint8_t *backbone, int32_t nb, int32_t B, int32_t *widths_local, const int32_t *widths_global,
                     uint8_t **out_full, uint8_t **out_masked, int64_t *out_rows, int64_t *out_width) {
  const bool widths_only = widths_local != nullptr && out_full == nullptr;
  if (!q_off || !q_row || nq < 0 || nb < 0 || B < 1 || (nq > 0 && !codes) ||
      (!widths_only && (!backbone && nb > 0)) || (!widths_only && (!out_full || !out_masked || !out_rows || !out_width || (nq > 0 && !q_text)))) {
    set_error("wh_merge: bad argument");
    return WH_EINVAL;
  }
  if (g_device < 0 && wh_init(device) != WH_OK) return WH_EHIP;
  HIPCHK(hipSetDevice(device));
  // the merge needs no model: its device buffers live for this call only
  struct Bufs { DevBuf b[12]; ~Bufs() { for (DevBuf &x : b) x.release(); } } bufs;
  DevBuf *m_buf = bufs.b;
  const int64_t total = nq > 0 ? q_off[nq] : 0;
  std::vector<int64_t> row_q;
  for (int64_t q = 0; q < nq; q++) {
    if (q_row[q] < -2) { set_error("wh_merge: q_row[%lld] = %d", (long long)q, q_row[q]); return WH_EINVAL; }
    if (q_row[q] >= 0) row_q.push_back(q);
  }
  for (int64_t r = 0; r < total; r++)
    if (codes[r] >= B || codes[r] < -1 - B) { set_error("wh_merge: code %d of residue %lld outside the backbone (%d columns)", codes[r], (long long)r, B); return WH_EINVAL; }
  const int64_t nrows = (int64_t)nb + (int64_t)row_q.size();
  enum { mTEXT = 0, mOFF, mCODES, mQROW, mROWQ, mBB, mW, mGAP, mK, mLAY, mFULL, mMASK };
  const size_t by[10] = {(size_t)total, sizeof(int64_t) * (size_t)(nq + 1), sizeof(int32_t) * (size_t)total, sizeof(int32_t) * (size_t)nq,
                         sizeof(int64_t) * row_q.size(), (size_t)nb * (size_t)B, sizeof(int32_t) * (size_t)(B + 1), sizeof(int32_t) * (size_t)total,
                         sizeof(int32_t) * (size_t)total, sizeof(long long) * (size_t)(2 * B + 2)};
  for (int t = 0; t < 10; t++) if (m_buf[t].ensure(by[t] + 16)) return WH_ENOMEM;
  const void *src[6] = {q_text, q_off, codes, q_row, row_q.data(), backbone};
  for (int t = 0; t < 6; t++) if (by[t] && src[t]) HIPCHK(hipMemcpy(m_buf[t].p, src[t], by[t], hipMemcpyHostToDevice));
  HIPCHK(hipMemset(m_buf[mW].p, 0, by[mW]));
  MergeArgs a;
  memset(&a, 0, sizeof a);
  a.q_text = (const uint8_t *)m_buf[mTEXT].p; a.q_off = (const int64_t *)m_buf[mOFF].p; a.codes = (const int32_t *)m_buf[mCODES].p;
  a.q_row = (const int32_t *)m_buf[mQROW].p; a.row_q = (const int64_t *)m_buf[mROWQ].p; a.nq = nq;
  a.bb = (const uint8_t *)m_buf[mBB].p; a.nb = nb; a.B = B;
  a.W = (int32_t *)m_buf[mW].p; a.res_gap = (int32_t *)m_buf[mGAP].p; a.res_k = (int32_t *)m_buf[mK].p;
  a.gap_start = (long long *)m_buf[mLAY].p; a.col_pos = a.gap_start + (B + 1); a.width = a.col_pos + B;
  hipError_t err = launch_merge_runs(a, nullptr);
  if (err != hipSuccess) { set_error("merge kernels failed to launch: %s", hipGetErrorString(err)); return WH_EHIP; }
  // sharded use (one process per GPU): the widest run per gap is a MAX over all ranks - the caller all-reduces
  // widths_local and hands the result back as widths_global; the layout is then the same on every rank
  if (widths_local) HIPCHK(hipMemcpy(widths_local, a.W, by[mW], hipMemcpyDeviceToHost));
  if (widths_only) return WH_OK;
  if (widths_global) {
    for (int g = 0; g <= B; g++) if (widths_global[g] < 0) { set_error("wh_merge: negative width at gap %d", g); return WH_EINVAL; }
    HIPCHK(hipMemcpy(a.W, widths_global, by[mW], hipMemcpyHostToDevice));
    err = launch_merge_layout(a, nullptr);
    if (err != hipSuccess) { set_error("merge layout kernel failed to launch: %s", hipGetErrorString(err)); return WH_EHIP; }
  }
  long long width = 0;
  HIPCHK(hipMemcpy(&width, a.width, sizeof width, hipMemcpyDeviceToHost));
  if (width < B || (double)width * (double)nrows > 6.0e10) { set_error("wh_merge: %lld rows x %lld columns is not a plausible alignment", (long long)nrows, width); return WH_ERANGE; }
  if (m_buf[mFULL].ensure((size_t)nrows * (size_t)width + 16) || m_buf[mMASK].ensure((size_t)nrows * (size_t)B + 16)) return WH_ENOMEM;
  a.out_full = (uint8_t *)m_buf[mFULL].p; a.out_masked = (uint8_t *)m_buf[mMASK].p;
  err = launch_merge_render(a, nrows, nullptr);
  if (err != hipSuccess) { set_error("merge render kernel failed to launch: %s", hipGetErrorString(err)); return WH_EHIP; }
  uint8_t *hf = (uint8_t *)malloc((size_t)nrows * (size_t)width + 1), *hm = (uint8_t *)malloc((size_t)nrows * (size_t)B + 1);
  if (!hf || !hm) { free(hf); free(hm); set_error("wh_merge: out of host memory"); return WH_ENOMEM; }
  hipError_t c1 = hipMemcpy(hf, a.out_full, (size_t)nrows * (size_t)width, hipMemcpyDeviceToHost);
  hipError_t c2 = hipMemcpy(hm, a.out_masked, (size_t)nrows * (size_t)B, hipMemcpyDeviceToHost);
  if (c1 != hipSuccess || c2 != hipSuccess) { free(hf); free(hm); set_error("wh_merge: copying the alignment back failed: %s", hipGetErrorString(c1 != hipSuccess ? c1 : c2)); return WH_EHIP; }
  *out_full = hf; *out_masked = hm; *out_rows = nrows; *out_width = width;
  return WH_OK;
}

static int max_query_len(const int64_t *offsets, int64_t nq) {
  int64_t m = 0;
  for (int64_t i = 0; i < nq; i++) m = std::max(m, offsets[i + 1] - offsets[i]);
  return (int)m;
}

int wh_score(wh_ehmm *e, const uint8_t *residues, const int64_t *offsets, int64_t nq, int32_t *decibits,
             uint8_t *flags, float *fwd_bits, wh_pair_detail *detail) {
  if (!e || !residues || !offsets || !decibits || !flags || nq < 0) { set_error("wh_score: bad argument"); return WH_EINVAL; }
  if (nq == 0) return WH_OK;
  HIPCHK(hipSetDevice(e->device));
  const int H = (int)e->hmms.size();
  const int64_t total = offsets[nq];
  for (int64_t i = 0; i < total; i++)
    if (residues[i] >= e->Kp) { set_error("residue code %d at position %lld is not in the alphabet", residues[i], (long long)i); return WH_EINVAL; }
  const size_t np = (size_t)nq * H;
  if (e->s_res.ensure((size_t)total + 16) || e->s_off.ensure(sizeof(int64_t) * (size_t)(nq + 1)) ||
      e->s_deci.ensure(sizeof(int32_t) * np) || e->s_flags.ensure(np) ||
      (fwd_bits && e->s_fwd.ensure(sizeof(float) * np)) || (detail && e->s_det.ensure(sizeof(wh_pair_detail) * np)))
    return WH_ENOMEM;
  HIPCHK(hipMemcpy(e->s_res.p, residues, (size_t)total, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->s_off.p, offsets, sizeof(int64_t) * (size_t)(nq + 1), hipMemcpyHostToDevice));
  int rc = wh_score_dev(e, (const uint8_t *)e->s_res.p, (const int64_t *)e->s_off.p, nq, total, max_query_len(offsets, nq),
                        (int32_t *)e->s_deci.p, (uint8_t *)e->s_flags.p, fwd_bits ? (float *)e->s_fwd.p : nullptr,
                        detail ? (wh_pair_detail *)e->s_det.p : nullptr, nullptr);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(decibits, e->s_deci.p, sizeof(int32_t) * np, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(flags, e->s_flags.p, np, hipMemcpyDeviceToHost));
  if (fwd_bits) HIPCHK(hipMemcpy(fwd_bits, e->s_fwd.p, sizeof(float) * np, hipMemcpyDeviceToHost));
  if (detail) HIPCHK(hipMemcpy(detail, e->s_det.p, sizeof(wh_pair_detail) * np, hipMemcpyDeviceToHost));
  return WH_OK;
}

// ------------------------------------------------------------------------------------ top-k
int wh_topk_dev(wh_ehmm *e, const int32_t *d_decibits, const uint8_t *d_flags, int64_t nq, int k, int32_t *d_idx,
                double *d_w, int32_t *d_n_kept, int32_t *d_n_used, void *stream) {
  if (!e || !d_decibits || !d_flags || !d_idx || !d_w || !d_n_kept || !d_n_used || k <= 0 || nq < 0) {
    set_error("wh_topk_dev: bad argument");
    return WH_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(e->device));
  if (timer_begin(e, 1, s)) return WH_EHIP;
  TopkArgs a;
  a.decibits = d_decibits; a.flags = d_flags; a.nseq = (const int32_t *)e->d_nseq.p;
  a.hmm_index = (const int32_t *)e->d_index.p; a.nq = nq; a.H = (int)e->hmms.size(); a.k = k;
  a.idx = d_idx; a.w = d_w; a.n_kept = d_n_kept; a.n_used = d_n_used;
  hipError_t err = launch_topk(a, s);
  if (err != hipSuccess) { set_error("topk kernel launch failed: %s", hipGetErrorString(err)); return WH_EHIP; }
  if (timer_end(e, 1, s, 1)) return WH_EHIP;
  return WH_OK;
}

int wh_topk(wh_ehmm *e, const int32_t *decibits, const uint8_t *flags, int64_t nq, int k, int32_t *idx, double *w,
            int32_t *n_kept, int32_t *n_used) {
  if (!e || !decibits || !flags || !idx || !w || !n_kept || !n_used || k <= 0 || nq < 0) { set_error("wh_topk: bad argument"); return WH_EINVAL; }
  if (nq == 0) return WH_OK;
  HIPCHK(hipSetDevice(e->device));
  const size_t np = (size_t)nq * e->hmms.size(), nk = (size_t)nq * (size_t)k;
  if (e->s_deci.ensure(sizeof(int32_t) * np) || e->s_flags.ensure(np) || e->s_idx.ensure(sizeof(int32_t) * nk) ||
      e->s_w.ensure(sizeof(double) * nk) || e->s_nk.ensure(sizeof(int32_t) * (size_t)nq) || e->s_nu.ensure(sizeof(int32_t) * (size_t)nq))
    return WH_ENOMEM;
  HIPCHK(hipMemcpy(e->s_deci.p, decibits, sizeof(int32_t) * np, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->s_flags.p, flags, np, hipMemcpyHostToDevice));
  int rc = wh_topk_dev(e, (const int32_t *)e->s_deci.p, (const uint8_t *)e->s_flags.p, nq, k, (int32_t *)e->s_idx.p,
                       (double *)e->s_w.p, (int32_t *)e->s_nk.p, (int32_t *)e->s_nu.p, nullptr);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(idx, e->s_idx.p, sizeof(int32_t) * nk, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(w, e->s_w.p, sizeof(double) * nk, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(n_kept, e->s_nk.p, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(n_used, e->s_nu.p, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost));
  return WH_OK;
}

// ------------------------------------------------------------------------------------ align
static int plan_align_block(int Q, int K, int Lcap, int *waves, int *SP, int *wave_lds, size_t *lds) {
  const int sp = (Lcap + 1 + 3) / 4 * 4;
  const int wl = kAlignSpecArrays * sp + (Lcap + 3) / 4 + 4;
  const size_t table = (size_t)(K + 2 * FW_NARR) * Q * kWave * sizeof(float);
  int w = 8;
  while (w >= 1 && table + (size_t)w * wl * sizeof(float) > kLdsBudget) w--;
  if (w < 1) return WH_ERANGE;
  *waves = w; *SP = sp; *wave_lds = wl; *lds = kLdsHeader + table + (size_t)w * wl * sizeof(float);
  return WH_OK;
}

int wh_align_dev(wh_ehmm *e, const uint8_t *d_residues, const int64_t *d_offsets, int64_t nq, int64_t total_residues,
                 int32_t max_len, const int64_t *d_pair_q, const int32_t *d_pair_h, int64_t npairs,
                 const int64_t *d_col_offsets, int32_t *d_cols, void *stream) {
  (void)nq; (void)total_residues;
  if (!e || !d_residues || !d_offsets || !d_pair_q || !d_pair_h || !d_col_offsets || !d_cols || npairs < 0 || max_len < 0) {
    set_error("wh_align_dev: bad argument");
    return WH_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(e->device));
  if (npairs == 0) { e->timers[2].launches = 0; e->timers[2].ms = 0; return WH_OK; }
  if (npairs > 0x7FFFFFFF) { set_error("too many pairs"); return WH_ERANGE; }
  // group the pairs by model on the host (the model's tables are shared through LDS by a workgroup)
  std::vector<int32_t> ph((size_t)npairs);
  HIPCHK(hipMemcpyAsync(ph.data(), d_pair_h, sizeof(int32_t) * (size_t)npairs, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const int H = (int)e->hmms.size();
  std::vector<int32_t> cnt((size_t)H + 1, 0);
  for (int64_t p = 0; p < npairs; p++) {
    if (ph[(size_t)p] < 0 || ph[(size_t)p] >= H) { set_error("pair %lld: model position %d out of range", (long long)p, ph[(size_t)p]); return WH_EINVAL; }
    cnt[(size_t)ph[(size_t)p] + 1]++;
  }
  for (int h = 0; h < H; h++) cnt[(size_t)h + 1] += cnt[(size_t)h];
  std::vector<int32_t> order((size_t)npairs), cursor(cnt.begin(), cnt.end() - 1);
  for (int64_t p = 0; p < npairs; p++) order[(size_t)cursor[(size_t)ph[(size_t)p]]++] = (int32_t)p;
  if (e->d_order.ensure(sizeof(int32_t) * (size_t)npairs)) return WH_ENOMEM;
  // pairs whose Backward sweep leaves float32 range are queued on the device and redone in log space
  const bool want_redo = !e->knobs.no_logspace;
  if (e->d_recs.ensure(sizeof(int32_t) * ((size_t)npairs + 4))) return WH_ENOMEM;
  int *d_redo_count = (int *)e->d_recs.p;
  int32_t *d_redo_list = (int32_t *)e->d_recs.p + 4;
  HIPCHK(hipMemsetAsync(d_redo_count, 0, sizeof(int), s));
  HIPCHK(hipMemsetAsync((int *)e->d_counter.p + 96, 0, 28 * sizeof(int), s));
  if (timer_begin(e, 2, s)) return WH_EHIP;
  int launches = 0;
  // one pass = plan the launches of every model class for the pairs in <order> (grouped by model,
  // cnt = prefix counts per model) and run them
  auto run_pass = [&](const std::vector<int32_t> &order, const std::vector<int32_t> &cnt, bool logsp) -> int {
  HIPCHK(hipMemcpyAsync(e->d_order.p, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice, s));
  std::vector<int32_t> items;   // all classes back to back: h, start, count
  std::vector<std::array<int, 7>> plans;   // Q, first item, n items, waves, SP, wave_lds, Klds
  std::vector<size_t> ldss;
  for (auto &kv : e->by_q) {
    const int Q = kv.first;
    int waves = 0, SP = 0, wave_lds = 0; size_t lds = 0;
    if (Q <= kMaxQFast && plan_align_block(Q, e->K, std::max(max_len, 1), &waves, &SP, &wave_lds, &lds) != WH_OK) waves = 0;
    int Klds = e->K;
    bool swap = Q > kMaxQFast;
    if (!swap && Q >= 20 && (waves < 4 || e->knobs.force_specg)) {
      // 20/24-cell models whose emission rows (protein: 20) do not fit beside BOTH orientations even
      // with the special states in HBM: pass-synchronous variant
      const size_t table2 = (size_t)(e->K + 2 * FW_NARR) * Q * kWave * sizeof(float);
      const size_t per_wave = (size_t)((std::max(max_len, 1) + 3) / 4 + 4) * sizeof(float);
      if (kLdsHeader + table2 + 4 * per_wave > kLdsBudget) swap = true;
    }
    if (swap) {   // long models: one orientation resident, 4 waves, special states in HBM
      SP = (std::max(max_len, 1) + 1 + 3) / 4 * 4;
      wave_lds = -((std::max(max_len, 1) + 3) / 4 + 4);
      waves = 4;
      size_t table = (size_t)(Klds + 8) * Q * kWave * sizeof(float);
      if (kLdsHeader + table + (size_t)waves * (size_t)(-wave_lds) * sizeof(float) > kLdsBudget) { Klds = 0; table = (size_t)8 * Q * kWave * sizeof(float); }
      lds = kLdsHeader + table + (size_t)waves * (size_t)(-wave_lds) * sizeof(float);
      if (lds > kLdsBudget) { set_error("query length %d with model class Q=%d does not fit in LDS", max_len, Q); return WH_ERANGE; }
    } else if (waves < 4 || e->knobs.force_specg) {   // long queries: special-state rows in HBM
      SP = (std::max(max_len, 1) + 1 + 3) / 4 * 4;
      wave_lds = -((std::max(max_len, 1) + 3) / 4 + 4);   // negative marks the HBM mode for the launch loop below
      const size_t table = (size_t)(e->K + 2 * FW_NARR) * Q * kWave * sizeof(float);
      waves = 8;
      while (waves >= 1 && kLdsHeader + table + (size_t)waves * (size_t)(-wave_lds) * sizeof(float) > kLdsBudget) waves--;
      if (waves < 1) { set_error("model class Q=%d does not fit in LDS", Q); return WH_ERANGE; }
      lds = kLdsHeader + table + (size_t)waves * (size_t)(-wave_lds) * sizeof(float);
    }
    const int first = (int)items.size() / 3;
    for (int h : kv.second) {
      int lo = cnt[(size_t)h], hi = cnt[(size_t)h + 1];
      for (int st = lo; st < hi; st += waves) { items.push_back(h); items.push_back(st); items.push_back(std::min(waves, hi - st)); }
    }
    const int n = (int)items.size() / 3 - first;
    if (n > 0) { plans.push_back({Q, first, n, waves, SP, wave_lds, Klds + (swap ? 1000 : 0)}); ldss.push_back(lds); }
  }
  size_t need_scratch = 0, need_spec = 0;
  const size_t nit = items.size() / 3;
  std::vector<int32_t> soa(items.size());
  for (size_t t = 0; t < nit; t++) { soa[t] = items[3 * t]; soa[nit + t] = items[3 * t + 1]; soa[2 * nit + t] = items[3 * t + 2]; }
  if (e->d_items.ensure(sizeof(int32_t) * soa.size() + 16)) return WH_ENOMEM;
  HIPCHK(hipMemcpyAsync(e->d_items.p, soa.data(), sizeof(int32_t) * soa.size(), hipMemcpyHostToDevice, s));
  for (int pass = 0; pass < 2; pass++) {        // pass 0: size the workspace of every class, allocate once; pass 1: launch
  if (pass == 1 && (e->d_ascratch.ensure(need_scratch) || (need_spec && e->d_spec.ensure(need_spec)))) return WH_ENOMEM;
  for (size_t pl = 0; pl < plans.size(); pl++) {
    const int Q = plans[pl][0], first = plans[pl][1], n = plans[pl][2], waves = plans[pl][3];
    AlignArgs a;
    memset(&a, 0, sizeof a);
    a.hmms = (const DevHMM *)e->d_hmms.p; a.tables = (const float *)e->d_tables.p;
    a.residues = d_residues; a.offsets = d_offsets; a.pair_q = d_pair_q;
    a.order = (const int32_t *)e->d_order.p;
    a.item_h = (const int32_t *)e->d_items.p + first;
    a.item_start = (const int32_t *)e->d_items.p + nit + first;
    a.item_count = (const int32_t *)e->d_items.p + 2 * nit + first;
    a.n_items = n;
    a.col_offsets = d_col_offsets; a.cols = d_cols;
    a.counter = (int *)e->d_counter.p + launches;
    a.Lcap = std::max(max_len, 1); a.SP = plans[pl][4]; a.wave_lds = std::abs(plans[pl][5]);
    a.K = e->K; a.Kp = e->Kp; a.Klds = plans[pl][6] % 1000; a.swap = plans[pl][6] >= 1000 ? 1 : 0;
    a.logsp = logsp ? 1 : 0;
    a.no_window = e->knobs.no_window ? 1 : 0;
    a.wstat = logsp ? nullptr : (int *)e->d_counter.p + 96;
    a.wcyc = (!logsp && (e->knobs.trace || e->knobs.stats)) ? reinterpret_cast<unsigned long long *>((int *)e->d_counter.p + 100) : nullptr;
    a.redo_count = (!logsp && want_redo) ? d_redo_count : nullptr;
    a.redo_list = (!logsp && want_redo) ? d_redo_list : nullptr;
    if (launches >= kMaxLaunches) { set_error("wh_align_dev: too many launches in one call"); return WH_ERANGE; }
    int blocks = std::min(n, e->cu_count * std::max(1, 8 / waves));
    a.scratch_stride = (size_t)(a.Lcap + 1) * 5 * Q * kWave;
    a.spec_stride = plans[pl][5] < 0 ? (size_t)kAlignSpecArrays * a.SP : 0;
    blocks = clamp_blocks(blocks, (size_t)waves * (a.scratch_stride + a.spec_stride) * sizeof(float), e->d_ascratch);
    if (pass == 0) {
      need_scratch = std::max(need_scratch, (size_t)blocks * waves * a.scratch_stride * sizeof(float));
      if (plans[pl][5] < 0) need_spec = std::max(need_spec, (size_t)blocks * waves * a.spec_stride * sizeof(float));
      continue;
    }
    blocks = (int)std::min<size_t>((size_t)blocks, e->d_ascratch.cap / ((size_t)waves * a.scratch_stride * sizeof(float)));
    if (plans[pl][5] < 0) blocks = (int)std::min<size_t>((size_t)blocks, e->d_spec.cap / ((size_t)waves * a.spec_stride * sizeof(float)));
    if (blocks < 1) { set_error("workspace planning failed (Q=%d)", Q); return WH_ENOMEM; }
    if (plans[pl][5] < 0) a.spec_scratch = (float *)e->d_spec.p;
    a.scratch = (float *)e->d_ascratch.p;
    HIPCHK(hipMemsetAsync(a.counter, 0, sizeof(int), s));
    hipError_t err = launch_align(Q, a, blocks, waves * kWave, ldss[pl], s);
    if (err != hipSuccess) { set_error("align kernel launch (Q=%d) failed: %s", Q, hipGetErrorString(err)); return WH_EHIP; }
    launches++;
  }
  }
  // the host vectors of this pass are consumed by async copies: drain before they go out of scope
  HIPCHK(hipStreamSynchronize(s));
  return WH_OK;
  };
  int rc = run_pass(order, cnt, false);
  if (rc != WH_OK) return rc;
  int n_redo = 0;
  if (want_redo) {
    HIPCHK(hipMemcpyAsync(&n_redo, d_redo_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  if (n_redo > 0) {
    std::vector<int32_t> redo((size_t)n_redo);
    HIPCHK(hipMemcpyAsync(redo.data(), d_redo_list, sizeof(int32_t) * (size_t)n_redo, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<int32_t> cnt2((size_t)H + 1, 0);
    for (int32_t p : redo) cnt2[(size_t)ph[(size_t)p] + 1]++;
    for (int h = 0; h < H; h++) cnt2[(size_t)h + 1] += cnt2[(size_t)h];
    std::vector<int32_t> order2((size_t)n_redo), cur2(cnt2.begin(), cnt2.end() - 1);
    std::sort(redo.begin(), redo.end());
    for (int32_t p : redo) order2[(size_t)cur2[(size_t)ph[(size_t)p]]++] = p;
    if (e->knobs.trace) fprintf(stderr, "[wh] align: %d of %lld pairs left float32 range, redone in log space\n", n_redo, (long long)npairs);
    rc = run_pass(order2, cnt2, true);
    if (rc != WH_OK) return rc;
  }
  {
    int ws[28] = {0};
    HIPCHK(hipMemcpyAsync(ws, (int *)e->d_counter.p + 96, sizeof ws, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int t = 0; t < 4; t++) e->last_align_paths[t] = ws[t];
    if (e->knobs.trace || e->knobs.stats) {
      unsigned long long cy[4];
      memcpy(cy, ws + 4, sizeof cy);
      fprintf(stderr, "[wh] align: %d + %d pairs on a 256- / 512-node window, %d windows rejected (full width), %d without a window; wave cycles of the window pairs: "
              "Forward %.3g, Backward+posteriors %.3g, OA fill %.3g, traceback %.3g\n", ws[0], ws[3], ws[1], ws[2], (double)cy[0], (double)cy[1], (double)cy[2], (double)cy[3]);
      fprintf(stderr, "[wh] align: window attempts by slack (lane blocks between the path's span with margins and the window, 0..7+): accepted");
      for (int t = 0; t < 8; t++) fprintf(stderr, " %d", ws[12 + t]);
      fprintf(stderr, "; rejected");
      for (int t = 0; t < 8; t++) fprintf(stderr, " %d", ws[20 + t]);
      fprintf(stderr, "\n");
    }
  }
  e->last_align_redo = n_redo;
  e->last_align_unaligned = 0;
  e->last_unaligned_pairs.clear();
  if (!e->generic.empty() || e->force_wide) {
    // pairs on models of more than 3072 nodes: the any-size float64 alignment kernel, one wavefront per pair
    std::vector<int32_t> gitems;
    const int Lc0 = std::max(max_len, 1);
    // models of 3 073 - 12 288 nodes: the several-waves-per-pair alignment kernel (wh_score_wide.hip); pairs that leave
    // float32 range there, longer queries and larger models go to the float64 kernel below
    const size_t walds = wide_align_lds_bytes(Lc0);
    const bool use_wide = walds <= kLdsBudget && !e->wide_by_w.empty() && !e->knobs.no_wide_align;
    std::map<int, std::vector<int32_t>> witems;
    for (int64_t p = 0; p < npairs; p++) {
      const DevHMM &dm = e->dev[(size_t)ph[(size_t)p]];
      if (use_wide && dm.wideW > 0 && (dm.Q > kMaxQ || e->force_wide)) witems[dm.wideQ * 16 + dm.wideW].push_back((int32_t)p);
      else if (dm.Q > kMaxQ) gitems.push_back((int32_t)p);
    }
    if (!witems.empty()) {
      if (e->d_recs.ensure(sizeof(int32_t) * ((size_t)npairs + 4))) return WH_ENOMEM;
      HIPCHK(hipMemsetAsync(e->d_recs.p, 0, sizeof(int32_t) * (size_t)npairs, s));
      size_t ooff = 0;
      std::vector<int32_t> all;
      for (auto &kv : witems) all.insert(all.end(), kv.second.begin(), kv.second.end());
      if (e->d_order.ensure(sizeof(int32_t) * (all.size() + (size_t)npairs))) return WH_ENOMEM;
      HIPCHK(hipMemcpyAsync(e->d_order.p, all.data(), sizeof(int32_t) * all.size(), hipMemcpyHostToDevice, s));
      int wclass = 0;
      for (auto &kv : witems) {
        const int W = kv.first & 15, wq = kv.first >> 4;
        WideAlignArgs wa;
        memset(&wa, 0, sizeof wa);
        wa.hmms = (const DevHMM *)e->d_hmms.p; wa.tables = (const float *)e->d_tables.p;
        wa.residues = d_residues; wa.offsets = d_offsets;
        wa.items = (const int32_t *)e->d_order.p + ooff; wa.n_items = (int)kv.second.size();
        ooff += kv.second.size();
        wa.pair_q = d_pair_q; wa.pair_h = d_pair_h; wa.col_off = d_col_offsets; wa.cols = d_cols;
        wa.status = (int32_t *)e->d_recs.p;
        if (wclass > 8) { set_error("too many classes of long models"); return WH_ERANGE; }
        wa.counter = (int *)e->d_counter.p + 80 + wclass++;
        wa.Lcap = Lc0; wa.SP = (Lc0 + 1 + 3) / 4 * 4;
        wa.K = e->K; wa.Kp = e->Kp;
        wa.scratch_stride = (size_t)(Lc0 + 1) * 5 * wq * W * kWave;
        int blocks = (int)std::min<size_t>(kv.second.size(), (size_t)e->cu_count);
        blocks = clamp_blocks(blocks, wa.scratch_stride * sizeof(float), e->d_wscratch);
        if (e->d_wscratch.ensure((size_t)blocks * wa.scratch_stride * sizeof(float))) return WH_ENOMEM;
        wa.scratch = (float *)e->d_wscratch.p;
        HIPCHK(hipMemsetAsync(wa.counter, 0, sizeof(int), s));
        if (e->knobs.trace) fprintf(stderr, "[wh] wide alignment: %zu pairs, %d waves per pair, %d workgroups, lds %zu, slab %zu MB per workgroup\n", kv.second.size(), W, blocks, walds, wa.scratch_stride * 4 >> 20);
        hipError_t werr = launch_align_wide(wq, wa, blocks, W, walds, s);
        if (werr != hipSuccess) { set_error("wide alignment kernel launch failed: %s", hipGetErrorString(werr)); return WH_EHIP; }
        launches++;
      }
      std::vector<int32_t> wst((size_t)npairs);
      HIPCHK(hipMemcpyAsync(wst.data(), e->d_recs.p, sizeof(int32_t) * wst.size(), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      int n_hand = 0;
      for (size_t p = 0; p < wst.size(); p++) if (wst[p] == 1) { gitems.push_back((int32_t)p); n_hand++; }
      std::sort(gitems.begin(), gitems.end());
      e->last_align_redo += n_hand;
      if (e->knobs.trace && n_hand) fprintf(stderr, "[wh] wide alignment: %d pairs left float32 range, handed to the float64 kernel\n", n_hand);
    }
    if (!gitems.empty()) {
      const int Lc = std::max(max_len, 1);
      GenericAlignArgs g;
      memset(&g, 0, sizeof g);
      g.hmms = (const DevHMM *)e->d_hmms.p; g.gtab = (const double *)e->d_gtab.p;
      g.residues = d_residues; g.offsets = d_offsets;
      HIPCHK(hipMemcpyAsync(e->d_order.p, gitems.data(), sizeof(int32_t) * gitems.size(), hipMemcpyHostToDevice, s));
      g.items = (const int32_t *)e->d_order.p; g.n_items = (int)gitems.size();
      g.pair_q = d_pair_q; g.pair_h = d_pair_h; g.col_off = d_col_offsets; g.cols = d_cols;
      if (e->d_recs.ensure(sizeof(int32_t) * ((size_t)npairs + 4))) return WH_ENOMEM;
      HIPCHK(hipMemsetAsync(e->d_recs.p, 0, sizeof(int32_t) * (size_t)npairs, s));
      g.status = (int32_t *)e->d_recs.p;
      g.counter = (int *)e->d_counter.p + 67;
      g.Lcap = Lc; g.Qmax = e->max_Q; g.Kp = e->Kp;
      g.slab_stride = (generic_align_doubles(Lc, e->max_Q) + 1) & ~(size_t)1;
      const size_t glds = (size_t)Lc + 64;
      if (glds > kLdsBudget) { set_error("query length %d does not fit the any-size kernel's LDS", max_len); return WH_ERANGE; }
      int blocks = (int)std::min<size_t>(gitems.size(), (size_t)e->cu_count * std::min<size_t>(12, kLdsBudget / glds));
      blocks = clamp_blocks(blocks, g.slab_stride * sizeof(double), e->d_rmx);
      if (e->d_rmx.ensure((size_t)blocks * g.slab_stride * sizeof(double))) return WH_ENOMEM;
      g.slab = (double *)e->d_rmx.p;
      HIPCHK(hipMemsetAsync(g.counter, 0, sizeof(int), s));
      if (e->knobs.trace) fprintf(stderr, "[wh] any-size alignment: %zu pairs, %d wavefronts, slab %zu MB per wave\n", gitems.size(), blocks, g.slab_stride * 8 >> 20);
      hipError_t gerr = launch_generic_align(g, blocks, glds, s);
      if (gerr != hipSuccess) { set_error("any-size alignment kernel launch failed: %s", hipGetErrorString(gerr)); return WH_EHIP; }
      launches++;
      std::vector<int32_t> st((size_t)npairs);
      HIPCHK(hipMemcpyAsync(st.data(), e->d_recs.p, sizeof(int32_t) * st.size(), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));     // gitems is a local
      int n_range = 0, n_log = 0;
      for (size_t p = 0; p < st.size(); p++) { n_range += st[p] == 3; n_log += st[p] == 4 || st[p] == 3; }
      e->last_align_unaligned = 0;        // (round 5: no pair is left unaligned for its range - see generic_align_kernel)
      e->last_align_redo += n_log;
      if (e->knobs.trace && n_log > 0) fprintf(stderr, "[wh] any-size alignment: %d pairs left float64 range, redone in log space\n", n_log);
      if (n_range > 0)
        fprintf(stderr, "[wh] note: on %d pair(s) on models of more than %d nodes the log-space Forward and Backward scores disagree; "
                        "aligned from the Forward-normalised posteriors, as hmmalign does\n", n_range, kMaxQ * kWave);
    }
  }
  if (timer_end(e, 2, s, launches)) return WH_EHIP;
  return WH_OK;
}

int wh_align(wh_ehmm *e, const uint8_t *residues, const int64_t *offsets, int64_t nq, const int64_t *pair_q,
             const int32_t *pair_h, int64_t npairs, const int64_t *col_offsets, int32_t *cols) {
  if (!e || !residues || !offsets || !pair_q || !pair_h || !col_offsets || !cols || nq < 0 || npairs < 0) {
    set_error("wh_align: bad argument");
    return WH_EINVAL;
  }
  if (npairs == 0) return WH_OK;
  HIPCHK(hipSetDevice(e->device));
  const int64_t total = offsets[nq];
  for (int64_t i = 0; i < total; i++)
    if (residues[i] >= e->Kp) { set_error("residue code %d at position %lld is not in the alphabet", residues[i], (long long)i); return WH_EINVAL; }
  for (int64_t p = 0; p < npairs; p++)
    if (pair_q[p] < 0 || pair_q[p] >= nq) { set_error("pair %lld: query %lld out of range", (long long)p, (long long)pair_q[p]); return WH_EINVAL; }
  const int64_t ncols = col_offsets[npairs];
  if (e->s_res.ensure((size_t)total + 16) || e->s_off.ensure(sizeof(int64_t) * (size_t)(nq + 1)) ||
      e->s_pq.ensure(sizeof(int64_t) * (size_t)npairs) || e->s_ph.ensure(sizeof(int32_t) * (size_t)npairs) ||
      e->s_co.ensure(sizeof(int64_t) * (size_t)(npairs + 1)) || e->s_cols.ensure(sizeof(int32_t) * (size_t)ncols + 16))
    return WH_ENOMEM;
  HIPCHK(hipMemcpy(e->s_res.p, residues, (size_t)total, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->s_off.p, offsets, sizeof(int64_t) * (size_t)(nq + 1), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->s_pq.p, pair_q, sizeof(int64_t) * (size_t)npairs, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->s_ph.p, pair_h, sizeof(int32_t) * (size_t)npairs, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->s_co.p, col_offsets, sizeof(int64_t) * (size_t)(npairs + 1), hipMemcpyHostToDevice));
  int rc = wh_align_dev(e, (const uint8_t *)e->s_res.p, (const int64_t *)e->s_off.p, nq, total, max_query_len(offsets, nq),
                        (const int64_t *)e->s_pq.p, (const int32_t *)e->s_ph.p, npairs, (const int64_t *)e->s_co.p,
                        (int32_t *)e->s_cols.p, nullptr);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(cols, e->s_cols.p, sizeof(int32_t) * (size_t)ncols, hipMemcpyDeviceToHost));
  return WH_OK;
}

// ------------------------------------------------------------------------------------ consensus
static const int kConsKmax = 4096;   // sanity bound only: the per-residue edge lists live in HBM scratch sized by the call's own k

int wh_consensus_dev(wh_ehmm *e, const int64_t *d_offsets, int64_t nq, int32_t max_len, const int64_t *d_qpair_off,
                     const int32_t *d_pair_h, const double *d_pair_w, const int64_t *d_col_offsets,
                     const int32_t *d_cols, const int64_t *d_ret_off, const int32_t *d_retained,
                     const int32_t *d_nongaps, int32_t backbone_length, int32_t max_pairs_per_query,
                     int32_t *d_out, int32_t *d_minmax, void *stream) {
  if (!e || !d_offsets || !d_qpair_off || !d_pair_h || !d_pair_w || !d_col_offsets || !d_cols || !d_ret_off ||
      !d_retained || !d_nongaps || !d_out || !d_minmax || nq < 0 || backbone_length <= 0 || max_len < 0) {
    set_error("wh_consensus_dev: bad argument");
    return WH_EINVAL;
  }
  if (max_pairs_per_query > kConsKmax || max_pairs_per_query < 0) { set_error("more than %d HMMs per query are not supported by the consensus kernel", kConsKmax); return WH_ERANGE; }
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(e->device));
  if (timer_begin(e, 3, s)) return WH_EHIP;
  if (nq > 0) {
    ConsArgs a;
    memset(&a, 0, sizeof a);
    a.offsets = d_offsets; a.nq = nq; a.qpair_off = d_qpair_off; a.pair_h = d_pair_h; a.pair_w = d_pair_w;
    a.col_offsets = d_col_offsets; a.cols = d_cols; a.ret_off = d_ret_off; a.retained = d_retained; a.nongaps = d_nongaps;
    a.backbone_length = backbone_length; a.out = d_out; a.minmax = d_minmax;
    // a residue collects at most one edge per kept HMM (the reference takes any -k, weighting.py:71-73)
    a.Lcap = std::max(max_len, 1); a.Wcap = backbone_length + 1; a.KMAX = std::max(4, (int)max_pairs_per_query);
    int waves = 4;
    while (waves >= 1 && (size_t)waves * (a.Wcap + 2) * sizeof(double) > kLdsBudget) waves--;
    const bool row_in_hbm = waves < 1;        // backbones beyond ~19 000 columns: the DP row moves to the wave's HBM region
    if (row_in_hbm) waves = 4;
    const size_t lds = row_in_hbm ? 0 : (size_t)waves * (a.Wcap + 2) * sizeof(double);
    int blocks = (int)std::min<int64_t>((nq + waves - 1) / waves, (int64_t)e->cu_count * 2);
    blocks = clamp_blocks(blocks, (size_t)waves * ((size_t)(a.Lcap + 1) * (a.Wcap + 2) + (size_t)a.Lcap * a.KMAX * 12 + (size_t)a.Lcap * 4 + (size_t)(a.Wcap + 2) * 8), e->d_back);
    const size_t nw = (size_t)blocks * waves;
    if (row_in_hbm) {
      if (e->d_crow.ensure(nw * (size_t)(a.Wcap + 2) * sizeof(double))) return WH_ENOMEM;
      a.rowg = (double *)e->d_crow.p;
    }
    if (e->d_back.ensure(nw * (size_t)(a.Lcap + 1) * (a.Wcap + 2)) || e->d_cwj.ensure(nw * (size_t)a.Lcap * a.KMAX * sizeof(int32_t)) ||
        e->d_cwv.ensure(nw * (size_t)a.Lcap * a.KMAX * sizeof(double)) || e->d_cwn.ensure(nw * (size_t)a.Lcap * sizeof(int32_t)))
      return WH_ENOMEM;
    a.back = (uint8_t *)e->d_back.p; a.cwj = (int32_t *)e->d_cwj.p; a.cwv = (double *)e->d_cwv.p; a.cwn = (int32_t *)e->d_cwn.p;
    a.counter = (int *)e->d_counter.p + 63;
    HIPCHK(hipMemsetAsync(a.counter, 0, sizeof(int), s));
    hipError_t err = launch_consensus(a, blocks, waves * kWave, lds, s);
    if (err != hipSuccess) { set_error("consensus kernel launch failed: %s", hipGetErrorString(err)); return WH_EHIP; }
  }
  if (timer_end(e, 3, s, nq > 0 ? 1 : 0)) return WH_EHIP;
  return WH_OK;
}

int wh_consensus(wh_ehmm *e, const int64_t *offsets, int64_t nq, const int64_t *qpair_off, const int32_t *pair_h,
                 const double *pair_w, const int64_t *col_offsets, const int32_t *cols, const int64_t *ret_off,
                 const int32_t *retained, const int32_t *nongaps, int32_t backbone_length, int32_t *out, int32_t *minmax) {
  if (!e || !offsets || !qpair_off || !pair_h || !pair_w || !col_offsets || !cols || !ret_off || !retained || !nongaps ||
      !out || !minmax || nq < 0) {
    set_error("wh_consensus: bad argument");
    return WH_EINVAL;
  }
  if (nq == 0) return WH_OK;
  HIPCHK(hipSetDevice(e->device));
  const int H = (int)e->hmms.size();
  const int64_t npairs = qpair_off[nq], ncols = col_offsets[npairs], nret = ret_off[H], total = offsets[nq];
  int maxpp = 0;
  for (int64_t q = 0; q < nq; q++) maxpp = std::max<int>(maxpp, (int)(qpair_off[q + 1] - qpair_off[q]));
  for (int64_t p = 0; p < npairs; p++)
    if (pair_h[p] < 0 || pair_h[p] >= H) { set_error("pair %lld: model position out of range", (long long)p); return WH_EINVAL; }
  for (int h = 0; h < H; h++)
    if (ret_off[h + 1] - ret_off[h] != e->hmms[(size_t)h].M) {
      set_error("model %d: %lld retained columns but %d match states", h, (long long)(ret_off[h + 1] - ret_off[h]), e->hmms[(size_t)h].M);
      return WH_EINVAL;
    }
  for (int64_t t = 0; t < nret; t++)
    if (retained[t] < 0 || retained[t] >= backbone_length) { set_error("retained column %d outside the backbone", retained[t]); return WH_EINVAL; }
  const void *src[10] = {offsets, qpair_off, pair_h, pair_w, col_offsets, cols, ret_off, retained, nongaps, nullptr};
  const size_t bytes[10] = {sizeof(int64_t) * (size_t)(nq + 1), sizeof(int64_t) * (size_t)(nq + 1), sizeof(int32_t) * (size_t)npairs,
                            sizeof(double) * (size_t)npairs, sizeof(int64_t) * (size_t)(npairs + 1), sizeof(int32_t) * (size_t)ncols,
                            sizeof(int64_t) * (size_t)(H + 1), sizeof(int32_t) * (size_t)nret, sizeof(int32_t) * (size_t)nret,
                            sizeof(int32_t) * (size_t)(total + 2 * nq)};
  for (int t = 0; t < 10; t++) {
    if (e->c_buf[t].ensure(bytes[t] + 16)) return WH_ENOMEM;
    if (src[t] && bytes[t]) HIPCHK(hipMemcpy(e->c_buf[t].p, src[t], bytes[t], hipMemcpyHostToDevice));
  }
  int32_t *d_out = (int32_t *)e->c_buf[9].p, *d_mm = d_out + total;
  int rc = wh_consensus_dev(e, (const int64_t *)e->c_buf[0].p, nq, max_query_len(offsets, nq), (const int64_t *)e->c_buf[1].p,
                            (const int32_t *)e->c_buf[2].p, (const double *)e->c_buf[3].p, (const int64_t *)e->c_buf[4].p,
                            (const int32_t *)e->c_buf[5].p, (const int64_t *)e->c_buf[6].p, (const int32_t *)e->c_buf[7].p,
                            (const int32_t *)e->c_buf[8].p, backbone_length, maxpp, d_out, d_mm, nullptr);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, d_out, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(minmax, d_mm, sizeof(int32_t) * (size_t)(2 * nq), hipMemcpyDeviceToHost));
  return WH_OK;
}

int wh_ehmm_max_query_len(const wh_ehmm *e) {
  if (!e) return WH_EINVAL;
  // Long queries keep their special-state rows in HBM; what remains in LDS per wave is the
  // residue buffer, so the bound is one wave's residues beside the largest model's tables
  // (the HBM workspace grows with L x M and may still fail with WH_ENOMEM).
  int best = 1 << 20;
  for (auto &kv : e->by_q) {
    const size_t table = (size_t)(e->K + 2 * FW_NARR) * kv.first * kWave * sizeof(float);
    const size_t left = kLdsBudget - kLdsHeader - table - (32 + kRegsInts + 8) * sizeof(float);
    best = std::min(best, (int)std::min<size_t>(left, 1u << 20));
  }
  return best;
}

}  // extern "C"
