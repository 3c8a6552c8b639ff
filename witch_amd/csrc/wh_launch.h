// Kernel argument blocks and launcher prototypes (host <-> device glue).
#pragma once
#include <hip/hip_runtime.h>

#include "wh_common.h"

namespace wh {

// A pair with a multidomain region, handed from the scoring kernels to resolve_kernel (wh_resolve.hip)
struct ResolveRec {
  int64_t q;
  int32_t h;
  float fwdsc;                 // float32 multihit Forward score of the whole sequence (nats)
  float fwd_bits;
  int32_t nreg;                // regions found; the first nenv are stored
  int32_t nenv;
  int32_t flags;               // WH_FLAG_* collected so far
  int32_t multi_mask;          // bit e: stored region e is multidomain
  int32_t ri[WH_MAX_ENVELOPES], rj[WH_MAX_ENVELOPES];
  float envsc[WH_MAX_ENVELOPES], domcorr[WH_MAX_ENVELOPES];   // single-domain regions: scored by the scoring kernel
};

struct ScoreArgs {
  const DevHMM *hmms;          // all models of the eHMM
  const float *tables;         // table buffer (fw / bw / em arrays of every model)
  const int32_t *hmm_list;     // model positions of this size class
  int n_list;
  const uint8_t *residues;
  const int64_t *offsets;
  int64_t nq;
  int QB;                      // queries per work item
  int n_qblocks;
  int n_items;                 // n_list * n_qblocks
  int *counter;                // work-queue head (zeroed by a memset node before the launch)
  int Lcap;                    // longest query of the batch
  int SP;                      // stride of the per-row special-state arrays (>= Lcap+1)
  int wave_lds;                // floats of LDS per wave
  int spec_arrays;             // per-row special-state arrays in that block the PLANNER assumed (the launcher refuses a mismatch with its build)
  float *scratch;              // per-wave Forward slabs
  size_t scratch_stride;       // floats per wave
  int32_t *decibits;
  uint8_t *flags;
  float *fwd_bits;
  wh_pair_detail *detail;
  float *spec_scratch;         // long-query mode: per-wave special-state rows in HBM (else NULL -> LDS)
  size_t spec_stride;          // floats per wave
  int H;
  int K, Kp;
  uint32_t degen[32];
  int Klds;                    // emission rows staged in LDS (= K, or 0: read from L2)
  int no_window;               // 1: envelope Backward sweeps at full width only (WH_NO_WINDOW)
  int p2win;                   // 1: the wave blocks hold three more per-row arrays and the multihit Backward sweep tries a node window first;
                               // 2: no room for them (20- / 24-cell models): the window sweep works IN PLACE on the block, after P1's rows
                               //    were copied to <p2_backup> (a doubtful scan brings them back for the full-width sweep)
  float *p2_backup;            // p2win == 2: per resident wave spec_arrays x SP floats
  size_t p2_backup_stride;     // floats per wave
  int dbg;                     // timing experiments only: 1 = skip Forward-row stores, 2 = skip Forward-row loads
  float keep_scale;            // Forward-row spill threshold relative to E(row); 0 = the kernel's default (2^-24)
  int spill_band;              // 1: an envelope's Forward sweep stores only the lane blocks around P1's dominant path (spill_band, wh_score7.hip)
  ResolveRec *rrecs;           // queue of pairs with a multidomain region (NULL: such regions become one envelope)
  int *rcount;                 // queue length (device counter)
  int rcap;
  unsigned long long *stats;   // WH_STATS: [4..11] wave cycles per phase (or NULL)
  unsigned long long *paths;   // 6 counters (always counted, one atomic per wave and counter at the end of the launch): envelope Backward
                               // sweeps on a 256-node window, on a 512-node window, windows that failed the certificate, full-width sweeps;
                               // multihit Backward sweeps kept from a node window, windows whose region scan was in doubt (redone at full width)
  const int32_t *qorder;       // long-model kernel: queries in descending length order (or NULL: input order)
};

// ---- staged scoring launches (wh_staged.hip): the five sweeps of a pair as kernels of their own, each at the
// occupancy it can use, over batches of pairs whose intermediate results live in HBM.
enum { ST_CLS_NONE = 0, ST_CLS_FULL = 1, ST_CLS_DENSE = 2, ST_CLS_W256 = 4, ST_CLS_W512 = 8 };   // what an envelope still needs
// Per pair of a batch:
struct StPair {
  float xC; int32_t ef;              // P1: C(L) and its scale exponent (the multihit Forward score)
  uint32_t um_lo, um_hi;             // P1: lane blocks the dominant alignment runs through (places the window of P2)
  uint32_t um_steady;                // P1: range of the steady blocks among them (the envelopes' band; forward_sweep, wh_device.h)
  int32_t state;                     // 0 nothing to do (empty / too long / no Forward mass: result written by P1), 1 P1 done,
                                     // 2 regions known, 3 the windowed region scan was in doubt (full-width P2 follows),
                                     // 4 the window of P2 needs 512 nodes
  int32_t nenv, nreg, flags;         // regions: flags = WH_FLAG_* | multidomain mask << 8
  int32_t path;                      // WH_PATH_* bits of the pair (wh_set_path_buffer)
  int32_t pad[3];
  int32_t regs[2 * WH_MAX_ENVELOPES];
  float envsc[WH_MAX_ENVELOPES], domcorr[WH_MAX_ENVELOPES];
  int32_t uid[WH_MAX_ENVELOPES];     // per envelope: its unit (Forward slab) in the batch
  uint8_t cls[WH_MAX_ENVELOPES];     // per envelope: ST_CLS_* - the Backward sweep it still needs
};
// Per envelope unit of a batch (one Forward slab each):
struct StUnit {
  float xC; int32_t ef;              // P3: unihit Forward of the envelope
  int32_t m0, pad;                   // first reversed node of the window P4 runs on
};
enum { ST_C_P1 = 0, ST_C_P2, ST_C_P2W8, ST_C_P2B, ST_C_P3, ST_N_UNITS, ST_C_256, ST_C_512, ST_C_FULL, ST_C_DENSE, ST_OVERFLOW, ST_NCOUNT };
struct StagedArgs {
  ScoreArgs a;                       // the class launch (tables, queries, outputs, items) as the fused kernel gets it
  int item0, n_items_b;              // this batch: work items [item0, item0 + n_items_b) of the class launch
  int NB;                            // pairs per batch = n_items_b * QB (per-pair arrays are indexed inside the batch)
  int NS;                            // envelope units (Forward slabs) a batch may use
  int G;                             // work items a workgroup draws at a time in THIS launch
  int cand_cap;                      // entries of the workgroup's candidate list in LDS (behind the wave blocks)
  StPair *pairs;
  float *p1spec; size_t p1stride;    // per pair: P1's six per-row arrays (floats per pair)
  StUnit *units;
  float *p3spec; size_t p3stride;    // per unit: P3's six per-row arrays
  float *slabs; size_t slab_stride;  // per unit: Forward rows (floats per unit)
  int *cnt;                          // ST_NCOUNT counters of the batch (zeroed before its first launch)
  uint8_t *pair_paths;               // optional [nq x H]: WH_PATH_* bits per pair (or NULL)
};
// <threads> = 64 x waves; <lds> covers the header, the tables of the kind, the wave blocks and the candidate list
hipError_t launch_staged_p1(int Q, const StagedArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_staged_p2win(int Q, int QB, const StagedArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_staged_p2full(int Q, const StagedArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_staged_p3(int Q, const StagedArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_staged_p4win(int Q, int QB, const StagedArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_staged_p4full(int Q, const StagedArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_staged_dense(int Q, const StagedArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_staged_env(int Q, const StagedArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_staged_assemble(const StagedArgs &a, hipStream_t s);

hipError_t launch_score_big(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_score7(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
hipError_t launch_score7b(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s);   // A/B slot
hipError_t launch_score7q(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s);   // four envelopes per Backward sweep (Q = 16)
// two queries per wavefront (wh_score9.hip): ScoreArgs::wave_lds = 2 x score9_block_floats(SP, Lcap), scratch_stride = two slabs
hipError_t launch_score9(int Q, const ScoreArgs &a, int blocks, int threads, size_t lds, hipStream_t s);
int score9_block_floats(int SP, int Lcap);

struct ResolveArgs {
  const DevHMM *hmms;
  const double *gtab;          // float64 tables of every model (DevHMM::gfw_off / gem_off)
  const float *ftab;           // float32 table buffer (DevHMM::emn_off: node-major emission odds)
  const uint8_t *residues;
  const int64_t *offsets;
  const ResolveRec *recs;
  const int32_t *order;        // queue positions in processing order (model by model, longest pairs first inside a model), or NULL
  const int32_t *chunks;       // the queue's segments, one per model: (first entry of <order>, entries, model position, cells per lane) x n_chunks
  int n_chunks;
  const int32_t *slots;        // what a workgroup draws: segment numbers, a model's in proportion to its share of the work
  int n_slots;
  int *cursors;                // per segment: next entry to hand out (zeroed before the launch)
  int lds_tables;              // > 0: a chunk's model of up to this many cells per lane has its float64 transition arrays staged in LDS
  int wave_lds_ints;           // 4-byte units of LDS per wave
  const int *count;            // number of queued pairs (device)
  int rec_cap;
  int *counter;                // work-queue head (chunks)
  int Lcap, Mmax;
  double *mx;                  // per-wave matrix slabs
  size_t mx_stride;            // doubles per wave
  size_t dc_off;               // offset (doubles, even) of the walk's threshold-line cache inside a wave's slab
  unsigned launch_id;          // differs from launch to launch of a handle: part of the cache's validation bits (the slabs persist)
  int null2_gather;            // WH_RES_NULL2_GATHER (environment, per call): null2 by trace from one table row per residue (the pre-round-4 path)
  int32_t *segs;               // per-wave segment arrays
  size_t seg_stride;           // ints per wave
  int seg_cap;
  int32_t *decibits;
  uint8_t *flags;
  wh_pair_detail *detail;
  int H, K, Kp;
  uint32_t degen[32];
  unsigned long long *stats;   // WH_STATS: wave cycles per phase [0] region Forward [1] traces [2] clustering [3] cluster statistics [4] envelope Forward
  int dbg;                     // WH_RDBG > 0: print the first <dbg> sampled segments and the cluster statistics of every region
  int *err;                    // device counter: records dropped because their model is not the segment's (the host turns it into WH_EHIP)
  const int32_t *rext;         // long-list pass (wh_api.hip): record t's regions are NOT in the record but here, at rext + t * rext_stride:
  int64_t rext_stride;         // kRextInts ints per region (first row, last row, envsc bits, domcorr bits, multidomain), ResolveRec::nenv of them
};
constexpr int kRextInts = 5;
hipError_t launch_resolve(const ResolveArgs &a, int blocks, int waves, size_t lds, hipStream_t s);
size_t resolve_lds_header_bytes(int Qt);
// cost estimate of every queued pair (cells of its multidomain regions) for the longest-first order
hipError_t launch_resolve_keys(const ResolveRec *recs, int n, const DevHMM *hmms, float *keys, int32_t *models, hipStream_t s,
                               const int32_t *rext = nullptr, int64_t rext_stride = 0);
size_t resolve_lds_bytes(int Lcap, int Mmax);
int resolve_seg_cap();
int resolve_waves_per_cu();
size_t resolve_seg_ints(int Lcap, int Mmax);
size_t resolve_dcache_doubles();
size_t resolve_tail_row_doubles();

// models of any size (wh_generic.hip)
// several wavefronts per pair (wh_score_wide.hip): models of 3 073 - 12 288 nodes
struct WideArgs {
  const DevHMM *hmms;
  const float *tables;
  const int32_t *hmm_list;     // model positions of this launch (all with the same wideW)
  int n_list;
  const uint8_t *residues;
  const int64_t *offsets;
  int64_t nq;
  int *counter;
  int Lcap, SP;
  float *scratch;              // per-workgroup Forward slabs
  size_t scratch_stride;       // floats per workgroup
  int32_t *decibits;
  uint8_t *flags;
  float *fwd_bits;
  wh_pair_detail *detail;
  int H, K, Kp;
  uint32_t degen[32];
  ResolveRec *rrecs;
  int *rcount;
  int rcap;
  const int32_t *qorder;
  int em_lds;                  // 12-cell kernels: the emission rows of the canonical residues are staged in LDS (behind the block)
  unsigned long long *stats;   // WH_STATS: [0..4] cycles of the first wave in P1, P2, region scan, P3, P4; [5] whole items (or NULL)
  int sparse;                  // envelope Forward rows: only the lanes above 2^-24 of the row's E are stored (+ masks behind the slab)
};
struct WideAlignArgs {
  const DevHMM *hmms;
  const float *tables;
  const uint8_t *residues;
  const int64_t *offsets;
  const int32_t *items;        // pair indices served by this launch (models with the same wideW)
  int n_items;
  const int64_t *pair_q;
  const int32_t *pair_h;
  const int64_t *col_off;
  int32_t *cols;
  int32_t *status;             // per pair: set to 1 when the pair left float32 range (the float64 kernel then redoes it)
  int *counter;
  int Lcap, SP;
  float *scratch;              // per workgroup: Forward/posterior rows, then OA rows
  size_t scratch_stride;       // floats per workgroup
  int K, Kp;
};
size_t wide_align_lds_bytes(int Lcap);
hipError_t launch_align_wide(int Q, const WideAlignArgs &a, int blocks, int waves, size_t lds, hipStream_t s);
size_t wide_lds_bytes(int Lcap, size_t em_floats = 0);
hipError_t launch_score_wide(int Q, const WideArgs &a, int blocks, int waves, size_t lds, hipStream_t s);

struct GenericArgs {
  const DevHMM *hmms;
  const double *gtab;
  const int32_t *hmm_list;     // model positions served by the generic kernels
  int n_list;
  const uint8_t *residues;
  const int64_t *offsets;
  int64_t nq;
  int *counter;                // work-queue head
  int Lcap, Qmax;
  double *slab;                // per-wave workspace
  size_t slab_stride;          // doubles per wave
  int32_t *decibits;
  uint8_t *flags;
  float *fwd_bits;
  wh_pair_detail *detail;
  int H, K, Kp;
  uint32_t degen[32];
  ResolveRec *rrecs;           // EVERY pair with a region is finished by resolve_kernel
  int *rcount;
  int rcap;
  // long-list pass: the pairs whose region list did not fit WH_MAX_ENVELOPES in the kernel that scored them (WH_FLAG_TRUNC)
  // are scored again here with a region list in HBM.  Work item t = pair_list[t] (q * H + h), its record is rrecs[t] and its
  // regions are rext + t * rext_stride (kRextInts ints each, up to ext_cap of them).  NULL: the hmm_list x nq mode above.
  const int64_t *pair_list;
  int64_t n_pairs;
  int32_t *rext;
  int64_t rext_stride;
  int ext_cap;
};
// pairs flagged WH_FLAG_TRUNC, appended to <list> (up to <cap>; <count> keeps counting)
hipError_t launch_trunc_list(const uint8_t *flags, int64_t npairs, int *count, int64_t *list, int cap, hipStream_t s);
struct GenericAlignArgs {
  const DevHMM *hmms;
  const double *gtab;
  const uint8_t *residues;
  const int64_t *offsets;
  const int32_t *items;        // pair indices served by this launch
  int n_items;
  const int64_t *pair_q;
  const int32_t *pair_h;
  const int64_t *col_off;
  int32_t *cols;
  int32_t *status;             // per pair: 0 ok, 1 no path has probability, 2 traceback found no cell (or NULL)
  int *counter;
  int Lcap, Qmax, Kp;
  double *slab;
  size_t slab_stride;
};
hipError_t launch_generic_front(const GenericArgs &a, int blocks, size_t lds, hipStream_t s);
hipError_t launch_generic_align(const GenericAlignArgs &a, int blocks, size_t lds, hipStream_t s);
size_t generic_front_doubles(int Lcap, int Qmax);
size_t generic_align_doubles(int Lcap, int Qmax);
size_t generic_lds_bytes(int Lcap);

// final transitive merge (wh_merge.hip)
struct MergeArgs {
  const uint8_t *q_text;       // query characters as given (ASCII), concatenated
  const int64_t *q_off;        // [nq+1]
  const int32_t *codes;        // consensus kernel output per residue
  const int32_t *q_row;        // per query: >= 0 the query gets a row; -1 widens the gaps only; -2 no alignment
  const int64_t *row_q;        // per query row (in output order after the backbone rows): its query
  int64_t nq;
  const uint8_t *bb;           // backbone rows [nb][B], upper case
  int32_t nb, B;
  int32_t *W;                  // [B+1] widest insertion run per gap (zeroed before merge_runs)
  int32_t *res_gap, *res_k;    // per residue: gap of an insertion (or -1) and position inside its run
  long long *gap_start, *col_pos, *width;   // [B+1], [B], [1]
  uint8_t *out_full, *out_masked;
};
hipError_t launch_merge_runs(const MergeArgs &a, hipStream_t s);
hipError_t launch_merge_layout(const MergeArgs &a, hipStream_t s);
hipError_t launch_merge_render(const MergeArgs &a, int64_t nrows, hipStream_t s);

struct TopkArgs {
  const int32_t *decibits;
  const uint8_t *flags;
  const int32_t *nseq;         // [H]
  const int32_t *hmm_index;    // [H]
  int64_t nq;
  int H, k;
  int32_t *idx;
  double *w;
  int32_t *n_kept;
  int32_t *n_used;
};
hipError_t launch_topk(const TopkArgs &a, hipStream_t s);

constexpr int kAlignSpecArrays = 14;   // special-state rows per wave of the alignment kernel (AL_NARR)
struct AlignArgs {
  const DevHMM *hmms;
  const float *tables;
  const uint8_t *residues;
  const int64_t *offsets;
  const int64_t *pair_q;
  const int32_t *order;        // pair indices grouped by model
  const int32_t *item_h;       // work items: model position, first entry of <order>, entry count
  const int32_t *item_start;
  const int32_t *item_count;
  int n_items;
  const int64_t *col_offsets;
  int32_t *cols;
  int *counter;
  int Lcap, SP, wave_lds;
  float *scratch;              // per-wave slabs: Forward/posterior rows, then OA rows
  size_t scratch_stride;       // floats per wave
  float *spec_scratch;         // long-query mode: per-wave special-state rows in HBM (else NULL -> LDS)
  size_t spec_stride;
  int Klds;                    // emission rows staged in LDS (= K, or 0: read from L2)
  int K, Kp;
  int *redo_count;             // pairs whose Backward sweep saturated (float32 range) are appended to
  int32_t *redo_list;          // redo_list for the log-space pass (NULL in that pass)
  int logsp;                   // 1: this launch is the log-space pass
  int swap;                    // 1: pass-synchronous variant (one table orientation resident in LDS)
  int no_window;               // 1: Backward / OA / traceback at full width only (WH_NO_WINDOW)
  int *wstat;                  // NULL or 4 counters: pairs aligned on a 256-node window, window rejected, window not tried, 512-node window
  unsigned long long *wcyc;    // NULL or 4 wave-cycle sums of the window pairs: Forward, Backward + posteriors, OA fill, traceback
};
hipError_t launch_align(int Q, const AlignArgs &a, int blocks, int threads, size_t lds, hipStream_t s);

struct ConsArgs {
  const int64_t *offsets;
  int64_t nq;
  const int64_t *qpair_off;
  const int32_t *pair_h;
  const double *pair_w;
  const int64_t *col_offsets;
  const int32_t *cols;
  const int64_t *ret_off;
  const int32_t *retained;
  const int32_t *nongaps;
  int backbone_length;
  int32_t *out;
  int32_t *minmax;
  int *counter;
  int Lcap, Wcap, KMAX;
  uint8_t *back;               // per wave (Lcap+1) x (Wcap+2) back-pointers
  int32_t *cwj;                // per wave Lcap x KMAX edge columns
  double *cwv;                 // per wave Lcap x KMAX edge weights
  int32_t *cwn;                // per wave Lcap edge counts
  double *rowg;                // per wave Wcap+2 doubles: the DP row when it does not fit in LDS (else NULL)
};
hipError_t launch_consensus(const ConsArgs &a, int blocks, int threads, size_t lds, hipStream_t s);

}  // namespace wh
