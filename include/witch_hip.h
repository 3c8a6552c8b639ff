/*
 * witch_hip.h - C ABI of libwitch_hip.so: MI355X-native query-vs-eHMM scoring,
 * weighting/top-k and optimal-accuracy alignment for WITCH.
 *
 * Every entry point replaces a piece of the reference's HMMER-subprocess path
 * (paths relative to the c5shen/WITCH checkout):
 *
 *   wh_ehmm_load   <- HMMSubset.__init__ reading NSEQ + the HMM text files that
 *                     hmmsearch/hmmalign parse      witch_msa/gcmm/loader.py:17-65
 *   wh_score*      <- SearchAlgorithm.search / subset_frag_chunk_hmmsearch running
 *                     "hmmsearch --cpu 1 --noali -E 99999999 --max" per (HMM, chunk)
 *                     and evalHMMSearchOutput        witch_msa/gcmm/algorithm.py:273-336,482-544,579-605
 *   wh_topk*       <- readAndRankBitscoreMP + calculateWeights/writeWeights
 *                                                    witch_msa/gcmm/loader.py:299-332, weighting.py:58-74,121-169
 *                     and the 0.999 cumulative-weight cut   witch_msa/gcmm/aligner.py:58-63
 *   wh_align*      <- getBackbones running "hmmalign -o OUT HMM QUERY" per chosen HMM
 *                     and decoding the Stockholm row witch_msa/gcmm/aligner.py:96-142
 *
 * Conventions: plain pointers and sizes; the caller owns every input and output
 * buffer; the library owns wh_ehmm handles and its device workspace.  Functions
 * return 0 on success or a negative WH_E* code; wh_last_error() gives the message of
 * the calling thread's last failure.  Calls on one handle must not overlap.
 * "*_dev" variants take DEVICE pointers and a hipStream_t (passed as void*) and only
 * enqueue work; the plain variants take HOST pointers and block.
 *
 * Data contract:
 *   decibits  int32  [nq x H]  bit-score x 10 exactly as hmmsearch's "%6.1f" prints it
 *   flags     uint8  [nq x H]  WH_FLAG_* (REPORTED = the pair is listed by hmmsearch)
 *   top-k     int32 idx[nq x k] (the caller's hmm_index values, -1 padded),
 *             double w[nq x k] (0 padded), n_kept[nq], n_used[nq] (0.999 prefix length)
 *             ordered by (-weight, -decibits, +hmm_index)
 *   cols      int32, CSR over the residues of each pair: 0-based match column or -1
 */
#ifndef WITCH_HIP_H
#define WITCH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WH_OK          0
#define WH_EINVAL     -1   /* bad argument                                   */
#define WH_EIO        -2   /* cannot open / parse an HMM file                */
#define WH_ENODEV     -3   /* no usable HIP device                           */
#define WH_EHIP       -4   /* a HIP runtime call failed                      */
#define WH_ERANGE     -5   /* model (> 16384 nodes) or query longer than this build supports */
#define WH_ENOMEM     -6

#define WH_ALPH_DNA    0
#define WH_ALPH_RNA    1
#define WH_ALPH_AMINO  2

#define WH_FLAG_REPORTED  1   /* pair appears in hmmsearch's per-sequence table          */
#define WH_FLAG_MULTI     2   /* a region was multidomain (HMMER's stochastic class)     */
#define WH_FLAG_OVERRIDE  4   /* reconstruction score overrode the Forward score         */
#define WH_FLAG_TRUNC     8   /* a list of the pair overflowed and part of it was dropped: see below       */
#define WH_FLAG_EXACT    16   /* an envelope failed the sparse-spill certificate and was redone dense */

/* Envelopes a wh_pair_detail record LISTS.  The SCORE of a pair has no such limit (hmmsearch has none: SURVEY A.4, called at
 * witch_msa/gcmm/algorithm.py:526-532): a pair with more regions than the scoring kernels' list holds is scored a second
 * time inside the same call by the long-list pass (float64 front end with the region list in HBM + a resolver launch of
 * its own; wh_last_score_counters out8[7] counts such pairs), every region and every envelope enters its score, and it
 * comes back WITHOUT WH_FLAG_TRUNC; its detail record lists the first WH_MAX_ENVELOPES envelopes, nregions is the full
 * count.  What can still set WH_FLAG_TRUNC: ONE multidomain region with more than 32 domains in a sampled trace or more
 * than 64 significant clusters (tandem repeats of up to 28 copies are tested), more than four million overflowing pairs
 * in one call, or the development knobs WH_NO_LONG_LIST / WH_NO_RESOLVE. */
#define WH_MAX_ENVELOPES 16

/* Which code path a pair took through the scoring kernels (optional per-pair byte, wh_set_path_buffer; written by
 * the staged launches only).  A pair's result does not depend on the path except in the last bits of the float32
 * null2 correction (window vs full width): the tests draw their oracle samples per path. */
#define WH_PATH_P2_WIN     1   /* regions from the multihit Backward sweep on a node window (certified)  */
#define WH_PATH_P2_FULL    2   /* ... from the full-width sweep (no window fitted, or a decision in doubt) */
#define WH_PATH_P4_W256    4   /* an envelope's Backward sweep kept from a 256-node window                */
#define WH_PATH_P4_W512    8   /* ... from a 512-node window                                              */
#define WH_PATH_P4_WFAIL  16   /* a window failed the mass certificate and was redone at full width       */
#define WH_PATH_P4_FULL   32   /* an envelope's Backward sweep at full width                              */
#define WH_PATH_DENSE     64   /* an envelope redone with every Forward row stored (WH_FLAG_EXACT)        */
#define WH_PATH_MULTI    128   /* finished by the multidomain resolver                                    */

typedef struct wh_ehmm wh_ehmm;

/* Optional per-pair diagnostics (tests compare them with the oracle stage by stage). */
typedef struct wh_pair_detail {
  float   fwd_bits;        /* (Forward - null1) / ln 2, multihit local               */
  float   seq_score;       /* final float32 score before the deci-bit print          */
  float   pre_score;       /* score before the null2 correction                      */
  float   seqbias_nats;
  int32_t nregions;
  int32_t nenv;
  int32_t env_i[WH_MAX_ENVELOPES];
  int32_t env_j[WH_MAX_ENVELOPES];
  float   envsc[WH_MAX_ENVELOPES];      /* nats */
  float   domcorr[WH_MAX_ENVELOPES];    /* nats */
} wh_pair_detail;

const char *wh_version(void);
const char *wh_last_error(void);

/* Select the HIP device for the calling process (one process per GPU). */
int wh_init(int device);
int wh_device_info(char *name, int name_len, int *cu_count, int64_t *hbm_bytes);

/* Digitise text residues with HMMER/Easel's alphabet rules (case-insensitive, U->T,
 * X->N for nucleic acids).  Codes >= K are degenerate; 255 marks an illegal character. */
int wh_digitize(int alphabet, const char *text, int64_t n, uint8_t *out);

/* Parse n HMMER3/f text models, configure the local profiles and upload them.
 * hmm_index[i] is the caller's label for model i (WITCH's A_0_<idx>); nseq may be NULL
 * (then NSEQ from each file header is used, as loader.py:48-53 does). */
wh_ehmm *wh_ehmm_load(const char *const *hmm_paths, const int32_t *hmm_index, const int32_t *nseq, int n);
void     wh_ehmm_free(wh_ehmm *e);
int      wh_ehmm_count(const wh_ehmm *e);
int      wh_ehmm_alphabet(const wh_ehmm *e);
int      wh_ehmm_info(const wh_ehmm *e, int32_t *M, int32_t *nseq, int32_t *hmm_index);
/* MAP annotation of model h: alignment column (1-based) of match state k=1..M, 0 if absent. */
int      wh_ehmm_map(const wh_ehmm *e, int h, int32_t *map_cols);
int      wh_ehmm_max_query_len(const wh_ehmm *e);

/* All-vs-all scoring: nq queries (digital residues, CSR offsets[nq+1]) x H models.  One call serves fewer than 2^31
 * pairs (WH_ERANGE beyond: feed the queries in chunks, as the reference feeds hmmsearch 20 000 sequences at a time,
 * witch_msa/gcmm/algorithm.py:209,280-284; witch_amd.gcmm.QueryAlignmentEngine.run does). */
int wh_score(wh_ehmm *e, const uint8_t *residues, const int64_t *offsets, int64_t nq,
             int32_t *decibits, uint8_t *flags, float *fwd_bits, wh_pair_detail *detail);
int wh_score_dev(wh_ehmm *e, const uint8_t *d_residues, const int64_t *d_offsets, int64_t nq,
                 int64_t total_residues, int32_t max_len,
                 int32_t *d_decibits, uint8_t *d_flags, float *d_fwd_bits, wh_pair_detail *d_detail,
                 void *stream);

/* Weights over the reported models of each query, deterministic top-k and 0.999 prefix. */
int wh_topk(wh_ehmm *e, const int32_t *decibits, const uint8_t *flags, int64_t nq, int k,
            int32_t *idx, double *w, int32_t *n_kept, int32_t *n_used);
int wh_topk_dev(wh_ehmm *e, const int32_t *d_decibits, const uint8_t *d_flags, int64_t nq, int k,
                int32_t *d_idx, double *d_w, int32_t *d_n_kept, int32_t *d_n_used, void *stream);

/* Optimal-accuracy alignment of query pair_q[p] against model position pair_h[p]
 * (0..H-1, the position in the load order, not the hmm_index label).
 * cols is CSR: pair p writes cols[col_offsets[p] .. col_offsets[p] + len(query)). */
int wh_align(wh_ehmm *e, const uint8_t *residues, const int64_t *offsets, int64_t nq,
             const int64_t *pair_q, const int32_t *pair_h, int64_t npairs,
             const int64_t *col_offsets, int32_t *cols);
int wh_align_dev(wh_ehmm *e, const uint8_t *d_residues, const int64_t *d_offsets, int64_t nq,
                 int64_t total_residues, int32_t max_len,
                 const int64_t *d_pair_q, const int32_t *d_pair_h, int64_t npairs,
                 const int64_t *d_col_offsets, int32_t *d_cols, void *stream);

/* Outcome classes of the last wh_align / wh_align_dev call on this handle (the call itself returns WH_OK for
 * them): n_logspace = pairs that left the float range and were redone in log space (same columns as hmmalign's
 * own log-space fallback); n_unaligned = pairs returned with ALL columns -1 where hmmalign (aligner.py:96-142) would
 * have produced an alignment: always 0 since round 5 (a pair on a model of more than 3072 nodes whose log-space Forward
 * and Backward scores disagree used to be dropped; it is now aligned from the Forward-normalised posteriors, as hmmalign
 * does, counted in n_logspace and noted on stderr).  The argument stays for binary compatibility; unaligned_pairs is
 * not written. */
int wh_last_align_status(wh_ehmm *e, int64_t *n_logspace, int64_t *n_unaligned, int64_t *unaligned_pairs, int64_t cap);

/* Which sweeps the register-kernel pairs (models of up to 3072 nodes) of the last wh_align / wh_align_dev call went
 * through - same columns either way, the split only prices the call (bench.py):
 * paths4[0] = Backward / posteriors / OA / traceback on a 256-node window around the dominant path, [3] = on a
 * 512-node window, [1] = window result not accepted (mass certificate) and redone at full width, [2] = full width
 * from the start (query too long for a window, no dominant path, models of fewer than 8 nodes per lane). */
int wh_last_align_paths(wh_ehmm *e, int64_t *paths4);

/* How the Backward sweeps of the last wh_score call ran, counted on the device by the one-wavefront-per-pair scoring
 * kernels (models of up to 24 cells per lane; the pass-synchronous, several-waves-per-pair and any-size kernels have no
 * window and are not counted).  Envelope sweeps (unihit Backward + posterior accumulation -> null2):
 * paths6[0] = envelopes whose sweep ran on a 256-node window around the dominant alignment and passed the mass
 * certificate, [1] = the same on a 512-node window, [2] = windows that failed the certificate (each then ran again at
 * full width), [3] = full-width sweeps (no window tried, a failed window, or the dense redo of a sparse spill).
 * Multihit sweeps (Backward + domain decoding): [4] = pairs whose regions come from a sweep on a node
 * window (every threshold decision of the region scan beyond the window's slack), [5] = pairs whose window left a
 * decision in doubt and whose sweep ran again at full width (pairs that never tried a window are in neither).
 * Waits for the device.  (What a window is: DESIGN.md section 4.1; WH_NO_WINDOW switches both off.) */
int wh_last_score_paths(wh_ehmm *e, int64_t *paths6);
/* The same six counters and, in out8[6], the BYTES of Forward rows the envelope sweeps of the last scoring call stored to
 * their slabs (lane blocks kept by the sparse spill x 8 bytes per cell: what the kernels ASKED the memory system to write,
 * counted on the device with one scalar add per row; the Backward sweeps read about three quarters of it back).  bench.py
 * reports it live beside the HBM-level traffic of the stamped profile: a regression of the spill shows in the driver's own
 * line (one-wavefront-per-pair kernels only).  out8[7] = pairs of the last call that went through the long-list pass
 * (more regions than WH_MAX_ENVELOPES; see there). */
int wh_last_score_counters(wh_ehmm *e, int64_t *out8);

/* Optional per-PAIR record of the same: a device array of nq x H bytes that the scoring calls made after this one fill
 * with WH_PATH_* bits (NULL switches it off again).  Written by the staged launches only (WH_SCORE_KERNEL=10; pairs
 * of the other kernels keep whatever the array held).  The buffer belongs to the caller and must stay valid for as many
 * pairs as the calls score.  tests/test_gpu_parity.py draws its headline-size oracle samples per path from it. */
int wh_set_path_buffer(wh_ehmm *e, uint8_t *d_paths);

/* Scoring passes the last wh_score call REPEATED (0 or 1).  The queue that hands pairs with a multidomain region to the
 * resolver stage is sized by estimate (5 % of the pairs, or 1.25 x the largest share an earlier call on the handle
 * queued); a call that needs more slots counts them, grows the queue and scores once more - same results, about twice
 * the scoring time of that one call.  Negative: error. */
int wh_last_queue_reruns(wh_ehmm *e);

/* Weighted consensus of each query's per-HMM alignments (witch-ng merge DP; replaces the Python
 * loops of alignSubQueriesNew, witch_msa/gcmm/aligner.py:376-473).  Pairs are grouped by
 * query in top-k order: query q owns pairs qpair_off[q] .. qpair_off[q+1]; pair p aligned
 * the query to model pair_h[p] (position 0..H-1) with weight pair_w[p] and per-residue match
 * columns cols[col_offsets[p] ..] as produced by wh_align.  retained / nongaps are the
 * reference's subset_to_retained_columns / subset_to_nongaps_per_column
 * (witch_msa/gcmm/algorithm.py:423-429), CSR over models by ret_off[H+1].
 * out (CSR by <offsets>, one int per residue): backbone column >= 0 for a match, -1 - nc for
 * an insertion placed before backbone column nc.  minmax[2q], minmax[2q+1]: first/last
 * backbone column touched (max < 0: nothing aligned). */
int wh_consensus(wh_ehmm *e, const int64_t *offsets, int64_t nq, const int64_t *qpair_off,
                 const int32_t *pair_h, const double *pair_w, const int64_t *col_offsets, const int32_t *cols,
                 const int64_t *ret_off, const int32_t *retained, const int32_t *nongaps,
                 int32_t backbone_length, int32_t *out, int32_t *minmax);
int wh_consensus_dev(wh_ehmm *e, const int64_t *d_offsets, int64_t nq, int32_t max_len, const int64_t *d_qpair_off,
                     const int32_t *d_pair_h, const double *d_pair_w, const int64_t *d_col_offsets,
                     const int32_t *d_cols, const int64_t *d_ret_off, const int32_t *d_retained,
                     const int32_t *d_nongaps, int32_t backbone_length, int32_t max_pairs_per_query,
                     int32_t *d_out, int32_t *d_minmax, void *stream);

/* Duration (ms) and launch count of the kernels of the last *_dev/plain call, measured
 * with HIP events on the stream the kernels ran on: which = 0 scoring kernels, 1 topk, 2 align, 3 consensus,
 * 4 multidomain resolver (the second part of wh_score: stage time of scoring = 0 + 4). */
int wh_last_kernel_ms(wh_ehmm *e, int which, double *ms, int *launches);
/* Timing mode only: the scoring launches of the last wh_score[_dev] call, in launch order - the cells-per-lane class of the
 * launch's models (16 = models of 961..1024 nodes ...), the kernel family (0 phase-call wh::k7::score_kernel7, 1
 * pass-synchronous wh::score_big_kernel, 2 any-size wh::generic_front_kernel, 3 several-waves-per-pair wh::wide::score_wide_kernel with
 * cells_per_lane = 24 x waves) and its HIP-event duration.  Returns the number
 * of launches (the first <cap> are written); bench.py names the measured dominant kernel from it. */
int wh_last_score_launches(wh_ehmm *e, int32_t *cells_per_lane, int32_t *kind, double *ms, int cap);
/* When enabled, every kernel launch is bracketed by HIP events (bench/roofline use). */
int wh_set_timing(wh_ehmm *e, int enabled);
/* Development knobs (WH_SCORE_KERNEL, WH_KEEP_LOG2, WH_MAX_WAVES, WH_FORCE_SPECG,
 * WH_NO_LOGSPACE, WH_STATS, WH_TRACE, WH_DBG).  The environment is read ONCE, in wh_ehmm_load; this call
 * changes a knob on a live handle (A/B harness tools/ab_score.py).  Production needs none of them. */
int wh_set_option(wh_ehmm *e, const char *name, const char *value);

/* ---- final transitive merge ---------------------------------------------------
 * Replaces mergeAlignmentsCollapsed -> ExtendedAlignment.merge_in per query (witch_msa/gcmm/merger.py:40-131,
 * helpers/alignment_tools.py:1183-1316) and the masked writer (alignment_tools.py:1140-1156, merger.py:100-103),
 * fed directly with wh_consensus' per-residue codes (code >= 0 backbone column; -1 - g insertion in front of
 * column g).  q_text: the queries' characters as given (case is normalised as the reference does: aligned
 * residues upper, insertions lower); q_row[q]: >= 0 the query gets a row (rows follow the backbone rows in query
 * order), -1 its insertions widen the gaps but it gets no row (its name exists already), -2 no alignment.
 * backbone: nb rows of B characters (already upper-cased).  Returns two malloc'ed row-major byte matrices
 * (release with wh_free_text): the full alignment rows x width and the masked one rows x B.  Needs no model
 * handle: device = the HIP device to run on. */
int wh_merge(int device, const uint8_t *q_text, const int64_t *q_off, int64_t nq, const int32_t *codes, const int32_t *q_row,
             const uint8_t *backbone, int32_t nb, int32_t B, uint8_t **out_full, uint8_t **out_masked, int64_t *out_rows,
             int64_t *out_width);
/* One process per GPU (queries sharded): the width of a gap is the MAX over all ranks' queries - the merge's one
 * exchange step.  Call once with widths_local != NULL and out_full == NULL (fills widths_local[B+1] from this
 * rank's queries, renders nothing; backbone may be NULL), all-reduce MAX over the ranks (RCCL), call again with
 * widths_global: this rank's rows (nb backbone rows first; pass nb = 0 on the ranks that do not write them) are
 * rendered in the global layout. */
int wh_merge_sharded(int device, const uint8_t *q_text, const int64_t *q_off, int64_t nq, const int32_t *codes, const int32_t *q_row,
                     const uint8_t *backbone, int32_t nb, int32_t B, int32_t *widths_local, const int32_t *widths_global,
                     uint8_t **out_full, uint8_t **out_masked, int64_t *out_rows, int64_t *out_width);

/* ---- eHMM construction (host code, no GPU needed) -------------------------------
 * Replaces the reference's per-subset call
 *     hmmbuild --cpu 1 --<molecule> --ere 0.59 --symfrac 0.0 --informat afa -o /dev/null MODEL SUBSET.fasta
 * (witch_msa/gcmm/algorithm.py:463-470).  rows: nseq aligned sequences of alen characters each (aligned
 * FASTA text, '-' '.' '_' gaps; no terminator needed), molecule "dna" | "rna" | "amino".  Writes the model as
 * HMMER3/f text (same probability fields, MAP / CONS annotation, COMPO, NSEQ / EFFN / CKSUM as hmmbuild
 * 3.1b2, MAXL included for nucleotide models; no STATS lines - see wh_hmmbuild2) into a malloc'ed buffer the caller releases with wh_free_text.  out_M /
 * out_neff (optional): model length and effective sequence number. */
int  wh_hmmbuild(const char *molecule, int32_t nseq, int64_t alen, const char *const *rows, const char *name,
                 double ere, double symfrac, double fragthresh, char **out_text, int64_t *out_len,
                 int32_t *out_M, double *out_neff);
/* Same, with options.  flags: WH_BUILD_STATS adds hmmbuild's three "STATS LOCAL MSV / VITERBI / FORWARD" lines
 * (E-value calibration on 3 x 200 random sequences, generator seeded with 42 as hmmbuild does: the MSV and Viterbi
 * and Forward locations come out in hmmbuild's printed digits on all 47 golden model files; the reference's bundled
 * hmmsearch then prints the same report, E-values included, as for hmmbuild's own file).  WITCH never reads them
 * (hmmsearch -E 99999999, only bit scores are parsed); stock HMMER refuses a file without them.  Costs about 0.5 s
 * per 1 000-node model on one core, against 15 ms for the build itself. */
#define WH_BUILD_STATS 1
int  wh_hmmbuild2(const char *molecule, int32_t nseq, int64_t alen, const char *const *rows, const char *name,
                  double ere, double symfrac, double fragthresh, int32_t flags, char **out_text, int64_t *out_len,
                  int32_t *out_M, double *out_neff);
void wh_free_text(char *text);

#ifdef __cplusplus
}
#endif
#endif /* WITCH_HIP_H */
