// Internal declarations shared by the host side (parser, profile tables, C ABI) and
// the HIP kernels of libwitch_hip.so.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/witch_hip.h"

namespace wh {

constexpr int kWave = 64;          // CDNA wavefront width
constexpr int kMaxQ = 48;          // cells per lane of the register-resident kernels -> M <= 64*kMaxQ = 3072
constexpr int kMaxQGen = 256;      // ... of the any-size float64 kernels (wh_generic.hip) -> M <= 16384
constexpr int kMaxQFast = 24;      // up to here both transition orientations stay resident in LDS
constexpr int kQRegMax = 16;       // up to this Q the transition tables live in VGPRs
constexpr int kWideQ = 24;         // cells per lane of the several-waves-per-pair scoring kernel
constexpr int kWideQReg = 12;      // ... its variant with the transition tables in registers: models of up to 8 x 64 x 12 = 6 144 nodes
constexpr int kWideQReg2 = 16;     // ... and the same with 16 cells per lane: up to 8 192 nodes (no room for the row-ahead requests of P4)
constexpr int kWideWavesMax = 8;   // ... and its largest workgroup: models of up to 8 x 64 x 24 = 12 288 nodes
// per-wave LDS block of the scoring kernels: region list (i, j) x WH_MAX_ENVELOPES, 8 spare ints, and the
// envelope results (envsc, domcorr) x WH_MAX_ENVELOPES staged for the multidomain resolver's record
constexpr int kRegsInts = 5 * WH_MAX_ENVELOPES;

void set_error(const char *fmt, ...);

// ---- one parsed HMMER3/f model + its configured local profile (host, float64) ----
struct HostHMM {
  int M = 0, K = 0, Kp = 0, alphabet = -1, nseq = 0, index = 0;
  std::string name, path;
  std::vector<double> t;      // [(M+1)*7]  MM MI MD IM II DM DD, as in the file (node 0..M)
  std::vector<double> mat;    // [(M+1)*K]
  std::vector<int32_t> map;   // [M+1]
  // configured profile
  std::vector<double> pt;     // [(M+1)*7], node 0 and node M zeroed
  std::vector<double> entry;  // [M+2]
  std::vector<double> odds;   // [Kp*(M+1)]
};

int  alphabet_sizes(int alphabet, int *K, int *Kp);
void degen_masks(int alphabet, uint32_t *mask /*[32]*/);
int  digitize(int alphabet, const char *text, int64_t n, uint8_t *out);
int  parse_hmm_file(const std::string &path, HostHMM &h);   // 0 or WH_E*
void configure_profile(HostHMM &h);

// ---- device-side model descriptor -------------------------------------------------
// Lane-blocked layout: lane r of a wavefront owns model nodes k = r*Q + q + 1, q in [0,Q).
// Tables are float arrays [arr][Q/4][64 lanes][4] so that a lane fetches 4 consecutive
// cells with one 16-byte access and a wavefront's access is one contiguous 1 KiB line.
//   fw: 8 arrays (tMM,tIM,tDM into node k; entry_k; tMI_k, tII_k; tMD,tDD into D_k)
//   bw: 8 arrays in REVERSED node order u = Mpad - k (tMM_k,tIM_k,tDM_k,tMI_k,tII_k,tMD_k,tDD_k,entry_k)
//   em: Kp arrays, odds ratio e_k(a)/f(a) in forward node order
struct DevHMM {
  int32_t M, Q, Mpad, K, Kp, nseq, index, qclass;
  int64_t fw_off, bw_off, em_off;    // offsets (in floats) into the table buffer
  int64_t gfw_off, gem_off;          // offsets (in doubles) into the float64 table buffer of the resolver
  int64_t emn_off;                   // node-major float32 emission odds [M+1][K] in the float table buffer (resolver: null2 by trace)
  int64_t esum_off;                  // ... and their prefix sums over the nodes, [M+1][K] doubles in the float64 table buffer
  int32_t wideQ, wideW;              // > 0: float32 tables laid out over wideW x 64 lanes of wideQ cells (wh_score_wide.hip) at ...
  int64_t wfw_off, wbw_off, wem_off; // ... these offsets (floats) of the table buffer
};

enum { FW_A = 0, FW_B, FW_C, FW_E, FW_MI, FW_II, FW_D1, FW_D2, FW_NARR, FW_P = FW_NARR };
enum { BW_MM = 0, BW_IM, BW_DM, BW_MI, BW_II, BW_MD, BW_DD, BW_E, BW_NARR, BW_P = BW_NARR };
// FW_P / BW_P: a 9th array per orientation in the table buffer with the in-lane running products
// of the D->D coefficients; it turns the carry fix-up of the D chain into one FMA per cell.
// Kernels copy it to LDS only if they ask for it (wh_score7.hip, WH_K7_MAXQP); measured on MI355X
// it does not pay: +3 % kernel time at two waves per SIMD, +0.5 % at three.

int  choose_Q(int M);     // cells per lane for a model of M nodes, or -1 if unsupported
void build_tables(const HostHMM &h, int Q, std::vector<float> &fw, std::vector<float> &bw,
                  std::vector<float> &em, int lanes = kWave);   // lanes > 64: the several-waves-per-pair kernel (wh_score_wide.hip)
// float64 tables of the multidomain resolver: 8 forward arrays and Kp emission rows, value of node
// k = lane*Q + q + 1 at arr*Q*64 + ((q/2)*64 + lane)*2 + q%2 (the two nodes of a pair adjacent: 16-byte accesses)
void build_tables_f64(const HostHMM &h, int Q, std::vector<double> &fw, std::vector<double> &em);

}  // namespace wh
