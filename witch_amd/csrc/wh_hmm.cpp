// Host side of the eHMM: HMMER3/f text parser, local-profile configuration and the
// lane-blocked float tables the kernels consume.
//
// Replaces what the reference leaves to the HMMER binaries when they open a model
// (hmmsearch/hmmalign invoked from witch_msa/gcmm/algorithm.py:526-532 and
// witch_msa/gcmm/aligner.py:96-100) and HMMSubset's NSEQ read (gcmm/loader.py:39-58).
// Format: SURVEY.md Appendix B.1; configuration: Appendix A.1.
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "wh_common.h"

namespace wh {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char *last_error() { return g_err; }

static const char *kDnaSym = "ACGT-RYMKSWHBVDN*~";
static const char *kAminoSym = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";

static const double kAminoBg[20] = {
    0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198,
    0.0590092, 0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639,
    0.0540978, 0.0683364, 0.0540687, 0.0673417, 0.0114135, 0.0304133};

int alphabet_sizes(int alphabet, int *K, int *Kp) {
  if (alphabet == WH_ALPH_AMINO) { *K = 20; *Kp = 29; return 0; }
  if (alphabet == WH_ALPH_DNA || alphabet == WH_ALPH_RNA) { *K = 4; *Kp = 18; return 0; }
  return WH_EINVAL;
}

void degen_masks(int alphabet, uint32_t *mask) {
  for (int i = 0; i < 32; i++) mask[i] = 0;
  if (alphabet == WH_ALPH_AMINO) {
    for (int i = 0; i < 20; i++) mask[i] = 1u << i;
    auto bit = [](char c) { return 1u << (uint32_t)(strchr(kAminoSym, c) - kAminoSym); };
    mask[21] = bit('N') | bit('D');   // B
    mask[22] = bit('I') | bit('L');   // J
    mask[23] = bit('Q') | bit('E');   // Z
    mask[24] = bit('K');              // O (pyrrolysine scored as lysine)
    mask[25] = bit('C');              // U (selenocysteine scored as cysteine)
    mask[26] = 0xFFFFFu;              // X
  } else {
    for (int i = 0; i < 4; i++) mask[i] = 1u << i;
    const uint32_t A = 1, C = 2, G = 4, T = 8;
    mask[5] = A | G;  mask[6] = C | T;  mask[7] = A | C;  mask[8] = G | T;  mask[9] = C | G;
    mask[10] = A | T; mask[11] = A | C | T; mask[12] = C | G | T; mask[13] = A | C | G;
    mask[14] = A | G | T; mask[15] = A | C | G | T;
  }
}

int digitize(int alphabet, const char *text, int64_t n, uint8_t *out) {
  const char *sym = (alphabet == WH_ALPH_AMINO) ? kAminoSym : kDnaSym;
  int bad = 0;
  for (int64_t i = 0; i < n; i++) {
    int c = toupper((unsigned char)text[i]);
    if (alphabet != WH_ALPH_AMINO) {
      if (c == 'U') c = 'T';
      else if (c == 'X') c = 'N';
      else if (c == 'I') c = 'A';
    }
    if (c == '_' || c == '.') c = '-';
    const char *p = c ? strchr(sym, c) : nullptr;
    if (p) out[i] = (uint8_t)(p - sym);
    else { out[i] = 255; bad++; }
  }
  return bad;
}

static bool next_tokens(std::ifstream &f, std::vector<std::string> &tok) {
  std::string line;
  tok.clear();
  if (!std::getline(f, line)) return false;
  std::istringstream ss(line);
  std::string w;
  while (ss >> w) tok.push_back(w);
  return true;
}

static double tok_prob(const std::string &s) { return s[0] == '*' ? 0.0 : std::exp(-atof(s.c_str())); }

int parse_hmm_file(const std::string &path, HostHMM &h) {
  std::ifstream f(path);
  if (!f) { set_error("cannot open HMM file %s", path.c_str()); return WH_EIO; }
  std::vector<std::string> tok;
  if (!next_tokens(f, tok) || tok.empty() || tok[0].rfind("HMMER3/", 0) != 0) {
    set_error("%s: not a HMMER3 text model", path.c_str());
    return WH_EIO;
  }
  h.path = path;
  bool body = false;
  while (next_tokens(f, tok)) {
    if (tok.empty()) continue;
    const std::string &key = tok[0];
    if (key == "NAME" && tok.size() > 1) h.name = tok[1];
    else if (key == "LENG" && tok.size() > 1) h.M = atoi(tok[1].c_str());
    else if (key == "NSEQ" && tok.size() > 1) h.nseq = atoi(tok[1].c_str());
    else if (key == "ALPH" && tok.size() > 1) {
      std::string a = tok[1];
      for (auto &c : a) c = (char)tolower((unsigned char)c);
      if (a == "dna") h.alphabet = WH_ALPH_DNA;
      else if (a == "rna") h.alphabet = WH_ALPH_RNA;
      else if (a == "amino") h.alphabet = WH_ALPH_AMINO;
      else { set_error("%s: unsupported alphabet %s", path.c_str(), tok[1].c_str()); return WH_EIO; }
    } else if (key == "HMM") { body = true; break; }
  }
  if (!body || h.M <= 0 || h.alphabet < 0) { set_error("%s: truncated header", path.c_str()); return WH_EIO; }
  alphabet_sizes(h.alphabet, &h.K, &h.Kp);
  const int M = h.M, K = h.K;
  h.t.assign((size_t)(M + 1) * 7, 0.0);
  h.mat.assign((size_t)(M + 1) * K, 0.0);
  h.map.assign((size_t)M + 1, 0);
  auto bad = [&](const char *what, int k) {
    set_error("%s: malformed %s at node %d", path.c_str(), what, k);
    return WH_EIO;
  };
  if (!next_tokens(f, tok)) return bad("transition header", 0);
  if (!next_tokens(f, tok)) return bad("node 0", 0);
  if (!tok.empty() && tok[0] == "COMPO") { if (!next_tokens(f, tok)) return bad("node 0", 0); }
  // tok = node-0 insert emissions (insert odds are hardwired to 1 by the profile: ignored)
  if (!next_tokens(f, tok) || tok.size() < 7) return bad("node-0 transitions", 0);
  for (int x = 0; x < 7; x++) h.t[x] = tok_prob(tok[x]);
  for (int k = 1; k <= M; k++) {
    if (!next_tokens(f, tok) || (int)tok.size() < K + 1 || atoi(tok[0].c_str()) != k) return bad("match line", k);
    for (int x = 0; x < K; x++) h.mat[(size_t)k * K + x] = tok_prob(tok[1 + x]);
    if ((int)tok.size() > K + 1 && tok[K + 1][0] != '-') h.map[k] = atoi(tok[K + 1].c_str());
    if (!next_tokens(f, tok)) return bad("insert line", k);
    if (!next_tokens(f, tok) || tok.size() < 7) return bad("transition line", k);
    for (int x = 0; x < 7; x++) h.t[(size_t)k * 7 + x] = tok_prob(tok[x]);
  }
  configure_profile(h);
  return WH_OK;
}

// Appendix A.1: local entry B->M_k = occ_k / Z, local exit = 1, node-0 and node-M
// transitions are not part of the local profile, odds ratios against the background,
// degenerate residues by background-weighted mean log-odds.
void configure_profile(HostHMM &h) {
  enum { tMM = 0, tMI, tMD, tIM, tII, tDM, tDD };
  const int M = h.M, K = h.K, Kp = h.Kp;
  double bg[20];
  for (int a = 0; a < K; a++) bg[a] = (h.alphabet == WH_ALPH_AMINO) ? kAminoBg[a] : 0.25;
  uint32_t mask[32];
  degen_masks(h.alphabet, mask);
  h.pt.assign((size_t)(M + 1) * 7, 0.0);
  h.entry.assign((size_t)M + 2, 0.0);
  h.odds.assign((size_t)Kp * (M + 1), 0.0);
  for (int k = 1; k < M; k++)
    for (int x = 0; x < 7; x++) h.pt[(size_t)k * 7 + x] = h.t[(size_t)k * 7 + x];
  std::vector<double> occ((size_t)M + 2, 0.0);
  occ[1] = h.t[tMI] + h.t[tMM];
  for (int k = 2; k <= M; k++)
    occ[k] = occ[k - 1] * (h.t[(size_t)(k - 1) * 7 + tMM] + h.t[(size_t)(k - 1) * 7 + tMI]) +
             (1.0 - occ[k - 1]) * h.t[(size_t)(k - 1) * 7 + tDM];
  double Z = 0.0;
  for (int k = 1; k <= M; k++) Z += occ[k] * (double)(M - k + 1);
  for (int k = 1; k <= M; k++) h.entry[k] = occ[k] / Z;
  for (int k = 1; k <= M; k++) {
    double sc[20];
    for (int a = 0; a < K; a++) {
      double e = h.mat[(size_t)k * K + a];
      sc[a] = e > 0.0 ? std::log(e / bg[a]) : -INFINITY;
      h.odds[(size_t)a * (M + 1) + k] = e > 0.0 ? e / bg[a] : 0.0;
    }
    for (int x = K; x < Kp; x++) {
      if (!mask[x]) continue;   // gap, '*', '~': impossible
      double num = 0.0, den = 0.0;
      for (int a = 0; a < K; a++)
        if (mask[x] & (1u << a)) { num += sc[a] * bg[a]; den += bg[a]; }
      h.odds[(size_t)x * (M + 1) + k] = std::exp(num / den);
    }
  }
}

int choose_Q(int M) {
  int q = (M + kWave - 1) / kWave;
  q = (q + 3) / 4 * 4;
  if (q < 4) q = 4;
  // instantiated classes: 4..24 (both orientations resident in LDS), 28..48 (pass-synchronous swap); beyond:
  // the any-size float64 kernels (wh_generic.hip), which take Q as a run-time value
  if (q > kMaxQGen) return -1;
  return q;
}

void build_tables(const HostHMM &h, int Q, std::vector<float> &fw, std::vector<float> &bw,
                  std::vector<float> &em, int lanes) {
  enum { tMM = 0, tMI, tMD, tIM, tII, tDM, tDD };
  const int M = h.M, Mpad = Q * lanes, Q4 = Q / 4;
  auto at = [&](int arr, int pos) -> size_t {   // pos = lane*Q + q
    int lane = pos / Q, q = pos % Q;
    return (((size_t)arr * Q4 + q / 4) * lanes + lane) * 4 + (q % 4);
  };
  fw.assign((size_t)(FW_NARR + 1) * Mpad, 0.f);   // + FW_P
  bw.assign((size_t)(BW_NARR + 1) * Mpad, 0.f);   // + BW_P
  em.assign((size_t)h.Kp * Mpad, 0.f);
  for (int k = 1; k <= M; k++) {
    const double *tp = &h.pt[(size_t)(k - 1) * 7];
    const double *tk = &h.pt[(size_t)k * 7];
    int pos = k - 1;
    fw[at(FW_A, pos)] = (float)tp[tMM];
    fw[at(FW_B, pos)] = (float)tp[tIM];
    fw[at(FW_C, pos)] = (float)tp[tDM];
    fw[at(FW_E, pos)] = (float)h.entry[k];
    fw[at(FW_MI, pos)] = (float)tk[tMI];
    fw[at(FW_II, pos)] = (float)tk[tII];
    fw[at(FW_D1, pos)] = (float)tp[tMD];
    fw[at(FW_D2, pos)] = (float)tp[tDD];
    int u = Mpad - k;   // reversed coordinate
    bw[at(BW_MM, u)] = (float)tk[tMM];
    bw[at(BW_IM, u)] = (float)tk[tIM];
    bw[at(BW_DM, u)] = (float)tk[tDM];
    bw[at(BW_MI, u)] = (float)tk[tMI];
    bw[at(BW_II, u)] = (float)tk[tII];
    bw[at(BW_MD, u)] = (float)tk[tMD];
    bw[at(BW_DD, u)] = (float)tk[tDD];
    bw[at(BW_E, u)] = (float)h.entry[k];
    for (int x = 0; x < h.Kp; x++) em[at(x, pos)] = (float)h.odds[(size_t)x * (M + 1) + k];
  }
  // FW_P / BW_P: running products of the D->D coefficients inside each lane's block, in float32
  // like the kernels' own products (the in-lane carry of the D chain becomes one FMA per cell)
  for (int lane = 0; lane < lanes; lane++) {
    float pf = 1.0f, pb = 1.0f;
    for (int q = 0; q < Q; q++) {
      pf *= fw[at(FW_D2, lane * Q + q)];
      pb *= bw[at(BW_DD, lane * Q + q)];
      fw[at(FW_P, lane * Q + q)] = pf;
      bw[at(BW_P, lane * Q + q)] = pb;
    }
  }
}

void build_tables_f64(const HostHMM &h, int Q, std::vector<double> &fw, std::vector<double> &em) {
  enum { tMM = 0, tMI, tMD, tIM, tII, tDM, tDD };
  const int M = h.M, Mpad = Q * kWave;
  // nodes 2j, 2j+1 of a lane adjacent inside every array (wh_resolve.hip ofs2: one 16-byte access per pair)
  auto at = [&](int arr, int k) -> size_t { const int q = (k - 1) % Q, ln = (k - 1) / Q; return (size_t)arr * Q * kWave + ((((size_t)(q >> 1) * kWave + ln) << 1) + (q & 1)); };
  fw.assign((size_t)8 * Mpad, 0.0);
  em.assign((size_t)h.Kp * Mpad, 0.0);
  for (int k = 1; k <= M; k++) {
    const double *tp = &h.pt[(size_t)(k - 1) * 7];
    const double *tk = &h.pt[(size_t)k * 7];
    fw[at(0, k)] = tp[tMM]; fw[at(1, k)] = tp[tIM]; fw[at(2, k)] = tp[tDM]; fw[at(3, k)] = h.entry[k];
    fw[at(4, k)] = tk[tMI]; fw[at(5, k)] = tk[tII]; fw[at(6, k)] = tp[tMD]; fw[at(7, k)] = tp[tDD];
    for (int x = 0; x < h.Kp; x++) em[at(x, k)] = h.odds[(size_t)x * (M + 1) + k];
  }
}

}  // namespace wh
