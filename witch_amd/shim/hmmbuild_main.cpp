// Level-0 executable for WITCH's `hmmbuildpath` configuration key (default.config:15-17):
// accepts the command line the reference issues (witch_msa/gcmm/algorithm.py:463-470)
//     hmmbuild --cpu 1 --<dna|rna|amino> --ere X --symfrac X --informat afa -o /dev/null MODEL ALIGNMENT
// and writes the model with wh_hmmbuild (witch_amd/csrc/wh_build.cpp, compiled into this program: pure host code,
// no GPU and no libwitch_hip.so needed).  Options that would change HMMER's model (weighting, priors, --fast
// off, --hand, fragments ...) are refused instead of being silently ignored.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/witch_hip.h"

namespace wh {
static char g_err[1024];
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
}  // namespace wh

static int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "hmmbuild (witch_hip): ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
  return 1;
}

int main(int argc, char **argv) {
  std::string mol, out_summary, name;
  double ere = -1.0, symfrac = 0.5;
  bool stats = true;                               // like hmmbuild: the file carries its three STATS LOCAL lines
  std::vector<std::string> pos;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&](const char *what) -> const char * { if (i + 1 >= argc) { fail("option %s needs a value", what); exit(1); } return argv[++i]; };
    if (a == "--dna" || a == "--rna" || a == "--amino") mol = a.substr(2);
    else if (a == "--cpu") need("--cpu");
    else if (a == "--ere") ere = atof(need("--ere"));
    else if (a == "--symfrac") symfrac = atof(need("--symfrac"));
    else if (a == "--informat") { const std::string f = need("--informat"); if (f != "afa" && f != "AFA") return fail("only --informat afa is supported (got %s)", f.c_str()); }
    else if (a == "-o") out_summary = need("-o");
    else if (a == "-n") name = need("-n");
    else if (a == "--fast" || a == "--wpb" || a == "--eent") continue;                   // HMMER's defaults, as implemented
    else if (a == "--nostats") stats = false;      // (extension) skip the E-value calibration: WITCH never reads the STATS lines
    else if (a == "-h") { printf("hmmbuild (witch_hip): hmmbuild [--cpu N] --dna|--rna|--amino [--ere X] [--symfrac X] [--nostats] --informat afa [-o FILE] [-n NAME] MODEL ALIGNMENT\n"); return 0; }
    else if (!a.empty() && a[0] == '-' && a != "-") return fail("option %s is not supported by this build (it would change the model HMMER builds)", a.c_str());
    else pos.push_back(a);
  }
  if (pos.size() != 2) return fail("expected MODEL and ALIGNMENT arguments");
  if (mol.empty()) return fail("one of --dna, --rna, --amino is required (the alphabet is not guessed)");
  if (ere < 0.0) ere = mol == "amino" ? 0.59 : 0.62;     // HMMER's defaults; the reference always passes --ere
  // ---- aligned FASTA
  std::ifstream in(pos[1]);
  if (!in) return fail("cannot open %s", pos[1].c_str());
  std::vector<std::string> rows;
  std::string line;
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
    if (line.empty()) continue;
    if (line[0] == '>') rows.emplace_back();
    else {
      if (rows.empty()) return fail("%s: sequence data before the first '>' line", pos[1].c_str());
      for (char c : line) if (c != ' ' && c != '\t') rows.back().push_back(c);
    }
  }
  if (rows.empty()) return fail("%s: no sequences", pos[1].c_str());
  for (size_t i = 1; i < rows.size(); i++)
    if (rows[i].size() != rows[0].size()) return fail("%s: sequence %zu has %zu columns, the first has %zu", pos[1].c_str(), i + 1, rows[i].size(), rows[0].size());
  if (name.empty()) {                                     // HMMER: the alignment file's name without path and extension
    name = pos[1];
    const size_t sl = name.find_last_of('/');
    if (sl != std::string::npos) name = name.substr(sl + 1);
    const size_t dot = name.find_last_of('.');
    if (dot != std::string::npos && dot > 0) name = name.substr(0, dot);
  }
  std::vector<const char *> ptr;
  for (auto &r : rows) ptr.push_back(r.c_str());
  char *text = nullptr;
  int64_t n = 0;
  int32_t M = 0;
  double neff = 0.0;
  const int rc = wh_hmmbuild2(mol.c_str(), (int32_t)rows.size(), (int64_t)rows[0].size(), ptr.data(), name.c_str(), ere, symfrac, 0.5,
                              stats ? WH_BUILD_STATS : 0, &text, &n, &M, &neff);
  if (rc != 0) return fail("%s", wh::g_err);
  FILE *f = fopen(pos[0].c_str(), "w");
  if (!f) return fail("cannot write %s", pos[0].c_str());
  fwrite(text, 1, (size_t)n, f);
  fclose(f);
  wh_free_text(text);
  if (!out_summary.empty() && out_summary != "/dev/null") {
    FILE *o = fopen(out_summary.c_str(), "w");
    if (o) {
      fprintf(o, "# hmmbuild (witch_hip) :: profile HMM construction from a multiple alignment\n");
      fprintf(o, "# idx name                  nseq  alen  mlen eff_nseq\n");
      fprintf(o, "%-5d %-20s %5zu %5zu %5d %8.2f\n", 1, name.c_str(), rows.size(), rows[0].size(), M, neff);
      fclose(o);
    }
  }
  return 0;
}
