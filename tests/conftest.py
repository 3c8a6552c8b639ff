import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["dna_hmmbuild", "dna_synth", "amino_hmmbuild", "example_sub30", "example_ehmm", "amino_multidomain"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_fasta(path):
    names, seqs = [], []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            names.append(line[1:].split()[0])
            seqs.append("")
        elif line:
            seqs[-1] += line
    return names, seqs


class GoldenCase:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(GOLDEN, name)
        with gzip.open(os.path.join(self.dir, "golden.json.gz"), "rt") as f:
            self.g = json.load(f)
        self.qnames, self.qseqs = read_fasta(os.path.join(self.dir, "queries.fasta"))
        self.hmm_paths = [self._materialize(hf) for hf in self.g["hmm_files"]]
        self.hmm_files = self.g["hmm_files"]
        self.hmm_index = self.g["hmm_index"]
        self.nseq = self.g["nseq"]
        self.k = self.g["k"]
        self.alphabet = self.g["alphabet"]


    def _materialize(self, hf):
        """Big model files are committed gzipped; unpack them once into the temp dir."""
        path = os.path.join(self.dir, hf)
        if os.path.exists(path) or not os.path.exists(path + ".gz"):
            return path
        import tempfile
        d = os.path.join(tempfile.gettempdir(), "witch_golden_%d_%s" % (os.getuid(), self.name))
        os.makedirs(d, exist_ok=True)
        out = os.path.join(d, os.path.basename(hf))
        if not os.path.exists(out):
            tmp = "%s.tmp.%d" % (out, os.getpid())          # several ranks may unpack at once
            with gzip.open(path + ".gz", "rb") as fi, open(tmp, "wb") as fo:
                fo.write(fi.read())
            os.replace(tmp, out)
        return out


_cache = {}


def load_case(name):
    if name not in _cache:
        _cache[name] = GoldenCase(name)
    return _cache[name]


@pytest.fixture(params=CASES)
def golden_case(request):
    return load_case(request.param)
