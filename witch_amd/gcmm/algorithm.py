"""search() for the '-p <hmmdir>' style hand-off: writes the per-(HMM, chunk) result files
the reference's readers expect (witch_msa/gcmm/algorithm.py:524-537, loader.py:277-294):
<hmmdir>/root/A_0_<idx>/hmmsearch.results.A_0_<idx>.fragment_chunk_<i> containing
str({taxon: (evalue, bitscore)}).  E-values are never used by WITCH (loader.py:293); a
placeholder 0.0 is written."""
import os
import re

from .engine import current_engine


def search(hmm_dirs, chunk_of_taxon=None):
    """hmm_dirs: {hmm index: directory A_0_<idx>}.  chunk_of_taxon: optional {taxon: chunk};
    default puts every query into chunk 0.  Returns the list of files written."""
    eng = current_engine()
    written = []
    for col, label in enumerate(eng.hmm_index.tolist()):
        d = hmm_dirs[label]
        os.makedirs(d, exist_ok=True)
        per_chunk = {}
        for row, taxon in enumerate(eng.taxa):
            if eng.flags[row, col] & 1:
                c = 0 if chunk_of_taxon is None else chunk_of_taxon[taxon]
                per_chunk.setdefault(c, {})[taxon] = (0.0, float(eng.decibits[row, col]) / 10.0)
        if not per_chunk:
            per_chunk = {0: {}}
        for c, res in per_chunk.items():
            path = '{}/hmmsearch.results.A_0_{}.fragment_chunk_{}'.format(d, label, c)
            with open(path, 'w') as f:
                f.write(str(res))
            written.append(path)
    return written


def evalHMMSearchOutput(path):
    """Parser of hmmsearch's per-sequence table, same contract as algorithm.py:579-605
    (used by the level-0 shim tests to prove the text we emit is readable)."""
    results = {}
    pattern = re.compile(r"(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)")
    start = False
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not start and line.startswith("E-value"):
                start = True
            elif start and line == "":
                break
            elif start:
                m = pattern.search(line)
                if m is not None and m.group(0).find("--") == -1:
                    results[m.group(9).strip()] = (float(m.group(1)), float(m.group(2)))
    return results
