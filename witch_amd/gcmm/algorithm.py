"""search(): the all-against-all step of the reference (SearchAlgorithm.search,
witch_msa/gcmm/algorithm.py:273-336) answered from the batched GPU run.

Writes what the reference's readers expect (algorithm.py:524-537, loader.py:277-294):
<hmmdir>/root/A_0_<idx>/hmmsearch.results.<label>.fragment_chunk_<i> containing
str({taxon: (evalue, bitscore)}), one file per (HMM, query chunk), with the reference's chunk
layout: num_chunks = lcm(#HMMs, #cpus) // #HMMs (algorithm.py:280-284), at most 20 000
sequences per chunk (:209), queries dealt round-robin names[i::chunks] in input order
(helpers/alignment_tools.py:674-686); empty chunks are skipped and the remaining ones keep
their own number (:376-383).  Query names with blanks or tabs are rejected with the
reference's ValueError (:351-359).  E-values are never used downstream (loader.py:293:
only scores[1]); a placeholder 0.0 is written.
"""
import math
import os

from .engine import current_engine

MAX_CHUNK_SIZE = 20000          # algorithm.py:209 (default in SEPP/UPP)


def check_query_names(names):
    """algorithm.py:351-359."""
    bad = [n for n in names if (' ' in n) or ('\t' in n)]
    if bad:
        raise ValueError(
            "Your input fragment file contains {} sequences, ".format(len(bad)) +
            "which names contain either whitespaces or tabs '\\t'. " +
            "Their names are:\n {}".format("'\n'  ".join(bad)))


def num_chunks_for(n_hmms, num_cpus):
    """lcm(#HMMs, #cpus) // #HMMs (algorithm.py:280-284, helpers/math_utils.py)."""
    return (n_hmms * num_cpus // math.gcd(n_hmms, num_cpus)) // n_hmms


def divide_to_equal_chunks(names, chunks, max_chunk_size=MAX_CHUNK_SIZE):
    """Round-robin chunks in input order; None for an empty chunk (alignment_tools.py:674-686)."""
    names = list(names)
    if max_chunk_size and len(names) / chunks > max_chunk_size:
        chunks = len(names) // max_chunk_size + 1
    return [names[i:len(names):chunks] or None for i in range(chunks)]


def search(hmm_dirs, num_cpus=1, chunk_of_taxon=None, fragment_chunk_dir=None, sequences=None):
    """hmm_dirs: {hmm index: directory A_0_<idx>}.  Returns (result files written, fragment chunk
    FASTA paths) - the second list is what SearchAlgorithm.search returns (algorithm.py:336); the
    FASTA files are written only if <fragment_chunk_dir> and <sequences> ({taxon: text}) are given.
    chunk_of_taxon overrides the layout (tests)."""
    eng = current_engine()
    if eng.world > 1:
        raise RuntimeError("gcmm.search writes the result files of the WHOLE batch: run it on a one-rank engine")
    check_query_names(eng.taxa)
    if chunk_of_taxon is None:
        chunks = divide_to_equal_chunks(eng.taxa, num_chunks_for(len(hmm_dirs), max(1, int(num_cpus))))
        chunk_of_taxon = {t: i for i, c in enumerate(chunks) if c for t in c}
        live = [i for i, c in enumerate(chunks) if c]
    else:
        live = sorted(set(chunk_of_taxon.values()))
    frag_paths = []
    if fragment_chunk_dir is not None and sequences is not None:
        os.makedirs(fragment_chunk_dir, exist_ok=True)
        for i in live:
            path = '{}/fragment_chunk_{}.fasta'.format(fragment_chunk_dir, i)
            with open(path, 'w') as f:
                for t in eng.taxa:
                    if chunk_of_taxon[t] == i:
                        f.write('>{}\n{}\n'.format(t, sequences[t]))
            frag_paths.append(path)
    written = []
    for col, label in enumerate(eng.hmm_index.tolist()):
        d = hmm_dirs[label]
        os.makedirs(d, exist_ok=True)
        per_chunk = {i: {} for i in live} or {0: {}}
        for row, taxon in enumerate(eng.taxa):
            if eng.flags[row, col] & 1:
                per_chunk[chunk_of_taxon[taxon]][taxon] = (0.0, float(eng.decibits[row, col]) / 10.0)
        for c, res in per_chunk.items():
            path = '{}/hmmsearch.results.A_0_{}.fragment_chunk_{}'.format(d, label, c)
            with open(path, 'w') as f:
                f.write(str(res))
            written.append(path)
    return written, frag_paths
