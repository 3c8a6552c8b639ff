"""Checkpoint wire format of the per-query alignments.

The reference appends one gzip member per finished query to <outdir>/checkpoint_alignments.txt.gz
(witch_msa/gcmm/callback.py:19-26: the line 'taxon\\tsequence\\n', utf-8, gzip.open(path, 'ab'))
and on resume reads the file back (witch_msa/gcmm/loader.py:95-150): every line but the text after
the last newline; taxon = everything before the LAST tab; column labels rebuilt from the case of
the characters (lowercase = insertion -1, -2, ..., anything else = backbone column 0, 1, ...).
A later line for the same taxon replaces an earlier one (dict assignment, loader.py:141-142).
The batched GPU path produces all strings at once, so writeCheckpointAlignments emits them in one
pass - as the same concatenation of single-line gzip members, byte-compatible with what the
reference's callback leaves behind and with what its reader accepts.
"""
import gzip

from .merge import QueryAlignment


def callback_queryAlignment(success, ignored, retry, i_retry, query, index, taxon_name, checkpoint_path):
    """Same contract as callback.py:9-29 for ONE finished query."""
    if (not query) and i_retry > 0:
        retry.append(index)
        return
    if (not query) or len(query) == 0:
        ignored.append(taxon_name)
        return
    if len(query) != 1:
        return
    line = '{}\t{}\n'.format(taxon_name, query[taxon_name])
    with gzip.open(checkpoint_path, 'ab') as f:
        f.write(line.encode('utf-8'))
    success.append(query)


def writeCheckpointAlignments(queries, checkpoint_path, mode='ab'):
    """Append every non-empty query alignment ({taxon: string} objects, 'skipped' entries ignored):
    one gzip member per query, exactly the bytes the reference's callback would have appended."""
    n = 0
    with open(checkpoint_path, mode) as raw:
        for q in queries:
            if q == 'skipped' or not q or len(q) != 1:
                continue
            taxon = next(iter(q.keys()))
            raw.write(gzip.compress('{}\t{}\n'.format(taxon, q[taxon]).encode('utf-8')))
            n += 1
    return n


def readCheckpointAlignments(path, pool=None, lock=None):
    """{taxon: query alignment} like loader.py:117-150 (the pool only parallelises the parse there)."""
    with gzip.open(path, 'rb') as f:
        lines = f.read().decode('utf-8').split('\n')[:-1]
    out = {}
    for line in lines:
        parts = line.split('\t')
        taxon, seq = '\t'.join(parts[:-1]), parts[-1]
        q = QueryAlignment()
        q[taxon] = seq                               # labels follow from the case of the characters, lazily
        out[taxon] = q
    return out
