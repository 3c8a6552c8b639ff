"""Per-query merge with the reference's shape (witch_msa/gcmm/aligner.py:350-538,
witch-ng mode): the weighted consensus DP runs on the GPU for all queries at once
(wh_consensus); this module rebuilds the reference's strings and labels from its output."""
import re

from .engine import current_engine


def compressInsertions(seq):
    """helpers/alignment_tools.py:1356-1384: lowercase letters in front of the first and after
    the last aligned block are packed against the block (gaps moved outward)."""
    alns = [(m.start(), m.end()) for m in re.finditer(r'[A-Z]+', seq)]
    if len(alns) == 0:
        return seq
    f_end, b_start = alns[0][0], alns[-1][1]
    front = seq[:f_end].replace('-', '')
    back = seq[b_start:].replace('-', '')
    return front + '-' * (f_end - len(front)) + seq[f_end:b_start] + '-' * (len(seq) - b_start - len(back)) + back


def trace_to_string(seq, codes, backbone_length):
    """codes[r] >= 0: residue r sits in that backbone column (uppercase); codes[r] = -1 - nc: it is
    an insertion in front of backbone column nc (lowercase).  Gaps fill untouched columns."""
    parts, c = [], 0
    for ch, code in zip(seq, codes):
        if code >= 0:
            parts.append('-' * (code - c))
            parts.append(ch.upper())
            c = code + 1
        else:
            nc = -1 - code
            parts.append('-' * (nc - c))
            parts.append(ch.lower())
            c = nc
    parts.append('-' * (backbone_length - c))
    return compressInsertions(''.join(parts))


class QueryAlignment(dict):
    """The part of the reference's ExtendedAlignment that the callers of alignSubQueriesNew
    use: {taxon: aligned string}, _col_labels, get_length()."""
    def __init__(self):
        super().__init__()
        self._col_labels = []

    def get_length(self):
        return len(next(iter(self.values()))) if len(self) else 0


def alignSubQueriesNew(backbone_path, backbone_length, index_to_hmm, lock, timeout,
                       taxon, seq, query_weights, index):
    """Returns (query alignment, index, taxon) like aligner.py:350-538.  An empty alignment means
    the query has no weights (reference: 'does not have any matching HMMs')."""
    eng = current_engine()
    query = QueryAlignment()
    if len(query_weights) == 0 or eng.merged is None:
        return query, index, taxon
    row = eng._local(eng.taxon_row[taxon], "the consensus alignment")
    lo, hi = eng.query_offsets[row], eng.query_offsets[row + 1]
    combined = trace_to_string(seq, eng.merged[lo:hi].tolist(), backbone_length)
    query[taxon] = combined
    insertion, regular = -1, 0                      # aligner.py:489-495
    for ch in combined:
        if ch.islower():
            query._col_labels.append(insertion)
            insertion -= 1
        else:
            query._col_labels.append(regular)
            regular += 1
    if query.get_length() < backbone_length:        # aligner.py:514,533-538: failure -> empty alignment
        return QueryAlignment(), index, taxon
    return query, index, taxon
