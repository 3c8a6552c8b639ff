"""Per-query merge with the reference's shape (witch_msa/gcmm/aligner.py:350-538,
witch-ng mode): the weighted consensus DP runs on the GPU for all queries at once
(wh_consensus); this module rebuilds the reference's strings and labels from its output."""
import re

import numpy as np

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
    an insertion in front of backbone column nc (lowercase).  Gaps fill untouched columns; leading /
    trailing insertions are packed to the ends like compressInsertions does (numpy, no per-character loop)."""
    codes = np.asarray(codes, dtype=np.int64)
    n = codes.size
    if n == 0:
        return '-' * backbone_length
    ch = np.frombuffer(seq.encode('ascii'), dtype=np.uint8)
    is_alpha = ((ch >= 65) & (ch <= 90)) | ((ch >= 97) & (ch <= 122))
    ins = codes < 0
    col = np.where(ins, -1 - codes, codes)
    ins_before = np.cumsum(ins) - ins                 # insertions emitted before residue r
    pos = col + ins_before
    out = np.full(backbone_length + int(ins.sum()), 45, dtype=np.uint8)
    out[pos] = np.where(ins, np.where(is_alpha, ch | 32, ch), np.where(is_alpha, ch & 0xDF, ch))
    # compressInsertions (alignment_tools.py:1356-1384): blocks are runs of A-Z
    upper = (out >= 65) & (out <= 90)
    if upper.any():
        f_end = int(np.argmax(upper))
        b_start = len(out) - int(np.argmax(upper[::-1]))
        if f_end:
            front = out[:f_end]
            keep = front[front != 45]
            out[:f_end] = 45
            out[:keep.size] = keep
        if b_start < len(out):
            back = out[b_start:]
            keep = back[back != 45]
            out[b_start:] = 45
            if keep.size:
                out[len(out) - keep.size:] = keep
    return out.tobytes().decode('ascii')


def column_labels(text):
    """aligner.py:489-495: lowercase characters are insertion columns -1, -2, ...; every other one is the next
    backbone column."""
    a = np.frombuffer(text.encode('ascii'), dtype=np.uint8)
    low = (a >= 97) & (a <= 122)
    return np.where(low, -np.cumsum(low), np.cumsum(~low) - 1).tolist()


class QueryAlignment(dict):
    """The part of the reference's ExtendedAlignment that the callers of alignSubQueriesNew
    use: {taxon: aligned string}, _col_labels, get_length().  The labels are derived from the string when
    first asked for (the reference's merger reads them; this package's closed-form merger does not)."""
    def __init__(self):
        super().__init__()
        self._labels = None

    @property
    def _col_labels(self):
        if self._labels is None:
            self._labels = column_labels(next(iter(self.values()))) if len(self) else []
        return self._labels

    @_col_labels.setter
    def _col_labels(self, v):
        self._labels = v

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
    combined = trace_to_string(seq, eng.merged[lo:hi], backbone_length)
    query[taxon] = combined                         # labels (aligner.py:489-495) come lazily from the string
    if query.get_length() < backbone_length:        # aligner.py:514,533-538: failure -> empty alignment
        return QueryAlignment(), index, taxon
    return query, index, taxon
