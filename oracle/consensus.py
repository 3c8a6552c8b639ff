"""CPU oracle for the first "next" row: the witch-ng weighted
consensus of the per-HMM alignments of one query (TEST INFRASTRUCTURE ONLY).

Restates witch_msa/gcmm/aligner.py:376-495 (edge weights, max-weight trace DP, traceback)
and witch_msa/helpers/alignment_tools.py:1356-1384 (compressInsertions) with numpy float64
in the reference's own operation order, so results are bit-identical to the reference's
Python floats.  Pinned by tests/golden/*/golden.json.gz["merged"], which holds the strings
the reference's alignSubQueriesNew returned in the build container.
"""
import re

import numpy as np


def consensus_trace(seq_len, aligned, weights, retained, nongaps, backbone_length):
    """aligned: list of (hmm label, [match col or -1] * seq_len) in top-k order;
    weights: {label: float64}; retained/nongaps: {label: sequence}.
    Returns per-residue codes: backbone column (>= 0) for a match, -1 - nc for an insertion
    placed before backbone column nc; plus (min_col, max_col)."""
    combined = {}
    min_col, max_col = backbone_length + 1, -1
    for label, cols in aligned:                      # aligner.py:399-418
        w = np.float64(weights[label])
        for i in range(seq_len):
            c = cols[i]
            if c == -1:
                continue
            j = int(retained[label][c])
            add = nongaps[label][c] * w
            combined[(i, j)] = combined.get((i, j), np.float64(0.0)) + add
            min_col = min(min_col, j)
            max_col = max(max_col, j)
    if max_col < 0:                                  # nothing aligned: every residue is an insertion
        return [-1 - 0] * seq_len, (0, -1)
    W = max_col + 2
    graph = np.zeros((seq_len + 1, W), dtype=np.float64)
    back = np.zeros((seq_len + 1, W), dtype=np.int8)
    for i in range(1, seq_len + 1):                  # aligner.py:426-448
        for j in range(min_col + 1, max_col + 2):
            cw = combined.get((i - 1, j - 1), 0.0)
            values = (graph[i - 1, j - 1] + cw, graph[i - 1, j], graph[i, j - 1])
            cur_max, cur_bt = 0.0, 0
            for ind, val in enumerate(values):
                if ind == 0 and cw <= 0:
                    cur_bt = 1
                    continue
                if val > cur_max:
                    cur_max, cur_bt = val, ind
            graph[i, j] = cur_max
            back[i, j] = cur_bt
    out = [0] * seq_len
    i, j = seq_len, max_col + 1                      # aligner.py:452-473
    while i > 0 and j > min_col:
        bt = back[i, j]
        if bt == 0:
            out[i - 1] = j - 1
            i -= 1
            j -= 1
        elif bt == 1:
            out[i - 1] = -1 - j
            i -= 1
        else:
            j -= 1
    while i > 0:
        out[i - 1] = -1 - j
        i -= 1
    return out, (min_col, max_col)


def trace_to_string(seq, codes, backbone_length):
    """Rebuild the reference's aligned string: uppercase = match, lowercase = insertion,
    '-' = backbone column without a residue; then compressInsertions."""
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
    return compress_insertions(''.join(parts))


def compress_insertions(s):
    """alignment_tools.py:1356-1384: leading/trailing insertions are packed against the ends."""
    alns = [(m.start(), m.end()) for m in re.finditer(r'[A-Z]+', s)]
    if not alns:
        return s
    f_end, b_start = alns[0][0], alns[-1][1]
    front = s[:f_end].replace('-', '')
    back = s[b_start:].replace('-', '')
    return front + '-' * (f_end - len(front)) + s[f_end:b_start] + '-' * (len(s) - b_start - len(back)) + back
