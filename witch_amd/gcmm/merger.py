"""Final transitive merge with the reference's shape (witch_msa/gcmm/merger.py:40-131).

The reference merges the query alignments into the backbone ONE AT A TIME with
ExtendedAlignment.merge_in (helpers/alignment_tools.py:1183-1316): two cursors walk the column
labels and every new run of insertion columns is spliced into EVERY sequence collected so far
(`seq[me:me] = ins`), i.e. O(queries x rows x width) byte moves - the step SURVEY.md section 8f
ranks as the second host bottleneck at 100k queries.

What that loop computes has a closed form, used here:
  * every query alignment carries all B backbone columns (labels 0..B-1) plus insertion
    columns (lowercase, negative labels) in the B+1 gaps before / between / after them;
  * when both sides hold insertion columns at the same gap the runs are walked together from
    the left ("We both have a series of insertion columns"), the longer run's tail is appended;
    so after all merges gap g is W[g] = max over queries of their run length there, and every
    query's run is LEFT-justified in it; everybody else shows '-';
  * rows: the backbone sequences in file order, then the queries in list order (a query whose
    name already exists is not added, alignment_tools.py:1226-1230, but its insertion columns
    still widen the gaps); 'skipped' entries and empty alignments contribute nothing;
  * renamed taxa are popped and re-inserted under their original name, which moves them to
    the end in the order of the rename map (merger.py:84-93);
  * <name>.masked.fasta is the same matrix without the insertion columns (merger.py:100-103).
The result is byte-identical to the reference's two output files (tests/test_merger_host.py
against vectors produced by the reference's own function).
"""
import time

import numpy as np

_DASH = ord('-')


def masked_path(outpath):
    """merger.py:47-56: '<name>.masked.<suffix>' for .fa/.fasta, else '<outpath>.masked.fasta'."""
    suffix = outpath.split('.')[-1]
    if suffix in ('fa', 'fasta'):
        return '.'.join(outpath.split('.')[:-1]) + '.masked.' + suffix
    return outpath + '.masked.fasta'


def read_fasta_upper(path):
    """Alignment.read_file_object (alignment_tools.py:716-733): names verbatim, sequences
    upper-cased; a repeated name replaces the earlier sequence but keeps its position."""
    rows, name, chunks = {}, None, []
    opener = open
    if str(path).endswith('.gz'):
        import gzip
        opener = gzip.open
    with opener(path, 'rt') as f:
        for line in f:
            line = line.strip()
            if line.startswith('>'):
                if name is not None:
                    rows[name] = ''.join(chunks).upper()
                name, chunks = line[1:], []
            elif line and name is not None:
                chunks.append(line)
    if name is not None:
        rows[name] = ''.join(chunks).upper()
    return rows


def _split_query(text, B):
    """(column characters uint8[B], gap index of every lowercase residue, the residues)."""
    a = np.frombuffer(text.encode('ascii'), dtype=np.uint8)
    low = (a >= 97) & (a <= 122)                     # str.islower() of a single ASCII character
    cols = a[~low]
    if cols.size != B:
        raise ValueError("query alignment has %d backbone columns, the backbone has %d" % (cols.size, B))
    gap_of = np.cumsum(~low)[low]                    # columns seen before the residue = its gap 0..B
    return cols, gap_of, a[low]


def merge_collapsed(backbone_rows, queries, renamed_taxa=None):
    """backbone_rows: {name: aligned string} (insertion-ordered); queries: iterable of
    {taxon: aligned string} alignments (the objects alignSubQueriesNew returns), 'skipped'
    markers or empty alignments.  Returns (names, full uint8 matrix, backbone column positions)."""
    names = list(backbone_rows.keys())
    if not names:
        raise ValueError("empty backbone alignment")
    B = len(backbone_rows[names[0]])
    parsed = []                                       # (name, cols, gap_of, residues, add_row)
    seen = set(names)
    W = np.zeros(B + 1, dtype=np.int64)
    for q in queries:
        if isinstance(q, str) or q is None or len(q) == 0:
            continue                                  # 'skipped' / failed query (merger.py:75-77, merge_in:1210)
        for name, text in q.items():
            cols, gap_of, res = _split_query(text, B)
            if gap_of.size:
                run = np.bincount(gap_of, minlength=B + 1)
                np.maximum(W, run, out=W)
            add = name not in seen
            seen.add(name)
            parsed.append((name, cols, gap_of, res, add))
    # column layout: gap g (W[g] insertion columns) precedes backbone column g; gap B closes the row
    gap_start = np.concatenate(([0], np.cumsum(W)[:-1])) + np.arange(B + 1)
    col_pos = gap_start[:B] + W[:B]
    width = int(B + W.sum())
    rows_q = [p for p in parsed if p[4]]
    out = np.full((len(names) + len(rows_q), width), _DASH, dtype=np.uint8)
    for r, n in enumerate(names):
        s = np.frombuffer(backbone_rows[n].encode('ascii'), dtype=np.uint8)
        if s.size != B:
            raise ValueError("backbone row %s has %d columns, expected %d" % (n, s.size, B))
        out[r, col_pos] = s
    for r, (name, cols, gap_of, res, _) in enumerate(rows_q, start=len(names)):
        out[r, col_pos] = cols
        if gap_of.size:
            # left-justified inside the gap: k-th residue of a run sits at gap_start + k
            first = np.concatenate(([True], gap_of[1:] != gap_of[:-1]))
            run_begin = np.maximum.accumulate(np.where(first, np.arange(gap_of.size), 0))
            out[r, gap_start[gap_of] + (np.arange(gap_of.size) - run_begin)] = res
        names.append(name)
    # merger.py:84-93: rename back; popped entries move to the end in rename-map order
    order = list(range(len(names)))
    if renamed_taxa:
        name_map = {v: k for k, v in renamed_taxa.items()}
        pos = {n: i for i, n in enumerate(names)}
        for name, ori in name_map.items():
            if name in pos:
                i = pos.pop(name)
                order.remove(i)
                if ori in pos:                        # overwriting an existing key keeps that key's slot
                    order[order.index(pos[ori])] = i
                else:
                    order.append(i)
                pos[ori] = i
                names[i] = ori
    return [names[i] for i in order], out[order], col_pos


def write_fasta_matrix(path, names, mat):
    with open(path, 'wb') as f:
        for n, row in zip(names, mat):
            f.write(b'>' + n.encode() + b'\n' + row.tobytes() + b'\n')


def mergeAlignmentsCollapsed(backbone_alignment_path, queries, renamed_taxa, pool, output_path=None,
                             log=None):
    """Same arguments as merger.py:40 (pool is unused there too) plus the output path the
    reference takes from Configs.output_path.  Writes <output_path> and its masked twin; returns
    (output_path, masked_output_path)."""
    if output_path is None:
        raise ValueError("output_path is required (the reference reads Configs.output_path)")
    start = time.time()
    if not len(queries) > 0:
        raise SystemExit('No query alignment provided to merger!')   # merger.py:60-62 prints and exits
    backbone = read_fasta_upper(backbone_alignment_path)
    names, mat, col_pos = merge_collapsed(backbone, queries, renamed_taxa)
    write_fasta_matrix(output_path, names, mat)
    mpath = masked_path(output_path)
    write_fasta_matrix(mpath, names, mat[:, col_pos])
    if log is not None:
        log('Time to merge all outputs (s): {}'.format(time.time() - start))
    return output_path, mpath


def world_of(eng):
    return int(getattr(eng, "world", 1))


def mergeAlignmentsDevice(backbone_alignment_path, renamed_taxa=None, output_path=None, engine=None, taxa=None, log=None,
                          group=None):
    """The final merge straight from the consensus kernel's codes (wh_merge, witch_amd/csrc/wh_merge.hip): what
    `[alignSubQueriesNew(...) for every query]` + `mergeAlignmentsCollapsed(...)` write, without building the
    per-query strings on the host.  `taxa`: the queries to merge, in the order the reference would append them
    (default: every query that has weights, in batch order); the others are left out like the reference's 'ignored'
    ones.  Returns the two paths (merger.py:95-103).

    One process per GPU (engine.world > 1, torch.distributed initialised): every rank calls this with the same
    arguments.  The width of a gap is the maximum over ALL queries, so the ranks all-reduce (MAX) their local gap
    widths - the merge's one exchange step - render their own rows in the common layout, and rank 0 gathers the
    rows and writes the two files (the other ranks return the paths without writing)."""
    import ctypes as C
    from .._lib import lib, check
    from .engine import current_engine
    eng = engine or current_engine()
    if eng.merged is None:
        raise RuntimeError("the engine ran without the consensus step (subset_to_retained_columns not given)")
    if output_path is None:
        raise ValueError("output_path is required")
    s1 = time.time()
    backbone = read_fasta_upper(backbone_alignment_path)
    names = list(backbone.keys())
    B = len(backbone[names[0]])
    bb = np.frombuffer("".join(backbone[n] for n in names).encode("ascii"), dtype=np.uint8)
    if bb.size != len(names) * B:
        raise ValueError("backbone rows differ in length")
    lo, hi = eng.row_lo, eng.row_hi
    nloc = hi - lo
    if taxa is None:
        # a query has weights iff at least one HMM reported it: n_kept of the (gathered) top-k table
        t_lo, t_hi = eng.topk_rows
        if world_of(eng) > 1 and (t_lo, t_hi) != (0, len(eng.taxa)):
            raise RuntimeError("call engine.gather() first: the merge needs every rank's top-k records")
        taxa = [eng.taxa[r] for r in range(t_lo, t_hi) if eng.n_kept[r - t_lo] > 0]
    # rows in append order, decided over ALL given taxa (every rank computes the same list); a name that exists
    # already widens the gaps but gets no row (alignment_tools.py:1226-1230)
    q_row = np.full(nloc, -2, dtype=np.int32)
    seen = set(names)
    appended = []                                     # (global query row, name) in append order
    for t in taxa:
        g = eng.taxon_row[t]
        new = t not in seen
        seen.add(t)
        if new:
            appended.append((g, t))
        if lo <= g < hi:
            q_row[g - lo] = 0 if new else -1
    codes = np.ascontiguousarray(eng.merged, dtype=np.int32)
    offs = np.ascontiguousarray(eng.query_offsets, dtype=np.int64)
    text = np.ascontiguousarray(eng.query_text, dtype=np.uint8)
    from ..distributed import collectives_forced
    world = world_of(eng)
    exchange = world > 1 or collectives_forced()      # forced: the RCCL smoke of a one-GPU box takes the N-rank path
    W = None
    if exchange:
        import torch
        import torch.distributed as dist
        Wl = np.zeros(B + 1, dtype=np.int32)
        check(lib().wh_merge_sharded(int(eng.device), None, offs.ctypes.data, nloc, codes.ctypes.data, q_row.ctypes.data,
                                     None, 0, B, Wl.ctypes.data, None, None, None, None, None), "wh_merge_sharded (widths)")
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        wt = torch.from_numpy(Wl).to(dev)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX, group=group)
        W = np.ascontiguousarray(wt.cpu().numpy(), dtype=np.int32)
    write_bb = world == 1 or eng.rank == 0
    nb = len(names) if write_bb else 0
    pf, pm, nr, wd = C.c_void_p(), C.c_void_p(), C.c_int64(0), C.c_int64(0)
    check(lib().wh_merge_sharded(int(eng.device), text.ctypes.data, offs.ctypes.data, nloc, codes.ctypes.data, q_row.ctypes.data,
                                 bb.ctypes.data if nb else None, nb, B, None, W.ctypes.data if W is not None else None,
                                 C.byref(pf), C.byref(pm), C.byref(nr), C.byref(wd)), "wh_merge")
    try:
        nrows, width = int(nr.value), int(wd.value)
        full = np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_uint8)), shape=(max(nrows * width, 1),))[:nrows * width].reshape(nrows, width).copy()
        masked = np.ctypeslib.as_array(C.cast(pm, C.POINTER(C.c_uint8)), shape=(max(nrows * B, 1),))[:nrows * B].reshape(nrows, B).copy()
    finally:
        lib().wh_free_text(pf)
        lib().wh_free_text(pm)
    # this rank's query rows come out in QUERY order
    mine = sorted(g for g, _ in appended if lo <= g < hi)
    local_row = {g: nb + i for i, g in enumerate(mine)}
    mpath = masked_path(output_path)
    if exchange:
        # Only rank 0 writes, so only rank 0 receives: ONE gather of a uint8 tensor per rank - its rendered rows, full and
        # masked side by side (the row length is global after the MAX all-reduce above), padded to the largest shard.  How
        # many rows every rank brings follows from the append list and the shard ranges, which every rank computes alike:
        # nothing is exchanged about sizes and nothing is pickled (until round 4: all_gather_object of a dictionary of
        # byte strings, every rank receiving every row).
        import torch
        import torch.distributed as dist
        from ..distributed import shard_range
        bounds = [shard_range(len(eng.taxa), r, world) for r in range(world)] if world > 1 else [(lo, hi)]
        per_rank = [sorted(g for g, _ in appended if a_ <= g < b_) for a_, b_ in bounds]
        nmax = max(1, max(len(v) for v in per_rank))
        rowlen = width + B
        send = np.zeros((nmax, rowlen), dtype=np.uint8)
        if mine:
            sel = np.array([local_row[g] for g in mine], dtype=np.int64)
            send[:len(mine), :width] = full[sel]
            send[:len(mine), width:] = masked[sel]
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        send_t = torch.from_numpy(send).to(dev)
        recv = [torch.empty_like(send_t) for _ in range(world)] if eng.rank == 0 else None
        dist.gather(send_t, recv, dst=0, group=group)
        if eng.rank != 0:
            return output_path, mpath
        where = {}
        for r, gs in enumerate(per_rank):
            block = recv[r].cpu().numpy()
            for i, g in enumerate(gs):
                where[g] = block[i]
        full_rows = [full[i].tobytes() for i in range(nb)] + [where[g][:width].tobytes() for g, _ in appended]
        masked_rows = [masked[i].tobytes() for i in range(nb)] + [where[g][width:].tobytes() for g, _ in appended]
    else:
        full_rows = [full[i].tobytes() for i in range(nb)] + [full[local_row[g]].tobytes() for g, _ in appended]
        masked_rows = [masked[i].tobytes() for i in range(nb)] + [masked[local_row[g]].tobytes() for g, _ in appended]
    all_names = names + [t for _, t in appended]
    order = list(range(len(all_names)))
    if renamed_taxa:                                   # merger.py:84-93
        name_map = {v: k for k, v in renamed_taxa.items()}
        pos = {n: i for i, n in enumerate(all_names)}
        for name, ori in name_map.items():
            if name in pos:
                i = pos.pop(name)
                order.remove(i)
                if ori in pos:
                    order[order.index(pos[ori])] = i
                else:
                    order.append(i)
                pos[ori] = i
                all_names[i] = ori
    for path, rows_ in ((output_path, full_rows), (mpath, masked_rows)):
        with open(path, "wb") as f:
            for i in order:
                f.write(b">" + all_names[i].encode() + b"\n" + rows_[i] + b"\n")
    if log:
        log("Finished merging all GCM subproblems on the device, runtime (s): %s" % (time.time() - s1))
    return output_path, mpath
