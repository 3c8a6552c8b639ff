"""The N>1 path on CPU: world_size-2 gloo processes shard the queries and all-gather the
top-k records; the result must equal the unsharded tables."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, k, outdir):
    import torch
    import torch.distributed as dist
    from witch_amd.distributed import gather_topk, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, 200, size=(nq, k)).astype(np.int32)
    w = rng.random((nq, k))
    nk = rng.integers(0, k + 1, size=nq).astype(np.int32)
    nu = np.minimum(nk, rng.integers(0, k + 1, size=nq)).astype(np.int32)
    lo, hi = shard_range(nq, rank, world)
    ok = True
    for n_total in (nq, None):        # shard sizes from shard_range (one collective), or exchanged first
        got = gather_topk(torch.from_numpy(idx[lo:hi]), torch.from_numpy(w[lo:hi]), torch.from_numpy(nk[lo:hi]),
                          torch.from_numpy(nu[lo:hi]), n_total=n_total)
        ok = ok and all(np.array_equal(g.numpy(), f) and g.dtype == torch.from_numpy(f).dtype for g, f in zip(got, (idx, w, nk, nu)))
    open(os.path.join(outdir, "rank%d" % rank), "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


@pytest.mark.parametrize("nq,world", [(101, 2), (8, 2), (103, 8), (5, 8)])
def test_sharded_topk_gather(tmp_path, nq, world):
    """world 2 and 8 (gloo): ragged shards, and at 8 ranks x 5 queries shards of zero rows"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, nq, 10, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / ("rank%d" % r)).read() == "ok"


def _mismatch_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    from witch_amd.distributed import gather_topk, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nq, k = 11, 4
    lo, hi = shard_range(nq, rank, world)
    n = hi - lo + (1 if rank == 1 else 0)                 # rank 1 brings one row too many
    args = (torch.zeros((n, k), dtype=torch.int32), torch.zeros((n, k), dtype=torch.float64), torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int32))
    try:
        gather_topk(*args, n_total=nq)
        out = "returned"
    except ValueError:
        out = "raised"
    try:                                                  # wrong dtype: refused before anything is packed (on every rank alike)
        gather_topk(args[0], args[1].float(), args[2], args[3], n_total=None)
        out += " returned"
    except TypeError:
        out += " typeerror"
    open(os.path.join(outdir, "rank%d" % rank), "w").write(out)
    dist.destroy_process_group()


def test_a_rank_with_the_wrong_row_count_fails_on_every_rank(tmp_path):
    """One rank's table is not its shard's: every rank raises (they agree on the failure with one all-reduce) instead of
    one raising and the others waiting in the collective until the backend times out."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_mismatch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / ("rank%d" % r)).read() == "raised typeerror"


def test_shard_ranges_cover_and_are_contiguous():
    from witch_amd.distributed import shard_range
    for n in (0, 1, 7, 100000):
        for world in (1, 2, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1


def _engine_worker(rank, world, port, outdir):
    """Each rank holds the golden tables of ITS block of queries (shard_range), like a rank of a
    multi-GPU run; after the gather, the reference-shaped functions must give the unsharded answers."""
    import pickle
    import torch.distributed as dist
    from tests.conftest import load_case
    from tests.test_gcmm_host import _Sub, _engine_from_golden
    from witch_amd import gcmm
    from witch_amd.distributed import shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = load_case("dna_hmmbuild")
    full = _engine_from_golden(case)                      # the one-rank engine: reference answers
    index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(case.hmm_index, case.hmm_paths, case.nseq)}
    want_ranked = gcmm.rankBitscores(index_to_hmm, {})
    want_weights = gcmm.writeWeights(index_to_hmm, want_ranked)
    lo, hi = shard_range(len(case.qnames), rank, world)
    # this rank's shard: rows [lo, hi) of every table; aligned columns only for its own queries
    pairs = {k: p for k, p in full.pair_of.items() if lo <= k[0] < hi}
    eng = gcmm.QueryAlignmentEngine.from_results(
        case.qnames, case.hmm_index, case.nseq, full.decibits[lo:hi], full.flags[lo:hi], case.k,
        topk=(full.topk_idx[lo:hi], full.topk_w[lo:hi], full.n_kept[lo:hi], full.n_used[lo:hi]),
        aligned=(full.cols, full.col_offsets, pairs), rows=(lo, hi), world=world, rank=rank)
    gcmm.install(eng)
    ok = True
    # before the gather a rank cannot answer for foreign queries - loudly
    foreign = 0 if rank == 1 else len(case.qnames) - 1
    try:
        eng.weights(foreign)
        ok = False
    except KeyError:
        pass
    eng.gather()
    ranked = gcmm.rankBitscores(index_to_hmm, {})         # this rank's block
    ok &= all(lo <= eng.taxon_row[t] < hi for t in ranked) and all(ranked[t] == want_ranked[t] for t in ranked)
    ok &= set(ranked) == {t for t in want_ranked if lo <= eng.taxon_row[t] < hi}
    weights = gcmm.writeWeights(index_to_hmm, want_ranked)   # every query, on every rank
    ok &= weights.keys() == want_weights.keys()
    ok &= all([(i, float(x)) for i, x in weights[t]] == [(i, float(x)) for i, x in want_weights[t]] for t in weights)
    # aligned columns: own queries answer, foreign ones raise
    for (row, label), p in list(full.pair_of.items())[:200]:
        if lo <= row < hi:
            ok &= eng.aligned_columns(row, label) == full.aligned_columns(row, label)
        else:
            try:
                eng.aligned_columns(row, label)
                ok = False
            except KeyError:
                pass
    # the owners' per-query results travel as Python objects, as INTEGRATION.md section 5 shows
    mine = {t: weights[t] for t in ranked}
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    merged = {}
    for g in gathered:
        merged.update(g)
    ok &= merged.keys() == want_weights.keys()
    open(os.path.join(outdir, "erank%d" % rank), "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_sharded_engine_answers_like_the_unsharded_one(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_engine_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / ("erank%d" % r)).read() == "ok"


def _forced_worker(rank, world, port, outdir):
    os.environ["WITCH_FORCE_COLLECTIVES"] = "1"
    _worker(rank, world, port, 37, 10, outdir)


def test_forced_collectives_in_a_group_of_one(tmp_path):
    """WITCH_FORCE_COLLECTIVES=1 (the RCCL smoke of a one-GPU box, tests/test_gpu_parity.py) sends the gather through
    the collectives even at world 1: same tables back."""
    import torch.multiprocessing as mp
    mp.spawn(_forced_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(tmp_path / "rank0").read() == "ok"
