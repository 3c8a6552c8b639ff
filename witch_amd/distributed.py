"""One process per GPU; queries are sharded contiguously, the eHMM is replicated, and the
only exchange step of the path is the gather of the per-query top-k records
(int32 idx[k], float64 w[k], n_kept, n_used = 124 B/query at k=10) over RCCL/xGMI
(backend "nccl" on ROCm) - SURVEY.md section 8e.  Works with the gloo backend on CPU
tensors too (that is what the CPU tests use)."""
from __future__ import annotations


def collectives_forced() -> bool:
    """WITCH_FORCE_COLLECTIVES=1: run the path's exchange steps (top-k all-gather, the merge's MAX all-reduce and
    row gather) through torch.distributed even in a group of ONE rank - the RCCL smoke test of a one-GPU box
    (tests/test_gpu_parity.py::test_rccl_collectives_on_one_gpu): same code path, same results as without."""
    import os
    return os.environ.get("WITCH_FORCE_COLLECTIVES") == "1"


def shard_range(n: int, rank: int, world: int):
    """Contiguous block of queries owned by <rank>."""
    return n * rank // world, n * (rank + 1) // world


def gather_topk(idx, w, nk, nu, group=None, n_total=None):
    """all-gather the per-rank top-k tables; every rank gets the full tables in query order.

    ONE collective: a rank packs its rows into fixed-size byte records (w float64[k] | idx int32[k] | n_kept | n_used =
    12k + 8 bytes, 128 B at k = 10), pads to the largest shard and calls all_gather_into_tensor once; the shard sizes
    follow from shard_range(n_total, r, world), so nothing has to be exchanged about them and nothing is read back to
    the host in between.  Without n_total (a caller that shards differently) the sizes are gathered first."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not collectives_forced():
        return idx, w, nk, nu
    n_local, k = int(idx.shape[0]), int(idx.shape[1])
    # the record layout below is byte arithmetic: refuse anything but the types it was written for
    if w.dtype != torch.float64 or idx.dtype != torch.int32 or nk.dtype != torch.int32 or nu.dtype != torch.int32:
        raise TypeError("gather_topk: expects w float64, idx / n_kept / n_used int32 (got %s, %s, %s, %s)" % (w.dtype, idx.dtype, nk.dtype, nu.dtype))
    if n_total is not None:
        sizes = [shard_range(int(n_total), r, world)[1] - shard_range(int(n_total), r, world)[0] for r in range(world)]
        # A rank whose row count is not its shard's must not raise ALONE: the others would wait in the collective until
        # the backend times out.  The ranks agree on the failure first (one MAX all-reduce of a flag, 8 bytes) and then
        # all raise - or, if nobody disagrees, nothing else was exchanged about the sizes.
        bad = torch.tensor([1 if sizes[dist.get_rank(group)] != n_local else 0], device=idx.device, dtype=torch.int64)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        if int(bad.item()):
            raise ValueError("gather_topk: a rank's row count is not its shard's (this rank: %d local rows, shard_range(%d, rank, %d) owns %d)"
                             % (n_local, n_total, world, sizes[dist.get_rank(group)]))
    else:
        mine = torch.tensor([n_local], device=idx.device, dtype=torch.int64)
        got = torch.empty(world, device=idx.device, dtype=torch.int64)
        dist.all_gather_into_tensor(got, mine, group=group)
        sizes = [int(x) for x in got.tolist()]
    nmax = max(sizes)
    R = 12 * k + 8
    rec = torch.zeros((nmax, R), dtype=torch.uint8, device=idx.device)
    if n_local:
        rec[:n_local, :8 * k] = w.contiguous().view(torch.uint8).reshape(n_local, 8 * k)
        rec[:n_local, 8 * k:12 * k] = idx.contiguous().view(torch.uint8).reshape(n_local, 4 * k)
        rec[:n_local, 12 * k:12 * k + 4] = nk.contiguous().view(torch.uint8).reshape(n_local, 4)
        rec[:n_local, 12 * k + 4:] = nu.contiguous().view(torch.uint8).reshape(n_local, 4)
    flat = torch.empty((world * nmax, R), dtype=torch.uint8, device=idx.device)    # rank r's records at rows [r * nmax, (r + 1) * nmax)
    dist.all_gather_into_tensor(flat, rec, group=group)
    out = flat.reshape(world, nmax, R)
    allrec = torch.cat([out[r, :sizes[r]] for r in range(world)], 0) if len(set(sizes)) > 1 else flat
    n = allrec.shape[0]
    g_w = allrec[:, :8 * k].contiguous().view(w.dtype).reshape(n, k)
    g_idx = allrec[:, 8 * k:12 * k].contiguous().view(idx.dtype).reshape(n, k)
    g_nk = allrec[:, 12 * k:12 * k + 4].contiguous().view(nk.dtype).reshape(n)
    g_nu = allrec[:, 12 * k + 4:].contiguous().view(nu.dtype).reshape(n)
    return g_idx, g_w, g_nk, g_nu
