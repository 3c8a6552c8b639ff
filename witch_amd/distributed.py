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


def gather_topk(idx, w, nk, nu, group=None):
    """all-gather the ragged (per-rank row count) top-k tables; every rank gets the full
    tables in query order.  Uses padded fixed-size all_gathers (ring collectives on xGMI are
    per-link bound; four ~MB-sized messages are far below the latency/bandwidth knee)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not collectives_forced():
        return idx, w, nk, nu
    n_local = torch.tensor([idx.shape[0]], device=idx.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)

    def pad(t):
        if t.shape[0] == nmax:
            return t.contiguous()
        p = torch.zeros((nmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        p[:t.shape[0]] = t
        return p

    outs = []
    for t in (idx, w, nk, nu):
        pt = pad(t)
        buf = [torch.empty_like(pt) for _ in range(world)]
        dist.all_gather(buf, pt, group=group)
        outs.append(torch.cat([b[:s] for b, s in zip(buf, sizes)], 0))
    return tuple(outs)
