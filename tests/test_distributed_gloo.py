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
    got = gather_topk(torch.from_numpy(idx[lo:hi]), torch.from_numpy(w[lo:hi]), torch.from_numpy(nk[lo:hi]),
                      torch.from_numpy(nu[lo:hi]))
    ok = all(np.array_equal(g.numpy(), f) for g, f in zip(got, (idx, w, nk, nu)))
    open(os.path.join(outdir, "rank%d" % rank), "w").write("ok" if ok else "bad")
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [101, 8])
def test_sharded_topk_gather_world2(tmp_path, nq):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, nq, 10, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / ("rank%d" % r)).read() == "ok"


def test_shard_ranges_cover_and_are_contiguous():
    from witch_amd.distributed import shard_range
    for n in (0, 1, 7, 100000):
        for world in (1, 2, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1
