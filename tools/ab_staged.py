#!/usr/bin/env python3
"""Staged launches (WH_SCORE_KERNEL=10) against the fused kernel, pair by pair, with the per-pair detail records:
the first field that differs tells which stage is wrong.   usage: tools/ab_staged.py NQ [workload] [NH]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    nq = int(sys.argv[1])
    wl = sys.argv[2] if len(sys.argv) > 2 else "dna_100k_x200"
    nh = int(sys.argv[3]) if len(sys.argv) > 3 else None
    import torch
    from witch_amd.ehmm import EHMM, pack_queries
    wd = tempfile.mkdtemp(prefix="witch_abst_")
    try:
        fam, se, names, seqs, k = bench.make_workload(wl, wd, nq, nh)
        e = EHMM(se.paths, hmm_index=se.index, nseq=se.nseq, device=0)
        res, offs = pack_queries([s.astype(np.uint8) for s in seqs])
        out = {}
        e.set_timing(True)
        if os.environ.get("AB_STATS"):
            e.set_option("WH_STATS", "1")
        for rep in range(2):
            for name, kern in [("fused", "7")] + [("staged%s" % k, k) for k in os.environ.get("AB_KERNELS", "10").split(",")]:
                e.set_option("WH_SCORE_KERNEL", kern)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                deci, flags, fwd, det = e.score(res, offs, want_fwd=True, want_detail=True)
                dt = time.perf_counter() - t0
                ms, _ = e.last_kernel_ms(0)
                print("rep %d %-6s kernels %9.3f ms  (call %.3f s)  paths %s  reruns %d" % (rep, name, ms, dt, e.last_score_paths(), e.last_queue_reruns()), flush=True)
                out[name] = (deci, flags, fwd, det)
        for key in [k for k in out if k != "fused"]:
            compare(out["fused"], out[key], key)
        e.close()
    finally:
        shutil.rmtree(wd, ignore_errors=True)


def compare(a, b, key):
    if True:
        print("== fused vs", key)
        print("pairs %d  decibit diffs %d  flag diffs %d  fwd_bits diffs %d" % (a[0].size, int((a[0] != b[0]).sum()), int((a[1] != b[1]).sum()),
              int((a[2].view(np.int32) != b[2].view(np.int32)).sum())))
        da, db = np.ctypeslib.as_array(a[3]), np.ctypeslib.as_array(b[3])
        for name in da.dtype.names:
            x, y = da[name], db[name]
            if x.dtype.kind == "f":
                neq = (x.view(np.int32) != y.view(np.int32))
            else:
                neq = x != y
            n = int(neq.sum())
            print("  detail.%-12s %d entries differ" % (name, n))
            if n:
                idx = np.argwhere(neq)[:5]
                for i in idx:
                    print("     at", tuple(int(v) for v in i), x[tuple(i)], y[tuple(i)])


if __name__ == "__main__":
    main()
