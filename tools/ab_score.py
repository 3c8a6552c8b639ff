#!/usr/bin/env python3
"""A/B harness for the scoring kernel: same process, same GPU, same inputs.

Each variant is a set of WH_* knobs (set on the live handle with wh_set_option);
prints the kernel time of every variant and how many (query, HMM) deci-bit scores / flags
differ from the first variant.   usage: tools/ab_score.py NQ "K=V,K=V" "K=V" ...
"""
import os
import sys
import tempfile
import shutil
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    nq = int(sys.argv[1])
    wl = "dna_100k_x200"
    variants = sys.argv[2:] or [""]
    if variants and variants[0].startswith("workload="):
        wl = variants[0].split("=")[1]
        variants = variants[1:] or [""]
    import torch
    from witch_amd.ehmm import EHMM, pack_queries
    wd = tempfile.mkdtemp(prefix="witch_ab_")
    try:
        fam, se, names, seqs, k = bench.make_workload(wl, wd, nq, int(os.environ.get("AB_NH", "0")) or None)
        e = EHMM(se.paths, hmm_index=se.index, nseq=se.nseq, device=0)
        res, offs = pack_queries([s.astype(np.uint8) for s in seqs])
        maxlen = int(np.max(np.diff(offs)))
        res_t = torch.from_numpy(res).cuda()
        off_t = torch.from_numpy(offs).cuda()
        cells = float(np.diff(offs).sum()) * float(e.M.astype(np.int64).sum())
        knobs = set()
        for v in variants:
            for kv in filter(None, v.split(",")):
                knobs.add(kv.split("=")[0])
        ref = None
        e.set_timing(True)
        for rep in range(2):
            for v in variants:
                for kn in knobs:
                    e.set_option(kn, "")
                for kv in filter(None, v.split(",")):
                    a, b = kv.split("=")
                    e.set_option(a, b)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                deci, flags = e.score_t(res_t, off_t, maxlen)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) * 1e3
                ms, n = e.last_kernel_ms(0)
                rms, _ = e.last_kernel_ms(4)
                paths = e.last_score_paths()
                d = deci.cpu().numpy()
                f = flags.cpu().numpy()
                if ref is None:
                    ref = (d, f)
                nd = int((d != ref[0]).sum())
                nf = int(((f & 7) != (ref[1] & 7)).sum())
                mx = int(np.abs(d.astype(np.int64) - ref[0]).max())
                print("rep %d  %-44s kernel %9.3f ms (wall %9.3f)  %.3g cells/s  resolver %9.3f ms for %d multidomain pairs  decibit diffs vs first: %d (max %d)  flag diffs: %d  dense redos: %d  paths %s"
                      % (rep, v or "(default)", ms, dt, cells / (ms * 1e-3), rms, int(((f & 2) != 0).sum()), nd, mx, nf, int(((f & 16) != 0).sum()), paths), flush=True)
    finally:
        shutil.rmtree(wd, ignore_errors=True)


if __name__ == "__main__":
    main()
