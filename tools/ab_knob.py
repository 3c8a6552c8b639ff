#!/usr/bin/env python3
"""One knob against the default build, pair by pair: kernel time, live spill bytes, score differences.
usage: tools/ab_knob.py NQ KNOB=VALUE[,KNOB=VALUE...] [workload] [NH]"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    nq = int(sys.argv[1])
    knobs = [kv.split("=") for kv in sys.argv[2].split(",")]
    wl = sys.argv[3] if len(sys.argv) > 3 else "dna_100k_x200"
    nh = int(sys.argv[4]) if len(sys.argv) > 4 else None
    import torch
    from witch_amd.ehmm import EHMM, pack_queries
    wd = tempfile.mkdtemp(prefix="witch_abk_")
    try:
        fam, se, names, seqs, k = bench.make_workload(wl, wd, nq, nh)
        res, offs = pack_queries([s.astype(np.uint8) for s in seqs])
        out = {}
        for name, kv in (("default", []), ("knob", knobs)):
            e = EHMM(se.paths, hmm_index=se.index, nseq=se.nseq, device=0)
            e.set_timing(True)
            for key, val in kv:
                e.set_option(key, val)
            for rep in range(3):
                torch.cuda.synchronize()
                deci, flags = e.score(res, offs)[:2]
                ms, _ = e.last_kernel_ms(0)
                print("%-8s rep %d kernels %9.3f ms  spill %.4g B  paths %s" % (name, rep, ms, e.last_score_spill_bytes(), e.last_score_paths()), flush=True)
            out[name] = (np.asarray(deci).copy(), np.asarray(flags).copy())
            M = e.M.copy()
            e.close()
        a, b = out["default"], out["knob"]
        d = a[0].astype(np.int64) - b[0].astype(np.int64)
        print("dense redos (WH_FLAG_EXACT): default %d  knob %d" % (int((a[1] & 16 != 0).sum()), int((b[1] & 16 != 0).sum())))
        ra, rb = (a[1] & 16 != 0).sum(axis=0), (b[1] & 16 != 0).sum(axis=0)
        cells = -(-M // 64)
        for q in sorted(set(cells.tolist())):
            sel = cells == q
            print("  models of %2d cells per lane: %3d HMMs  redos default %6d knob %6d of %d pairs" % (q, int(sel.sum()), int(ra[sel].sum()), int(rb[sel].sum()), int(sel.sum()) * a[1].shape[0]))
        print("pairs %d  decibit diffs %d (max |d| %d)  flag diffs %d" % (a[0].size, int((d != 0).sum()), int(np.abs(d).max()), int((a[1] != b[1]).sum())))
    finally:
        shutil.rmtree(wd, ignore_errors=True)


if __name__ == "__main__":
    main()
