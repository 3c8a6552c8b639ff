#!/usr/bin/env python3
"""bench.py - WITCH query-vs-eHMM hot path on MI355X (metric of BASELINE.json).

One "step" = one pass of the whole hot path over one batch of synthetic queries that are
already resident in HBM: score every query against every HMM of the ensemble
(wh_score_dev), weights + deterministic top-k + 0.999 prefix (wh_topk_dev), MEA alignment
against the kept HMMs (wh_align_dev), result gather to the host.  With N > 1 ranks the
queries are sharded contiguously (the eHMM is replicated), and the only collective is
the all-gather of the per-query top-k records over RCCL.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (alphabet, family seed, root_len, leaves, sub_rate, indel_rate, n_hmms, n_queries, qlen, k)
    # configs[2] of BASELINE.json (headline): 100k x 150 nt DNA queries x 200-HMM eHMM, k=10
    "dna_100k_x200": ("dna", 20251205, 900, 1024, 0.03, 1e-4, 200, 100000, 150, 10),
    # configs[1]: 1k queries x 10 HMMs, k=4 (parity-test sized)
    "dna_1k_x10": ("dna", 20251205, 1000, 256, 0.03, 2e-3, 10, 1000, 150, 4),
    # development only: longer models (24 and 12 DP cells per lane)
    "dna_m1450": ("dna", 20251205, 1450, 256, 0.03, 1e-4, 20, 2048, 150, 4),
    "dna_m700": ("dna", 20251205, 700, 256, 0.03, 1e-4, 20, 2048, 150, 4),
    "dna_m1250": ("dna", 20251205, 1250, 256, 0.03, 1e-4, 40, 4096, 150, 4),
    "dna_m1900_long": ("dna", 20251205, 1900, 256, 0.03, 1e-4, 10, 256, (900, 1800), 4),
    "dna_m1250_long": ("dna", 20251205, 1250, 256, 0.03, 1e-4, 20, 512, (600, 1200), 4),
    "dna_m1450_long": ("dna", 20251205, 1450, 256, 0.03, 1e-4, 20, 512, (600, 1400), 4),
    # shaped like the reference's examples/data (rRNA backbone of 2574 columns, full-length queries)
    "dna_rrna_like": ("dna", 20251207, 2400, 256, 0.03, 2e-4, 10, 256, (1500, 2400), 4),
    # SURVEY.md section 8d config 5 (reported in DESIGN.md, not the headline): protein family,
    # 500-HMM eHMM, 50k queries of 50..2000 residues built from family windows and random flanks
    "aa_50k_x500": ("amino", 20251206, 600, 2048, 0.03, 1e-4, 500, 50000, (50, 2000), 10),
}


def make_workload(name, workdir, nq_override=None, nh_override=None):
    from witch_amd import synth
    alph, seed, root_len, leaves, sub, indel, n_hmms, nq, qlen, k = WORKLOADS[name]
    if nq_override:
        nq = nq_override
    if nh_override:
        n_hmms = nh_override
    fam = synth.make_family(seed, root_len, leaves, alph, sub, indel)
    ehmm = synth.make_ehmm(fam, n_hmms, workdir)
    if isinstance(qlen, tuple):
        names, seqs = synth.make_queries(fam, seed + 1, nq, qlen, flank_frac=0.3)
    else:
        names, seqs = synth.make_queries(fam, seed + 1, nq, qlen)
    return fam, ehmm, names, seqs, k


def hot_path_step(e, res_t, off_t, maxlen, k, gather_topk=None):
    """One pass of the hot path; returns host-side results (top-k table, aligned columns)."""
    import torch
    deci, flags = e.score_t(res_t, off_t, maxlen)
    idx, w, nk, nu = e.topk_t(deci, flags, k)
    # pairs (query, kept model) for the 0.999 prefix (aligner.py:58-63)
    ar = torch.arange(k, device=idx.device, dtype=torch.int32)[None, :]
    keep = ar < nu[:, None]
    pq = torch.nonzero(keep, as_tuple=False)[:, 0].contiguous()
    lab = idx[keep]
    ph = e.label_to_pos_t(lab)
    lens = (off_t[1:] - off_t[:-1])[pq]
    co = torch.zeros(pq.numel() + 1, dtype=torch.int64, device=idx.device)
    torch.cumsum(lens, 0, out=co[1:])
    total_cols = int(co[-1].item())
    cols = e.align_t(res_t, off_t, maxlen, pq, ph, co, total_cols)
    if gather_topk is not None:
        idx, w, nk, nu = gather_topk(idx, w, nk, nu)
    out = (idx.cpu(), w.cpu(), nk.cpu(), nu.cpu(), cols.cpu(), co.cpu())
    hot_path_step.dense_redo = int(((flags & 16) != 0).sum().item())
    hot_path_step.multidomain = int(((flags & 2) != 0).sum().item())
    hot_path_step.reported = int(((flags & 1) != 0).sum().item())
    return out, int(pq.numel()), total_cols


def cpu_baseline(ehmm_paths, nseq, seqs, k, n_sample, threads):
    """The CPU oracle (a float64 port of the HMMER-driven path) timed on a bounded sample."""
    from oracle import oracle as orc
    hm = [orc.OracleHMM(p) for p in ehmm_paths]
    sample = seqs[:n_sample]
    res, offs = orc.pack([np.asarray(s, dtype=np.uint8) for s in sample])
    t0 = time.time()
    deci, flags, _, _ = orc.score_batch(hm, res, offs, nthreads=threads)
    pq, ph = [], []
    for q in range(len(sample)):
        ranked = orc.rank_bitscores(list(range(len(hm))), deci[q], flags[q] & 1)
        if not ranked:
            continue
        idxs = [r[0] for r in ranked]
        w = orc.calculate_weights(idxs, [r[1] for r in ranked], [nseq[i] for i in idxs], k)
        for i, _ in w[:orc.adaptive_cut(w)]:
            pq.append(q)
            ph.append(i)
    orc.align_batch(hm, res, offs, pq, ph, nthreads=threads)
    dt = time.time() - t0
    return len(sample) / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dna_100k_x200", choices=sorted(WORKLOADS))
    ap.add_argument("--nq", type=int, default=0, help="override the query count (development only)")
    ap.add_argument("--nh", type=int, default=0, help="override the HMM count (development only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # rehearsal hook for a one-GPU box: WITCH_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses
    # gloo for the gather (RCCL refuses two ranks on one device); the driver's runs never set it
    rehearsal = os.environ.get("WITCH_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from witch_amd.ehmm import EHMM, pack_queries
    workdir = tempfile.mkdtemp(prefix="witch_bench_%d_" % rank)
    try:
        fam, synth_ehmm, names, seqs, k = make_workload(args.workload, workdir, args.nq or None, args.nh or None)
        e = EHMM(synth_ehmm.paths, hmm_index=synth_ehmm.index, nseq=synth_ehmm.nseq, device=local_rank)
        nq_total = len(seqs)
        # contiguous query shards (weak scaling is NOT used: total work is fixed by the config)
        from witch_amd.distributed import shard_range
        lo, hi = shard_range(nq_total, rank, world)
        res, offs = pack_queries([s.astype(np.uint8) for s in seqs[lo:hi]])
        maxlen = int(np.max(np.diff(offs))) if hi > lo else 1
        res_t = torch.from_numpy(res).cuda()
        off_t = torch.from_numpy(offs).cuda()

        gather = None
        if world > 1:
            from witch_amd.distributed import gather_topk
            # the path's one exchange step: per-query top-k records to every rank over RCCL
            gather = gather_topk
            if rehearsal:
                def gather(idx, w, nk, nu):       # gloo gathers host tensors
                    return gather_topk(idx.cpu(), w.cpu(), nk.cpu(), nu.cpu())

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            hot_path_step(e, res_t, off_t, maxlen, k, gather)
        e.set_timing(True)
        barrier()
        t0 = time.perf_counter()
        kern_ms = [0.0, 0.0, 0.0]
        kern_n = [0, 0, 0]
        npairs = ncols = 0
        for _ in range(args.steps):
            out, npairs, ncols = hot_path_step(e, res_t, off_t, maxlen, k, gather)
            for which in range(3):
                ms, n = e.last_kernel_ms(which)
                kern_ms[which] += ms
                kern_n[which] += n
        barrier()
        dt = time.perf_counter() - t0
        e.set_timing(False)
        if world > 1:
            tmax = torch.tensor([dt], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())

        if rank == 0:
            H = e.H
            M = e.M.astype(np.float64)
            L = float(np.mean(np.diff(offs)))
            qps = nq_total * args.steps / dt
            # dominant kernel: the fused scoring kernel.  Algorithmic HBM bytes per (query,HMM)
            # pair: the Forward rows of the envelope are written once and
            # read once, 2 states x 4 B each way = 16 B per (residue x model node) cell,
            # plus the query residues and the 9 output bytes.
            n_local = hi - lo
            score_launches = max(kern_n[0], 1)
            score_ms = kern_ms[0] / score_launches
            bytes_per_launch = n_local * float(np.sum(16.0 * L * M + L + 9.0)) * args.steps / score_launches
            achieved = bytes_per_launch / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("workload") == args.workload and not args.nq and not args.nh:
                        traffic = tj.get("score_kernel_hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # VALU view of the same kernel (it is arithmetic-bound by design):
            # 5 DP sweeps (2 multihit parsers, 2 envelope sweeps, decoding) ~ 77 flop per cell
            flops = n_local * float(np.sum(L * M)) * 77.0 * args.steps / score_launches
            roofline = {"bound": "hbm", "kernel": "wh::k7::score_kernel7", "achieved": round(achieved, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                        "kernel_ms_avg": round(score_ms, 3), "launches": score_launches,
                        "valu_tflops": round(flops / (score_ms * 1e-3) / 1e12, 2) if score_ms > 0 else 0.0,
                        "valu_peak_tflops": 157.3}
            line = {
                "metric": "query-seqs aligned/sec (100k queries x 200-HMM eHMM)",
                "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": args.workload, "n_queries": nq_total, "n_hmms": H,
                           "query_len": int(round(L)), "model_len_min": int(M.min()), "model_len_max": int(M.max()),
                           "k": k, "aligned_pairs_per_step": npairs, "sharding": "queries/%d" % world,
                           "pairs_reported_rank0": hot_path_step.reported, "pairs_multidomain_rank0": hot_path_step.multidomain,
                           "pairs_dense_redo_rank0": hot_path_step.dense_redo},
                "stage_ms_per_step": {"score": round(kern_ms[0] / args.steps, 3), "topk": round(kern_ms[1] / args.steps, 3),
                                      "align": round(kern_ms[2] / args.steps, 3)},
                "roofline": roofline,
            }
            if not args.no_cpu_baseline:
                threads = min(os.cpu_count() or 1, 64)
                n_sample = 384 if H >= 100 else 2048      # about 15 s of CPU work on 64 host threads
                v, cdt = cpu_baseline(synth_ehmm.paths, synth_ehmm.nseq, seqs, k, min(n_sample, nq_total), threads)
                line["cpu_baseline"] = {"value": round(v, 3), "unit": "queries/s", "cores": threads, "kind": "port",
                                        "sample": "first %d queries x %d HMMs through the float64 oracle "
                                                  "(score + top-k + align), %.1f s" % (min(n_sample, nq_total), H, cdt)}
            print(json.dumps(line), flush=True)
        e.close()
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
