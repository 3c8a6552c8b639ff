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
    # the same configuration with SURVEY.md 8(d)'s sketch of the family (root 1000 nt, 0.2 % indels per branch:
    # models of ~1000-1100 nodes, 20 cells per lane for most of them); quoted beside the headline in DESIGN.md
    "dna_100k_x200_m1000": ("dna", 20251205, 1000, 1024, 0.03, 2e-3, 200, 100000, 150, 10),
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
    # development only: an eHMM whose models exceed 3072 nodes (16S-like backbone of ~6000 columns): every pair goes
    # through the any-size float64 kernels (wh_generic.hip)
    "dna_6k_nodes": ("dna", 20251208, 6000, 64, 0.03, 1e-4, 8, 2000, (150, 1500), 4),
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


def hot_path_step(e, res_t, off_t, maxlen, k, gather_topk=None, keep_device=False):
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
    if keep_device:      # inputs of the next stage (weighted consensus), kept on the device
        hot_path_step.dev = dict(pq=pq, ph=ph, pw=w[keep].contiguous(), co=co, cols=cols, nu=nu, flags=flags)
    if gather_topk is not None:
        idx, w, nk, nu = gather_topk(idx, w, nk, nu)
    out = (idx.cpu(), w.cpu(), nk.cpu(), nu.cpu(), cols.cpu(), co.cpu())
    hot_path_step.dense_redo = int(((flags & 16) != 0).sum().item())
    hot_path_step.multidomain = int(((flags & 2) != 0).sum().item())
    hot_path_step.reported = int(((flags & 1) != 0).sum().item())
    hot_path_step.aligned_cells = float((lens.double() * torch.from_numpy(e.M.astype(np.float64)).to(ph.device)[ph.long()]).sum().item())
    hot_path_step.aligned_residues = float(lens.double().sum().item())
    hot_path_step.align_paths = e.last_align_paths()
    hot_path_step.long_list = e.last_long_list_pairs()           # pairs with more than 16 regions, scored in full by the long-list pass
    hot_path_step.spill_bytes = e.last_score_spill_bytes()      # device counter of the scoring call above (waits for the device: after the copies)
    return out, int(pq.numel()), total_cols


def consensus_stage(e, synth_ehmm, fam, off_t, maxlen, k):
    """Next row #1 (aligner.py:376-473) on the outputs of the last hot_path_step: HIP-event time of
    wh_consensus_dev over every query of this rank.  Reported as an extra stage, not part of <value>."""
    import torch
    d = hot_path_step.dev
    dev = off_t.device
    nq = off_t.numel() - 1
    qpo = torch.zeros(nq + 1, dtype=torch.int64, device=dev)
    torch.cumsum(d["nu"].long(), 0, out=qpo[1:])
    ret = [np.asarray(h.map_cols[1:] - 1, dtype=np.int32) for h in synth_ehmm.hmms]
    ng = [np.asarray(h.nongaps, dtype=np.int32) for h in synth_ehmm.hmms]
    ro = np.zeros(len(ret) + 1, dtype=np.int64)
    ro[1:] = np.cumsum([len(r) for r in ret])
    ro_t = torch.from_numpy(ro).to(dev)
    ret_t = torch.from_numpy(np.concatenate(ret)).to(dev)
    ng_t = torch.from_numpy(np.concatenate(ng)).to(dev)
    codes, _ = e.consensus_t(off_t, maxlen, qpo, d["ph"], d["pw"], d["co"], d["cols"], ro_t, ret_t, ng_t, fam.msa.shape[1], k)
    torch.cuda.synchronize()
    ms, _ = e.last_kernel_ms(3)
    consensus_stage.codes = codes
    return ms


def merge_stage(fam, seqs_local, off_t, nu_t, alphabet):
    """Next row #2 (merger.py:40-131) on the codes of consensus_stage: wall time of wh_merge (host buffers in,
    the two final matrices out: upload + three kernels + download) for this rank's queries merged into the 64 first
    backbone rows.  Reported as an extra stage, not part of <value>."""
    import ctypes as C
    from witch_amd import synth
    from witch_amd._lib import lib, check
    codes = consensus_stage.codes.cpu().numpy().astype(np.int32)
    offs = off_t.cpu().numpy().astype(np.int64)
    nq = len(offs) - 1
    sym = np.frombuffer((synth.symbols(alphabet) + "N").encode(), dtype=np.uint8)
    text = sym[np.minimum(np.concatenate([np.asarray(s, dtype=np.int64) for s in seqs_local]), len(sym) - 1)]
    q_row = np.where(nu_t.cpu().numpy() > 0, 0, -2).astype(np.int32)
    nb, B = 64, fam.msa.shape[1]
    bbm = fam.msa[:nb].astype(np.int64).copy()
    bbm[bbm < 0] = len(sym)
    bb = np.concatenate([sym, np.frombuffer(b"-", dtype=np.uint8)])[bbm].astype(np.uint8)
    pf, pm, nr, wd = C.c_void_p(), C.c_void_p(), C.c_int64(0), C.c_int64(0)
    t0 = time.perf_counter()
    check(lib().wh_merge(int(off_t.device.index or 0), text.ctypes.data, offs.ctypes.data, nq, codes.ctypes.data, q_row.ctypes.data,
                         np.ascontiguousarray(bb).ctypes.data, nb, B, C.byref(pf), C.byref(pm), C.byref(nr), C.byref(wd)), "wh_merge")
    ms = (time.perf_counter() - t0) * 1e3
    rows, width = int(nr.value), int(wd.value)
    lib().wh_free_text(pf)
    lib().wh_free_text(pm)
    return ms, rows, width


def level1_stage(fam, synth_ehmm, names, seqs, k, alphabet, workdir, nq=20000, device=0):
    """The LEVEL-1 path a WITCH maintainer calls (witch_amd.gcmm), end to end on the first
    <nq> queries of the workload: query TEXT in -> QueryAlignmentEngine.run (digitise, score, top-k, align, consensus)
    -> rankBitscores -> writeWeights -> mergeAlignmentsDevice -> the two merged FASTA files on disk.  Wall time of a
    warm process (library loaded, HIP context up); a fresh model handle, so eHMM parsing / upload and every
    first-call allocation are inside.  Reported beside <value>, never part of it."""
    from witch_amd import gcmm, synth

    class _Sub:
        def __init__(self, path, n):
            self.hmm_model_path, self.num_taxa = path, n
    nq = min(nq, len(seqs))
    index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(synth_ehmm.index, synth_ehmm.paths, synth_ehmm.nseq)}
    retained = {i: (h.map_cols[1:] - 1).tolist() for i, h in zip(synth_ehmm.index, synth_ehmm.hmms)}
    nongaps = {i: h.nongaps.tolist() for i, h in zip(synth_ehmm.index, synth_ehmm.hmms)}
    B = fam.msa.shape[1]
    texts = [synth.to_text(s_, alphabet) for s_ in seqs[:nq]]
    bpath = os.path.join(workdir, "level1_backbone.fasta")
    synth.write_msa_fasta(bpath, fam, 0, 64)
    gcmm.warm_up(device)
    t0 = time.perf_counter()
    eng = gcmm.install(gcmm.QueryAlignmentEngine.run(index_to_hmm, list(zip(names[:nq], texts)), k, device=device,
                                                     subset_to_retained_columns=retained, subset_to_nongaps_per_column=nongaps,
                                                     backbone_length=B))
    t1 = time.perf_counter()
    ranked = gcmm.rankBitscores(index_to_hmm, {})
    weights = gcmm.writeWeights(index_to_hmm, ranked)
    t2 = time.perf_counter()
    out = os.path.join(workdir, "level1_out.fasta")
    gcmm.mergeAlignmentsDevice(bpath, {}, output_path=out, taxa=[t for t in names[:nq] if t in weights])
    t3 = time.perf_counter()
    rows = sum(1 for line in open(out) if line.startswith(">"))
    return {"queries": nq, "seconds": round(t3 - t0, 3), "queries_per_s": round(nq / (t3 - t0), 1),
            "stages_s": {"engine_run": round(t1 - t0, 3), "rank_and_weights": round(t2 - t1, 3), "device_merge_and_files": round(t3 - t2, 3),
                         **{"engine_" + kk: round(v, 3) for kk, v in eng.timings.items()}},
            "rows_written": rows,
            "note": "text queries in -> two merged FASTA files out through witch_amd.gcmm (level 1), warm process, fresh model handle; outside the timed region"}


def crc_of(*arrays):
    import zlib
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


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



FAM_NAMES = {0: "wh::k7::score_kernel7", 1: "wh::score_big_kernel", 2: "wh::generic_front_kernel", 3: "wh::wide::score_wide_kernel", 4: "wh::staged launches (wh_staged.hip)"}


def lib_sha16():
    """First 16 hex digits of the sha256 of the loaded libwitch_hip.so: what a profile is stamped with."""
    import hashlib
    from witch_amd._lib import LIB_PATH
    return hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16]


def profile_stamp(workload, plain_run):
    """profiles/traffic.json (tools/profile_round.sh + tools/prof_summary.py): PMC figures of an EARLIER run of this
    command.  Used only when it was taken on this workload with THIS build of the library (sha256 stamp); a file from
    another build is refused, so the line cannot carry a stale traffic figure."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not (plain_run and os.path.exists(tpath)):
        return None, "no stamped profile for this run"
    try:
        tj = json.load(open(tpath))
    except Exception as ex:
        return None, "profiles/traffic.json unreadable: %s" % ex
    if tj.get("workload") != workload:
        return None, "profiles/traffic.json is for workload %s" % tj.get("workload")
    if tj.get("lib_sha16") != lib_sha16():
        return None, "profiles/traffic.json was taken on another build (lib %s, this one %s): refused" % (tj.get("lib_sha16"), lib_sha16())
    return tj, "profiles/traffic.json <- %s; counters of an earlier run of this command on this build (lib %s), NOT measured in this run" % (
        str(tj.get("source", "?")).split(" ")[0], tj.get("lib_sha16"))


def score_roofline(M, lens_local, class_ms, kern_ms0, kern_n0, steps, H, stamp, stamp_src):
    """VALU roofline of the dominant scoring launch class.  Unit = one DP cell (residue x model
    node): 32 flop/cell for the multihit Forward + Backward parsers + ~45 flop/cell for the envelope sweeps (unihit
    Forward, Backward, decoding, null2) = 77 flop/cell over L x M cells per pair, ALGORITHMIC: the price of the work the
    reference does, whatever the kernel executes for it.  Peak: 157.3 TFLOP/s fp32 vector (plain v_fma_f32 at two cycles
    per wave64 instruction reaches it on gfx950).  The dominant kernel = the launch class with the largest measured share
    of the scoring time (one launch per cells-per-lane class); its cells = the local residues x the nodes of ITS models."""
    def cls_of(m):      # cells-per-lane class of a model (witch_amd/csrc/wh_hmm.cpp choose_Q: next multiple of 4)
        return max(4, (-(-int(m) // 64) + 3) // 4 * 4)

    def wide_class(m):  # (cells per lane, waves per pair) of the several-waves-per-pair kernels (wh_api.hip wide_q_of: 12 cells up to 6 144 nodes, 16 up to 8 192, 24 beyond)
        wq = 12 if m <= 6144 else 16 if m <= 8192 else 24
        return wq, -(-int(m) // (64 * wq))
    if class_ms:
        (dom_q, dom_kind), (dom_ms, dom_n) = max(class_ms.items(), key=lambda kv: kv[1][0])
        if dom_kind == 2:
            dom_M = M[M > 3072]
        elif dom_kind == 3:         # several waves per pair: the class key is cells per lane x waves, as the library forms it
            dom_M = M[np.array([m > 3072 and wide_class(m)[0] * wide_class(m)[1] == dom_q for m in M])]
        else:
            dom_M = M[np.array([cls_of(m) == dom_q for m in M])]
        score_launches, score_ms = max(dom_n, 1), dom_ms / max(dom_n, 1)
        cells_launch = float(lens_local.sum() * dom_M.sum()) * steps / score_launches
        wq_w = wide_class(int(dom_M.max())) if dom_kind == 3 and len(dom_M) else (0, 0)
        dom_name = (FAM_NAMES[2] if dom_kind == 2 else "%s<%d cells per lane, %d waves per pair>" % ((FAM_NAMES[3],) + wq_w) if dom_kind == 3
                    else "%s<%d cells per lane>" % (FAM_NAMES.get(dom_kind, "?"), dom_q))
        dom_share = dom_ms / max(kern_ms0, 1e-9)
    else:
        score_launches = max(kern_n0, 1)
        score_ms = kern_ms0 / score_launches
        cells_launch = float(lens_local.sum() * M.sum()) * steps / score_launches
        dom_name, dom_share = "wh::k7::score_kernel7", 1.0
    s_tflops = cells_launch * 77.0 / (score_ms * 1e-3) / 1e12 if score_ms > 0 else 0.0
    L = float(np.mean(lens_local)) if len(lens_local) else 0.0
    r = {"bound": "valu", "kernel": dom_name, "kernel_share_of_scoring_time": round(dom_share, 4),
         "scoring_launch_classes": {"%s/%d" % (FAM_NAMES.get(kd, "?").split("::")[-1], qc): round(v[0] / steps, 3) for (qc, kd), v in sorted(class_ms.items())},
         "achieved": round(s_tflops, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(s_tflops / 157.3, 4),
         "traffic": stamp.get("score_kernel_hbm_bytes_per_launch") if stamp else None, "traffic_source": stamp_src,
         # measured LIVE in this run (wh_last_score_counters): bytes of Forward rows the envelope sweeps of one scoring call asked
         # the memory system to store (the Backward sweeps read about three quarters of them back); the HBM-level figure above
         # is the stamped profile's - a change of the spill shows here first
         "spill_bytes_stored_per_step_live": getattr(hot_path_step, "spill_bytes", None),
         "flop_per_cell": 77, "cells_per_launch": cells_launch, "cells_per_s": round(cells_launch / (score_ms * 1e-3), 1) if score_ms > 0 else 0.0,
         "kernel_ms_avg": round(score_ms, 3), "launches": score_launches,
         "algorithmic_hbm_bytes_per_launch": float(len(lens_local) * H * (L + 9.0)) * steps / score_launches}
    if stamp and stamp.get("score_kernel_valu_insts_per_launch"):
        # executed vector instructions x 64 lanes per cell (SQ_INSTS_VALU of the stamped profile), and the same in flop if
        # every one were an FMA (2 flop) - the figure to hold against the algorithmic 77: their ratio is the instruction
        # efficiency, frac / that ratio the share of cycles the vector ALUs issue
        slots = float(stamp["score_kernel_valu_insts_per_launch"]) * 64.0 / float(stamp.get("score_kernel_cells_per_launch", cells_launch))
        r["executed_valu_lane_ops_per_cell"] = round(slots, 2)
        r["executed_flop_per_cell"] = round(2.0 * slots, 1)
        r["instruction_efficiency"] = round(77.0 / (2.0 * slots), 3)
        r["valu_issue_share"] = round((s_tflops / 157.3) / (77.0 / (2.0 * slots)), 3)
    return r


def also_block(name, steps, warmup, device, nq=None, note=None, golden=None):
    """Another workload measured in the same run, reported beside <value> (never part of it): one-GPU hot-path rate,
    stage split and the VALU roofline of its dominant launch class.  Used for SURVEY.md 8(d)'s own sketch of config 3
    (dna_100k_x200_m1000: root 1000 nt, 0.2 % indels per branch; models of 996-2996 nodes in five size classes) at full
    size, and for a slice of BASELINE config 5 (aa_50k_x500: ALL 500 protein HMMs x the first <nq> of its 50 000
    mixed-length queries - a quarter of those pairs go through the multidomain resolver, reported as an entry of its own).
    <golden> = (case, rep): instead of a synthetic workload, the model files and query texts of tests/golden/<case> (data the
    reference's own example run produced; no oracle involved), the queries replicated <rep> times - the reference's example data
    is the one real-fragment shape in the line (16S fragments of ~400 nt on models of 1 278-2 574 nodes: long-query
    instantiations of the 20 / 24-cell classes, 28 % multidomain pairs), and the synthetic workloads did not show the round-5
    regression that this one did."""
    import torch
    from witch_amd.ehmm import EHMM, pack_queries
    wd = tempfile.mkdtemp(prefix="witch_bench_also_")
    try:
        if golden:
            from tests.conftest import load_case
            case = load_case(golden[0])
            k = int(case.k)
            e = EHMM(case.hmm_paths, hmm_index=case.hmm_index, nseq=case.nseq, device=device)
            seqs = [e.digitize(t) for t in case.qseqs] * int(golden[1])
        else:
            fam, se, names, seqs, k = make_workload(name, wd, nq)
            e = EHMM(se.paths, hmm_index=se.index, nseq=se.nseq, device=device)
        res, offs = pack_queries([s_.astype(np.uint8) for s_ in seqs])
        maxlen = int(np.max(np.diff(offs)))
        res_t, off_t = torch.from_numpy(res).cuda(), torch.from_numpy(offs).cuda()
        for _ in range(warmup):
            hot_path_step(e, res_t, off_t, maxlen, k)
        e.set_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kern_ms, kern_n, class_ms = [0.0] * 5, [0] * 5, {}
        for _ in range(steps):
            hot_path_step(e, res_t, off_t, maxlen, k)
            for which in (0, 1, 2, 4):
                ms, n = e.last_kernel_ms(which)
                kern_ms[which] += ms
                kern_n[which] += n
            for qc, kind, ms in e.last_score_launches():
                c = class_ms.setdefault((qc, kind), [0.0, 0])
                c[0] += ms
                c[1] += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        M = e.M.astype(np.float64)
        lens = np.diff(offs).astype(np.float64)
        out = {"workload": name, "value": round(len(seqs) * steps / dt, 2), "unit": "queries/s", "steps": steps, "warmup": warmup,
               "ms_per_step": round(dt / steps * 1e3, 3), "n_queries": len(seqs), "n_hmms": e.H,
               "model_len_min": int(M.min()), "model_len_max": int(M.max()), "model_len_mean": round(float(M.mean()), 1), "k": k,
               "stage_ms_per_step": {"scoring_kernels": round(kern_ms[0] / steps, 3), "multidomain_resolver": round(kern_ms[4] / steps, 3),
                                     "topk": round(kern_ms[1] / steps, 3), "align": round(kern_ms[2] / steps, 3)},
               "cells_per_s_all_classes": round(float(lens.sum() * M.sum()) * steps / (kern_ms[0] * 1e-3), 1) if kern_ms[0] > 0 else 0.0,
               "roofline": score_roofline(M, lens, class_ms, kern_ms[0], kern_n[0], steps, e.H, None, "not profiled"),
               "pairs_multidomain": hot_path_step.multidomain,
               "note": note or "SURVEY.md 8(d) config 3 as sketched there (root 1000 nt, 0.2 % indels per branch), same timed region as <value>, one GPU"}
        if nq and not golden:
            out["n_queries_of_config"] = WORKLOADS[name][7]
        if nq or golden:
            out["query_len_min_max"] = [int(lens.min()), int(lens.max())]
        if kern_ms[4] > 0 and hot_path_step.multidomain:
            # the resolver is a latency machine: one wavefront per queued pair, 200 stochastic traces whose
            # decisions are fetched as 1 KB threshold lines (16 B per decision x 64 lanes); its rate is (waves in flight) /
            # (wave time per pair)
            import torch as _t
            cu = _t.cuda.get_device_properties(device).multi_processor_count
            rs = kern_ms[4] / steps * 1e-3
            npair = float(hot_path_step.multidomain)
            out["resolver"] = {"pairs": int(npair), "ms_per_step": round(rs * 1e3, 3), "pairs_per_s": round(npair / rs, 1),
                               "waves_in_flight": cu * 8, "wave_ms_per_pair": round(rs * 1e3 * cu * 8 / npair, 3),
                               "threshold_line_bytes": 1024, "decisions_per_line": 64, "bytes_per_decision": 16,
                               "share_of_step": round(rs / (dt / steps), 4)}
        e.close()
        return out
    finally:
        shutil.rmtree(wd, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dna_100k_x200", choices=sorted(WORKLOADS))
    ap.add_argument("--nq", type=int, default=0, help="override the query count (development only)")
    ap.add_argument("--nh", type=int, default=0, help="override the HMM count (development only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-level1", action="store_true", help="skip the level-1 end-to-end stage (reported beside the hot-path value)")
    ap.add_argument("--no-also", action="store_true", help="skip the workloads reported beside the headline (dna_100k_x200_m1000 in full, a 3 000-query slice of aa_50k_x500, the reference's example data x 4)")
    args = ap.parse_args()
    if os.environ.get("WITCH_BENCH_WATCHDOG"):      # tests: a run that stops says where (every thread's stack on stderr), then goes on
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["WITCH_BENCH_WATCHDOG"]), exit=False)

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
    # WITCH_FORCE_COLLECTIVES=1: the RCCL group and the top-k all-gather also at world 1 (smoke test of the nccl path
    # on a one-GPU box, started under torch.distributed.run --nproc-per-node 1); the driver's runs never set it
    from witch_amd.distributed import collectives_forced
    use_dist = world > 1 or collectives_forced()
    if use_dist:
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
        if use_dist:
            from witch_amd.distributed import gather_topk
            # the path's one exchange step: per-query top-k records to every rank over RCCL
            # (ONE all_gather_into_tensor of packed 128-byte records; the shard sizes follow from shard_range)
            def gather(idx, w, nk, nu):
                if rehearsal:                     # gloo gathers host tensors
                    idx, w, nk, nu = idx.cpu(), w.cpu(), nk.cpu(), nu.cpu()
                return gather_topk(idx, w, nk, nu, n_total=nq_total)

        def barrier():
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            hot_path_step(e, res_t, off_t, maxlen, k, gather)
        e.set_timing(True)
        barrier()
        t0 = time.perf_counter()
        kern_ms = [0.0, 0.0, 0.0, 0.0, 0.0]     # scoring kernels, topk, align, (consensus: timed apart), multidomain resolver
        kern_n = [0, 0, 0, 0, 0]
        npairs = ncols = 0
        class_ms = {}                            # (cells per lane, kernel family) -> [ms, launches] of the scoring launches
        for st in range(args.steps):
            out, npairs, ncols = hot_path_step(e, res_t, off_t, maxlen, k, gather, keep_device=(st == args.steps - 1))
            for which in (0, 1, 2, 4):
                ms, n = e.last_kernel_ms(which)
                kern_ms[which] += ms
                kern_n[which] += n
            for qc, kind, ms in e.last_score_launches():
                cls = class_ms.setdefault((qc, kind), [0.0, 0])
                cls[0] += ms
                cls[1] += 1
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        # ---- outside the timed region: the next stage (weighted consensus) and the value distributions
        cons_ms = consensus_stage(e, synth_ehmm, fam, off_t, maxlen, k) if hi > lo else 0.0
        merge_info = None
        try:
            if hi > lo and rank == 0:
                merge_info = merge_stage(fam, seqs[lo:hi], off_t, hot_path_step.dev["nu"], WORKLOADS[args.workload][0])
        except Exception as ex:           # an extra stage must never take the bench line down
            merge_info = ("failed: %s" % ex, 0, 0)
        e.set_timing(False)

        if rank == 0:
            H = e.H
            M = e.M.astype(np.float64)
            lens_local = np.diff(offs).astype(np.float64)
            L = float(np.mean(lens_local)) if hi > lo else 0.0
            qps = nq_total * args.steps / dt
            n_local = hi - lo
            # ---- rooflines per SURVEY.md section 8(d) (score_roofline above)
            cells_step = float(lens_local.sum() * M.sum())              # every local query x every model
            stamp, stamp_src = profile_stamp(args.workload, not args.nq and not args.nh and world == 1)
            roofline = score_roofline(M, lens_local, class_ms, kern_ms[0], kern_n[0], args.steps, H, stamp, stamp_src)
            traffic_align = stamp.get("align_kernel_hbm_bytes_per_launch") if stamp else None
            traffic_src = stamp_src
            # Alignment.  Algorithmic bytes by sweep path (wh_last_align_paths): a pair aligned at full width moves
            # 52 B per L x M cell (Forward rows written + read 24, posteriors 16, OA rows 12);
            # a pair aligned on a node window moves 44 B per L x W cell, W = 256 or 512 nodes (its Forward cells
            # written 8 and read 8, posteriors 16, OA rows 12) - the cells outside the window are computed by the
            # Forward sweep but need not be stored.  Path shares are applied to the pass's totals (sum of L, sum of
            # L x M over the aligned pairs).  The stage time spans the pass's launches (one per model size class,
            # plus the log-space redo pass when pairs leave float32 range).
            align_ms = kern_ms[2] / args.steps
            ap = hot_path_step.align_paths
            n_ap = max(1, sum(ap.values()))
            f256, f512 = ap["window256"] / n_ap, ap["window512"] / n_ap
            a_bytes = (hot_path_step.aligned_residues * 44.0 * (256 * f256 + 512 * f512)
                       + hot_path_step.aligned_cells * 52.0 * (1.0 - f256 - f512)) if sum(ap.values()) else hot_path_step.aligned_cells * 52.0
            a_gbs = a_bytes / (align_ms * 1e-3) / 1e9 if align_ms > 0 else 0.0
            # SURVEY 8(d) prices alignment at 52 B per L x M cell whatever the kernel stores; with the node window the kernel
            # moves 3-4 x fewer bytes than that, so the survey's figure no longer describes it (it would read > 1 of the HBM
            # peak) - printed beside the window-based figure that the roofline uses, not instead of it
            s8d_bytes = hot_path_step.aligned_cells * 52.0
            s8d_gbs = s8d_bytes / (align_ms * 1e-3) / 1e9 if align_ms > 0 else 0.0
            roofline_align = {"bound": "hbm", "kernel": "wh::generic_align_kernel" if int(np.max(e.M)) > 3072 else "wh::align_kernel", "achieved": round(a_gbs, 1), "peak": 8000.0,
                              "unit": "GB/s", "frac": round(a_gbs / 8000.0, 4), "algorithmic_bytes_per_step": a_bytes,
                              "survey_8d": {"bytes_per_step": s8d_bytes, "bytes_per_cell": 52, "equivalent_GBps": round(s8d_gbs, 1), "frac_of_peak": round(s8d_gbs / 8000.0, 4),
                                            "note": "52 B x L x M per aligned pair as SURVEY.md 8(d) prices it; the kernel does not move these bytes (node window), so this is a rate of algorithmic work, not of HBM traffic"},
                              "bytes_per_cell": {"full_width": 52, "window": 44}, "pairs_by_path": ap,
                              "cells_per_step": hot_path_step.aligned_cells, "stage_ms": round(align_ms, 3),
                              "launches_per_step": kern_n[2] / args.steps, "traffic": traffic_align, "traffic_source": traffic_src}
            t_s, t_a = kern_ms[0] / args.steps, align_ms
            combined = (t_s * roofline["frac"] + t_a * roofline_align["frac"]) / (t_s + t_a) if t_s + t_a > 0 else 0.0
            # ---- the three distributions SURVEY.md 8(d) asks for with every run
            nu_h = np.bincount(out[3].numpy(), minlength=k + 1).tolist()
            dist3 = {"n_used": {str(i): int(c) for i, c in enumerate(nu_h) if c},
                     "multidomain_frac": round(hot_path_step.multidomain / max(1, n_local * H), 8)}
            ns = min(256, n_local)
            if ns > 0:
                _, _, det = e.score(res[:offs[ns]], offs[:ns + 1], want_detail=True)
                nreg = np.array([d.nregions for d in det])
                dist3["regions_per_pair"] = {str(i): int(c) for i, c in enumerate(np.bincount(nreg)) if c}
                dist3["regions_per_pair_sample"] = "first %d queries x %d HMMs" % (ns, H)
            line = {
                "metric": "query-seqs aligned/sec (100k queries x 200-HMM eHMM)",
                "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": args.workload, "n_queries": nq_total, "n_hmms": H,
                           "query_len": int(round(L)), "model_len_min": int(M.min()), "model_len_max": int(M.max()),
                           "model_len_mean": round(float(M.mean()), 1),
                           "k": k, "aligned_pairs_per_step": npairs, "sharding": "queries/%d" % world, "collective_backend": (dist.get_backend() if use_dist else None),
                           "pairs_reported_rank0": hot_path_step.reported, "pairs_multidomain_rank0": hot_path_step.multidomain,
                           "pairs_long_list_rank0": getattr(hot_path_step, "long_list", None),
                           "pairs_dense_redo_rank0": hot_path_step.dense_redo,
                           "topk_crc32": crc_of(out[0].numpy(), out[1].numpy(), out[2].numpy(), out[3].numpy())},
                "stage_ms_per_step": {"score": round((kern_ms[0] + kern_ms[4]) / args.steps, 3), "topk": round(kern_ms[1] / args.steps, 3),
                                      "align": round(kern_ms[2] / args.steps, 3),
                                      "score_parts": {"scoring_kernels": round(kern_ms[0] / args.steps, 3),
                                                      "multidomain_resolver": round(kern_ms[4] / args.steps, 3)}},
                "extra_stage_ms": {"consensus_rank0": round(cons_ms, 3),
                                   "merge_rank0_wall": (round(merge_info[0], 3) if merge_info and not isinstance(merge_info[0], str) else (merge_info[0] if merge_info else None)),
                                   "merge_rows_x_width": ([merge_info[1], merge_info[2]] if merge_info else None),
                                   "note": "weighted consensus DP (next row #1, HIP-event time) and final merge (next row #2, wh_merge wall time incl. upload of the codes and download of the two matrices) over this rank's queries, outside the timed region"},
                "roofline": roofline, "roofline_align": roofline_align,
                "roofline_time_weighted_frac": round(combined, 4),
                "distributions": dist3,
            }
            if not args.no_level1 and world == 1:
                try:
                    line["level1_e2e"] = level1_stage(fam, synth_ehmm, names, seqs, k, WORKLOADS[args.workload][0], workdir, 20000, local_rank)
                    line["level1_e2e_queries_per_s"] = line["level1_e2e"]["queries_per_s"]
                except Exception as ex:       # an extra stage must never take the bench line down
                    line["level1_e2e"] = {"failed": "%s: %s" % (type(ex).__name__, ex)}
            if not args.no_also and world == 1 and args.workload == "dna_100k_x200" and not args.nq and not args.nh:
                line["also"] = []
                for wl_, steps_, nq_, note_, golden_ in (
                        ("dna_100k_x200_m1000", 2, None, None, None),
                        ("aa_50k_x500", 1, 3000, "BASELINE.json configs[4] (SURVEY.md 8(d) config 5) as a SLICE: all 500 protein HMMs x its first 3 000 "
                                                 "mixed-length queries (50-2 000 aa), one warm-up + one step, same timed region as <value>, one GPU", None),
                        ("example_e2e_x4", 3, None, "the reference's own example data (tests/golden/example_e2e: 500 16S fragments x 15 HMMs of 1 278-2 574 nodes) "
                                                    "replicated 4 x, one warm-up + three steps, same timed region as <value>, one GPU; a launch-latency-sized batch: "
                                                    "read stage_ms_per_step, not <value>", ("example_e2e", 4))):
                    try:
                        line["also"].append(also_block(wl_, steps_, 1, local_rank, nq_, note_, golden_))
                    except Exception as ex:       # an extra workload must never take the bench line down
                        line["also"].append({"workload": wl_, "failed": "%s: %s" % (type(ex).__name__, ex)})
            if not args.no_cpu_baseline:
                threads = min(os.cpu_count() or 1, 64)
                # about 15 s of CPU work on 64 host threads: 384 queries of the headline (150 nt x 200 models of ~900
                # nodes = 1.0e10 cells); other workloads get the same number of cells
                mean_len = float(np.mean([len(x) for x in seqs[:2048]]))
                cells_per_query = mean_len * float(M.sum())
                n_sample = int(max(16, min(2048, 1.05e10 / max(cells_per_query, 1.0))))
                v, cdt = cpu_baseline(synth_ehmm.paths, synth_ehmm.nseq, seqs, k, min(n_sample, nq_total), threads)
                line["cpu_baseline"] = {"value": round(v, 3), "unit": "queries/s", "cores": threads, "kind": "port",
                                        "sample": "first %d queries x %d HMMs through the float64 oracle "
                                                  "(score + top-k + align), %.1f s" % (min(n_sample, nq_total), H, cdt)}
                # the reference's real CPU path (HMMER 3.1b2 binaries, the reference's process scheme) on a
                # subsample of the same seeded inputs, timed in the build container by
                # tests/tools/time_reference_cpu.py - the binaries cannot travel to the GPU box
                rpath = os.path.join(ROOT, "profiles", "cpu_reference_%s.json" % args.workload)
                if os.path.exists(rpath) and not args.nh:
                    rj = json.load(open(rpath))
                    line["cpu_baseline"]["reference"] = {
                        "value": rj["queries_per_s"], "unit": "queries/s", "cores": rj["cores"], "kind": "reference",
                        "host": rj["host"], "sample": rj["sample"], "seconds": rj["seconds"],
                        "source": "profiles/cpu_reference_%s.json (recorded in the build container, not this run)" % args.workload}
            print(json.dumps(line), flush=True)
        e.close()
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
        if use_dist:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
