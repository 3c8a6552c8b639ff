#!/usr/bin/env python3
"""The level-1 path of INTEGRATION.md section 5 on N ranks, end to end on the reference's example data
(tests/golden/example_e2e): sharded QueryAlignmentEngine.run -> gathered top-k -> mergeAlignmentsDevice (gap
widths all-reduced with MAX, rows gathered to rank 0).  Launch with
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/level1_ranks.py OUT.fasta
WITCH_LEVEL1_REHEARSAL=1: every rank uses cuda:0 and the gloo backend (rehearsal on a one-GPU box).
Rank 0 prints one JSON line with the sha256 of the two files."""
import gzip
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from tests.conftest import load_case
    from witch_amd import gcmm
    out = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("WITCH_LEVEL1_REHEARSAL") == "1"
    device = 0 if rehearsal else local
    from witch_amd.distributed import collectives_forced
    use_dist = world > 1 or collectives_forced()         # WITCH_FORCE_COLLECTIVES=1: the nccl path at one rank
    if use_dist:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    case = load_case("example_e2e")
    g = case.g

    class _Sub:
        def __init__(self, path, n):
            self.hmm_model_path, self.num_taxa = path, n
    index_to_hmm = {i: _Sub(p, n) for i, p, n in zip(case.hmm_index, case.hmm_paths, case.nseq)}
    retained = {int(k): v for k, v in g["retained"].items()}
    nongaps = {int(k): v for k, v in g["nongaps"].items()}
    B = g["backbone_length"]
    bpath = out + ".backbone.%d.fasta" % rank
    with gzip.open(os.path.join(case.dir, "backbone.fasta.gz"), "rt") as f, open(bpath, "w") as o:
        o.write(f.read())
    eng = gcmm.install(gcmm.QueryAlignmentEngine.run(
        index_to_hmm, list(zip(case.qnames, case.qseqs)), case.k, device=device, world=world, rank=rank,
        subset_to_retained_columns=retained, subset_to_nongaps_per_column=nongaps, backbone_length=B))
    full, masked = gcmm.mergeAlignmentsDevice(bpath, {}, output_path=out)
    if use_dist:
        dist.barrier()
    if rank == 0:
        print(json.dumps({"world": world, "backend": dist.get_backend() if use_dist else None, "rows_local": [eng.row_lo, eng.row_hi],
                          "full": hashlib.sha256(open(full, "rb").read()).hexdigest(),
                          "masked": hashlib.sha256(open(masked, "rb").read()).hexdigest()}), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
