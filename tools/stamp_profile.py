#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile_round.sh output directory: the flat per-launch figures bench.py puts in its
line (HBM bytes of the scoring / alignment / consensus / resolver kernels, executed vector instructions of the scoring
kernel) with the stamp of the library build they were measured on (bench.py refuses the file on any other build).
usage: tools/stamp_profile.py gpurun_out/<tag> <workload> <cells_per_launch of the dominant scoring class> [note]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d, workload, cells = sys.argv[1], sys.argv[2], float(sys.argv[3])
    t = json.load(open(os.path.join(d, "traffic.json")))
    tag = os.path.basename(os.path.normpath(d))
    sk = t.get("score_kernel7", {})
    out = {
        "workload": workload,
        "lib_sha16": t.get("lib_sha16"),
        "score_kernel_hbm_bytes_per_launch": sk.get("hbm_bytes_per_launch"),
        "score_kernel_valu_insts_per_launch": sk.get("sq", {}).get("SQ_INSTS_VALU"),
        "score_kernel_lds_insts_per_launch": sk.get("sq", {}).get("SQ_INSTS_LDS"),
        "score_kernel_cells_per_launch": cells,
        "score_kernel_sq": sk.get("sq"),
        "align_kernel_hbm_bytes_per_launch": t.get("align_kernel", {}).get("hbm_bytes_per_launch"),
        "consensus_kernel_hbm_bytes_per_launch": t.get("consensus_kernel", {}).get("hbm_bytes_per_launch"),
        "resolve_kernel_hbm_bytes_per_launch": t.get("resolve_kernel", {}).get("hbm_bytes_per_launch"),
        "source": "profiles/%s_traffic.json (tools/profile_round.sh %s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters, separate passes of "
                  "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-level1 --no-also`; gfx950 FETCH_SIZE correction applied)%s"
                  % (tag, tag, (" " + sys.argv[4]) if len(sys.argv) > 4 else ""),
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
