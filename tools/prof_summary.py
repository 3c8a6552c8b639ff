#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory: per-kernel time from the kernel-trace stats and
HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads -> doubled here; both counters are in KiB-like units
of 1 KB as rocprofv3 reports them).  Writes <dir>/kernel_stats.csv and <dir>/traffic.json."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    for key in ("score_kernel7", "score_kernel_big", "score_big", "resolve_kernel", "align_kernel", "topk_kernel", "consensus_kernel"):
        if key in name:
            return key
    return None


def main():
    d = sys.argv[1]
    stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if stats:
        # per kernel: every dispatch's duration; "full-size" = within 50 % of the longest one (bench.py also
        # scores a 256-query sample once, which rocprofv3's own kernel_stats.csv averages in)
        dur = collections.defaultdict(list)
        res = {}
        for r in csv.DictReader(open(stats[0])):
            if short(r["Kernel_Name"]):
                dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                res[r["Kernel_Name"]] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                                         r["Workgroup_Size_X"])
        with open(os.path.join(d, "kernel_stats.csv"), "w") as o:
            w = csv.writer(o)
            w.writerow(["Name", "Calls", "FullSizeCalls", "FullSizeAverageNs", "MinNs", "MaxNs", "VGPR", "AGPR", "SGPR", "LDS_Block_Size",
                        "Scratch_Size", "Workgroup_Size"])
            for k in sorted(dur, key=lambda k: -sum(dur[k])):
                full = [x for x in dur[k] if x >= 0.5 * max(dur[k])]
                w.writerow([k, len(dur[k]), len(full), "%.0f" % (sum(full) / len(full)), min(dur[k]), max(dur[k])] + list(res[k]))
    # per kernel: the LARGEST dispatch of each pass (bench.py also scores 256 queries once for the
    # regions-per-pair sample; that small launch must not be averaged into the full-size one)
    per = {"FETCH_SIZE": collections.defaultdict(lambda: collections.defaultdict(float)),
           "WRITE_SIZE": collections.defaultdict(lambda: collections.defaultdict(float))}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(d, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k and r["Counter_Name"] == c:
                    per[c][k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {}
    for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
        fetch = max(per["FETCH_SIZE"][k].values(), default=0.0)
        write = max(per["WRITE_SIZE"][k].values(), default=0.0)
        read_b = 2.0 * fetch * 1024.0                    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
        write_b = write * 1024.0
        out[k] = {"dispatches_seen": len(per["FETCH_SIZE"][k]), "FETCH_SIZE_raw_KB": fetch, "WRITE_SIZE_raw_KB": write,
                  "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
                  "hbm_bytes_per_launch": read_b + write_b,
                  "note": "largest dispatch of the kernel in each pass; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)"}
    # SQ / GRBM pass: counters of the LARGEST dispatch of each kernel (summed over the XCDs / SEs as rocprofv3 reports them)
    sq = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(os.path.join(d, "pmc_SQ", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                sq[k][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sq:
        big = max(sq[k].values(), key=lambda c: c.get("SQ_WAVE_CYCLES", 0.0))
        out.setdefault(k, {})["sq"] = dict(big)
    # stamp: the build these counters belong to (bench.py refuses a profile of another build)
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "witch_amd", "libwitch_hip.so")
    out["lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    print(open(os.path.join(d, "kernel_stats.csv")).read() if stats else "no kernel stats")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
