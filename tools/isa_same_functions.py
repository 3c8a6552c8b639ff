#!/usr/bin/env python3
"""Which device functions of a .hip file changed between two source trees: both are compiled to symbolic gfx950 assembly
(`hipcc --offload-device-only -S`, the flags of witch_amd/csrc/Makefile's K7 objects) and compared function by function
(labels renumbered, comments dropped).  Used at the end of round 5 to show that the row-layout change of the long-query
("SG") instantiations left every other function of wh_score7.hip instruction-identical (profiles/r05_v8_isa_diff.txt).
usage: tools/isa_same_functions.py <old tree> <new tree> [file under witch_amd/csrc, default wh_score7.hip]"""
import os
import re
import subprocess
import sys
import tempfile


def asm(tree, name, out):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-result",
           "-I" + os.path.join(tree, "include"), "-fno-slp-vectorize", "-DWH_SLIM_SPEC", "--offload-device-only", "-S",
           os.path.join(tree, "witch_amd", "csrc", name), "-o", out]
    return subprocess.Popen(cmd, stderr=subprocess.DEVNULL)


def functions(path):
    d, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            d[cur] = []
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is not None:
            s = line.strip()
            if not s or s.startswith(";") or s.startswith(".L"):
                continue
            d[cur].append(re.sub(r"\s*;.*$", "", re.sub(r"\.LBB\d+_\d+", ".LBB", s)))
    return d


def main():
    old, new = sys.argv[1], sys.argv[2]
    name = sys.argv[3] if len(sys.argv) > 3 else "wh_score7.hip"
    with tempfile.TemporaryDirectory() as t:
        pa, pb = asm(old, name, os.path.join(t, "a.S")), asm(new, name, os.path.join(t, "b.S"))
        assert pa.wait() == 0 and pb.wait() == 0, "hipcc failed"
        a, b = functions(os.path.join(t, "a.S")), functions(os.path.join(t, "b.S"))
    names = sorted(set(a) | set(b))
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    same = [dn for n, dn in zip(names, dem) if a.get(n) == b.get(n)]
    diff = [dn for n, dn in zip(names, dem) if a.get(n) != b.get(n)]
    print("%s: %d device functions instruction-identical, %d differ" % (name, len(same), len(diff)))
    print("differ:")
    for x in diff:
        print("   ", x[:170])
    print("identical:")
    for x in same:
        print("   ", x[:170])


if __name__ == "__main__":
    main()
