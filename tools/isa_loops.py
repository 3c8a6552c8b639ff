#!/usr/bin/env python3
"""Per-loop instruction mix of one kernel in a hipcc -save-temps .s file.

usage: tools/isa_loops.py FILE.s KERNEL_SUBSTRING
For every innermost loop (a label that a later branch jumps back to) prints its length and how many
scratch loads/stores, LDS reads, VALU, AGPR moves and s_waitcnt it holds - the quick check that a
sweep's row loop is free of register spills.
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^\S+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i:
                loops.append((labels[t], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    print("%s: %d instructions, %d loops (%d innermost)" % (key, len(body), len(loops), len(inner)))
    for a, b in sorted(set(loops)):
        seg = [l.strip() for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
        c = lambda pat: sum(1 for l in seg if re.match(pat, l))
        tag = "inner" if (a, b) in inner else "outer"
        print("  %-5s lines %5d-%5d  n=%5d  scratch_ld=%3d scratch_st=%3d ds_read=%3d ds_write=%2d global=%3d valu=%4d accvgpr=%3d waitcnt=%3d nop=%2d"
              % (tag, a, b, len(seg), c(r"scratch_load"), c(r"scratch_store"), c(r"ds_read|ds_load"), c(r"ds_write|ds_store"),
                 c(r"global_|buffer_|flat_"), c(r"v_(?!accvgpr)"), c(r"v_accvgpr"), c(r"s_waitcnt"), c(r"s_nop")))


if __name__ == "__main__":
    main()
