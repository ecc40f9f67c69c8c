#!/usr/bin/env python3
"""Developer tool: the loops of one function in a gfx950 assembly listing (-save-temps) with their instruction mix
(DPP moves, LDS, scratch, global memory, waits, AGPR moves).  usage: asm_loops.py file.s <substring of the mangled name>"""
import re
import sys


def main(path, key):
    src = open(path).read().split("\n")
    start = end = None
    for i, l in enumerate(src):
        if start is None and re.match(r"^_Z\w+:", l) and key in l:
            start = i
        if start is not None and "-- End function" in l:
            end = i
            break
    fn = src[start:end]
    print(fn[0], len(fn), "lines")
    lab = {l.split(":")[0]: i for i, l in enumerate(fn) if re.match(r"^\.LBB\d+_\d+:", l)}
    isins = lambda b: b.startswith("\t") and not b.startswith("\t.") and not b.startswith("\t;")
    for i, l in enumerate(fn):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            ins = [b for b in fn[lab[m.group(1)]:i + 1] if isins(b)]
            c = lambda p: sum(1 for b in ins if re.search(p, b))
            print("loop %-10s lines %6d-%6d: %5d ins | dpp %3d ds %3d scratch %3d global %3d waitcnt %3d accvgpr %4d" % (
                m.group(1), lab[m.group(1)], i, len(ins), c("dpp"), c(r"\sds_"), c("scratch_"), c("global_"), c("s_waitcnt"), c("accvgpr")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
