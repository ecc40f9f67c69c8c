#!/usr/bin/env python3
"""Per-function register / scratch / occupancy table of a gfx950 assembly listing made with
`hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -save-temps -c <file.hip>` (the *-gfx950.s file)."""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return n


def main(path):
    # the statistics block of a function follows its "-- End function" marker and ends with "; Occupancy"
    name, rows, cur = None, [], None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            continue
        if "-- End function" in line and name:
            cur = {}
            continue
        if cur is not None:
            m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)\D*(\d+)", line)
            if m:
                cur[m.group(1)] = int(m.group(2))
                if m.group(1) in ("Occupancy",) or (m.group(1) == "ScratchSize" and False):
                    rows.append((name, cur)); cur = None; name = None
            elif line.startswith("\t.") and "ScratchSize" in cur:      # a device function's block has no Occupancy line
                rows.append((name, cur)); cur = None; name = None
    print(f"{'function':62s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s} {'code':>7s}")
    for n, c in rows:
        print(f"{demangle(n)[-62:]:62s} {c.get('NumVgprs', 0):5d} {c.get('NumAgprs', 0):5d} {c.get('ScratchSize', 0):8d} "
              f"{c.get('Occupancy', 0):4d} {c.get('LDSByteSize', 0):7d} {c.get('codeLenInByte', 0):7d}")


if __name__ == "__main__":
    main(sys.argv[1])
