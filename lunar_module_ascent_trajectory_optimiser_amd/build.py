"""Builds csrc/libascent.so for gfx950 with hipcc (in-tree, no JIT cache)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libascent.so")
OBJ = os.path.join(CSRC, "_obj")
SOURCES = ["ascent_solver.hip", "ascent_pipeline.hip", "ascent_dense.hip", "ascent_blocktri.hip", "ascent_persist.hip", "ascent_hs.hip"]
HEADERS = ["ascent_device.hpp", "ascent_tile.hpp", "ascent_pipeline.hpp", "ascent_dense.hpp", "ascent_blocktri.hpp", "ascent_persist.hpp",
           "ascent_persist_dev.hpp", os.path.join(ROOT, "include", "ascent.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source under csrc/_obj (rebuilt when the source or any header is newer), compiled side by side, then linked."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h if os.path.isabs(h) else os.path.join(CSRC, h)) for h in HEADERS)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include")]
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, j in jobs:
        if j.wait() != 0:
            raise subprocess.CalledProcessError(j.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
