"""Diagnostic: per-pass shader-cycle shares of k_solve (separate -DASCENT_PROFILE build; never timed)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
csrc = os.path.join(ROOT, "lunar_module_ascent_trajectory_optimiser_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "libascent_prof.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DASCENT_PROFILE",
                       "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(csrc, "ascent_solver.hip")])
from lunar_module_ascent_trajectory_optimiser_amd import _lib
_lib.LIB_PATH = lib
import lunar_module_ascent_trajectory_optimiser_amd as A
L = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(round(B ** 0.5))
S = A.sweep_isp_drymass(n, n)
A.solve_batch(S, 200, want_traj=False)
out = (C.c_ulonglong * 8)()
L.ascent_debug_profile(out, 1)
r = A.solve_batch(S, 200, want_traj=False)
L.ascent_debug_profile(out, 1)
v = np.array(list(out), dtype=float)
names = ["E error", "B+F+A newton", "barrier_now", "T trials", "U update"]
tot = v[:5].sum()
nw = (B + 63) // 64
print(f"batch {B}: kernel {r.kernel_ms:.1f} ms, mean iters {r.iters.mean():.1f}, waves {nw}")
for nme, x in zip(names, v):
    print(f"  {nme:14s} {x/nw/r.iters.mean()/199:10.0f} cycles per stage-visit-iteration   {100*x/tot:5.1f} %")
print(f"  total cycles per wave {tot/nw:.3e}  -> {tot/nw/ (r.kernel_ms*1e-3)/1e9:.2f} GHz-equivalent busy")
