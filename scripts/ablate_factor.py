"""Diagnostic: timing-only ablations of q_decide_factor (results are wrong by construction; never shipped).
Builds variants with -DDF_EXP=n into gpurun_out/ and times 3-iteration solves under rocprofv3-free HIP events
by reading the kernel trace is not needed: we simply compare whole-solve device time with max_iter=3."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "lunar_module_ascent_trajectory_optimiser_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
exp = int(sys.argv[1])
lib = os.path.join(out, f"libascent_exp{exp}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DDF_EXP={exp}",
                       "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(csrc, "ascent_solver.hip"), os.path.join(csrc, "ascent_pipeline.hip")])
from lunar_module_ascent_trajectory_optimiser_amd import _lib
_lib.LIB_PATH = lib
import lunar_module_ascent_trajectory_optimiser_amd as A
os.environ["ASCENT_PIPELINE"] = "split"; os.environ["ASCENT_FACTOR_WAVES"] = "1"
S = A.sweep_isp_drymass()
for _ in range(4):
    r = A.solve_batch(S, 200, max_iter=3, want_traj=False)
print(f"DF_EXP={exp}: solve(max_iter=3) {r.kernel_ms:.2f} ms")
