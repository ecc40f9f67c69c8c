"""Diagnostic: where a step of q_decide_factor spends its cycles (separate -DDF_STAMPS build; shares only)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "lunar_module_ascent_trajectory_optimiser_amd", "csrc")
lib = os.path.join(ROOT, "gpurun_out", "libascent_stamps.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDF_STAMPS",
                       "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(csrc, "ascent_solver.hip"), os.path.join(csrc, "ascent_pipeline.hip")])
from lunar_module_ascent_trajectory_optimiser_amd import _lib
_lib.LIB_PATH = lib
import lunar_module_ascent_trajectory_optimiser_amd as A
L = _lib.load()
os.environ["ASCENT_PIPELINE"] = "split"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(round(B ** 0.5))
S = A.sweep_isp_drymass(n, B // n)
A.solve_batch(S, 200, want_traj=False)
out = (C.c_ulonglong * 6)()
L.ascent_debug_df_stamps(out, 1)
r = A.solve_batch(S, 200, want_traj=False)
L.ascent_debug_df_stamps(out, 1)
v = [float(x) for x in out]
nsweeps = v[5]
names = ["matrix part (incl. wait for matrix loads)", "rhs 0 (incl. wait for vector loads)", "rhs 1+2", "schur sums"]
print(f"batch {B}: solve {r.kernel_ms:.1f} ms; {nsweeps:.0f} wave-sweeps; cycles per step: total {v[4]/nsweeps/199:.0f}")
for nme, x in zip(names, v[:4]):
    print(f"   {nme:45s} {x/nsweeps/199:8.0f} cycles/step  {100*x/sum(v[:4]):5.1f} %")
