"""Developer tool: A/B of two builds of the library (PROF_LIB_A / PROF_LIB_B, default = in-tree build) on the config-3
batch: answers, iteration counts, time, for the 16-lane and the one-lane sweeps."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import lunar_module_ascent_trajectory_optimiser_amd as A
    from lunar_module_ascent_trajectory_optimiser_amd import _lib
    if os.environ.get("PROF_LIB"):
        _lib.LIB_PATH = os.path.abspath(os.environ["PROF_LIB"])
    S = A.sweep_isp_drymass()
    os.environ["ASCENT_PIPELINE"] = "split"
    out = {}
    for mode in ("wide", "lane"):
        os.environ["ASCENT_FACTOR"] = mode
        A.solve_batch(S, 200, want_traj=False)
        ms = []
        for _ in range(3):
            r = A.solve_batch(S, 200, want_traj=False); ms.append(A.last_kernel_ms())
        out[mode] = dict(ms=min(ms), conv=int(r.converged.sum()), iters=int(r.iters.sum()), tf=float(r.tf.sum()))
    print(json.dumps(out))
else:
    for name in ("PROF_LIB_A", "PROF_LIB_B"):
        env = dict(os.environ, PROF_LIB=os.environ.get(name, ""))
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(name, os.environ.get(name, "(in-tree)"), o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-400:])
