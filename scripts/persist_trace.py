"""Developer tool: iteration history of one NLP in the persistent kernel (diagnostic -DPERSIST_TRACE=<index> build, LIB=) beside the
C restatement's trace.  python scripts/persist_trace.py <index in the 64-NLP config-4 sample> <scheme> <tol>"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from lunar_module_ascent_trajectory_optimiser_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
import lunar_module_ascent_trajectory_optimiser_amd as A
from oracle import c_oracle as O
i, scheme, tol = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
S = A.sweep_config4()[::4099][:64]
os.environ["ASCENT_SMALL_BATCH"] = "off"
r = A.solve_batch(S[i:i+1], 200, tol=tol, scheme=scheme, max_iter=60)
print("gpu iters", r.iters, r.status, flush=True)
O.lib().oracle_set_trace(1)
o = O.solve_batch(S[i:i+1], 200, 60, tol, scheme=scheme); O.set_scheme(0)
O.lib().oracle_set_trace(0)
print("oracle iters", o["iters"])
