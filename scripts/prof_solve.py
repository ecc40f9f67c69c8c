"""Developer tool: PROF_N (default 3) default-dispatch solves of the config-3 sweep (PROF_BATCH NLPs, N=200, cold start, tol 1e-9,
trajectories written) and nothing else on the GPU -- the command rocprofv3 runs for profiles/traffic.json:

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 scripts/prof_solve.py
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 scripts/prof_solve.py      (and WRITE_SIZE, separately)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lunar_module_ascent_trajectory_optimiser_amd as A
B = int(os.environ.get("PROF_BATCH", "4096"))
MP = os.environ.get("PROF_MP", "1") == "1"        # the bench's default model: with the reference's DCOST = 1e-5 (PROF_MP=0: without)
S = A.sweep_isp_drymass()[:: max(1, 4096 // B)][:B] if B <= 4096 else A.sweep_config4()[:B]
S[:, 15] = 1e-5
for _ in range(int(os.environ.get("PROF_N", "3"))):
    r = A.solve_batch(S, 200, tol=1e-9, move_penalty=MP)
print("move_penalty", MP, "path", A.default_path(B, 200, move_penalty=MP), "iters", r.iters.mean(), "converged", (r.status == 0).sum(), "kernel ms", A.last_kernel_ms())
