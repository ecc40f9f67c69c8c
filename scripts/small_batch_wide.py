"""One NLP per wavefront (p_solve<.,.,.,.,1>, batches <= 1024) against four per wavefront (ASCENT_PERSIST_WIDE=0): kernel time and
iteration counts over batch sizes, with and without the move penalty."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lunar_module_ascent_trajectory_optimiser_amd as A
S = A.sweep_isp_drymass(); S[:, 15] = 1e-5
for mp in (False, True):
    for B in (1, 4, 64, 256, 1024, 2048):
        Sx = S[:: max(1, 4096 // B)][:B]
        out = []
        for wide in ("1", "0"):
            os.environ["ASCENT_PERSIST_WIDE"] = wide
            for rep in range(3):
                r = A.solve_batch(Sx, 200, tol=1e-9, want_traj=False, move_penalty=mp)
            out.append((r.kernel_ms, r.iters.copy(), r.tf.copy(), (r.status == 0).sum()))
        del os.environ["ASCENT_PERSIST_WIDE"]
        print(f"move_penalty {mp} batch {B:5d}: one NLP per wavefront {out[0][0]:7.3f} ms | four {out[1][0]:7.3f} ms | converged {out[0][3]}/{out[1][3]} "
              f"iterations equal {np.array_equal(out[0][1], out[1][1])} max |dtf| {np.abs(out[0][2] - out[1][2]).max():.1e}")
