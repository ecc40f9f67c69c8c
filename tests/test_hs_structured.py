"""CPU check of the structured Hermite-Simpson Newton step restated in tests/hs_structured.py (the algorithm of
csrc/ascent_hs.hip) against the generic sparse-LU step of the generalised oracle (sympy-generated derivatives)."""
import numpy as np
import pytest

from conftest import generic_lu_newton_step, random_interior_blob, params_of_row
import hs_structured as hs


@pytest.mark.parametrize("nt,seed,mu,dw", [(12, 0, 0.1, 0.0), (25, 1, 1e-2, 1e-4), (40, 2, 1e-3, 1e-2), (33, 3, 0.05, 1.0)])
def test_structured_hs_step_equals_generic_lu(coracle, nt, seed, mu, dw):
    from oracle.ascent_numpy import Params
    P = Params()
    p16 = coracle.pack_params(P)
    blob = random_interior_blob(nt, seed, p16, coracle)
    step, inertia = hs.newton_step(P, nt, blob, mu, dw)
    assert inertia == 0
    lu, _, _, _ = generic_lu_newton_step(P, nt, blob, mu, dw, scheme=2)
    K = nt - 1
    for lo, hi in ((0, 8 * K), (8 * K, 15 * K), (15 * K, 21 * K), (21 * K, 21 * K + 10)):
        assert np.abs(step[lo:hi] - lu[lo:hi]).max() <= 1e-8 * max(1.0, np.abs(lu[lo:hi]).max())
