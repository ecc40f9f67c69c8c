import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def coracle():
    """The plain-C oracle (built on demand with gcc)."""
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def nominal_oracle_solution(coracle):
    from oracle.ascent_numpy import Params
    p16 = coracle.pack_params(Params())
    r = coracle.solve_batch(p16[None], 200, 300, 1e-9, want_blob=True)
    assert r["status"][0] == 0
    return p16, r


def random_interior_blob(nt, seed, p16, coracle):
    """A strictly interior primal-dual iterate: a few oracle IP iterations from the cold start,
    then multipliers perturbed (seeded)."""
    rng = np.random.default_rng(seed)
    r = coracle.solve_batch(p16[None], nt, 3 + seed % 4, 1e-9, want_blob=True, coarse_nodes=-1)
    blob = r["blob"][0].copy()
    K = nt - 1
    blob[8 * K:15 * K] += 0.05 * rng.standard_normal(7 * K)           # lambda
    blob[15 * K:21 * K] *= rng.uniform(0.7, 1.3, 6 * K)               # bound multipliers
    return blob
