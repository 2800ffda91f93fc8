import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def spec_of(g):
    from oracle.mlp_ref import MLPSpec
    if "rdim" in g:
        from oracle.rnet_ref import spec_from_fixture
        return spec_from_fixture(g)
    return MLPSpec(tuple(int(v) for v in g["dims"]), str(g["activ"]))


def assert_chain_matches_fixture(res, g, c=None):
    """Chain of a sampler run vs the reference's fixture.  Acceptance indices and acceptance rate:
    bit-exact.  States / log-posteriors: bit-exact on the machine the fixture was generated on;
    elsewhere the AMCMC proposal goes through the host's LAPACK SVD (numpy's legacy
    multivariate_normal), whose last bits depend on the CPU, so 1e-9 is allowed."""
    pick = (lambda a: np.asarray(a)) if c is None else (lambda a: np.asarray(a)[c])
    gp = (lambda k: g[k]) if c is None else (lambda k: g[k][c])
    chain, ref = pick(res["chain"]), gp("chain")
    acc = (chain[1:] != chain[:-1]).any(axis=1)
    assert np.array_equal(acc, (ref[1:] != ref[:-1]).any(axis=1)), "acceptance indices differ"
    assert float(pick(res["accrate"])) == float(gp("accrate"))
    for k in ("chain", "logpost", "mapparams"):
        a, b = pick(res[k]), gp(k)
        if not np.array_equal(a, b):
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11, err_msg=k)
    a, b = pick(res["alphas"]), gp("alphas")
    fin = np.isfinite(b) & (b < 1e300)
    np.testing.assert_allclose(a[fin], b[fin], rtol=1e-6, atol=1e-300)
    assert np.array_equal(np.isfinite(a), np.isfinite(b))
