import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # DCA_AMD_TEST_LIB: run the suite against an experiment build of the HIP library (tools/gpu_heads_narrow_check.sh shows
    # that the parity tests reject a build with one matrix product fewer per fp32 product) -- test infrastructure, the product always loads its own library
    lib = os.environ.get('DCA_AMD_TEST_LIB')
    if lib:
        from dca_amd import build as _b
        _b.LIB = os.path.abspath(lib)
        _b.needs_build = lambda: False


@pytest.fixture(scope='session')
def biochemists():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'biochemists.npz'))
    return {k: g[k] for k in g.files}


def synth_counts(n, G, seed=0, dropout=0.3):
    """Gamma-Poisson counts with extra dropout, every gene / cell >= 1 count."""
    rng = np.random.Generator(np.random.PCG64(seed))
    m = rng.normal(-3.2, 1.6, size=G)
    lib = rng.lognormal(0.0, 0.4, size=n)
    lam = lib[:, None] * np.exp(m)[None, :] * rng.gamma(2.0, 0.5, size=(n, G)) * 40.0
    y = rng.poisson(lam).astype(np.float64)
    y *= rng.random((n, G)) >= dropout
    y[np.arange(n), rng.integers(0, G, n)] += 1
    y[rng.integers(0, n, G), np.arange(G)] += 1
    return y
