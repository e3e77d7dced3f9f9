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
def refcpu():
    """The oracle (CPU restatement of the reference's LLVM path), built on demand with gcc."""
    from oracle import refcpu as r
    r.build()
    return r


@pytest.fixture(scope="session")
def gpu_ctx():
    """One context for the whole session; fails loudly when the HIP library or the GPU is missing."""
    import exprgrad_amd as eg
    ctx = eg.newGpuContext()
    yield ctx
    ctx.sync()


def rel_err(got, want):
    """max|got - want| / max|want| — the comparison SURVEY.md §7 budgets at 1e-5 for float32."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    denom = np.max(np.abs(want))
    if denom == 0:
        return float(np.max(np.abs(got)))
    return float(np.max(np.abs(got - want)) / denom)


# float32 parity tolerance stated by BASELINE.json north_star: 1e-5 relative.
TOL = 1e-5
