import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One context on device 0.  GPU tests must run the HIP path: a missing
    library or device is an error, never a skip or a CPU substitute."""
    from montecarlo_amd import Context

    ctx = Context(0)
    yield ctx
    ctx.close()
