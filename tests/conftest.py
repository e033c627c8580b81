import os
import sys

import pytest

# torch bundles its own HIP runtime; load it BEFORE libfealess_hip.so (which links the system
# one) so that tests which hand device buffers between the two (top-k export, RCCL) share one
# runtime.  The other order leaves torch without a device ("No HIP GPUs are available").
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# fl_recognize_* compute the finer pyramid levels only in the tiles the candidates can touch.  Under test, everything
# outside those tiles is filled with 0xFF first, so a read the tile marking did not foresee cannot go unnoticed.
os.environ.setdefault("FL_DEV_POISON", "1")
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def ctx():
    """A fealess_hip context on device 0.  GPU tests fail loudly (no skip, no fallback) when the
    HIP library or the device is missing."""
    from fealess_amd import api
    c = api.Context(0)
    yield c
    c.close()
