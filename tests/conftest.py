import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lidar-global-registration_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def lgr():
    """Context on cuda:0 through the C ABI; fails loudly when the HIP extension is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from lgr_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
