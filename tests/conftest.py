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
    # threads the oracle may really use: the affinity mask capped by the cgroup CPU quota (a one-GPU box hands the job 16 of the host's
    # cores; OpenMP's default of one thread per visible core only oversubscribes them)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        tok = open("/sys/fs/cgroup/cpu.max").read().split()
        if tok[0] != "max":
            cores = min(cores, max(1, round(int(tok[0]) / int(tok[1]))))
    except Exception:
        pass
    o.set_num_threads(cores)
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
