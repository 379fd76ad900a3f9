import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores() -> int:
    """min(affinity mask, cgroup CPU quota): the oracle runs on torch's CPU kernels, and a thread
    pool sized for every core of the host under a 16-core quota makes it several times slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(_usable_cores())


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _inference_by_default():
    """Tests run under torch.no_grad() unless they open torch.enable_grad() themselves: like the
    reference's module, TemporalUnet returns a tensor with an autograd graph (training forward +
    explicit backward pass of the engine) whenever gradients are enabled and a parameter requires grad;
    the parity tests of the sampling path want the inference kernels."""
    import torch
    with torch.no_grad():
        yield
