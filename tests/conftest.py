import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _cpu_quota():
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU bandwidth limit (a GPU box hands its container
    16 CPUs' worth of a 256-core host: PyTorch's default of one thread per core then runs the CPU oracle several times SLOWER)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(period)))
    except (OSError, ValueError):
        pass
    return n


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle legs of the suite (the GPU suite's wall time is mostly oracle time) run on as many threads
    as the container owns, not as the host has."""
    import torch
    torch.set_num_threads(max(1, min(_cpu_quota(), 32)))
    yield


@pytest.fixture(scope="session")
def unet_params():
    from tfpnp_amd import synth
    from tests.golden_inputs import WEIGHT_SEED
    return synth.make_unet_params(WEIGHT_SEED)


def golden(name):
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
