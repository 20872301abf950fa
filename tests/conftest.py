import os
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (ROOT, os.path.join(ROOT, "openset-imagenet_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def usable_cpus():
    """CPUs' worth of time this process can get: min(os.cpu_count(), affinity mask, cgroup CPU quota). A one-GPU lease of the test pool
    shows 256 logical CPUs and a quota of 16; torch sizes its intra-op pool for the 256 (128 threads), and 128 threads throttled to 16
    CPUs run the fp64 reference convolutions of the parity tests 2 - 3.5x slower than 16 do (tools/cpu_conv_threads.py)."""
    n = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        n = min(n, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
        except (OSError, ValueError):
            pass
    return max(1, n)


@pytest.fixture(scope="session", autouse=True)
def _torch_threads_fit_the_lease():
    import torch
    torch.set_num_threads(usable_cpus())
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
