import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """min(affinity mask, cgroup cpu quota): os.cpu_count() is the whole host, and a CPU oracle that spawns 128 OpenMP threads
    on a 16-core share runs the full-depth parity cases 5 - 25 x slower than one thread per usable core (measured on the GPU
    box: SANA-1.6B bf16 forward + backward 42.9 -> 7.8 s, SD3.5-Medium 39.4 -> 2.1 s; the whole -m gpu suite 380 - 510 s -> < 200 s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(_usable_cores())
    except ImportError:
        pass


@pytest.fixture(scope="session")
def built_lib():
    """Build (cross-compile) libyat_hip.so once per session; returns its path."""
    from yat_amd.build import build
    return build(verbose=False)
