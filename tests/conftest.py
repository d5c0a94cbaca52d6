import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle on one thread per usable core (yat_amd/common/host.py): with the 128 OpenMP threads torch takes by default on a
    # 16-core share the full-depth parity cases run 5 - 25 x slower (SANA-1.6B bf16 forward + backward 42.9 -> 7.8 s)
    try:
        import torch
        from yat_amd.common.host import usable_cores
        torch.set_num_threads(usable_cores())
    except ImportError:
        pass


@pytest.fixture(scope="session")
def built_lib():
    """Build (cross-compile) libyat_hip.so once per session; returns its path."""
    from yat_amd.build import build
    return build(verbose=False)
