import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def host_threads(limit: int = 32) -> int:
    """Thread count for the CPU oracle legs of the GPU tests: all cores up to `limit` - the GPU boxes have 128+ hardware threads and
    torch's fp32 GEMMs get SLOWER when every one of them is used on the mid-sized matrices of these checks (bench.py probes the best
    count for the same reason)."""
    import torch
    n = max(1, min(limit, os.cpu_count() or 1))
    torch.set_num_threads(n)
    return n
