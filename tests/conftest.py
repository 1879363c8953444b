import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "neurips21-self-supervised-bug-detection-and-repair_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is thousands of small torch ops: on the 256-thread hosts of the GPU boxes torch's default (128 threads) spends
    # its time in the thread pool -- the 8-graph BASELINE-size case, forward + backward in fp64: 1.9 s at 16 threads, 6.0 s at 64,
    # 353 s at 256 (tools/experiments/oracle_threads.py, profiles/r06zzc_oracle_threads.log).  bench.py's cpu_baseline leg makes
    # the same choice.
    import torch

    torch.set_num_threads(min(16, os.cpu_count() or 8))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
