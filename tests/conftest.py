import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~8 s on the GPU box (round-6 timing: profiles/r06_gpu_suite_durations.txt); "
                                       "`-m 'gpu and not slow'` is the quick pass, the driver's `-m gpu` runs everything")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
