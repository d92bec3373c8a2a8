import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

import leclip_amd  # noqa: E402

leclip_amd.configure()      # the entry points' hardware-queue setting (before any device call): the GPU tests run the product schedule


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on 8 CPU cores")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
