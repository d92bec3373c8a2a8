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


@pytest.fixture(scope="session", autouse=True)
def _gemm_family_override():
    """LECLIP_TEST_GEMM_FAMILY=128 | 256 | 384: run the whole GPU suite with that GEMM kernel family forced for every call it can take
    (leclip_set_gemm_family; the families are bit-identical, so every parity / golden test must pass unchanged) - how a new family is
    exercised on all the shapes the models produce, not only on its own unit tests.  Tests that assert the DEFAULT dispatch are skipped by
    their own check of this variable.  Unset: the library's rate heuristic (the product)."""
    fam = os.environ.get("LECLIP_TEST_GEMM_FAMILY")
    if fam:
        from leclip_amd.hip import ops
        ops.set_gemm_family(int(fam))
    yield
