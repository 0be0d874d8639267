import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# lets tests force a GEMM kernel configuration through CCV_GEMM_RING / CCV_GEMM_SPLIT (read per call only when this
# is set before the library's first GEMM); with those two unset the built-in planner decides as usual
os.environ.setdefault("CCV_GEMM_TUNE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
