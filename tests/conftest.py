import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _real_backend_builder():
    """CPU tests point npbnn_amd.sampler._make_backend at an oracle-backed stand-in (tests/oracle_backend.serve_from_oracle);
    every test starts and ends with the product's own builder."""
    from npbnn_amd import sampler
    original = sampler._make_backend
    yield
    sampler._make_backend = original
