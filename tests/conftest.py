import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "slow: long-running (full-size soak)")
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end driver)")


@pytest.fixture(scope="session")
def hip_lib():
    """The HIP C-ABI library; GPU tests fail loudly (no skip, no fallback) when it is missing."""
    from linevis_amd import capi
    return capi.load()


@pytest.fixture(autouse=True)
def _oracle_default_form():
    """Every test starts with the oracle on the reference's literal roots (the library's default form); Case.oracle_params()
    switches it to what the case's settings select (common.Case.literal_form)."""
    from oracle import lvo
    lvo.set_default_intersection_form(True)
    yield
    lvo.set_default_intersection_form(True)
