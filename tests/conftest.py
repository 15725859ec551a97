import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tables():
    from tests._util import load_tables
    return load_tables()


@pytest.fixture(params=["exact", "fast"])
def sweep_mode(request, monkeypatch):
    """Run a GPU test once per sweep mode (c2r_params.sweep_mode): c2r_create honours C2R_SWEEP_MODE, so every
    context the test creates follows it; tests._util.tol()/assert_gamma() pick the mode's stated tolerances."""
    monkeypatch.setenv("C2R_SWEEP_MODE", "1" if request.param == "fast" else "0")
    return request.param
