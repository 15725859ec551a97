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


# The library reads no environment variable for its schedule switches (c2r_set_option is the only way in).  The GPU tests' A/B
# cases predate that and name their switches through these variables: THIS file (test infrastructure) hands them to every
# HipBackend a test creates.  Product code never reads them.
ENV_OPTIONS = {"C2R_GRAPH": "graph", "C2R_CHAIN_GRAPH": "chain_graph", "C2R_FUSED_ITER": "fused_iter", "C2R_FUSE_SMALL": "fuse_small",
               "C2R_FOLD_SOURCE_CELL": "fold_source_cell", "C2R_PAIR_SHELLS": "pair_shells", "C2R_SCHED_HINT": "sched_hint",
               "C2R_SPIN_WAIT": "spin_wait", "C2R_POLL_WAIT": "poll_wait", "C2R_STREAM_HINT": "stream_hint", "C2R_XCD_ORDER": "xcd_order",
               "C2R_XCD_MIN_PER_PLANE": "xcd_min_per_plane", "C2R_XCD_MIN_ALIVE": "xcd_min_alive", "C2R_XCD_QMIN": "xcd_qmin",
               "C2R_CHAINS": "chains", "C2R_BATCH_CAP": "batch_cap", "C2R_SPARSE_EXCHANGE": "sparse_exchange",
               "C2R_SPARSE_FRACTION": "sparse_fraction", "C2R_EXCHANGE_OVERLAP": "exchange_overlap",
               "C2R_EXCHANGE_OVERLAP_MIN": "exchange_overlap_min"}


@pytest.fixture(autouse=True, scope="session")
def _schedule_switches_from_the_test_environment():
    import __graft_entry__ as g
    pkg = g.load_package()
    orig = pkg.HipBackend.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        for env, name in ENV_OPTIONS.items():
            if os.environ.get(env, "") != "":
                self.set_option(name, float(os.environ[env]))
    pkg.HipBackend.__init__ = init
    yield
    pkg.HipBackend.__init__ = orig


@pytest.fixture(params=["exact", "fast"])
def sweep_mode(request, monkeypatch):
    """Run a GPU test once per sweep mode (c2r_params.sweep_mode): the hosts above the C ABI (HipBackend(fast=None), the Fortran
    shim, the native test harness) follow C2R_SWEEP_MODE -- the library itself reads no environment variable --, so every
    context the test creates runs in it; tests._util.tol()/assert_gamma() pick the mode's stated tolerances."""
    monkeypatch.setenv("C2R_SWEEP_MODE", "1" if request.param == "fast" else "0")
    return request.param
