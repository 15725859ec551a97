"""GPU test of the multi-rank path with two processes on the one GPU of the test box (gloo as the
collective, see tests/_gpu2_worker.py): sources sharded 1+rank,NumSrc,npr, Gamma + {photon loss, nbox}
all-reduced, global pass replicated -- against the reference fixture."""
import os
import subprocess
import sys
import numpy as np
import pytest
from tests._util import F, load_case

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]     # once per sweep mode (C2R_SWEEP_MODE reaches child processes too)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_match_reference(tmp_path):
    out = tmp_path / "out.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29741",
                           os.path.join(ROOT, "tests", "_gpu2_worker.py"), str(out)], env=env, cwd=ROOT, timeout=580)
    got = np.load(out)
    m, a = load_case("evolve32_std_bubbles")
    s = m["steps"]["step001"]
    for mode in ("python", "native", "balanced", "native_slab"):
        assert int(got[mode + "_niter"]) == s["niter"]
        assert list(got[mode + "_conv"]) == s["log"]["nonconv"]
        assert int(got[mode + "_nbox"]) == s["sum_nbox_all"]
        assert abs(float(got[mode + "_loss"]) - s["photon_loss_all"]) <= 1e-10 * abs(s["photon_loss_all"])
        assert np.max(np.abs(got[mode + "_xh"] - F(a["step001_xh_after"]))) < 1e-9
        ref = F(a["step001_phih_grid"])
        assert np.max(np.abs(got[mode + "_phih"] - ref) / np.maximum(ref, 1e-60)) < 1e-8
    # the non-isothermal step on two ranks (heating rates sharded + all-reduced, evolve.F90:604-609)
    mt, at = load_case("evolve32_thermal")
    st = mt["steps"]["step001"]
    for mode in ("thermal_python", "thermal_native", "thermal_native_slab", "thermal_native_det", "thermal_native_slab_det"):
        assert int(got[mode + "_niter"]) == st["niter"]
        assert list(got[mode + "_conv"]) == st["log"]["nonconv"]
        assert np.max(np.abs(got[mode + "_xh"] - F(at["step001_xh_after"]))) < 1e-9
        ref = F(at["step001_phiheat_grid"])
        assert np.array_equal(got[mode + "_heat"] == 0, ref == 0)
        assert np.max(np.abs(got[mode + "_heat"] - ref) / np.maximum(ref, 1e-60)) < 1e-8
        assert np.max(np.abs(got[mode + "_temper"].astype(np.float64) / at["step001_temper_after"] - 1)) <= 1.5e-7
    # slab chemistry from the Python host (Evolve(slab=True): reduce-scatter, slab global pass, all-gather) with the rates in
    # deterministic order: the same step, bit for bit, as the replicated pass -- rates included (gathered when the step ends)
    for k in ("xh", "xh_av", "temper", "heat", "phih"):
        assert np.array_equal(got["thermal_native_det_" + k], got["thermal_native_slab_det_" + k]), k
