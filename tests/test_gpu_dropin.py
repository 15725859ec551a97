"""GPU integration test of the drop-in boundary: the reference's OWN main program (C2Ray.F90 and
every set-up module, compiled unmodified) linked with c2-ray3dm_amd/fortran/evolve_hip.F90 +
libc2ray_hip.so IN PLACE OF its evolve modules (oracle/ref_build.sh: build_hip_dropin), run on its
own test problem for all 140 time steps, against the outputs of the pure reference run
(tests/golden/refrun32_onesrc.*).  The binary is built in the build container (it contains
compiled reference objects, so it lives in the git-ignored oracle/_ref/); skipped when absent."""
import json
import os
import shutil
import subprocess
import tempfile
import numpy as np
import pytest
from tests._util import GOLDEN

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]     # once per sweep mode (C2R_SWEEP_MODE reaches child processes too)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "N32", "hip", "c2ray_test_hip")


def read_sm3d(path, dtype):
    raw = open(path, "rb").read()
    n = np.frombuffer(raw, dtype=np.int32, count=3, offset=4)
    nb = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=20)[0])
    data = np.frombuffer(raw, dtype=dtype, count=nb // np.dtype(dtype).itemsize, offset=24)
    return data.reshape(tuple(int(v) for v in n), order="F")


@pytest.mark.skipif(not os.path.exists(EXE), reason="drop-in program not built (needs the reference: oracle/ref_build.sh 32)")
@pytest.mark.parametrize("case", ["refrun32_onesrc", "refrun32_std"])
def test_reference_driver_with_hip_evolve_matches_pure_reference_run(case):
    m = json.load(open(os.path.join(GOLDEN, case + ".json")))
    a = np.load(os.path.join(GOLDEN, case + ".npz"))
    d = tempfile.mkdtemp(prefix="c2r_dropin_")
    try:
        os.makedirs(d + "/results")
        open(d + "/answers", "w").write(m["answers"])
        with open(d + "/test_sources.dat", "w") as f:
            f.write("%d\n" % len(m["sources"]))
            for (i, j, k, flux) in m["sources"]:
                f.write("%d %d %d %.17e 0.0\n" % (i, j, k, flux))
        subprocess.check_call([EXE, "answers"], cwd=d, stdout=subprocess.DEVNULL, timeout=600)
        outs = sorted(f for f in os.listdir(d + "/results") if f.startswith("xfrac3D_"))
        assert outs == m["outputs"]
        nonconv = [int(l.split(":")[1]) for l in open(d + "/results/C2Ray.log")
                   if "Number of non-converged points:" in l]
        assert len(nonconv) == m["total_outer_iterations"]       # e.g. 648 outer iterations over 140 steps
        assert nonconv == m["nonconv"]
        for f in m["kept"]:
            z = f[len("xfrac3D_"):-4]
            x = read_sm3d(d + "/results/" + f, np.float64)
            assert np.max(np.abs(x - a["xfrac_" + z])) < 1e-8, f
        # PhotonCounts.out / PhotonCounts2.out (output.F90:504-606, 4 significant digits): the driver writes them from the
        # public variables of `photonstatistics` -- in the drop-in the shim's own module, fed by the sums the device took
        # (c2r_report) instead of the reference's three host loops over the mesh per step
        import __graft_entry__ as g
        fio = g.load_package().fileio
        for name in ("PhotonCounts", "PhotonCounts2"):
            got = np.array(fio.read_photon_counts(os.path.join(d, "results", name + ".out")))
            ref = np.array(m["photon_counts"][name])
            assert got.shape == ref.shape, name
            for j in range(ref.shape[1]):
                tolj = 5e-2 if (name == "PhotonCounts" and j == 7) else 2e-3        # (column 7: the photon-loss tail of the last iteration)
                assert np.all(np.abs(got[:, j] - ref[:, j]) <= tolj * np.abs(ref[:, j]) + 1e-12), (name, j)
        z = m["kept"][0][len("xfrac3D_"):-4]          # the final output: rates of the last step
        g = read_sm3d(d + "/results/IonRates3D_" + z + ".bin", np.float32)
        ref = a["ionrates_" + z]
        assert np.max(np.abs(g - ref) / np.maximum(np.abs(ref), 1e-30)) < 1e-5     # f32 output file
    finally:
        shutil.rmtree(d, ignore_errors=True)
