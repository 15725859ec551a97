"""Drivers built with -DALLFRAC (c2r_params.allfrac; ionfractions_module.F90:19-50): xh, xh_av, xh_intermed carry both fractions and
the NEUTRAL fraction is stored, not derived as 1 - x.  Fixtures from the reference compiled with that one flag (oracle/ref_build.sh
32:allfrac) on a field whose stored neutral fraction is deliberately NOT 1 - x (2e-3 of noise, 2 % stored zeros): a path that derived
it would not reproduce these numbers.  Both sweep modes; tolerances of tests/_util.py."""
import ctypes as C
import numpy as np
import pytest
from tests._util import F, load_case, oracle_for, relerr, tol, assert_gamma, oracle_pass, STATE_RTOL

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def _backend(pkg, tables, m, n, nd, xh, xh0):
    b = pkg.HipBackend(n, *tables, device=0, allfrac=True)
    b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
    b.set_sources(m["srcpos"], m["normflux"])
    b.set_rank(0, 1)
    b.load(ndens=nd, xh=xh, xh0=xh0)
    return b


@pytest.mark.parametrize("name", ["sweep32_allfrac", "sweep32_allfrac_zeros"])
def test_allfrac_sweep_vs_reference_fixture(pkg, tables, name):
    """evolve0D takes n_HI from the STORED neutral fraction (evolve_point.F90:131-134; `_zeros`: stored zeros raised to epsilon):
    sub-box counts exact, photon loss, rates and one source's column densities inside the mode's tolerances of the -DALLFRAC
    reference."""
    m, a = load_case(name)
    n = m["n"]
    nd, xh, xh0 = F(a["ndens"]), F(a["xh"]), F(a["xh0"])
    b = _backend(pkg, tables, m, n, nd, xh, xh0)
    b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert nbox == m["sum_nbox"]
    assert abs(loss - m["photon_loss"]) <= tol("loss") * abs(m["photon_loss"])
    phih, ref = b.fetch("phih_grid"), F(a["phih"])
    o = oracle_for(m, tables, n); o.enable_allfrac(xh0.copy())
    oloss, onb, ovis, ophih, w = oracle_pass(o, nd, xh, m["srcpos"], m["normflux"])
    assert np.array_equal(ophih, ref) and (onb, ovis) == (nbox, vis)
    assert_gamma(phih, ref, w, name)
    b.zero_rates()
    nb1, l1, v1, cd = b.do_source(m["ns_dump"], want_coldens=True)
    cref = F(a["coldensh_out"])
    assert np.array_equal(cd == 0, cref == 0) and relerr(cd, cref) < tol("cd")
    b.close()
    # the same inputs in a context that derives the neutral fraction are NOT these rates: the fixture tells the builds apart
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
    b.set_sources(m["srcpos"], m["normflux"]); b.set_rank(0, 1); b.load(ndens=nd, xh=xh)
    b.begin_step(); b.zero_rates(); b.pass_sources()
    assert relerr(b.fetch("phih_grid"), ref, floor=1e-60) > 1e-4
    with pytest.raises(pkg.C2RayHipError):
        b.fetch("xh0")                                           # arrays 7 - 9 exist in allfrac contexts only
    b.close()


def test_allfrac_whole_steps_vs_reference_fixture(pkg, tables):
    """evolve3D of the -DALLFRAC reference (both fractions read and written by evolve0D_global, evolve_point.F90:341-346 / :394-399;
    Test 2 on the stored neutral sum, evolve.F90:179-181; photon statistics on it): iteration count and non-converged sequence
    exact, both halves of xh within tol("x"), the statistics to 1e-9 -- through the device-resident entry and through the
    host-pointer entry the Fortran shim calls, whose arrays are then the driver's (mesh,0:1) ones."""
    m, a = load_case("evolve32_allfrac")
    n = m["n"]
    lib = pkg.load_library()
    for tag, s in m["steps"].items():
        nd, xh, xh0 = F(a[tag + "_ndens"]), F(a[tag + "_xh_before"]), F(a[tag + "_xh_before0"])
        b = _backend(pkg, tables, s, n, nd, xh, xh0)
        rep = b.evolve3d_native(s["dt"])
        assert rep.converged and rep.niter == s["niter"] and list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
        assert rep.sum_nbox_all == s["sum_nbox_all"]
        assert abs(rep.photon_loss_all - s["photon_loss_all"]) <= tol("loss") * abs(s["photon_loss_all"]) + 1e-300
        # Test 2's relative changes are differences of mesh sums: with every cell inside tol("x") they are good to 2 N^3 tol / sum
        # (the stored zeros of this field make single cells that sensitive; the shipped-build fixtures hold 1e-7 here)
        rel = np.array([rep.it_rel_change_xh1[:rep.niter], rep.it_rel_change_xh0[:rep.niter]]).T
        ref2 = np.array(s["log"]["test2"][1:])
        s1 = np.array(rep.it_sum_xh1[:rep.niter]); s0 = n ** 3 - s1
        assert np.all(np.abs(rel - ref2) <= 2.0 * n ** 3 * tol("x") / np.stack([s1, s0], axis=1))
        for k in ("totrec", "totcollisions", "total_ion"):
            assert abs(getattr(rep, k) / s[k] - 1) < 1e-9, (k, getattr(rep, k), s[k])
        assert abs(rep.dh0 - s["dh0"]) < 1e-9 * abs(s["total_ion"])
        for name in ("xh", "xh_av", "xh_intermed"):
            key = "xh_after" if name == "xh" else name
            assert np.max(np.abs(b.fetch(name) - F(a["%s_%s" % (tag, key)]))) < tol("x"), name
            assert np.max(np.abs(b.fetch(name + "0") - F(a["%s_%s0" % (tag, key)]))) < tol("x"), name + "0"
        # the host-pointer entry on (mesh,0:1) arrays: neutral half first
        x4 = np.concatenate([xh0, xh]); xav4 = np.zeros(2 * n ** 3); xint4 = np.zeros(2 * n ** 3); phih = np.zeros(n ** 3)
        rep2 = type(rep)()
        rc = lib.c2r_evolve3d(b.ctx, s["dt"], nd.ctypes.data, x4.ctypes.data, xav4.ctypes.data, xint4.ctypes.data, phih.ctypes.data, C.byref(rep2))
        assert rc == 0 and rep2.niter == s["niter"] and list(rep2.it_conv_flag[:rep2.niter]) == s["log"]["nonconv"]
        ncell = n ** 3
        assert np.max(np.abs(x4[ncell:] - F(a[tag + "_xh_after"]))) < tol("x") and np.max(np.abs(x4[:ncell] - F(a[tag + "_xh_after0"]))) < tol("x")
        assert np.max(np.abs(xav4[ncell:] - F(a[tag + "_xh_av"]))) < tol("x") and np.max(np.abs(xav4[:ncell] - F(a[tag + "_xh_av0"]))) < tol("x")
        b.close()


def test_allfrac_fortran_drop_in(pkg, tmp_path):
    """The fixture driver (oracle/ref_driver.F90) compiled with -DALLFRAC and linked with the shim's modules instead of the
    reference's (oracle/ref_build.sh 32:allfrac: ref_driver_hip; the shim sets c2r_params%allfrac and hands its (mesh,0:1) arrays
    over whole): two evolve3D steps from the fixture's inputs -- same iteration history, both halves of xh within 1e-9."""
    import json, os, sys
    from tests._util import GOLDEN
    sys.path.insert(0, GOLDEN)
    import inputs as gi
    exe = os.path.join(gi.REF, "N32_allfrac", "hip", "ref_driver_hip")
    if not os.path.exists(exe):
        pytest.skip("ref_driver_hip of the allfrac variant not built (needs the reference: oracle/ref_build.sh 32:allfrac)")
    m = json.load(open(os.path.join(GOLDEN, "evolve32_allfrac.json")))
    a = np.load(os.path.join(GOLDEN, "evolve32_allfrac.npz"))
    x, x0 = a["step001_xh_before"], a["step001_xh_before0"]
    d = gi.run_driver(32, gi.SRC_STD, {"mode": "'evolve'", "nsteps": 2, "dump_first": 1, "dump_last": 2, "x0_file": "'x0.f64'"},
                      dens=gi.density_factor(32, 11), xfield=x, variant="allfrac", extra_files={"x0.f64": lambda p: x0.T.tofile(p)},
                      hip=True, d=str(tmp_path / "run"))
    log = gi.parse_log(d + "/results/C2Ray.log")
    for k, tag in enumerate(("step001", "step002")):
        s = m["steps"][tag]
        assert np.array_equal(gi.rd(d, tag + "_xh_before0.f64", 32), a[tag + "_xh_before0"]) if k == 0 else True      # same inputs
        assert log[k]["nonconv"] == s["log"]["nonconv"]
        kv = gi.read_kv("%s/dump/%s_out.txt" % (d, tag))
        assert kv["sum_nbox_all"] == s["sum_nbox_all"]
        for name in ("xh_after", "xh_av", "xh_intermed"):
            assert np.max(np.abs(gi.rd(d, "%s_%s.f64" % (tag, name), 32) - a["%s_%s" % (tag, name)])) < 1e-9, (tag, name)
            assert np.max(np.abs(gi.rd(d, "%s_%s0.f64" % (tag, name), 32) - a["%s_%s0" % (tag, name)])) < 1e-9, (tag, name + "0")
        for key in ("totrec", "totcollisions", "total_ion"):
            assert abs(kv[key] / s[key] - 1) < 1e-9, key
