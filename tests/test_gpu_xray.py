"""The second ("P") source type of photoion_rates on the GPU (radiation_photoionrates.F90:133-137; builds of the driver
with use_xray_SED=.true., sed_parameters.f90:56): c2r_set_xray_tables / c2r_set_xray_sources, the XRAY variants of the sweep
kernels.  Fixtures: the reference rebuilt with that one parameter (oracle/ref_build.sh 32:xray), its X-ray tables set by the
fixture driver to the reference's own power-law tables (the reference integrates its X-ray tables over an array it never
fills, radiation_tables.F90:367: they are inputs of this path).  Both sweep modes."""
import json
import os
import shutil
import sys
import tempfile
import numpy as np
import pytest
from tests._util import F, GOLDEN, load_case, oracle_for, expand, relerr, tol, assert_gamma, oracle_pass

sys.path.insert(0, GOLDEN)
import inputs as gi      # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]

XSRC = [gi.SRC_STD[0] + (3e6,), gi.SRC_STD[1] + (0.0,), gi.SRC_STD[2] + (2e7,), gi.SRC_STD[3] + (0.0,), gi.SRC_STD[4] + (5e8,),
        gi.SRC_STD[5] + (0.0,), gi.SRC_STD[6] + (1e6,), gi.SRC_STD[7] + (0.0,), gi.SRC_STD[8] + (4e8,), gi.SRC_STD[9] + (0.0,)]
BUBBLES = [(18, 18, 18), (20, 10, 10), (6, 6, 18)]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_sweep_with_xray_sources_vs_reference(pkg, tables):
    m, a = load_case("sweep32_xray")
    n = m["n"]
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
    b.set_sources(m["srcpos"], m["normflux"])
    b.set_xray(a["xray_thick"], a["xray_thin"], m["normflux_xray"])
    b.set_rank(0, 1); b.load(ndens=nd, xh=xh); b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert nbox == m["sum_nbox"]
    assert abs(loss - m["photon_loss"]) <= tol("loss") * abs(m["photon_loss"])
    # the tolerance weight of the rates from the oracle (the checker), with both source types
    o = oracle_for(m, tables, n); o.enable_xray(a["xray_thick"], a["xray_thin"], m["normflux_xray"])
    _, _, _, phih_o, w = oracle_pass(o, nd, xh, m["srcpos"], m["normflux"])
    assert np.array_equal(phih_o, F(a["phih"]))                       # (the oracle IS the reference here)
    phih = b.fetch("phih_grid")
    assert_gamma(phih, F(a["phih"]), w, "X-ray sweep")
    # one source with an X-ray component through c2r_do_source: its column densities
    ns = m["ns_dump"]
    nb1, loss1, vis1, cd = b.do_source(ns, want_coldens=True)
    assert relerr(cd, F(a["coldensh_out"])) < tol("cd")
    # switched off again: the stellar-only rates of the same field (sweep32_bubbles is this field and list without column 5)
    b.set_xray(None); b.zero_rates()
    b.pass_sources()
    m0, a0 = load_case("sweep32_bubbles")
    o0 = oracle_for(m0, tables, n)
    _, _, _, p0, w0 = oracle_pass(o0, nd, xh, m0["srcpos"], m0["normflux"])
    assert_gamma(b.fetch("phih_grid"), p0, w0, "X-ray off again")
    b.close()


def test_evolve3d_with_xray_sources_vs_reference(pkg, tables):
    m, a = load_case("evolve32_xray")
    s, n = m["steps"]["step001"], m["n"]
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
    b.set_sources(s["srcpos"], s["normflux"])
    b.set_xray(a["xray_thick"], a["xray_thin"], s["normflux_xray"])
    b.load(ndens=F(a["step001_ndens"]), xh=F(a["step001_xh_before"]))
    rep = b.evolve3d_native(s["dt"])
    assert rep.niter == s["niter"] and list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
    assert rep.sum_nbox_all == s["sum_nbox_all"]
    assert abs(rep.photon_loss_all - s["photon_loss_all"]) <= tol("loss") * abs(s["photon_loss_all"])
    assert np.max(np.abs(b.fetch("xh") - F(a["step001_xh_after"]))) < tol("x")
    b.close()


def test_sweep_with_xray_heating_vs_reference(pkg, tables):
    """Non-isothermal context with the X-ray source type (EXT = 3 kernels): rates and heating rates against the reference
    rebuilt with both switches (fixture sweep32_xraythermal); c2r_set_xray_heat_tables is required before a pass."""
    from tests._util import load_thermal_tables, TOL, sweep_mode
    m, a = load_case("sweep32_xraythermal")
    n = m["n"]
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    tt = load_thermal_tables()
    o = oracle_for(m, tables, n)
    o.enable_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"], m["zred"], np.zeros((n ** 3, 3), dtype=np.float32))
    o.enable_xray(a["xray_thick"], a["xray_thin"], m["normflux_xray"]); o.enable_xray_heat(a["xray_heat_thick"], a["xray_heat_thin"])
    w, wh = o.enable_tolerance_weight(), o.enable_heat_tolerance_weight()
    o.pass_sources(nd, xh, np.zeros(n ** 3), m["srcpos"], m["normflux"])
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
    b.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
    b.set_sources(m["srcpos"], m["normflux"]); b.set_rank(0, 1)
    b.set_xray(a["xray_thick"], a["xray_thin"], m["normflux_xray"])
    b.load(ndens=nd, xh=xh, temperature_grid=np.full(n ** 3, 1e4, dtype=np.float32))
    b.begin_step(); b.zero_rates()
    with pytest.raises(pkg.C2RayHipError, match="c2r_set_xray_heat_tables"):
        b.pass_sources()
    b.set_xray_heat(a["xray_heat_thick"], a["xray_heat_thin"])
    loss, nbox, vis = b.pass_sources()
    assert nbox == m["sum_nbox"] and abs(loss - m["photon_loss"]) <= tol("loss") * abs(m["photon_loss"])
    assert_gamma(b.fetch("phih_grid"), F(a["phih"]), w, "X-ray + heating: Gamma")
    heat, ref = b.fetch("phiheat_grid"), F(a["phiheat"])
    t = TOL[sweep_mode()]
    assert np.array_equal(heat == 0, ref == 0)
    assert (np.abs(heat - ref) - (t["gamma_rtol"] * ref + t["gamma_wtol"] * wh)).max() <= 0
    b.close()


def test_xray_setters_check_their_arguments(pkg, tables):
    import ctypes as C
    lib = pkg.load_library()
    p = pkg.default_params(16)
    ctx = C.c_void_p()
    assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
    t = np.ones(2001)
    assert lib.c2r_set_xray_tables(ctx, t.ctypes.data, t.ctypes.data, 17) != 0           # wrong length
    assert lib.c2r_set_xray_tables(ctx, t.ctypes.data, None, 2001) != 0                  # one table only
    assert lib.c2r_set_xray_tables(ctx, t.ctypes.data, t.ctypes.data, 2001) == 0
    pos = np.array([[3, 3, 3], [8, 8, 8]], dtype=np.int32); nf = np.array([1e7, 1e7])
    assert lib.c2r_set_sources(ctx, pos.ctypes.data, nf.ctypes.data, 2) == 0
    assert lib.c2r_set_xray_sources(ctx, nf.ctypes.data, 3) != 0                         # one value per source
    assert lib.c2r_set_xray_sources(ctx, nf.ctypes.data, 2) == 0
    lib.c2r_destroy(ctx)


def test_fortran_drop_in_with_use_xray_sed(tables):
    """The xray build of the fixture driver linked with the shim (evolve_hip.F90 forwards xray_photo_*_table and NormFlux_xray
    when use_xray_SED is set): do_source per source against the fixture of the same driver with the reference's modules."""
    ex = os.path.join(gi.REF, "N32_xray", "hip", "ref_driver_hip")
    if not os.path.exists(ex):
        pytest.skip("ref_driver_hip of the xray variant not built (oracle/ref_build.sh 32:xray)")
    m = json.load(open(os.path.join(GOLDEN, "sweep32_xray.json")))
    a = np.load(os.path.join(GOLDEN, "sweep32_xray.npz"))
    d = tempfile.mkdtemp(prefix="c2r_xray_")
    try:
        def w(p):
            with open(p, "wb") as f:
                a["xray_thick"].tofile(f); a["xray_thin"].tofile(f)
        gi.run_driver(32, XSRC, {"mode": "'sweep'", "ns_dump": m["ns_dump"], "xray_tables": "'xray.f64'"}, dens=gi.density_factor(32, 5),
                      xfield=gi.bubble_xfield(32, BUBBLES, 7.0), variant="xray", extra_files={"xray.f64": w}, hip=True, d=d)
        kv = gi.read_kv(d + "/dump/step001_sweep.txt")
        assert kv["sum_nbox"] == m["sum_nbox"]
        assert abs(kv["photon_loss"] - m["photon_loss"]) <= 1e-10 * abs(m["photon_loss"])
        assert relerr(gi.rd(d, "step001_coldensh_out.f64", 32), a["coldensh_out"]) < 1e-11
        assert relerr(gi.rd(d, "step001_phih_grid.f64", 32), a["phih"]) < 1e-9
    finally:
        shutil.rmtree(d, ignore_errors=True)
