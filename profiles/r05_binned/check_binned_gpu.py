"""The source-binned far shells (csrc/kernels_binned.hpp): one launch per axis over the MESH PLANES -- a thread owns
mesh cells and loops over the sources whose face lies on its plane, the rates summed on chip and added without an atomic --
against the source-major launch (k_sweep_shell_fast) on the same inputs: sub-box counts and photon losses equal bit for bit (they
come out of the shell planes, which both paths must fill with the same bits), rates equal to the order of the adds; against the
oracle; with heating rates and X-ray sources; and as the engine of ordered (bit-reproducible) rates, where the ordered sum over
per-source grids covers only the shells that did not run binned.  C2R_BINNED / _QMIN / _COVER / _WX are read by c2r_create.
Reference: column_density.f90:108-140, evolve_point.F90:137-146, 283-286."""
import numpy as np
import pytest
from tests._util import F, oracle_for, assert_gamma, oracle_pass, tol, load_thermal_tables

pytestmark = [pytest.mark.gpu]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def case(pkg, n, nsrc, seed, mesh=None, x_mode="ionized"):
    rng = np.random.default_rng(seed)
    mesh = mesh or (n, n, n)
    tp = pkg.TestProblem(n)
    s = tp.step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(mesh) - 0.125)).astype(np.float32)
    if x_mode == "ionized":
        xh = 0.9995 * (1.0 - 1e-3 * rng.random(mesh))
    else:
        xh = np.clip(10.0 ** rng.uniform(-3.0, 0.0, mesh), 1e-6, 0.9995)
    pos = np.stack([rng.integers(1, m + 1, nsrc) for m in mesh], axis=1).astype(np.int32)
    nf = 10.0 ** rng.uniform(6.0, 9.0, nsrc)
    return s, F(nd), F(xh), pos, nf


def run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, binned, wx=64, qmin=2, det=False, thermal=None, xray=None,
             cover="0", scratch=0):
    monkeypatch.setenv("C2R_BINNED", "1" if binned else "0")
    monkeypatch.setenv("C2R_BINNED_QMIN", str(qmin))
    monkeypatch.setenv("C2R_BINNED_COVER", cover)
    monkeypatch.setenv("C2R_BINNED_WX", str(wx))
    b = pkg.HipBackend(mesh, *tables, device=0, fast=True, deterministic=det, scratch_bytes=scratch)
    b.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
    if thermal is not None:
        b.set_thermal(thermal["heat_thick"], thermal["heat_thin"], thermal["cool_logT"], thermal["cool_logL"])
    b.set_sources(pos, nf)
    b.set_rank(0, 1)
    if xray is not None:
        b.set_xray(xray["thick"], xray["thin"], xray["nfx"])
        if thermal is not None:
            b.set_xray_heat(xray["heat_thick"], xray["heat_thin"])
    b.load(ndens=nd, xh=xh)
    b.begin_step()
    b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    out = dict(loss=loss, nbox=nbox, vis=vis, per_src=b.last_nbox().copy(), phih=b.fetch("phih_grid"),
               heat=b.fetch("phiheat_grid") if thermal is not None else None, info=b.info())
    b.close()
    return out


def launches(info):
    return int(info.split("source-binned launches ")[1].split(";")[0])


def same_but_for_the_order_of_the_adds(a, b, key="phih", rel=5e-14):
    assert np.array_equal(a[key] == 0, b[key] == 0)
    assert np.max(np.abs(a[key] - b[key]) / np.maximum(np.abs(a[key]), 1e-300)) < rel


@pytest.mark.parametrize("wx,mesh,nsrc,x_mode", [(64, (48, 48, 48), 300, "ionized"), (16, (33, 40, 27), 200, "ionized"),
                                                 (32, (40, 40, 40), 150, "bubbles"), (64, (33, 40, 27), 200, "bubbles")])
def test_binned_equals_source_major(pkg, tables, monkeypatch, wx, mesh, nsrc, x_mode):
    s, nd, xh, pos, nf = case(pkg, mesh[0], nsrc, 7 + wx + nsrc, mesh, x_mode)
    a = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, False)
    b = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, wx)
    assert launches(a["info"]) == 0 and (launches(b["info"]) > 0 or x_mode == "bubbles")    # (neutral gas: most sources end in the fused first sub-box)
    assert (a["nbox"], a["vis"]) == (b["nbox"], b["vis"]) and np.array_equal(a["per_src"], b["per_src"])
    assert a["loss"] == b["loss"]                     # bit for bit: the planes the loss is read from are the same bits
    same_but_for_the_order_of_the_adds(a, b)


def test_binned_vs_oracle(pkg, tables, monkeypatch):
    monkeypatch.setenv("C2R_SWEEP_MODE", "1")
    mesh = (40, 40, 40)
    s, nd, xh, pos, nf = case(pkg, 40, 120, 99, mesh)
    o = oracle_for(s, tables, mesh)
    oloss, onb, ovis, phih_o, w = oracle_pass(o, nd, xh, pos, nf)
    r = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True)
    assert launches(r["info"]) > 0
    assert (r["nbox"], r["vis"]) == (onb, ovis)
    assert abs(r["loss"] - oloss) <= tol("loss") * abs(oloss)
    assert_gamma(r["phih"], phih_o, w)


def test_binned_with_heating_rates(pkg, tables, monkeypatch):
    """Non-isothermal passes: the binned launch sums the heating rates of a plane's sources in a second set of LDS slots (the
    source-major kernel pays a second atomic per visit there); default policy: binned where the coverage rule holds."""
    mesh = (48, 48, 48)
    s, nd, xh, pos, nf = case(pkg, 48, 260, 5, mesh)
    tt = load_thermal_tables()
    a = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, False, thermal=tt)
    b = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, thermal=tt)
    assert launches(b["info"]) > 0
    assert (a["nbox"], a["vis"], a["loss"]) == (b["nbox"], b["vis"], b["loss"])
    same_but_for_the_order_of_the_adds(a, b)
    assert (a["heat"] > 0).any()
    same_but_for_the_order_of_the_adds(a, b, "heat")
    # the library's own choice (C2R_BINNED unset): a non-isothermal context in the tolerance mode runs binned where it pays
    monkeypatch.delenv("C2R_BINNED", raising=False)
    monkeypatch.setenv("C2R_BINNED_COVER", "0.2")
    monkeypatch.setenv("C2R_BINNED_QMIN", "8")
    bk = pkg.HipBackend(mesh, *tables, device=0, fast=True)
    bk.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
    bk.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
    bk.set_sources(pos, nf); bk.set_rank(0, 1); bk.load(ndens=nd, xh=xh)
    bk.begin_step(); bk.zero_rates()
    loss, nbox, vis = bk.pass_sources()
    assert launches(bk.info()) > 0 and (nbox, vis, loss) == (a["nbox"], a["vis"], a["loss"])
    bk.close()


def test_binned_with_xray_sources_and_heating(pkg, tables, monkeypatch):
    from tests._util import load_case
    mesh = (40, 40, 40)
    s, nd, xh, pos, nf = case(pkg, 40, 160, 17, mesh)
    a = load_case("sweep32_xraythermal")[1]          # the X-ray tables of the reference fixture
    xr = dict(thick=a["xray_thick"], thin=a["xray_thin"], heat_thick=a["xray_heat_thick"], heat_thin=a["xray_heat_thin"])
    rng = np.random.default_rng(3)
    xr["nfx"] = nf * 10.0 ** rng.uniform(-2.0, 0.0, len(nf)) * (rng.random(len(nf)) < 0.7)
    tt = load_thermal_tables()
    for thermal in (None, tt):
        a = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, False, thermal=thermal, xray=xr)
        b = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, thermal=thermal, xray=xr)
        assert launches(b["info"]) > 0
        assert (a["nbox"], a["vis"], a["loss"]) == (b["nbox"], b["vis"], b["loss"])
        same_but_for_the_order_of_the_adds(a, b)
        if thermal is not None:
            same_but_for_the_order_of_the_adds(a, b, "heat")


def test_binned_as_the_engine_of_ordered_rates(pkg, tables, monkeypatch):
    """deterministic_rates = 1: shells that run binned add their rates in bin order (fixed), the others go through the per-source
    grids and k_gamma_reduce's source-ordered sum, which leaves the binned shells' cells out.  Two runs: the same bits.  Against
    ordered rates without the binned kernel: equal to the order of the adds (the ORDER differs: bin order for the binned shells,
    not evolve_point.F90:283's source order).  In batches (a scratch budget of ~70 sources): still reproducible."""
    mesh = (48, 48, 48)
    s, nd, xh, pos, nf = case(pkg, 48, 220, 23, mesh)
    plain = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, False, det=True)
    r1 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, det=True, qmin=12)
    r2 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, det=True, qmin=12)
    assert launches(r1["info"]) > 0
    assert (plain["nbox"], plain["vis"], plain["loss"]) == (r1["nbox"], r1["vis"], r1["loss"])
    assert np.array_equal(r1["phih"], r2["phih"])                      # bit-reproducible
    same_but_for_the_order_of_the_adds(plain, r1)
    atomics = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, False)
    same_but_for_the_order_of_the_adds(atomics, r1)
    per_src = 2 * 6 * (2 * 24 + 1) ** 2 * 8 + 2 * 48 ** 3 * 8
    b1 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, det=True, qmin=12, scratch=70 * per_src)
    b2 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, det=True, qmin=12, scratch=70 * per_src)
    assert np.array_equal(b1["phih"], b2["phih"]) and b1["loss"] == plain["loss"]
    same_but_for_the_order_of_the_adds(plain, b1)
    # ... and with heating rates
    tt = load_thermal_tables()
    h0 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, False, det=True, thermal=tt)
    h1 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, det=True, thermal=tt, qmin=12)
    h2 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True, det=True, thermal=tt, qmin=12)
    assert launches(h1["info"]) > 0
    assert np.array_equal(h1["phih"], h2["phih"]) and np.array_equal(h1["heat"], h2["heat"])
    same_but_for_the_order_of_the_adds(h0, h1); same_but_for_the_order_of_the_adds(h0, h1, "heat")


def test_whole_steps_with_binned_shells(pkg, tables, monkeypatch):
    """evolve3D from a pre-ionised start (the sources reach the far shells) with 150 sources: iteration count, non-converged-cell
    history and sub-box history equal, xh equal to the order of the adds."""
    n, S = 32, 150
    tp = pkg.TestProblem(n); s = tp.step(1)
    nd, xh = tp.fields(1)
    xh = np.full_like(xh, 0.999)
    pos, nf = pkg.seeded_sources(n, S, seed=3)
    reps = []
    for binned in (0, 1):
        monkeypatch.setenv("C2R_BINNED", str(binned))
        monkeypatch.setenv("C2R_BINNED_QMIN", "2"); monkeypatch.setenv("C2R_BINNED_COVER", "0")
        b = pkg.HipBackend(n, *tables, device=0, fast=True)
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
        b.set_sources(pos, nf); b.load(ndens=nd, xh=xh)
        rep = b.evolve3d_native(s["dt"])
        reps.append((rep.niter, list(rep.it_conv_flag[:rep.niter]), list(rep.it_sum_nbox[:rep.niter]), rep.photon_loss_all,
                     b.fetch("xh"), launches(b.info())))
        b.close()
    a, c = reps
    assert a[5] == 0 and c[5] > 0
    assert a[0] == c[0] and a[1] == c[1] and a[2] == c[2]
    assert abs(a[3] - c[3]) <= 1e-12 * abs(a[3])
    assert np.max(np.abs(a[4] - c[4])) < 1e-11
