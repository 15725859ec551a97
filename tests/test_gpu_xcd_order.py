"""The XCD-aware, plane-ordered block mapping of the far shells (csrc/kernels_sweep.hpp k_sweep_shell_xcd): block b
works for XCD group b % 8 on the (b / 8)-th item of that group's list -- for every face the group's eighth of the batch's sources
sorted along the face's axis -- so that sources whose faces lie on one mesh plane run on one XCD at about the same time and share
the plane's n_HI in its L2.  WHICH workgroup does a (source, face, tile) changes, nothing else: sub-box counts, visited cells and
the photon loss are bit-identical to the plain (tile, face, source) grid, the rates equal to the order of the atomics; also with
sources that retire early (their blocks return at once), with a source count that is no multiple of eight, with zero-flux
sources, heating rates, X-ray sources and ordered (deterministic) rates -- those bit-identical.  C2R_XCD_ORDER / _QMIN / _MIN_ALIVE reach the library as c2r_set_option calls (tests/conftest.py)."""
import numpy as np
import pytest
from tests._util import F, oracle_for, assert_gamma, oracle_pass, tol, load_thermal_tables, load_case

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]     # once per sweep mode


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def case(pkg, n, nsrc, seed, mesh=None, x_mode="ionized", dark=0):
    rng = np.random.default_rng(seed)
    mesh = mesh or (n, n, n)
    tp = pkg.TestProblem(n)
    s = tp.step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(mesh) - 0.125)).astype(np.float32)
    if x_mode == "ionized":
        xh = 0.9995 * (1.0 - 1e-3 * rng.random(mesh))
    elif x_mode == "mixed":            # an ionised half and a neutral half: sources retire at very different sub-boxes
        xh = 0.9995 * (1.0 - 1e-3 * rng.random(mesh))
        xh[: mesh[0] // 2] = 10.0 ** rng.uniform(-3.0, -1.0, xh[: mesh[0] // 2].shape)
    else:
        xh = np.clip(10.0 ** rng.uniform(-3.0, 0.0, mesh), 1e-6, 0.9995)
    pos = np.stack([rng.integers(1, m + 1, nsrc) for m in mesh], axis=1).astype(np.int32)
    nf = 10.0 ** rng.uniform(6.0, 9.0, nsrc)
    if dark:
        nf[rng.choice(nsrc, dark, replace=False)] = 0.0          # sources that are never traced
    return s, F(nd), F(xh), pos, nf


def run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, order, qmin=2, alive="0", thermal=None, xray=None, det=False):
    monkeypatch.setenv("C2R_CHAINS", "1")
    monkeypatch.setenv("C2R_XCD_ORDER", str(order))
    monkeypatch.setenv("C2R_XCD_QMIN", str(qmin))
    monkeypatch.setenv("C2R_XCD_MIN_ALIVE", alive)
    b = pkg.HipBackend(mesh, *tables, device=0, deterministic=det)
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
    return int(info.split("plane-ordered launches ")[1].split(";")[0])


def same_to_the_order_of_the_atomics(a, b, key="phih"):
    assert np.array_equal(a[key] == 0, b[key] == 0)
    assert np.max(np.abs(a[key] - b[key]) / np.maximum(np.abs(a[key]), 1e-300)) < 1e-13


def same_integers_and_loss(a, b):
    assert (a["nbox"], a["vis"]) == (b["nbox"], b["vis"]) and np.array_equal(a["per_src"], b["per_src"])
    assert a["loss"] == b["loss"]          # bit for bit: same launches' loss partials, summed in the same shape


@pytest.mark.parametrize("mesh,nsrc,x_mode,dark", [((48, 48, 48), 301, "ionized", 0), ((33, 40, 27), 203, "ionized", 9),
                                                   ((40, 40, 40), 150, "mixed", 0), ((64, 64, 64), 97, "mixed", 5)])
def test_plane_ordered_equals_plain_grid(pkg, tables, monkeypatch, mesh, nsrc, x_mode, dark):
    s, nd, xh, pos, nf = case(pkg, mesh[0], nsrc, 11 + nsrc, mesh, x_mode, dark)
    a = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 0)
    b = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 1)
    assert launches(a["info"]) == 0 and launches(b["info"]) > 0
    if x_mode == "mixed":
        assert len(set(int(v) for v in a["per_src"])) > 1          # sources retire at different sub-boxes
    same_integers_and_loss(a, b)
    same_to_the_order_of_the_atomics(a, b)


def test_plane_ordered_only_while_most_sources_are_alive(pkg, tables, monkeypatch):
    """The library's own rule: the plane-ordered mapping while >= 90 % of the batch's sources are still traced (its lists are made
    once per batch: retired sources' blocks are idle), the compact active list after that -- and not at all below 1.5 sources
    per mesh plane."""
    mesh = (40, 40, 40)
    s, nd, xh, pos, nf = case(pkg, 40, 150, 161, mesh, "mixed")
    a = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 0)
    forced = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 1, alive="0")
    ruled = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 1, alive="0.9")
    assert 0 <= launches(ruled["info"]) < launches(forced["info"])
    same_integers_and_loss(a, ruled); same_to_the_order_of_the_atomics(a, ruled)
    monkeypatch.delenv("C2R_XCD_ORDER")
    mesh = (48, 48, 48)
    few = case(pkg, 48, 70, 3, mesh)                                # 70 sources on 48 planes: below the rule (72)
    monkeypatch.setenv("C2R_CHAINS", "1")
    bk = pkg.HipBackend(mesh, *tables, device=0)
    bk.set_step((few[0]["dr1"], few[0]["dr2"], few[0]["dr3"]), few[0]["vol"], few[0]["coldensh_LLS"], few[0]["clumping"])
    bk.set_sources(few[3], few[4]); bk.set_rank(0, 1); bk.load(ndens=few[1], xh=few[2]); bk.begin_step(); bk.zero_rates()
    bk.pass_sources()
    assert launches(bk.info()) == 0
    bk.close()


def test_plane_ordered_vs_oracle(pkg, tables, monkeypatch):
    mesh = (40, 40, 40)
    s, nd, xh, pos, nf = case(pkg, 40, 120, 99, mesh)
    o = oracle_for(s, tables, mesh)
    oloss, onb, ovis, phih_o, w = oracle_pass(o, nd, xh, pos, nf)
    r = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 1)
    assert launches(r["info"]) > 0
    assert (r["nbox"], r["vis"]) == (onb, ovis)
    assert abs(r["loss"] - oloss) <= tol("loss") * abs(oloss)
    assert_gamma(r["phih"], phih_o, w)


def test_plane_ordered_with_heating_rates_and_xray_sources(pkg, tables, monkeypatch):
    mesh = (40, 40, 40)
    s, nd, xh, pos, nf = case(pkg, 40, 160, 17, mesh)
    fx = load_case("sweep32_xraythermal")[1]          # the X-ray tables of the reference fixture
    rng = np.random.default_rng(3)
    xr = dict(thick=fx["xray_thick"], thin=fx["xray_thin"], heat_thick=fx["xray_heat_thick"], heat_thin=fx["xray_heat_thin"],
              nfx=nf * 10.0 ** rng.uniform(-2.0, 0.0, len(nf)) * (rng.random(len(nf)) < 0.7))
    tt = load_thermal_tables()
    for thermal, xray in ((tt, None), (None, xr), (tt, xr)):
        a = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 0, thermal=thermal, xray=xray)
        b = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 1, thermal=thermal, xray=xray)
        assert launches(b["info"]) > 0
        same_integers_and_loss(a, b); same_to_the_order_of_the_atomics(a, b)
        if thermal is not None:
            assert (a["heat"] > 0).any()
            same_to_the_order_of_the_atomics(a, b, "heat")


def test_plane_ordered_with_ordered_rates(pkg, tables, monkeypatch):
    """deterministic_rates = 1: the per-source grids are written by whichever workgroup the mapping gives a tile to and summed
    in source order afterwards -- the rates are the SAME BITS with and without the mapping (and with heating rates)."""
    mesh = (40, 40, 40)
    s, nd, xh, pos, nf = case(pkg, 40, 170, 29, mesh, "mixed", 4)
    tt = load_thermal_tables()
    for thermal in (None, tt):
        a = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 0, det=True, thermal=thermal)
        b = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, 1, det=True, thermal=thermal)
        assert launches(a["info"]) == 0 and launches(b["info"]) > 0
        same_integers_and_loss(a, b)
        assert np.array_equal(a["phih"], b["phih"])
        if thermal is not None:
            assert np.array_equal(a["heat"], b["heat"])


def test_whole_steps_plane_ordered(pkg, tables, monkeypatch):
    """evolve3D from a pre-ionised start with 150 sources: iteration count, non-converged-cell history and sub-box history equal."""
    n, S = 32, 150
    tp = pkg.TestProblem(n); s = tp.step(1)
    nd, xh = tp.fields(1)
    xh = np.full_like(xh, 0.999)
    pos, nf = pkg.seeded_sources(n, S, seed=3)
    reps = []
    for order in (0, 1):
        monkeypatch.setenv("C2R_CHAINS", "1")
        monkeypatch.setenv("C2R_XCD_ORDER", str(order)); monkeypatch.setenv("C2R_XCD_QMIN", "2")
        b = pkg.HipBackend(n, *tables, device=0)
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
