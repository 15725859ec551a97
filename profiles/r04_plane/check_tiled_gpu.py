"""GPU tests of the mesh-tile far shells (k_sweep_tile_fast): the launch over tiles of the mesh planes against the
source-major launch on the same inputs (sub-box counts and photon losses equal bit for bit -- they come out of the shell
planes, which both paths must fill with the same bits --, rates equal to the order of the adds) and against the oracle.
c2r_create."""
import numpy as np
import pytest
from tests._util import F, oracle_for, assert_gamma, oracle_pass, tol

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


def run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, binned, lls=None, qmin=2):
    monkeypatch.setenv("C2R_TILED", "1" if binned else "0")
    monkeypatch.setenv("C2R_TILED_QMIN", str(qmin))
    monkeypatch.setenv("C2R_TILED_COVER", "0")
    b = pkg.HipBackend(mesh, *tables, device=0, fast=True)
    b.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
    if lls is not None:
        b.set_lls(*lls)
    b.set_sources(pos, nf)
    b.set_rank(0, 1)
    b.load(ndens=nd, xh=xh)
    b.begin_step()
    b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    per_src = b.last_nbox().copy()
    phih = b.fetch("phih_grid")
    b.close()
    return loss, nbox, vis, per_src, phih


@pytest.mark.parametrize("mesh,nsrc,x_mode", [((48, 48, 48), 300, "ionized"), ((33, 40, 27), 200, "ionized"),
                                              ((40, 40, 40), 150, "bubbles"), ((300, 20, 24), 400, "ionized"),
                                              ((20, 270, 26), 300, "ionized")])
def test_tiled_equals_source_major(pkg, tables, monkeypatch, mesh, nsrc, x_mode):
    """(the last two meshes are wider than a tile's 256 columns along one plane axis: two tiles per row, runs cut by the periodic seam)"""
    s, nd, xh, pos, nf = case(pkg, mesh[0], nsrc, 7 + nsrc, mesh, x_mode)
    l0, nb0, v0, ps0, g0 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, False)
    l1, nb1, v1, ps1, g1 = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True)
    assert (nb0, v0) == (nb1, v1) and np.array_equal(ps0, ps1)
    assert l0 == l1                                  # bit for bit: the planes the loss is read from are the same bits
    assert np.array_equal(g0 == 0, g1 == 0)
    assert np.max(np.abs(g1 - g0) / np.maximum(g0, 1e-300)) < 5e-14      # the order of the adds only


def test_tiled_vs_oracle(pkg, tables, monkeypatch):
    monkeypatch.setenv("C2R_SWEEP_MODE", "1")
    mesh = (40, 40, 40)
    s, nd, xh, pos, nf = case(pkg, 40, 120, 99, mesh)
    o = oracle_for(s, tables, mesh)
    oloss, onb, ovis, phih_o, w = oracle_pass(o, nd, xh, pos, nf)
    loss, nbox, vis, _, phih = run_pass(pkg, tables, s, mesh, nd, xh, pos, nf, monkeypatch, True)
    assert (nbox, vis) == (onb, ovis)
    assert abs(loss - oloss) <= tol("loss") * abs(oloss)
    assert_gamma(phih, phih_o, w)
