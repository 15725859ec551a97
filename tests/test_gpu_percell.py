"""GPU tests of the reference's per-cell call surface through the C ABI (include/c2ray_hip.h): c2r_evolve0d_host =
evolve0D(dt,rtpos,ns,niter) (evolve_point.F90:83-299) and c2r_global_pass_cell_host = evolve0D_global(dt,pos,conv_flag)
(:305-406), driven the way the reference's own sweep routines drive them -- cell by cell, in a causal order, on the
caller's arrays -- against the batch path (c2r_do_source, c2r_global_pass_host) and the oracle."""
import ctypes as C
import numpy as np
import pytest
from tests._util import F, load_case, oracle_for, expand, tol

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def _i3(v):
    return np.ascontiguousarray(v, dtype=np.int32)


@pytest.mark.parametrize("lls,thermal", [(2, False), (3, False), (1, True)])
def test_evolve0d_variants_equal_do_source(pkg, tables, lls, thermal):
    """The per-cell entry with the non-default physics switches -- a position-dependent LLS column (type_of_LLS = 2), the hard
    barrier (3), a non-isothermal context (heating rates into phiheat_grid) -- against the batch path on a 24^3 mesh, first
    sub-box: column densities (bit for bit in the exact mode), rates and heating rates."""
    from tests._util import load_thermal_tables
    rng = np.random.default_rng(40 + lls)
    n = 24
    tp = pkg.TestProblem(32); s = tp.step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(n ** 3))).astype(np.float32)
    xh = np.clip(0.9995 * (1.0 - 1e-3 * rng.random(n ** 3)), 1e-6, 1 - 1e-9)
    pos = np.array([[7, 9, 11]], dtype=np.int32); nf = np.array([3e8])
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    if lls == 2:
        b.set_lls(2, (s["coldensh_LLS"] * np.exp(rng.standard_normal(n ** 3))).astype(np.float32), 0.0)
    elif lls == 3:
        b.set_lls(3, None, 6.0 * float(np.ravel(s["dr1"])[0]))
    if thermal:
        tt = load_thermal_tables()
        b.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
        b.set_redshift(9.0)
    b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh)
    if thermal:
        b.load(temperature_grid=np.full(3 * n ** 3, 1e4, dtype=np.float32))
    b.begin_step(); b.zero_rates()
    nbox_ref, loss_ref, vis_ref, cd_ref = b.do_source(1, want_coldens=True)
    phih_ref = b.fetch("phih_grid")
    heat_ref = b.fetch("phiheat_grid") if thermal else None
    cd = np.zeros(n ** 3); phih = np.zeros(n ** 3); heat = np.zeros(n ** 3)
    src = pos[0].astype(np.int64)
    hl, hr = n // 2, n // 2 - 1 + n % 2
    for nbox in range(1, nbox_ref + 1):
        ext_l, ext_r = min(5 * nbox, hl), min(5 * nbox, hr)
        last_l, last_r = _i3(src - ext_l), _i3(src + ext_r)
        loss = C.c_double(0.0)
        cells = sorted((max(abs(i), abs(j), abs(k)), k, j, i) for k in range(-ext_l, ext_r + 1) for j in range(-ext_l, ext_r + 1)
                       for i in range(-ext_l, ext_r + 1))
        for q, k, j, i in cells:
            rt = _i3(src + np.array([i, j, k]))
            rc = b.lib.c2r_evolve0d_host(b.ctx, 1, rt.ctypes.data, last_l.ctypes.data, last_r.ctypes.data, nd.ctypes.data, xh.ctypes.data,
                                         cd.ctypes.data, phih.ctypes.data, heat.ctypes.data if thermal else None, C.byref(loss))
            assert rc == 0, b.lib.c2r_last_error(b.ctx)
    assert np.array_equal(cd != 0, cd_ref != 0)
    assert np.max(np.abs(cd - cd_ref) / np.maximum(cd_ref, 1e-300)) < tol("cd")
    nz = phih_ref != 0
    assert np.array_equal(phih != 0, nz) and np.max(np.abs(phih[nz] / phih_ref[nz] - 1)) < 1e-9
    assert abs(loss.value - loss_ref) <= tol("loss") * abs(loss_ref) + 1e-300
    if thermal:
        hz = heat_ref != 0
        assert np.array_equal(heat != 0, hz) and np.max(np.abs(heat[hz] / heat_ref[hz] - 1)) < 1e-9
    b.close()


def test_evolve0d_cell_by_cell_equals_do_source(pkg, tables):
    """One source of the 32^3 fixture traced by calling evolve0D for every cell of sub-boxes 1 and 2 in shell order (any order
    that visits a cell after its upstream neighbours is valid: evolve_source.F90:227-591), the sub-box limits moving as
    do_source moves them (:128-136): coldensh_out equal to the batch path's (bit for bit in the exact mode -- and to the
    oracle's, i.e. the Fortran's), the rates and the photon loss through the sub-box surface equal to the stated tolerances."""
    m, a = load_case("sweep32_bubbles")
    n = m["n"]
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    pos, nf = np.asarray(m["srcpos"], dtype=np.int32), np.asarray(m["normflux"], dtype=np.float64)
    ns = 1
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
    b.set_sources(pos, nf)
    b.set_rank(0, 1)
    b.load(ndens=nd, xh=xh)
    b.begin_step()
    b.zero_rates()
    nbox_ref, loss_ref, vis_ref, cd_ref = b.do_source(ns, want_coldens=True)
    phih_ref = b.fetch("phih_grid")
    assert nbox_ref >= 2
    # the same source, cell by cell
    cd = np.zeros(n ** 3); phih = np.zeros(n ** 3)
    src = pos[ns - 1].astype(np.int64)
    hl, hr = n // 2, n // 2 - 1 + n % 2
    losses = []
    for nbox in range(1, nbox_ref + 1):
        ext_l, ext_r = min(5 * nbox, hl), min(5 * nbox, hr)
        last_l, last_r = _i3(src - ext_l), _i3(src + ext_r)
        loss = C.c_double(0.0)
        cells = [(max(abs(i), abs(j), abs(k)), k, j, i) for k in range(-ext_l, ext_r + 1) for j in range(-ext_l, ext_r + 1)
                 for i in range(-ext_l, ext_r + 1)]
        cells.sort()
        for q, k, j, i in cells:
            rt = _i3(src + np.array([i, j, k]))
            rc = b.lib.c2r_evolve0d_host(b.ctx, ns, rt.ctypes.data, last_l.ctypes.data, last_r.ctypes.data, nd.ctypes.data,
                                         xh.ctypes.data, cd.ctypes.data, phih.ctypes.data, None, C.byref(loss))
            assert rc == 0, b.lib.c2r_last_error(b.ctx)
        losses.append(loss.value)
    assert np.array_equal(cd != 0, cd_ref != 0)
    assert np.max(np.abs(cd - cd_ref) / np.maximum(cd_ref, 1e-300)) < tol("cd")
    nz = phih_ref != 0
    assert np.array_equal(phih != 0, nz)
    assert np.max(np.abs(phih[nz] / phih_ref[nz] - 1)) < 1e-9
    assert abs(losses[-1] - loss_ref) <= tol("loss") * abs(loss_ref) + 1e-300
    # ... and the oracle, i.e. the Fortran's coldensh_out, BIT FOR BIT in either sweep mode of the context: the per-cell entry
    # always takes cinterp in the reference's operation order
    o = oracle_for(m, tables, n)
    onb, oloss, ovis, ocd = o.do_source(nd, xh, np.zeros(o.ncell), pos[ns - 1], nf[ns - 1])
    assert onb == nbox_ref and np.array_equal(cd, ocd)
    assert abs(losses[-1] - oloss) <= tol("loss") * abs(oloss) + 1e-300
    b.close()


def test_evolve0d_global_cell_by_cell_equals_the_global_pass(pkg, tables):
    """evolve0D_global for every cell of a 12^3 mesh, one call each, against global_pass over the mesh: xh_av, xh_intermed and
    the count of non-converged cells, bit for bit."""
    rng = np.random.default_rng(12)
    n = 12
    tp = pkg.TestProblem(32)
    s = tp.step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(n ** 3))).astype(np.float32)
    xh = np.clip(10.0 ** rng.uniform(-4, 0, n ** 3), 1e-6, 0.9999)
    phih = 10.0 ** rng.uniform(-16, -11, n ** 3) * (rng.random(n ** 3) < 0.6)
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    xav_ref, xint_ref = xh.copy(), xh.copy()
    conv_ref = C.c_int64(0)
    rc = b.lib.c2r_global_pass_host(b.ctx, s["dt"], nd.ctypes.data, xh.ctypes.data, xav_ref.ctypes.data, xint_ref.ctypes.data,
                                    phih.ctypes.data, C.byref(conv_ref))
    assert rc == 0
    xav, xint = xh.copy(), xh.copy()
    conv = C.c_int32(0)
    for k in range(1, n + 1):
        for j in range(1, n + 1):
            for i in range(1, n + 1):
                p3 = _i3([i, j, k])
                rc = b.lib.c2r_global_pass_cell_host(b.ctx, s["dt"], p3.ctypes.data, nd.ctypes.data, xh.ctypes.data, xav.ctypes.data,
                                                     xint.ctypes.data, phih.ctypes.data, None, None, C.byref(conv))
                assert rc == 0, b.lib.c2r_last_error(b.ctx)
    assert conv.value == conv_ref.value and conv.value > 0
    assert np.array_equal(xav, xav_ref) and np.array_equal(xint, xint_ref)
    b.close()
