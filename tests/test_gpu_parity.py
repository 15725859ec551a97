"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
  (a) the golden fixtures recorded from the compiled reference, and
  (b) the oracle on the same seeded inputs.
Every test runs once per sweep mode (c2r_params.sweep_mode: exact / fast; fixture `sweep_mode`).
Tolerances (tests/_util.TOL, per mode): integer results (nbox, visited, conv_flag, niter) exact;
  coldensh_out  rel 1e-11,  photon loss  rel 1e-10,  xh  abs 1e-9  (north_star asks for 1e-5),
  Gamma:  |dGamma| <= rtol Gamma + wtol W  with W the oracle's per-cell tolerance weight (oracle_cfg.tolw):
          rtol 1e-13 / 1e-12 (exact / fast), wtol 2e-14."""
import numpy as np
import pytest
from tests._util import F, load_case, oracle_for, expand, relerr, tol, assert_gamma, oracle_pass, oracle_step, STATE_RTOL

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def make_backend(pkg, tables, m, n, nd, xh, **kw):
    b = pkg.HipBackend(n, *tables, device=0, **kw)
    b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
    b.set_sources(m["srcpos"], m["normflux"])
    b.set_rank(0, 1)
    b.load(ndens=nd, xh=xh)
    return b



@pytest.mark.parametrize("name", ["sweep32_std_x999", "sweep33_std_x999", "sweep32_bubbles"])
def test_sweep_vs_reference_fixture(pkg, tables, name):
    m, a = load_case(name)
    n = m["n"]
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    b = make_backend(pkg, tables, m, n, nd, xh)
    b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert nbox == m["sum_nbox"]
    assert abs(loss - m["photon_loss"]) <= tol("loss") * abs(m["photon_loss"])
    phih = b.fetch("phih_grid")
    ref = F(a["phih"])
    assert np.count_nonzero(phih) == m["phih_nonzero"]
    oloss, onb, ovis, ophih, w = oracle_pass(oracle_for(m, tables, n), nd, xh, m["srcpos"], m["normflux"])
    assert np.array_equal(ophih, ref)             # the oracle is pinned to the fixture (tests/test_oracle.py): W is the fixture's
    assert_gamma(phih, ref, w, name)
    # one source alone: its full coldensh_out grid
    ns = m["ns_dump"]
    b.zero_rates()
    nb1, l1, v1, cd = b.do_source(ns, want_coldens=True)
    cref = F(a["coldensh_out"])
    assert np.array_equal(cd == 0, cref == 0)
    assert relerr(cd, cref) < tol("cd")
    b.close()


def test_sweep64_planes_vs_reference_fixture(pkg, tables):
    m, a = load_case("sweep64_bubbles")
    n = m["n"]
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    b = make_backend(pkg, tables, m, n, nd, xh)
    b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert nbox == m["sum_nbox"]
    assert abs(loss - m["photon_loss"]) <= tol("loss") * abs(m["photon_loss"])
    p3 = b.fetch("phih_grid").reshape((n, n, n), order="F")
    s = [(p - 1) % n for p in m["srcpos"][m["ns_dump"] - 1]]
    w3 = oracle_pass(oracle_for(m, tables, n), nd, xh, m["srcpos"], m["normflux"])[4].reshape((n, n, n), order="F")
    assert_gamma(p3[s[0]], a["phih_px"], w3[s[0]], "px")
    assert_gamma(p3[:, s[1]], a["phih_py"], w3[:, s[1]], "py")
    assert_gamma(p3[:, :, s[2]], a["phih_pz"], w3[:, :, s[2]], "pz")
    assert np.count_nonzero(p3) == m["phih_nonzero"]
    assert abs(float(np.sum(p3, dtype=np.longdouble)) / m["phih_sum"] - 1) < 1e-10
    b.close()


@pytest.mark.parametrize("name,native", [("evolve32_onesrc", True), ("evolve32_onesrc", False),
                                         ("evolve32_std_bubbles", True), ("evolve32_std_bubbles", False),
                                         ("evolve64_std_bubbles", True)])
def test_evolve3d_vs_reference_fixture(pkg, tables, name, native):
    """Whole time steps: same outer-iteration count, same non-converged-cell sequence, xh within
    tol("x") of the Fortran.  native=True runs the loop inside the C ABI (c2r_evolve3d_dev, what the
    Fortran shim calls) through the backend, native=False through the Python host (Evolve.evolve3D: since round 6 a thin host
    of the SAME C loop -- its report as a dict, dumps from the iteration hook, restarts through start_from_dump)."""
    m, a = load_case(name)
    n = m["n"]
    for tag, s in m["steps"].items():
        xh0 = F(a[tag + "_xh_before"]); nd = F(a[tag + "_ndens"])
        b = make_backend(pkg, tables, s, n, nd, xh0)
        if native:
            rep = b.evolve3d_native(s["dt"])
            niter, conv_seq = rep.niter, list(rep.it_conv_flag[:rep.niter])
            nbox_all, loss_all, converged = rep.sum_nbox_all, rep.photon_loss_all, rep.converged
            rel = np.array([rep.it_rel_change_xh1[:niter], rep.it_rel_change_xh0[:niter]]).T
            phot = {k: getattr(rep, k) for k in ("totrec", "totcollisions", "dh0", "total_ion")}
        else:
            ev = pkg.Evolve(b)
            r = ev.evolve3D(0.0, s["dt"], 0)
            niter, conv_seq = r["niter"], [e["conv_flag"] for e in r["log"]]
            nbox_all, loss_all, converged = r["sum_nbox_all"], r["photon_loss_all"], r["converged"]
            rel = np.array([[e["rel_change_xh1"], e["rel_change_xh0"]] for e in r["log"]])
            phot = r["photon_statistics"]
        assert converged
        assert niter == s["niter"], (tag, niter, s["niter"])
        assert conv_seq == s["log"]["nonconv"]
        assert nbox_all == s["sum_nbox_all"]
        assert abs(loss_all - s["photon_loss_all"]) <= tol("loss") * abs(s["photon_loss_all"]) + 1e-300
        assert relerr(rel, np.array(s["log"]["test2"][1:])) < 1e-7
        # photon statistics of the step (photonstatistics.F90 module variables after evolve3D);
        # dh0 is a difference of two ~1e70 sums, so its error is absolute at the 1e-13 * h0 level
        for k in ("totrec", "totcollisions", "total_ion"):
            assert abs(phot[k] / s[k] - 1) < 1e-9, (k, phot[k], s[k])
        assert abs(phot["dh0"] - s["dh0"]) < 1e-9 * abs(s["total_ion"])
        xh = b.fetch("xh")
        assert np.max(np.abs(xh - F(a[tag + "_xh_after"]))) < tol("x")
        if tag + "_phih_grid" in a:
            # W of the step's last pass from the (pinned) oracle's own run of the step
            orep, oxh, oxav, ophih, w = oracle_step(oracle_for(s, tables, n), s["dt"], nd, xh0, s["srcpos"], s["normflux"])
            assert np.array_equal(ophih, F(a[tag + "_phih_grid"]))
            assert_gamma(b.fetch("phih_grid"), ophih, w, tag, state_rtol=STATE_RTOL)
            assert np.max(np.abs(b.fetch("xh_av") - F(a[tag + "_xh_av"]))) < tol("x")
        b.close()


def _random_case(n, nsrc, seed, pkg):
    rng = np.random.default_rng(seed)
    tp = pkg.TestProblem(n)
    s = tp.step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(n ** 3))).astype(np.float32)
    xh = np.clip(10.0 ** rng.uniform(-4, 0, n ** 3) * 0.9999, 1e-6, 0.9999)
    # smooth the ionized field a little: blocks of 4^3 share a value, like bubbles
    x3 = xh.reshape((n, n, n), order="F")
    c = max(1, n // 8)
    x3[:] = np.repeat(np.repeat(np.repeat(x3[::c, ::c, ::c], c, 0), c, 1), c, 2)[:n, :n, :n]
    pos, nf = pkg.seeded_sources(n, nsrc, seed=seed)
    return s, nd, F(x3), pos, nf


@pytest.mark.parametrize("n,nsrc,seed", [(24, 7, 1), (40, 20, 2), (48, 33, 3), (21, 5, 4), (40, 64, 5), (40, 65, 6)])   # 32 | 33: graph + look-ahead pairs or not; 64 | 65: one-wave decision kernel or k_box_decide
def test_pass_and_global_vs_oracle_seeded(pkg, tables, n, nsrc, seed):
    """Seeded random density / ionization / sources at meshes the fixtures do not cover
    (odd and even, sources anywhere incl. next to the periodic seam)."""
    s, nd, xh, pos, nf = _random_case(n, nsrc, seed, pkg)
    o = oracle_for(s, tables, n)
    oloss, onb, ovis, phih_o, w = oracle_pass(o, nd, xh, pos, nf)
    b = make_backend(pkg, tables, dict(s, srcpos=pos, normflux=nf), n, nd, xh)
    b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert (nbox, vis) == (onb, ovis)
    assert abs(loss - oloss) <= tol("loss") * abs(oloss) + 1e-300
    phih = b.fetch("phih_grid")
    assert_gamma(phih, phih_o, w)
    xav, xint = xh.copy(), xh.copy()
    oconv = o.global_pass(s["dt"], nd, xh, xav, xint, phih_o)
    conv, sum1 = b.global_pass(s["dt"])
    assert conv == oconv
    assert np.max(np.abs(b.fetch("xh_intermed") - xint)) < tol("x")
    assert np.max(np.abs(b.fetch("xh_av") - xav)) < tol("x")
    assert abs(sum1 - o.sum(xint)) < 1e-9 * n ** 3
    b.close()


def test_small_scratch_batches_equal_one_batch(pkg, tables):
    """Sources processed in several batches (scratch cap) give the same rates as one batch."""
    n, nsrc = 32, 12
    s, nd, xh, pos, nf = _random_case(n, nsrc, 9, pkg)
    res = []
    for cap in (0, 1):      # 0 = everything at once; 1 byte = one source per batch
        b = make_backend(pkg, tables, dict(s, srcpos=pos, normflux=nf), n, nd, xh, scratch_bytes=cap)
        b.begin_step(); b.zero_rates()
        out = b.pass_sources()
        res.append((out, b.fetch("phih_grid")))
        b.close()
    assert res[0][0][1:] == res[1][0][1:]
    assert abs(res[0][0][0] - res[1][0][0]) <= 1e-13 * abs(res[0][0][0])
    assert relerr(res[0][1], res[1][1], floor=1e-60) < 1e-12


def test_fused_first_subboxes_equal_per_shell_launches(pkg, tables, monkeypatch):
    """k_sweep_box_fused (sub-boxes ending at q <= 10: one launch per sub-box, one workgroup per source)
    runs the same per-cell code as the per-shell launches: identical column densities and sub-box
    counts, rates equal to rounding (atomics), the loss equal up to its summation order."""
    n, nsrc = 48, 9
    s, nd, xh, pos, nf = _random_case(n, nsrc, 21, pkg)
    res = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("C2R_FUSE_SMALL", fuse)          # -> c2r_set_option "fuse_small" (tests/conftest.py)
        b = make_backend(pkg, tables, dict(s, srcpos=pos, normflux=nf), n, nd, xh)
        b.begin_step(); b.zero_rates()
        out = b.pass_sources()
        cds = [b.do_source(k + 1, want_coldens=True) for k in range(nsrc)]
        res.append((out, b.fetch("phih_grid"), cds))
        b.close()
    assert res[0][0][1:] == res[1][0][1:]                                        # sum_nbox, visited
    assert abs(res[0][0][0] - res[1][0][0]) <= 1e-13 * abs(res[0][0][0])        # photon loss
    for (nb0, l0, v0, cd0), (nb1, l1, v1, cd1) in zip(res[0][2], res[1][2]):
        assert nb0 == nb1 and v0 == v1
        assert np.array_equal(cd0, cd1)                                          # bit for bit
        assert abs(l0 - l1) <= 1e-13 * abs(l0) + 1e-300


def test_edge_cases(pkg, tables):
    """No sources; a zero-flux source; a source outside [1,N] (wrapped); error codes."""
    n = 16
    s, nd, xh, pos, nf = _random_case(n, 3, 5, pkg)
    o = oracle_for(s, tables, n)
    # zero-flux + out-of-range position
    pos2 = np.array([[5, 5, 5], [n + 3, -2, 2 * n + 1], [7, 9, 11]], dtype=np.int32)
    nf2 = np.array([1e8, 1e9, 0.0])
    oloss, onb, ovis, phih_o, w = oracle_pass(o, nd, xh, pos2, nf2)
    b = make_backend(pkg, tables, dict(s, srcpos=pos2, normflux=nf2), n, nd, xh)
    b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert (nbox, vis) == (onb, ovis)
    assert abs(loss - oloss) <= tol("loss") * abs(oloss)
    assert_gamma(b.fetch("phih_grid"), phih_o, w)
    # no sources at all: nothing traced, rates stay zero, the global pass still runs
    b.set_sources(np.zeros((0, 3), dtype=np.int32), np.zeros(0))
    b.zero_rates()
    assert b.pass_sources() == (0.0, 0, 0)
    assert not b.fetch("phih_grid").any()
    with pytest.raises(pkg.C2RayHipError):
        b.do_source(1)
    b.close()


def test_division_helpers_are_ieee_exact(pkg, tables):
    """The kernels replace hipcc's generic f64 division expansion by its bare Newton-Raphson core and,
    for launch-invariant divisors, by a reciprocal multiply with an exact remainder step; both must
    return the same bits as `/` (4 x 2^22 operand sets on the device)."""
    b = pkg.HipBackend(8, *tables, device=0)
    assert b.selftest() == 0
    b.close()


def test_runs_are_reproducible_to_rounding(pkg, tables):
    """Gamma is accumulated with f64 atomics: the order of equal-distance sources may differ
    between runs, everything else is deterministic (loss, nbox, coldens are bit-identical)."""
    n, nsrc = 32, 16
    s, nd, xh, pos, nf = _random_case(n, nsrc, 11, pkg)
    b = make_backend(pkg, tables, dict(s, srcpos=pos, normflux=nf), n, nd, xh)
    outs = []
    for _ in range(2):
        b.begin_step(); b.zero_rates()
        r = b.pass_sources()
        outs.append((r, b.fetch("phih_grid")))
    assert outs[0][0] == outs[1][0]
    assert relerr(outs[0][1], outs[1][1], floor=1e-60) < 1e-14
    b.close()


@pytest.mark.parametrize("name", ["restart32_std_bubbles", "restart32_onesrc"])
@pytest.mark.parametrize("native", [True, False])
def test_restart_from_iteration_dump(pkg, tables, tmp_path, name, native):
    """evolve3D(restart=3) on the GPU against what the reference did from the same dump file."""
    m, a = load_case(name)
    n = m["n"]
    b = make_backend(pkg, tables, m, n, F(a["ndens"]), F(a["xh_before"]))
    if native:
        b.load(xh_av=a["dump_xh_av"], xh_intermed=a["dump_xh_intermed"], phih_grid=a["dump_phih"])
        rep = b.evolve3d_native(m["dt"], restart_niter=m["dump_niter"], restart_photon_loss=m["dump_photon_loss_all"])
        conv = list(rep.it_conv_flag[m["dump_niter"]:rep.niter])
        assert rep.converged and conv == m["log"]["nonconv"][1:]
    else:
        pkg.fileio.write_iteration_dump(str(tmp_path / "iterdump.bin"), m["dump_niter"], m["dump_photon_loss_all"],
                                        a["dump_phih"], a["dump_xh_av"], a["dump_xh_intermed"])
        ev = pkg.Evolve(b); ev.dump_dir = str(tmp_path)
        r = ev.evolve3D(0.0, m["dt"], 3)
        assert r["converged"] and [e["conv_flag"] for e in r["log"]] == m["log"]["nonconv"]
    assert np.max(np.abs(b.fetch("xh") - F(a["xh_after"]))) < tol("x")
    # the step's last pass in the (pinned) oracle, continued from the same dump, gives the tolerance weight
    o = oracle_for(m, tables, n)
    w = o.enable_tolerance_weight()
    oxh, oxav, oxint, ophih = F(a["xh_before"]), F(a["dump_xh_av"]), F(a["dump_xh_intermed"]), F(a["dump_phih"])
    o.evolve3d_restart(m["dt"], F(a["ndens"]), oxh, oxav, oxint, ophih, m["srcpos"], m["normflux"], m["dump_niter"])
    assert np.array_equal(ophih, F(a["phih_grid"]))
    assert_gamma(b.fetch("phih_grid"), ophih, w, name, state_rtol=STATE_RTOL)
    b.close()


def test_deterministic_rates_mode(pkg, tables):
    """deterministic_rates=1: per-source Gamma grids summed in source order (the serial reference's
    order) instead of atomics: bit-identical from run to run, equal to the atomic mode to rounding,
    same parity with the reference fixture."""
    m, a = load_case("sweep32_bubbles")
    n = m["n"]
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    runs = []
    for det in (True, True, False):
        b = make_backend(pkg, tables, m, n, nd, xh, deterministic=det)
        b.begin_step(); b.zero_rates()
        r = b.pass_sources()
        runs.append((r, b.fetch("phih_grid")))
        b.close()
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])        # bitwise
    assert runs[0][0] == runs[2][0]
    assert relerr(runs[0][1], runs[2][1], floor=1e-60) < 1e-13
    ref = F(a["phih"])
    w = oracle_pass(oracle_for(m, tables, n), nd, xh, m["srcpos"], m["normflux"])[4]
    assert_gamma(runs[0][1], ref, w)


def test_deterministic_mode_whole_step_and_batches(pkg, tables):
    """A whole evolve3D step in deterministic mode (iteration history of the reference), and
    several source batches (scratch cap) giving bit-identical rates to one batch."""
    m, a = load_case("evolve32_std_bubbles")
    s = m["steps"]["step001"]
    n = m["n"]
    out = []
    for cap in (0, 600000):          # 600 kB: one source per batch at 32^3 (2 x 262 kB of Gamma grids each)
        b = make_backend(pkg, tables, s, n, F(a["step001_ndens"]), F(a["step001_xh_before"]),
                         deterministic=True, scratch_bytes=cap)
        rep = b.evolve3d_native(s["dt"])
        assert rep.niter == s["niter"] and list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
        assert np.max(np.abs(b.fetch("xh") - F(a["step001_xh_after"]))) < tol("x")
        out.append(b.fetch("phih_grid"))
        b.close()
    assert np.array_equal(out[0], out[1])


@pytest.mark.parametrize("name", ["evolve32_lls2", "evolve32_lls3", "evolve32_clump5"])
def test_physics_variants_vs_reference(pkg, tables, name):
    """type_of_LLS=2 / 3 and a clumping grid (c2r_set_lls, c2r_set_clumping_grid) against the reference
    rebuilt with that one parameter changed."""
    m, a = load_case(name)
    n = m["n"]
    s = m["steps"]["step001"]
    b = make_backend(pkg, tables, s, n, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    if s["type_of_LLS"] != 1:
        b.set_lls(s["type_of_LLS"], a["lls_grid"] if "lls_grid" in a else None, s["R_max_LLS"])
    if "clump_grid" in a:
        b.set_clumping_grid(a["clump_grid"])
    rep = b.evolve3d_native(s["dt"])
    assert rep.converged and rep.niter == s["niter"]
    assert list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
    assert rep.sum_nbox_all == s["sum_nbox_all"]
    assert abs(rep.photon_loss_all - s["photon_loss_all"]) <= tol("loss") * abs(s["photon_loss_all"]) + 1e-300
    assert np.max(np.abs(b.fetch("xh") - F(a["step001_xh_after"]))) < tol("x")
    o = oracle_for(s, tables, n, lls_grid=a["lls_grid"] if "lls_grid" in a else None,
                   clump_grid=a["clump_grid"] if "clump_grid" in a else None)
    orep, oxh, oxav, ophih, w = oracle_step(o, s["dt"], F(a["step001_ndens"]), F(a["step001_xh_before"]), s["srcpos"], s["normflux"])
    assert np.array_equal(ophih, F(a["step001_phih_grid"]))
    assert_gamma(b.fetch("phih_grid"), ophih, w, name, state_rtol=STATE_RTOL)
    for k in ("totrec", "totcollisions", "total_ion"):
        assert abs(getattr(rep, k) / s[k] - 1) < 1e-9, k
    b.close()


@pytest.mark.parametrize("mesh,seed", [((24, 20, 16), 61), ((17, 32, 23), 62), ((1, 16, 12), 63), ((12, 1, 9), 64),
                                       ((16, 16, 2), 65), ((3, 3, 3), 66), ((2, 9, 8), 67)])
def test_non_cubic_mesh_vs_oracle(pkg, tables, mesh, seed):
    """mesh(1:3) need not be equal (sizes.f90:33) nor dr(1:3): trace limits, clipping and the sub-box
    surface are per axis; the loop condition looks at z only (evolve_source.F90:130-131)."""
    from oracle.oracle import Oracle
    rng = np.random.default_rng(seed)
    ncell = mesh[0] * mesh[1] * mesh[2]
    s = pkg.TestProblem(32).step(1)
    dr = (s["dr1"], 1.3 * s["dr1"], 0.8 * s["dr1"])
    vol = dr[0] * dr[1] * dr[2]
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(ncell))).astype(np.float32)
    xh = np.clip(10.0 ** rng.uniform(-4, 0, ncell) * 0.9999, 1e-6, 0.9999)
    nsrc = 9
    pos = np.stack([rng.integers(1, mesh[d] + 1, nsrc) for d in range(3)], axis=1).astype(np.int32)
    nf = 10.0 ** rng.uniform(6, 9, nsrc)
    o = Oracle(mesh, dr, vol, s["coldensh_LLS"], *tables)
    oloss, onb, ovis, phih_o, w = oracle_pass(o, nd, xh, pos, nf)
    b = pkg.HipBackend(mesh, *tables, device=0)
    b.set_step(dr, vol, s["coldensh_LLS"], 1.0)
    b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh)
    b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert (nbox, vis) == (onb, ovis)
    assert abs(loss - oloss) <= tol("loss") * abs(oloss) + 1e-300
    phih = b.fetch("phih_grid")
    assert_gamma(phih, phih_o, w)
    xav, xint = xh.copy(), xh.copy()
    oconv = o.global_pass(s["dt"], nd, xh, xav, xint, phih_o)
    conv, _ = b.global_pass(s["dt"])
    assert conv == oconv and np.max(np.abs(b.fetch("xh_intermed") - xint)) < tol("x")
    b.close()
