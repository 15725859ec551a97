"""GPU parity tests on BASELINE.json's own configurations, once per sweep mode:

  configs[2]  256^3 x 100 sources as WHOLE evolve3D steps against fixtures recorded from the Fortran reference
              (cold start and a late field: tests/golden/evolve256_100src_{cold,late});
  configs[3]  the bench workload itself -- 256^3 x 1000 seeded sources, x = 0.999 -- one pass: a 16-source subset
              against the oracle, the other 984 through additivity (Gamma, loss, sub-box counts and visited cells of a
              pass are sums over sources), and the expected trace-to-the-limit sub-box count;
  configs[1]  128^3, ONE source: the single-source sweep and a whole step (five outer iterations, hipGraph replay and
              eager launches) against fixtures recorded from the Fortran reference;
  configs[4]  504^3 (log-normal density, sigma = 1): two sources traced to the limits against the oracle; 48 sources
              through a scratch that holds 17 (three batches), two vs the oracle, the rest by additivity.

The serial oracle needs ~0.15 us per visited (cell, source) pair, so its results are computed once and shared by
the two modes (module cache)."""
import hashlib
import os
import numpy as np
import pytest
from tests._util import F, load_case, oracle_for, expand, tol, assert_gamma, oracle_pass, STATE_RTOL

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]

_cache = {}


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def backend(pkg, tables, s, n, nd, xh, pos, nf):
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
    b.set_sources(pos, nf)
    b.set_rank(0, 1)
    b.load(ndens=nd, xh=xh)
    return b


def _planes(p3, m):
    n = m["n"]
    s = [(q - 1) % n for q in m["srcpos"][0]]
    return {"px": p3[s[0]], "py": p3[:, s[1]], "pz": p3[:, :, s[2]]}


@pytest.mark.parametrize("name", ["evolve256_100src_cold", "evolve256_100src_late"])
def test_evolve3d_256_100src_vs_reference_fixture(pkg, tables, name):
    """BASELINE configs[2] as a whole time step (evolve.F90:83-281): outer-iteration count, the sequence of
    non-converged-cell counts, sum_nbox, photon statistics, planes of xh and Gamma through source 1, checksums."""
    from tests.golden.inputs import bubble_xfield
    m, a = load_case(name)
    n = m["n"]
    nd = F(expand(a["ndens"], n))
    pos, nf = pkg.seeded_sources(n, 100)
    assert [tuple(p) for p in pos.tolist()] == [tuple(p) for p in m["srcpos"]]
    if "xh_before" in a:
        xh0 = F(expand(a["xh_before"], n))
    else:       # too large to commit: regenerate from the recipe and prove it is the field the reference ran on
        if "xfield_" + name not in _cache:       # (30 s of numpy for 100 bubbles on 256^3: once for both sweep modes)
            _cache["xfield_" + name] = F(bubble_xfield(n, [tuple(int(v) for v in p) for p in pos], 14.0))
            assert hashlib.sha256(_cache["xfield_" + name].tobytes()).hexdigest() == m["xh_before_sha256"]
        xh0 = _cache["xfield_" + name].copy()
    b = backend(pkg, tables, m, n, nd, xh0, m["srcpos"], m["normflux"])
    # keep the xh_av each pass started from (ping-pong): the state before the LAST pass gives the tolerance weight
    import torch
    keep = [torch.empty_like(b.xh_av), torch.empty_like(b.xh_av)]
    b.set_iteration_hook(lambda niter, loss: keep[niter % 2].copy_(b.xh_av))
    rep = b.evolve3d_native(m["dt"])
    b.set_iteration_hook(None)
    assert rep.converged and rep.niter == m["niter"]
    assert list(rep.it_conv_flag[:rep.niter]) == m["log"]["nonconv"]
    assert rep.sum_nbox_all == m["sum_nbox_all"]
    assert abs(rep.photon_loss_all - m["photon_loss_all"]) <= tol("loss") * abs(m["photon_loss_all"]) + 1e-300
    x3 = b.fetch("xh").reshape((n, n, n), order="F")
    p3 = b.fetch("phih_grid").reshape((n, n, n), order="F")
    for tag, sl in _planes(x3, m).items():
        assert np.max(np.abs(sl - a["xh_" + tag])) < tol("x"), tag
    assert np.count_nonzero(p3) == m["phih_nonzero"]
    assert abs(float(np.sum(x3, dtype=np.longdouble)) / m["xh_sum"] - 1) < 1e-12
    assert abs(x3.min() - m["xh_min"]) < tol("x") and abs(x3.max() - m["xh_max"]) < tol("x")
    for k in ("totrec", "totcollisions"):
        assert abs(getattr(rep, k) / m[k] - 1) < 1e-9
    assert abs(rep.total_ion - m["total_ion"]) < n ** 3 * 2.3e-16 * rep.h0_before
    # Gamma of the last pass: the oracle's pass over the state that pass started from (the GPU's own xh_av of
    # the iteration before, equal to the reference's to ~1e-14) gives the rates and the tolerance weight
    xav_prev = keep[(rep.niter - 1) % 2].cpu().numpy() if rep.niter > 1 else xh0
    oloss, onb, ovis, ophih, w = oracle_pass(oracle_for(m, tables, n), nd, xav_prev, m["srcpos"], m["normflux"])
    assert onb == m["sum_nbox_all"]
    wp = _planes(w.reshape((n, n, n), order="F"), m)
    for tag, sl in _planes(p3, m).items():
        assert_gamma(sl, a["phih_" + tag], wp[tag], name + " " + tag, state_rtol=STATE_RTOL)       # the Fortran's planes
    assert_gamma(p3.ravel(order="F"), ophih, w, name + " whole mesh vs oracle")
    b.close()


def test_bench_workload_256_x_1000(pkg, tables):
    """The workload bench.py times (256^3, 1000 seeded sources, x = 0.999, first pass) is checked here."""
    n, S, nsub = 256, 1000, 16
    tp = pkg.TestProblem(n)
    s = tp.step(1)
    nd, xh = tp.fields(1, 0.999)
    pos, nf = pkg.seeded_sources(n, S)
    b = backend(pkg, tables, s, n, nd, xh, pos, nf)
    b.begin_step()
    b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    whole = b.fetch("phih_grid")
    per_src = b.last_nbox()
    assert per_src.sum() == nbox and int(np.sum(pkg.box_cost(per_src, (n, n, n)))) == vis
    assert np.all(whole > 0)                       # every cell is reached by some source
    # subset against the oracle
    if "bench" not in _cache:
        _cache["bench"] = oracle_pass(oracle_for(s, tables, n), nd, xh, pos[:nsub], nf[:nsub])
    oloss, onb, ovis, ophih, w = _cache["bench"]
    b.set_sources(pos[:nsub], nf[:nsub])
    b.zero_rates()
    l_a, nb_a, v_a = b.pass_sources()
    g_a = b.fetch("phih_grid")
    assert (nb_a, v_a) == (onb, ovis)
    assert np.array_equal(b.last_nbox(), per_src[:nsub])
    assert abs(l_a - oloss) <= tol("loss") * abs(oloss)
    assert_gamma(g_a, ophih, w, "16-source subset")
    # the other 984 sources: additivity
    b.set_sources(pos[nsub:], nf[nsub:])
    b.zero_rates()
    l_b, nb_b, v_b = b.pass_sources()
    g_b = b.fetch("phih_grid")
    assert np.array_equal(b.last_nbox(), per_src[nsub:])
    assert (nb_a + nb_b, v_a + v_b) == (nbox, vis)
    assert abs(l_a + l_b - loss) <= 1e-12 * abs(loss)
    assert np.max(np.abs(g_a + g_b - whole) / whole) < 1e-12       # order of the atomic adds only
    b.close()


def test_504_cubep3m_format_two_sources_vs_oracle(pkg, tables, tmp_path):
    """BASELINE configs[4]'s mesh (504^3, HBM-resident) fed the way the reference feeds it: a coarsened cubep3m
    density slice `<z>n_all.dat` (nbody_cubep3m.F90:87-107: 3 x int32 + N^3 float32 stream; log-normal, sigma = 1, in
    units of a fine N-body cell's mean mass) -> read_density_file -> scale_density (density_module.F90:203-287).  Two
    sources traced out to q = 252 against the oracle; 24-bit index arithmetic, plane pitch 505, 1.0 GB grids."""
    n, n_box = 504, 10976                       # the 425 Mpc/h run's fine mesh (nbody_cubep3m.F90:17-18)
    fio = pkg.fileio
    tp = pkg.TestProblem(n)        # cell size of the 100 Mpc/h box: 0.028 of an LLS mean free path per cell, rays reach the limits
    s = tp.step(1)
    _, xh = tp.fields(1, 0.9995)
    rng = np.random.default_rng(20261003)
    raw = (np.exp(rng.standard_normal(n ** 3, dtype=np.float32) - 0.5) * np.float32((n_box / n) ** 3)).reshape((n, n, n), order="F")
    raw[3, 4, 5] = 0.0                          # an empty cell: 0.1 particles (density_module.F90:281)
    path = str(tmp_path / fio.cubep3m_density_name("", s["zred"]).strip())
    fio.write_density(path, raw)
    nd = fio.scale_density(fio.read_density(path, mesh=n), s["zred"], n, n_box).ravel(order="F")
    assert abs(float(nd.mean(dtype=np.float64)) / s["ndens"] - 1) < 2e-3          # the mean IGM density of the test problem
    del raw
    pos = np.array([[17, 480, 252], [300, 301, 302]], dtype=np.int32)
    nf = np.array([3e8, 1e9])
    if "504" not in _cache:                      # (the field is seeded: the same in both modes)
        _cache["504"] = oracle_pass(oracle_for(s, tables, n), nd, xh, pos, nf)
    oloss, onb, ovis, ophih, w = _cache["504"]
    b = backend(pkg, tables, s, n, nd, xh, pos, nf)
    b.begin_step()
    b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert (nbox, vis) == (onb, ovis)
    assert list(b.last_nbox()) == [51, 51] and vis == 2 * n ** 3          # both traced out to q = 252: the whole mesh
    assert abs(loss - oloss) <= tol("loss") * abs(oloss) + 1e-300
    assert_gamma(b.fetch("phih_grid"), ophih, w, "504^3")
    b.close()


def test_128_one_source_sweep_vs_reference_fixture(pkg, tables):
    """BASELINE configs[1]: 128^3, ONE source (inputs/test_sources_onesrc.dat), x = 0.999 -- the single-source sweep out to
    sub-box 11 against values recorded from the Fortran reference: sub-box count, photon loss, the source's column
    densities and rates on three planes through it, checksums (exactly the quantity the 0.44 ms / iteration of the
    launch-bound regime is quoted for)."""
    m, a = load_case("sweep128_onesrc_x999")
    n = m["n"]
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    b = backend(pkg, tables, m, n, nd, xh, m["srcpos"], m["normflux"])
    b.begin_step()
    b.zero_rates()
    nb, loss, vis, cd = b.do_source(1, want_coldens=True)
    assert nb == m["sum_nbox"] == 11 and vis == int(pkg.box_cost(np.array([nb]), (n, n, n))[0])
    assert abs(loss - m["photon_loss"]) <= tol("loss") * abs(m["photon_loss"])
    c3 = cd.reshape((n, n, n), order="F")
    assert np.count_nonzero(c3) == m["cd_nonzero"]
    for tag, sl in _planes(c3, m).items():
        ref = a["cd_" + tag]
        assert np.max(np.abs(sl - ref) / np.maximum(ref, 1e-300)) < tol("cd"), tag
    p3 = b.fetch("phih_grid").reshape((n, n, n), order="F")
    assert np.count_nonzero(p3) == m["phih_nonzero"]
    assert abs(float(np.sum(p3, dtype=np.longdouble)) / m["phih_sum"] - 1) < 1e-10
    w3 = oracle_pass(oracle_for(m, tables, n), nd, xh, m["srcpos"], m["normflux"])[4].reshape((n, n, n), order="F")
    wp = _planes(w3, m)
    for tag, sl in _planes(p3, m).items():
        assert_gamma(sl, a["phih_" + tag], wp[tag], tag)
    # the pass over the (one-source) list takes the batch path (fused sub-boxes, per-shell launches, decisions): same numbers
    b.zero_rates()
    l2, nb2, v2 = b.pass_sources()
    assert (nb2, v2) == (nb, vis) and l2 == loss
    assert np.array_equal(b.fetch("phih_grid").reshape((n, n, n), order="F"), p3)
    b.close()


@pytest.mark.parametrize("graph", ["1", "0"])
def test_128_one_source_whole_step_vs_reference_fixture(pkg, tables, monkeypatch, graph):
    """BASELINE configs[1] as a whole evolve3D step (evolve.F90:83-281) from a field with a 30-cell ionized bubble: one
    source, so conv_criterion = 0 and five outer iterations of six sub-boxes follow.  From the second pass on the library
    replays the batch's launch sequence as a hipGraph (C2R_GRAPH=1, the default) -- the same history, sub-box counts and
    photon loss must come out of the replay and of eager launches (C2R_GRAPH=0), and both must be the reference's."""
    import hashlib
    from tests.golden.inputs import bubble_xfield
    monkeypatch.setenv("C2R_GRAPH", graph)
    m, a = load_case("evolve128_onesrc_bubble")
    n = m["n"]
    nd = F(expand(a["ndens"], n))
    xh0 = F(bubble_xfield(n, [(50, 50, 50)], 30.0))
    assert hashlib.sha256(xh0.tobytes()).hexdigest() == m["xh_before_sha256"]
    b = backend(pkg, tables, m, n, nd, xh0, m["srcpos"], m["normflux"])
    losses = []
    b.set_iteration_hook(lambda niter, loss: losses.append(loss))
    rep = b.evolve3d_native(m["dt"])
    b.set_iteration_hook(None)
    assert rep.converged and rep.niter == m["niter"] == 5
    assert list(rep.it_conv_flag[:rep.niter]) == m["log"]["nonconv"]
    assert list(rep.it_sum_nbox[:rep.niter]) == [6] * 5 and rep.sum_nbox_all == m["sum_nbox_all"]
    assert abs(rep.photon_loss_all - m["photon_loss_all"]) <= tol("loss") * abs(m["photon_loss_all"])
    x3 = b.fetch("xh").reshape((n, n, n), order="F")
    p3 = b.fetch("phih_grid").reshape((n, n, n), order="F")
    for tag, sl in _planes(x3, m).items():
        assert np.max(np.abs(sl - a["xh_" + tag])) < tol("x"), tag
    assert np.count_nonzero(p3) == m["phih_nonzero"]
    assert abs(float(np.sum(x3, dtype=np.longdouble)) / m["xh_sum"] - 1) < 1e-12
    for k in ("totrec", "totcollisions"):
        assert abs(getattr(rep, k) / m[k] - 1) < 1e-9
    # graph replay and eager launches: the per-iteration photon losses are the same numbers, bit for bit
    key = "onesrc_losses_" + os.environ.get("C2R_SWEEP_MODE", "0")
    if key in _cache:
        assert _cache[key] == losses
    _cache[key] = losses
    b.close()


def test_504_many_sources_in_three_batches(pkg, tables):
    """BASELINE configs[4]'s mesh with more sources than the sweep scratch holds at once: 504^3 (log-normal density), 48
    sources, scratch_bytes sized for 17 of them (24.5 MB of shell planes each) -> three batches through the same scratch
    (the path the 10 000-source run takes four times over).  Two sources against the oracle, the other 46 by additivity,
    and the three-batch pass against a one-batch pass of the same 48."""
    n, S = 504, 48
    tp = pkg.TestProblem(n)
    s = tp.step(1)
    _, xh = tp.fields(1, 0.9995)
    rng = np.random.default_rng(20261003)
    nd = (np.float32(s["ndens"]) * np.exp(rng.standard_normal(n ** 3, dtype=np.float32) - 0.5)).astype(np.float32)
    pos, nf = pkg.seeded_sources(n, S, seed=504)
    pos[0] = (17, 480, 252); pos[1] = (300, 301, 302)
    nf[0], nf[1] = 3e8, 1e9
    per_src = 2 * 6 * 505 * 505 * 8 + 6 * ((505 * 505 + 255) // 256) * 8 + 64
    res = {}
    for tag, scratch in (("three", 17 * per_src + 4096), ("one", 0)):
        b = pkg.HipBackend(n, *tables, device=0, scratch_bytes=scratch)
        b.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
        b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh)
        b.begin_step(); b.zero_rates()
        loss, nbox, vis = b.pass_sources()
        res[tag] = (loss, nbox, vis, b.fetch("phih_grid"), b.last_nbox().copy())
        if tag == "three":
            # the first two sources alone, against the oracle; then the other 46
            if "504many" not in _cache:
                _cache["504many"] = oracle_pass(oracle_for(s, tables, n), nd, xh, pos[:2], nf[:2])
            oloss, onb, ovis, ophih, w = _cache["504many"]
            b.set_sources(pos[:2], nf[:2]); b.zero_rates()
            l_a, nb_a, v_a = b.pass_sources()
            g_a = b.fetch("phih_grid")
            assert (nb_a, v_a) == (onb, ovis) and abs(l_a - oloss) <= tol("loss") * abs(oloss) + 1e-300
            assert_gamma(g_a, ophih, w, "504^3, 2 of 48")
            b.set_sources(pos[2:], nf[2:]); b.zero_rates()
            l_b, nb_b, v_b = b.pass_sources()
            g_b = b.fetch("phih_grid")
            assert (nb_a + nb_b, v_a + v_b) == (nbox, vis)
            assert abs(l_a + l_b - loss) <= 1e-12 * abs(loss)
            whole = res[tag][3]
            assert np.all(whole > 0) and np.max(np.abs(g_a + g_b - whole) / whole) < 1e-12
        b.close()
    assert list(res["three"][4]) == [51] * S and res["three"][2] == S * n ** 3       # every source traces the whole mesh
    assert res["three"][1:3] == res["one"][1:3] and np.array_equal(res["three"][4], res["one"][4])
    assert abs(res["three"][0] / res["one"][0] - 1) < 1e-12
    assert np.max(np.abs(res["three"][3] / res["one"][3] - 1)) < 1e-12


def test_504_ten_thousand_sources(pkg, tables):
    """BASELINE configs[4] at its own size: 504^3, log-normal density, **10 000 sources** -- one pass through the ~4 scratch
    batches the library cuts by itself (24.5 MB of shell planes per source against a quarter of the free HBM).  Every source
    traces the whole mesh (51 sub-boxes, N^3 cells each), every cell receives a rate, two of the sources are checked
    against the oracle (cached: the same two as in the 48-source test) and the other 9 998 by additivity."""
    n, S = 504, 10000
    tp = pkg.TestProblem(n)
    s = tp.step(1)
    _, xh = tp.fields(1, 0.9995)
    rng = np.random.default_rng(20261003)
    nd = (np.float32(s["ndens"]) * np.exp(rng.standard_normal(n ** 3, dtype=np.float32) - 0.5)).astype(np.float32)
    pos, nf = pkg.seeded_sources(n, S, seed=504)
    pos[0] = (17, 480, 252); pos[1] = (300, 301, 302)
    nf[0], nf[1] = 3e8, 1e9
    b = backend(pkg, tables, s, n, nd, xh, pos, nf)
    b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    assert nbox == 51 * S and vis == S * n ** 3 and np.all(b.last_nbox() == 51)
    whole = b.fetch("phih_grid")
    assert np.all(whole > 0)
    if "504many" not in _cache:
        _cache["504many"] = oracle_pass(oracle_for(s, tables, n), nd, xh, pos[:2], nf[:2])
    oloss, onb, ovis, ophih, w = _cache["504many"]
    b.set_sources(pos[:2], nf[:2]); b.zero_rates()
    l_a, nb_a, v_a = b.pass_sources()
    g_a = b.fetch("phih_grid")
    assert (nb_a, v_a) == (onb, ovis) and abs(l_a - oloss) <= tol("loss") * abs(oloss) + 1e-300
    assert_gamma(g_a, ophih, w, "504^3, 2 of 10 000")
    b.set_sources(pos[2:], nf[2:]); b.zero_rates()
    l_b, nb_b, v_b = b.pass_sources()
    g_a += b.fetch("phih_grid")
    assert (nb_a + nb_b, v_a + v_b) == (nbox, vis)
    assert abs(l_a + l_b - loss) <= 1e-12 * abs(loss)
    assert np.max(np.abs(g_a - whole) / whole) < 1e-11       # 10 000 atomic adds per cell in either order
    b.close()


@pytest.mark.parametrize("case,tag", [("evolve32_std_bubbles", "step001"), ("evolve32_onesrc", "step001"), ("evolve64_std_bubbles", "step001")])
def test_graph_replay_equals_eager_launches_while_the_sub_boxes_change(pkg, tables, monkeypatch, case, tag):
    """Batches of <= 32 sources replay their launch sequence as a hipGraph up to the sub-box the previous pass ended at
    (box_hint) and continue eagerly beyond it.  On steps whose sub-box counts change from iteration to iteration -- sources
    retire before the hint or grow past it -- the replay (C2R_GRAPH=1), eager launches (C2R_GRAPH=0) and the schedule without
    the hint (C2R_SCHED_HINT=0) must give the same iteration history, per-iteration sub-box sums and, with ordered rates,
    the same xh, Gamma and photon loss bit for bit."""
    m, a = load_case(case)
    n, s = m["n"], m["steps"][tag]
    nd, xh0 = F(a[tag + "_ndens"]), F(a[tag + "_xh_before"])
    out = []
    for env in ({"C2R_GRAPH": "1"}, {"C2R_GRAPH": "0"}, {"C2R_SCHED_HINT": "0"}):
        for k in ("C2R_GRAPH", "C2R_SCHED_HINT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b = pkg.HipBackend(n, *tables, device=0, deterministic=True)
        b.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
        b.set_sources(s["srcpos"], s["normflux"]); b.set_rank(0, 1); b.load(ndens=nd, xh=xh0)
        losses = []
        b.set_iteration_hook(lambda niter, loss: losses.append(loss))
        rep = b.evolve3d_native(s["dt"])
        b.set_iteration_hook(None)
        assert rep.niter == s["niter"] and list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"]
        out.append((list(rep.it_sum_nbox[:rep.niter]), losses, b.fetch("xh"), b.fetch("phih_grid")))
        b.close()
    if case == "evolve64_std_bubbles":
        assert len(set(out[0][0])) > 1                                   # the sub-box sums do change during the step (30, 32, 32, 33, 33)
    for o in out[1:]:
        assert o[0] == out[0][0] and o[1] == out[0][1]
        assert np.array_equal(o[2], out[0][2]) and np.array_equal(o[3], out[0][3])
