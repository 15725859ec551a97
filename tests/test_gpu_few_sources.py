"""GPU tests of the few-source (launch-bound) schedule: the fused outer iteration (c2r_iterate: one replayed hipGraph and
one host wait per iteration, the launches behind the pass gated on the device) and the look-ahead pairs (k_sweep_pair_fast:
two shells per launch, the second recomputing the first's column densities) must change NOTHING -- same sub-box counts,
same photon loss, same rates and fractions bit for bit (rates compared in deterministic-rates mode, where their order is
fixed, and to the rounding of the atomic order otherwise) -- against the plain three-step iteration with one launch
per shell, which is what tests/test_gpu_parity.py and tests/test_gpu_configs.py pin to the reference."""
import numpy as np
import pytest
from tests._util import F, load_case, load_tables, load_thermal_tables

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def _backend(pkg, n, S, seed, x, fast, det, monkeypatch, env, lls=1, thermal=False):
    """A context over a seeded test problem; env: schedule switches (C2R_PAIR_SHELLS, C2R_FUSED_ITER) that tests/conftest.py hands to c2r_set_option."""
    for k in ("C2R_PAIR_SHELLS", "C2R_FUSED_ITER", "C2R_GRAPH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tables = load_tables()
    rng = np.random.default_rng(seed)
    tp = pkg.TestProblem(n)
    s = tp.step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(n ** 3))).astype(np.float32)
    xh = np.clip(x * (1.0 - 1e-3 * rng.random(n ** 3)), 1e-6, 1 - 1e-9)
    pos, nf = pkg.seeded_sources(n, S, seed=seed)
    b = pkg.HipBackend(n, *tables, device=0, fast=fast, deterministic=det)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    if lls == 2:
        b.set_lls(2, (s["coldensh_LLS"] * np.exp(rng.standard_normal(n ** 3))).astype(np.float32), 0.0)
    elif lls == 3:
        b.set_lls(3, None, 0.3 * n * float(np.ravel(s["dr1"])[0]))
    if thermal:
        tt = load_thermal_tables()
        b.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
        b.set_redshift(9.0)
    b.set_sources(pos, nf)
    b.set_rank(0, 1)
    b.load(ndens=nd, xh=xh)
    if thermal:
        b.load(temperature_grid=np.full(n ** 3, 1e4, dtype=np.float32))
    b.begin_step()
    return b, s


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("n,S,x,lls,thermal", [(64, 5, 0.9995, 1, False), (96, 12, 0.9995, 2, False), (64, 9, 0.999, 3, False),
                                               (64, 6, 0.9995, 1, True), (130, 3, 0.99995, 1, False), (48, 30, 0.99, 2, True),
                                               # ... and passes that cross the pairs' work limit (n_active x cells of the second
                                               # shell <= 160000): pairs near the source, single launches beyond
                                               (256, 1, 0.99995, 1, False), (128, 4, 0.9999, 1, False)])
def test_lookahead_pairs_equal_one_launch_per_shell(pkg, monkeypatch, fast, n, S, x, lls, thermal):
    """One pass with C2R_PAIR_SHELLS=0 (one launch per shell) and =1 (the default): every template variant of the pair kernels
    (sweep mode x LLS type x heating) leaves the same sub-box counts, the same loss and -- rates in source order -- the same bits."""
    out = []
    for pair in ("0", "1"):
        b, _ = _backend(pkg, n, S, 11 * n + S, x, fast, True, monkeypatch, {"C2R_PAIR_SHELLS": pair}, lls, thermal)
        b.zero_rates()
        loss, nb, vis = b.pass_sources()
        loss2, nb2, vis2 = b.pass_sources()              # a second pass takes the hipGraph path (box_hint known)
        out.append((loss, nb, vis, loss2, nb2, b.last_nbox().copy(), b.fetch("phih_grid"),
                    b.fetch("phiheat_grid") if thermal else None))
        b.close()
    a, t = out
    assert a[:5] == t[:5] and np.array_equal(a[5], t[5])
    assert np.array_equal(a[6], t[6])
    assert a[6].max() > 0
    if thermal:
        assert np.array_equal(a[7], t[7]) and a[7].max() > 0


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("n,S,x,thermal", [(64, 4, 2e-4, False), (128, 1, 0.9995, False), (64, 7, 0.5, True)])
def test_iterate_equals_zero_rates_pass_global_pass(pkg, monkeypatch, fast, n, S, x, thermal):
    """c2r_iterate against its three steps over six outer iterations from the same state: from a cold or half-ionised start
    the sub-boxes grow from iteration to iteration (the gated tail stays shut and the iteration finishes the slow way), then
    the steady state replays one graph -- every iteration returns the same numbers, and the arrays end bit-equal."""
    res = []
    for mode in ("steps", "iterate"):
        b, s = _backend(pkg, n, S, 5 * n + S, x, fast, True, monkeypatch, {}, 1, thermal)
        hist = []
        for _ in range(6):
            if mode == "steps":
                b.zero_rates()
                loss, nb, vis = b.pass_sources()
                conv, s1 = b.global_pass(s["dt"])
            else:
                loss, nb, vis, conv, s1 = b.iterate(s["dt"])
            hist.append((loss, nb, vis, conv, s1))
        res.append((hist, b.fetch("phih_grid"), b.fetch("xh_av"), b.fetch("xh_intermed"),
                    b.fetch("temperature_grid") if thermal else None, b.last_nbox().copy()))
        b.close()
    a, t = res
    assert a[0] == t[0]
    if (n, S) == (128, 1):       # 10, 7, 7, 8, 8, 8 sub-boxes: the graph runs past the source's last box, then stops short of it
        nbs = [h[1] for h in a[0]]
        assert min(nbs) < nbs[0] and any(y > x_ for x_, y in zip(nbs[1:], nbs[2:]))
    for k in (1, 2, 3):
        assert np.array_equal(a[k], t[k])
    if thermal:
        assert np.array_equal(a[4], t[4])
    assert np.array_equal(a[5], t[5])


@pytest.mark.parametrize("thermal", [False, True])
def test_setters_between_fused_iterations_take_effect(pkg, monkeypatch, thermal):
    """The fused iteration replays a captured launch sequence: nothing a setter changes between two iterations may stay
    behind in it.  The step's scalars (dr, vol, LLS column, dt, clumping, temperature, redshift) reach the kernels through the
    device-resident step block, so the graph is NOT re-captured when they change -- and must still see them; the clumping
    grid's pointer is a kernel argument, so the graph IS re-captured when the grid appears or goes.  Same sequence of calls
    with C2R_FUSED_ITER=0 (three steps, arguments rebuilt on every call): the same numbers and bits."""
    n, S = 48, 3
    res = []
    for env in ({}, {"C2R_FUSED_ITER": "0"}):
        b, s = _backend(pkg, n, S, 77, 0.9995, True, True, monkeypatch, env, 1, thermal)
        rng = np.random.default_rng(5)
        grid = (1.0 + rng.random(n ** 3)).astype(np.float32)
        dr = np.ravel(s["dr1"])[0]
        hist = []
        def it(dt):
            hist.append(b.iterate(dt))
        captures = lambda: int(b.info().split("graph captures ")[1])
        for _ in range(3): it(s["dt"])                                   # steady state: the graph replays
        c0 = captures()
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 2.5, 2.0e4)     # clumping and temperature
        it(s["dt"]); it(s["dt"])
        it(0.5 * s["dt"]); it(0.5 * s["dt"])                              # dt
        c1 = captures()
        b.set_step(1.01 * dr, 1.01 ** 3 * s["vol"], 0.9 * s["coldensh_LLS"], 2.5, 2.0e4)    # the mesh expands, the LLS column changes
        it(0.5 * s["dt"]); it(0.5 * s["dt"])
        if not env:
            assert c0 >= 1 and c1 == c0          # new step scalars: the same captured sequence (re-captured only if a sub-box count moved)
        b.set_clumping_grid(grid)                                         # a clumping grid appears ...
        it(s["dt"]); it(s["dt"])
        b.set_clumping_grid(2.0 * grid)                                   # ... is refilled ...
        it(s["dt"])
        b.set_clumping_grid(None)                                         # ... and goes
        it(s["dt"]); it(s["dt"])
        if thermal:
            b.set_redshift(7.5)
            it(s["dt"]); it(s["dt"])
        res.append((hist, b.fetch("phih_grid"), b.fetch("xh_av"), b.fetch("xh_intermed"),
                    b.fetch("temperature_grid") if thermal else None))
        b.close()
    a, t = res
    assert a[0] == t[0]
    assert len({h[4] for h in a[0]}) >= 5                                 # (the changes did change the chemistry)
    for k in (1, 2, 3):
        assert np.array_equal(a[k], t[k])
    if thermal:
        assert np.array_equal(a[4], t[4])


def test_iterate_is_the_single_rank_call(pkg, monkeypatch):
    """With more than one rank a collective belongs between the pass and the global pass: C2R_ESTATE, with a message."""
    import ctypes as C
    b, s = _backend(pkg, 32, 2, 3, 0.9, True, False, monkeypatch, {})
    b.set_rank(0, 2, allreduce=lambda t: None)
    loss, nb, vis, conv, s1 = C.c_double(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
    rc = b.lib.c2r_iterate(b.ctx, s["dt"], C.byref(loss), C.byref(nb), C.byref(vis), C.byref(conv), C.byref(s1))
    assert rc == -2 and b"single-rank" in b.lib.c2r_last_error(b.ctx)
    b.close()


def test_whole_step_fused_and_paired_vs_reference_fixture(pkg, monkeypatch):
    """The 128^3 x 1-source step of the reference (BASELINE configs[1]) through the Python loop, which runs on
    Evolve.iteration -> c2r_iterate with look-ahead pairs: the fixture's iteration history and sub-box counts, and the same
    numbers bit for bit as the same loop with both switched off."""
    from tests._util import expand
    from tests.golden.inputs import bubble_xfield
    m, a = load_case("evolve128_onesrc_bubble")
    n = m["n"]
    nd = F(expand(a["ndens"], n))
    xh0 = F(bubble_xfield(n, [(50, 50, 50)], 30.0))
    got = []
    for env in ({}, {"C2R_PAIR_SHELLS": "0", "C2R_FUSED_ITER": "0"}):
        for k in ("C2R_PAIR_SHELLS", "C2R_FUSED_ITER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b = pkg.HipBackend(n, *load_tables(), device=0, fast=True, deterministic=True)
        b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
        b.set_sources(m["srcpos"], m["normflux"])
        b.load(ndens=nd, xh=xh0)
        r = pkg.Evolve(b).evolve3D(0.0, m["dt"], 0)
        got.append((r["niter"], [e["conv_flag"] for e in r["log"]], [e["sum_nbox"] for e in r["log"]], r["photon_loss_all"],
                    b.fetch("xh"), b.fetch("phih_grid")))
        b.close()
    f, p = got
    assert f[0] == m["niter"] == 5 and f[1] == m["log"]["nonconv"] and f[2] == [6] * 5
    assert abs(f[3] - m["photon_loss_all"]) <= 1e-9 * abs(m["photon_loss_all"])
    assert abs(float(np.sum(f[4], dtype=np.longdouble)) / m["xh_sum"] - 1) < 1e-11
    assert f[:4] == p[:4] and np.array_equal(f[4], p[4]) and np.array_equal(f[5], p[5])
