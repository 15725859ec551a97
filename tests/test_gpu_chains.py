"""A pass of 64 - 512 sources runs as several chains in flight (csrc/sweep.hip run_chains: the sources in contiguous shares,
each with its own stream and scratch, driven in lock-step) -- what one GPU's share of a multi-GPU run looks like.  The chains
change WHEN launches run, not what they compute: per-source sub-box counts, visited cells, the photon loss (bit for bit: the
shape of its sums is chosen by the pass's source count, not by the batch's) are those of one chain; Gamma differs only by the
order in which the f64 atomics of different sources land.  C2R_CHAINS=n forces n chains (tests/conftest.py hands it to c2r_set_option "chains": the library reads no environment)."""
import numpy as np
import pytest
from tests._util import F, tol, assert_gamma, oracle_pass

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def _case(pkg, n, S, seed):
    from tests.golden.inputs import bubble_xfield, density_factor
    tp = pkg.TestProblem(n); s = tp.step(1)
    nd = (tp.fields(1)[0].reshape((n, n, n), order="F") * density_factor(n, 21)).astype(np.float32)
    pos, nf = pkg.seeded_sources(n, S, seed=seed)
    xh = bubble_xfield(n, [tuple(int(v) for v in q) for q in pos[:40]], 9.0)
    return s, F(nd), F(xh), pos, nf


def _pass(pkg, tables, monkeypatch, chains, n, s, nd, xh, pos, nf, thermal=False):
    if chains is None:
        monkeypatch.delenv("C2R_CHAINS", raising=False)
    else:
        monkeypatch.setenv("C2R_CHAINS", str(chains))
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh); b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    out = (loss, nbox, vis, b.last_nbox().copy(), b.fetch("phih_grid"))
    b.close()
    return out


@pytest.mark.parametrize("S", [96, 130, 200])
def test_chains_give_one_chains_results(pkg, tables, monkeypatch, S):
    n = 64
    s, nd, xh, pos, nf = _case(pkg, n, S, 77)
    one = _pass(pkg, tables, monkeypatch, 1, n, s, nd, xh, pos, nf)
    assert len(set(int(v) for v in one[3])) > 2                     # sources retire at different sub-boxes
    for chains in (None, 2, 3, 4):                                  # None: the library's own rule (2 below 192 sources, 3 from there)
        r = _pass(pkg, tables, monkeypatch, chains, n, s, nd, xh, pos, nf)
        assert r[0] == one[0], (chains, r[0], one[0])               # photon loss: the same bits
        assert r[1:3] == one[1:3] and np.array_equal(r[3], one[3])  # sum of sub-boxes, visited pairs, per-source sub-box counts
        live = one[4] > 0
        assert np.array_equal(r[4] > 0, live)
        assert np.max(np.abs(r[4][live] - one[4][live]) / one[4][live]) < 1e-13      # atomics in another order


def test_chains_over_several_rounds_of_a_small_scratch(pkg, tables, monkeypatch):
    """200 sources through a scratch that holds 90 at a time (C2R_BATCH_CAP): rounds of 90 (two chains of 45), 90 and 20 (below
    64: one chain) -- the same results as everything at once on one chain."""
    n, S = 64, 200
    s, nd, xh, pos, nf = _case(pkg, n, S, 77)
    one = _pass(pkg, tables, monkeypatch, 1, n, s, nd, xh, pos, nf)
    monkeypatch.setenv("C2R_BATCH_CAP", "90")
    r = _pass(pkg, tables, monkeypatch, None, n, s, nd, xh, pos, nf)
    assert r[0] == one[0] and r[1:3] == one[1:3] and np.array_equal(r[3], one[3])
    live = one[4] > 0
    assert np.array_equal(r[4] > 0, live) and np.max(np.abs(r[4][live] - one[4][live]) / one[4][live]) < 1e-13


def test_chains_in_a_non_isothermal_context(pkg, tables, monkeypatch):
    """The HEAT kernels under chains: 90 sources, the rates AND the heating rates (a second accumulator pair all chains add
    into) as one chain leaves them, to the order of the atomics."""
    from tests._util import load_thermal_tables
    n, S = 48, 90
    s, nd, xh, pos, nf = _case(pkg, n, S, 11)
    tt = load_thermal_tables()
    res = []
    for chains in (1, 3):
        monkeypatch.setenv("C2R_CHAINS", str(chains))
        b = pkg.HipBackend(n, *tables, device=0)
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
        b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh)
        b.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
        b.begin_step(); b.zero_rates()
        loss, nbox, vis = b.pass_sources()
        res.append((loss, nbox, vis, b.fetch("phih_grid"), b.fetch("phiheat_grid")))
        b.close()
    a, c = res
    assert a[:3] == c[:3]
    for k in (3, 4):
        live = a[k] > 0
        assert live.any() and np.array_equal(c[k] > 0, live)
        assert np.max(np.abs(c[k][live] - a[k][live]) / a[k][live]) < 1e-13


def test_chained_pass_against_the_oracle(pkg, tables, monkeypatch, sweep_mode):
    """The default rule's chains at 100 sources on 48^3 against the pinned oracle: integers equal, Gamma inside the mode's tolerance."""
    n, S = 48, 100
    s, nd, xh, pos, nf = _case(pkg, n, S, 5)
    r = _pass(pkg, tables, monkeypatch, None, n, s, nd, xh, pos, nf)
    from oracle.oracle import Oracle
    o = Oracle(n, s["dr1"], s["vol"], s["coldensh_LLS"], *tables)
    oloss, onb, ovis, phih, w = oracle_pass(o, nd, xh, pos, nf)
    assert (r[1], r[2]) == (onb, ovis)
    assert abs(r[0] - oloss) <= tol("loss") * abs(oloss)
    assert_gamma(r[4], phih, w, "chained pass")


def test_whole_steps_with_and_without_chains(pkg, tables, monkeypatch):
    """evolve3D over a cold start with 80 sources: iteration count, non-converged-cell history, sub-box history and xh equal."""
    n, S = 32, 80
    tp = pkg.TestProblem(n); s = tp.step(1)
    nd, xh = tp.fields(1)
    pos, nf = pkg.seeded_sources(n, S, seed=3)
    reps = []
    for chains in (1, 2, 3):
        monkeypatch.setenv("C2R_CHAINS", str(chains))
        b = pkg.HipBackend(n, *tables, device=0)
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
        b.set_sources(pos, nf); b.load(ndens=nd, xh=xh)
        rep = b.evolve3d_native(s["dt"])
        reps.append((rep.niter, list(rep.it_conv_flag[:rep.niter]), list(rep.it_sum_nbox[:rep.niter]), rep.photon_loss_all, b.fetch("xh")))
        b.close()
    for r in reps[1:]:
        assert r[0] == reps[0][0] and r[1] == reps[0][1] and r[2] == reps[0][2]
        assert abs(r[3] - reps[0][3]) <= 1e-12 * abs(reps[0][3])
        assert np.max(np.abs(r[4] - reps[0][4])) < 1e-11


# ---- round 6: a chain's pass as ONE replayed launch sequence (csrc/sweep.hip run_chains, include/c2ray_hip.h option chain_graph) --------

def _info_counts(b):
    """(replayed, halted, launch by launch) chain passes so far, from c2r_info."""
    import re
    m = re.search(r"chain passes replayed (\d+) \(halted (\d+)\), launch by launch (\d+)", b.info())
    return tuple(int(v) for v in m.groups())


def _passes(pkg, tables, monkeypatch, chain_graph, n, s, nd, fields, pos, nf, chains=None):
    """One context, one pass per entry of `fields` (each loaded as xh_av before its pass): per pass (loss, sum_nbox, visited,
    per-source sub-boxes, Gamma), and the context's chain counters at the end."""
    monkeypatch.setenv("C2R_CHAIN_GRAPH", "1" if chain_graph else "0")
    if chains is None:
        monkeypatch.delenv("C2R_CHAINS", raising=False)
    else:
        monkeypatch.setenv("C2R_CHAINS", str(chains))
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=fields[0]); b.begin_step()
    out = []
    for x in fields:
        b.load(xh_av=x)
        b.zero_rates()
        loss, nbox, vis = b.pass_sources()
        out.append((loss, nbox, vis, b.last_nbox().copy(), b.fetch("phih_grid")))
    counts = _info_counts(b)
    b.close()
    return out, counts


def _same(r, ref, what):
    assert r[0] == ref[0], (what, r[0], ref[0])                                  # photon loss: the same bits
    assert r[1:3] == ref[1:3] and np.array_equal(r[3], ref[3]), what             # sum of sub-boxes, visited pairs, per-source sub-boxes
    live = ref[4] > 0
    assert np.array_equal(r[4] > 0, live), what
    assert np.max(np.abs(r[4][live] - ref[4][live]) / ref[4][live]) < 1e-13, what      # atomics in another order


@pytest.mark.parametrize("S,chains", [(96, None), (130, None), (200, None), (130, 4)])
def test_replayed_chain_passes_equal_launch_by_launch(pkg, tables, monkeypatch, S, chains):
    """Four passes over the same field: the first is driven launch by launch (nothing is known about the counts), the others are
    replays of the captured sequence -- per-source sub-box counts, visited cells and the photon loss of every pass are the bits
    of the launch-by-launch schedule, no replay halts, and one capture serves them all."""
    n = 64
    s, nd, xh, pos, nf = _case(pkg, n, S, 77)
    ref, c0 = _passes(pkg, tables, monkeypatch, False, n, s, nd, [xh] * 4, pos, nf, chains)
    got, c1 = _passes(pkg, tables, monkeypatch, True, n, s, nd, [xh] * 4, pos, nf, chains)
    assert len(set(int(v) for v in ref[0][3])) > 2                               # sources retire at different sub-boxes
    assert c0[0] == 0 and c0[2] > 0
    nch = c1[2]                                                                  # the first pass: every chain launch by launch
    assert c1 == (3 * nch, 0, nch), c1
    for k in range(4):
        _same(got[k], ref[k], (S, chains, k))
        _same(got[k], ref[0], (S, chains, k))


def test_replay_when_the_counts_move(pkg, tables, monkeypatch):
    """The field changes between passes: sources trace further than the captured sequence was sized for (the device halts the
    replay at the first decision that keeps more sources than the next launches hold, the host goes on from there), go beyond its
    last sub-box (the host continues launch by launch), or retire earlier (surplus launches return at once).  Every pass: the bits
    of a fresh launch-by-launch context on that field."""
    from tests.golden.inputs import bubble_xfield
    n, S = 64, 130
    s, nd, xh, pos, nf = _case(pkg, n, S, 77)
    centres = [tuple(int(v) for v in q) for q in pos[:40]]
    more = F(bubble_xfield(n, [tuple(int(v) for v in q) for q in pos], 14.0))            # every source in a large bubble: all trace further
    less = F(bubble_xfield(n, centres[:10], 5.0))                                         # nearly neutral: all retire in the first sub-boxes
    fields = [xh, xh, more, more, less, xh, xh, more]
    got, c = _passes(pkg, tables, monkeypatch, True, n, s, nd, fields, pos, nf)
    assert c[0] > 0 and c[1] > 0, c                                                       # replays happened, and at least one was halted
    for k, x in enumerate(fields):
        ref, _ = _passes(pkg, tables, monkeypatch, False, n, s, nd, [x], pos, nf)
        _same(got[k], ref[0], k)
    assert len({tuple(g[3]) for g in got}) >= 3                                           # the fields do move the sub-box counts


def test_iterations_with_a_device_gated_tail(pkg, tables, monkeypatch):
    """c2r_iterate over replayed chains: from the iteration whose chains all replay, the totals, the fold of the transposed rates and
    the global pass are enqueued behind them gated on the device (k_chain_gate) and the host waits once.  Iteration by iteration:
    the same photon loss, sub-box sum, visited pairs and non-converged count as with chains driven launch by launch, xh_av to the
    order of the atomics -- on a field that stays put (the gate opens) and after it has been changed (the gate stays shut for an
    iteration, the tail runs the ordinary way)."""
    import re
    from tests.golden.inputs import bubble_xfield
    n, S = 64, 130
    s, nd, xh, pos, nf = _case(pkg, n, S, 77)
    more = F(bubble_xfield(n, [tuple(int(v) for v in q) for q in pos], 14.0))
    hist = {}
    for cg in ("0", "1"):
        monkeypatch.setenv("C2R_CHAIN_GRAPH", cg)
        b = pkg.HipBackend(n, *tables, device=0)
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
        b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh); b.begin_step()
        out = []
        for k in range(9):
            b.load(xh_av=more if k >= 4 else xh)                    # from k = 4 every source traces further: halts, a shut gate
            b.load(xh_intermed=xh)                                  # (so that every iteration starts from the same chemistry state)
            out.append(tuple(b.iterate(s["dt"])) + (b.fetch("xh_av"),))
        hist[cg] = out
        tails = int(re.search(r"device-gated tail (\d+)", b.info()).group(1))
        assert (tails == 0) if cg == "0" else (3 <= tails <= 8), (cg, b.info())
        b.close()
    for k, (a, r) in enumerate(zip(hist["0"], hist["1"])):
        assert r[0] == a[0] and r[1:4] == a[1:4], (k, r[:5], a[:5])       # loss (bits), sum_nbox, visited, conv_flag
        assert abs(r[4] - a[4]) <= 1e-9 * abs(a[4]) and np.max(np.abs(r[5] - a[5])) < 1e-11, k


def test_whole_steps_replayed_and_launch_by_launch(pkg, tables, monkeypatch):
    """evolve3D over a cold start with 80 sources: the counts change from iteration to iteration (captures, halts, continuations
    all occur); iteration count, non-converged-cell history, sub-box history equal, xh to the order of the atomics."""
    n, S = 32, 80
    tp = pkg.TestProblem(n); s = tp.step(1)
    nd, xh = tp.fields(1)
    pos, nf = pkg.seeded_sources(n, S, seed=3)
    reps = []
    for cg in ("0", "1"):
        monkeypatch.setenv("C2R_CHAIN_GRAPH", cg)
        monkeypatch.setenv("C2R_CHAINS", "2")
        b = pkg.HipBackend(n, *tables, device=0)
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
        b.set_sources(pos, nf); b.load(ndens=nd, xh=xh)
        rep = b.evolve3d_native(s["dt"])
        reps.append((rep.niter, list(rep.it_conv_flag[:rep.niter]), list(rep.it_sum_nbox[:rep.niter]), rep.photon_loss_all, b.fetch("xh"), _info_counts(b)))
        b.close()
    a, r = reps
    assert a[5][0] == 0 and r[5][0] > 0, (a[5], r[5])
    assert r[0] == a[0] and r[1] == a[1] and r[2] == a[2]
    assert abs(r[3] - a[3]) <= 1e-12 * abs(a[3])
    assert np.max(np.abs(r[4] - a[4])) < 1e-11


def test_replayed_chains_and_gated_tail_in_a_non_isothermal_context(pkg, tables, monkeypatch):
    """The HEAT kernels under replayed chains with the gated tail (the fold of BOTH transposed accumulators, the thermal global
    pass): 90 sources, five iterations of c2r_iterate -- integers and photon loss as launch by launch, xh_av / temperatures to the
    order of the atomics."""
    from tests._util import load_thermal_tables
    n, S = 48, 90
    s, nd, xh, pos, nf = _case(pkg, n, S, 11)
    tt = load_thermal_tables()
    res = {}
    for cg in ("0", "1"):
        monkeypatch.setenv("C2R_CHAIN_GRAPH", cg)
        b = pkg.HipBackend(n, *tables, device=0)
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
        b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh)
        b.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
        b.set_redshift(9.0)
        b.load(temperature_grid=np.full(n ** 3, 1e4, dtype=np.float32))
        b.begin_step()
        out = []
        for _ in range(5):                       # (every iteration from the same state: the counts settle, the gate opens)
            b.load(xh_av=xh, xh_intermed=xh, temperature_grid=np.full(n ** 3, 1e4, dtype=np.float32))
            out.append(tuple(b.iterate(s["dt"])[:4]))
        res[cg] = (out, b.fetch("xh_av"), b.fetch("temperature_grid"), b.fetch("phiheat_grid"), _info_counts(b), b.info())
        b.close()
    a, r = res["0"], res["1"]
    assert r[4][0] > 0 and "device-gated tail 0" not in r[5], r[5]
    assert r[0] == a[0], (r[0], a[0])
    assert np.max(np.abs(r[1] - a[1])) < 1e-11
    assert np.max(np.abs(r[2] - a[2]) / a[2]) < 1e-5                    # f32 temperatures: a last-bit difference at most
    live = a[3] > 0
    assert np.array_equal(r[3] > 0, live) and np.max(np.abs(r[3][live] - a[3][live]) / a[3][live]) < 1e-12


def test_sources_on_request_in_chunks_of_chains(pkg, tables, monkeypatch):
    """c2r_set_source_queue on one rank with 200 sources handed out 70 at a time (rounds of 70 / 70 / 60: two chains, two chains,
    one chain below 64): per-source sub-box counts, visited cells and the photon loss of the static pass, bit for bit; three passes,
    every source taken once per pass."""
    import ctypes as C
    from c2ray3dm_amd import _capi
    n, S = 64, 200
    s, nd, xh, pos, nf = _case(pkg, n, S, 77)
    monkeypatch.delenv("C2R_CHAINS", raising=False)
    ref = _pass(pkg, tables, monkeypatch, None, n, s, nd, xh, pos, nf)
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh); b.begin_step()
    state = {"pass": None, "next": 0, "asked": []}

    def nxt(user, pass_id, want, first, count):
        if state["pass"] != pass_id:
            state["pass"], state["next"] = pass_id, 0
        k = min(want, S - state["next"])
        first[0], count[0] = state["next"], max(0, k)
        state["next"] += max(0, k)
        state["asked"].append((int(pass_id), int(first[0]), int(count[0])))
        return 0
    b.set_source_queue(nxt, 70)
    for _ in range(3):
        b.zero_rates()
        loss, nbox, vis = b.pass_sources()
        assert loss == ref[0] and (nbox, vis) == ref[1:3] and np.array_equal(b.last_nbox(), ref[3])
        assert np.array_equal(b.local_sources(), np.arange(S))
        g = b.fetch("phih_grid"); live = ref[4] > 0
        assert np.array_equal(g > 0, live) and np.max(np.abs(g[live] - ref[4][live]) / ref[4][live]) < 1e-13
    per_pass = {}
    for p_, f_, c_ in state["asked"]:
        per_pass.setdefault(p_, []).append((f_, c_))
    assert len(per_pass) == 3 and all(v[:3] == [(0, 70), (70, 70), (140, 60)] and len(v) == 4 and v[3][1] == 0 for v in per_pass.values()), per_pass
    b.set_source_queue(None)                                                 # off again: the static rule
    b.zero_rates()
    assert b.pass_sources()[1:] == ref[1:3]
    b.close()
