"""GPU tests of the C ABI's contract: call-order errors, argument checks, host-pointer entry point
(c2r_evolve3d, what the Fortran shim calls), context-owned buffers and streams."""
import ctypes as C
import numpy as np
import pytest
from tests._util import F, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_call_order_and_argument_errors(pkg, tables):
    lib = pkg.load_library()
    p = pkg.default_params(16)
    ctx = C.c_void_p()
    assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
    loss, nb, vis = C.c_double(), C.c_int64(), C.c_int64()
    # no tables / step scalars yet: C2R_ESTATE (-2), with a message
    assert lib.c2r_pass_sources(ctx, C.byref(loss), C.byref(nb), C.byref(vis)) == -2
    assert b"c2r_set_tables" in lib.c2r_last_error(ctx)
    thick, thin = tables
    assert lib.c2r_set_tables(ctx, thick.ctypes.data, thin.ctypes.data, 17) == -1          # wrong length
    assert lib.c2r_set_tables(ctx, thick.ctypes.data, thin.ctypes.data, 2001) == 0
    assert lib.c2r_pass_sources(ctx, C.byref(loss), C.byref(nb), C.byref(vis)) == -2
    assert b"c2r_set_step" in lib.c2r_last_error(ctx)
    dr = (C.c_double * 3)(1e24, 1e24, 1e24)
    assert lib.c2r_set_step(ctx, C.byref(dr), -1.0, 1e16, 1.0, 1e4) == -1                  # vol <= 0
    assert lib.c2r_set_step(ctx, C.byref(dr), 1e72, 1e16, 1.0, 1e4) == 0
    assert lib.c2r_set_rank(ctx, 2, 2, pkg._capi.ALLREDUCE_FN(0), None) == -1            # rank >= nranks
    assert lib.c2r_set_rank(ctx, 0, 2, pkg._capi.ALLREDUCE_FN(0), None) == -1            # nranks > 1 without a callback
    assert lib.c2r_set_lls(ctx, 2, None, 0.0) == -1 and lib.c2r_set_lls(ctx, 3, None, 0.0) == -1
    assert lib.c2r_do_source(ctx, 1, None, None, None, None) == -1                          # no sources set
    # no sources: a pass is legal and does nothing
    assert lib.c2r_pass_sources(ctx, C.byref(loss), C.byref(nb), C.byref(vis)) == 0
    assert (loss.value, nb.value, vis.value) == (0.0, 0, 0)
    lib.c2r_destroy(ctx)
    bad = pkg.default_params(0)
    assert lib.c2r_create(C.byref(ctx), C.byref(bad)) == -1
    # the sweep addresses cells with 32-bit byte offsets: meshes of 2^29 cells and more (813^3, 1024^3 -- which
    # would fit the 288 GB of HBM) are refused before anything is allocated; 812^3 is the largest cube
    for n in (813, 1024):
        ctx = C.c_void_p()
        assert lib.c2r_create(C.byref(ctx), C.byref(pkg.default_params(n))) == -1
        assert b"mesh too large" in lib.c2r_last_error(ctx)
        lib.c2r_destroy(ctx)
    big = pkg.default_params(16); big.sweep_mode = 7
    assert lib.c2r_create(C.byref(ctx), C.byref(big)) == -1                                # unknown sweep mode


def test_host_pointer_evolve3d_is_what_the_shim_calls(pkg, tables):
    """c2r_evolve3d with the driver's host arrays (context-owned device buffers, own stream):
    same results as the device-resident path, inputs other than xh untouched."""
    lib = pkg.load_library()
    m, a = load_case("evolve32_std_bubbles")
    s = m["steps"]["step001"]
    n = m["n"]
    p = pkg.default_params(n)
    ctx = C.c_void_p()
    assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
    thick, thin = tables
    assert lib.c2r_set_tables(ctx, thick.ctypes.data, thin.ctypes.data, 2001) == 0
    dr = (C.c_double * 3)(s["dr1"], s["dr2"], s["dr3"])
    assert lib.c2r_set_step(ctx, C.byref(dr), s["vol"], s["coldensh_LLS"], s["clumping"], 1e4) == 0
    pos = np.ascontiguousarray(s["srcpos"], dtype=np.int32); nf = np.ascontiguousarray(s["normflux"], dtype=np.float64)
    assert lib.c2r_set_sources(ctx, pos.ctypes.data, nf.ctypes.data, len(nf)) == 0
    nd = F(a["step001_ndens"]); xh = F(a["step001_xh_before"]); nd0 = nd.copy()
    xav, xint, phih = np.empty(n ** 3), np.empty(n ** 3), np.empty(n ** 3)
    rep = pkg.Report()
    assert lib.c2r_evolve3d(ctx, s["dt"], nd.ctypes.data, xh.ctypes.data, xav.ctypes.data, xint.ctypes.data,
                            phih.ctypes.data, C.byref(rep)) == 0
    assert rep.niter == s["niter"] and rep.converged == 1
    assert np.array_equal(nd, nd0)
    assert np.max(np.abs(xh - F(a["step001_xh_after"]))) < 1e-9
    assert np.max(np.abs(xav - F(a["step001_xh_av"]))) < 1e-9
    assert np.array_equal(xint, xh)                 # evolve.F90:218 xh = xh_intermed on convergence
    ref = F(a["step001_phih_grid"])
    assert np.max(np.abs(phih - ref) / np.maximum(ref, 1e-60)) < 1e-8
    # where the call's wall time went: the two groups of copies (HIP events), the whole call (host clock)
    assert 0.0 < rep.seconds_upload < rep.seconds_total and 0.0 < rep.seconds_download < rep.seconds_total
    assert rep.seconds_sweep + rep.seconds_chem < rep.seconds_total
    # optional outputs may be NULL -- then they are neither copied back nor touched (what the Fortran shim does with the work
    # arrays xh_av / xh_intermed); the same list of sources again keeps what the last pass learnt (no second scratch allocation)
    assert lib.c2r_set_sources(ctx, pos.ctypes.data, nf.ctypes.data, len(nf)) == 0
    xh2 = F(a["step001_xh_before"])
    rep2 = pkg.Report()
    assert lib.c2r_evolve3d(ctx, s["dt"], nd.ctypes.data, xh2.ctypes.data, None, None, None, C.byref(rep2)) == 0
    assert np.max(np.abs(xh2 - xh)) < 1e-13
    assert rep2.niter == rep.niter and list(rep2.it_conv_flag[:rep.niter]) == list(rep.it_conv_flag[:rep.niter])
    assert rep2.seconds_download < rep.seconds_download            # 8 B per cell instead of 32
    # the photon statistics the shim's photonstatistics module is fed with (c2r_report) = the reference's module variables
    for k in ("totrec", "totcollisions", "dh0", "total_ion"):
        assert abs(getattr(rep2, k) / s[k] - 1) < 1e-9, k
    lib.c2r_destroy(ctx)


def _ctx_for(pkg, tables, s, n):
    lib = pkg.load_library()
    p = pkg.default_params(n)
    ctx = C.c_void_p()
    assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
    thick, thin = tables
    assert lib.c2r_set_tables(ctx, thick.ctypes.data, thin.ctypes.data, 2001) == 0
    dr = (C.c_double * 3)(s["dr1"], s["dr2"], s["dr3"])
    assert lib.c2r_set_step(ctx, C.byref(dr), s["vol"], s["coldensh_LLS"], s["clumping"], 1e4) == 0
    pos = np.ascontiguousarray(s["srcpos"], dtype=np.int32); nf = np.ascontiguousarray(s["normflux"], dtype=np.float64)
    assert lib.c2r_set_sources(ctx, pos.ctypes.data, nf.ctypes.data, len(nf)) == 0
    return lib, ctx


def test_iteration_hook_sees_every_iteration_and_a_dumpable_state(pkg, tables):
    """c2r_set_iteration_hook: the reference's iteration-dump point (evolve.F90:271-275).  A dump taken
    in the hook at iteration k and handed to c2r_evolve3d_restart repeats that iteration's global pass
    bit for bit -- after which, as in the reference (fixture restart32_std_bubbles), no cell has changed,
    Test 1 passes at once and the step ends with the dump's xh_intermed."""
    m, a = load_case("evolve32_std_bubbles")
    s, n = m["steps"]["step001"], m["n"]
    lib, ctx = _ctx_for(pkg, tables, s, n)
    nd = F(a["step001_ndens"]); xh = F(a["step001_xh_before"])
    seen, dump = [], {}
    HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_double)

    def hook(user, niter, loss):
        seen.append((niter, loss))
        if niter == 2:
            for which, key in ((2, "xh_av"), (3, "xh_intermed"), (4, "phih")):
                dump[key] = np.empty(n ** 3)
                assert lib.c2r_download(ctx, which, dump[key].ctypes.data) == 0
            dump["loss"] = loss
        return 0
    cb = HOOK(hook)
    assert lib.c2r_set_iteration_hook(ctx, cb, None) == 0
    xav, xint, phih = np.empty(n ** 3), np.empty(n ** 3), np.empty(n ** 3)
    rep = pkg.Report()
    assert lib.c2r_evolve3d(ctx, s["dt"], nd.ctypes.data, xh.ctypes.data, xav.ctypes.data, xint.ctypes.data,
                            phih.ctypes.data, C.byref(rep)) == 0
    assert [k for k, _ in seen] == list(range(1, rep.niter + 1))
    assert seen[-1][1] == rep.photon_loss_all
    # resume from the iteration-2 state in a fresh context
    assert lib.c2r_set_iteration_hook(ctx, None, None) == 0
    lib.c2r_destroy(ctx)
    lib, ctx = _ctx_for(pkg, tables, s, n)
    xh2 = F(a["step001_xh_before"])
    rep2 = pkg.Report()
    assert lib.c2r_evolve3d_restart(ctx, s["dt"], 2, dump["loss"], nd.ctypes.data, xh2.ctypes.data,
                                    dump["xh_av"].ctypes.data, dump["xh_intermed"].ctypes.data,
                                    dump["phih"].ctypes.data, C.byref(rep2)) == 0
    assert rep2.niter == 2 and rep2.converged == 1 and rep2.conv_flag == 0
    assert np.array_equal(xh2, dump["xh_intermed"])
    assert rep2.photon_loss_all == dump["loss"]
    # a failing hook aborts the step with C2R_ECALLBACK
    bad = HOOK(lambda u, k, l: 1)
    assert lib.c2r_set_iteration_hook(ctx, bad, None) == 0
    xh3 = F(a["step001_xh_before"])
    assert lib.c2r_evolve3d(ctx, s["dt"], nd.ctypes.data, xh3.ctypes.data, None, None, None, None) == -4
    lib.c2r_destroy(ctx)


def test_host_array_pieces_do_source_and_global_pass(pkg, tables):
    """c2r_do_source_host / c2r_global_pass_host (what the Fortran shim's do_source calls, and the
    global_pass counterpart) against the oracle on the same arrays."""
    m, a = load_case("evolve32_std_bubbles")
    s, n = m["steps"]["step001"], m["n"]
    lib, ctx = _ctx_for(pkg, tables, s, n)
    from tests._util import oracle_for
    o = oracle_for(s, tables, n)
    nd = F(a["step001_ndens"]); xh = F(a["step001_xh_before"]); xav = xh.copy()
    phih = np.zeros(n ** 3); phih_o = np.zeros(n ** 3)
    for ns in range(1, len(s["normflux"]) + 1):
        cd = np.empty(n ** 3); loss = C.c_double(); nb = C.c_int32()
        assert lib.c2r_do_source_host(ctx, ns, nd.ctypes.data, xav.ctypes.data, phih.ctypes.data, cd.ctypes.data,
                                      C.byref(loss), C.byref(nb)) == 0
        nbo, losso, _, cdo = o.do_source(nd, xav, phih_o, s["srcpos"][ns - 1], s["normflux"][ns - 1])
        assert nb.value == nbo and abs(loss.value - losso) <= 1e-10 * abs(losso)
        assert np.array_equal(cd == 0, cdo == 0)
        assert np.max(np.abs(cd - cdo) / np.maximum(cdo, 1e-300)) < 1e-11
    assert np.max(np.abs(phih - phih_o) / np.maximum(phih_o, 1e-60)) < 1e-9
    xint = np.empty(n ** 3); conv = C.c_int64()
    xav_o, xint_o = xav.copy(), np.empty(n ** 3)
    assert lib.c2r_global_pass_host(ctx, s["dt"], nd.ctypes.data, xh.ctypes.data, xav.ctypes.data, xint.ctypes.data,
                                    phih_o.ctypes.data, C.byref(conv)) == 0
    conv_o = o.global_pass(s["dt"], nd, xh, xav_o, xint_o, phih_o)
    assert conv.value == conv_o
    assert np.max(np.abs(xint - xint_o)) < 1e-12 and np.max(np.abs(xav - xav_o)) < 1e-12
    lib.c2r_destroy(ctx)


def test_per_iteration_conservation_line_comes_out_of_the_global_pass(pkg, tables):
    """c2r_report.it_photcons[k]: the conservation ratio the reference logs after every global pass
    (evolve.F90:570 calculate_photon_statistics(dt,xh_intermed,xh_av) + report_photonstatistics).  Its four mesh sums
    are accumulated inside k_global_pass, in the order of the separate kernel (c2r_photon_sums): the entry of the LAST
    iteration must equal, bit for bit, what c2r_photon_sums(xh_intermed, xh_av) gives on the arrays the step leaves."""
    m, a = load_case("evolve32_std_bubbles")
    s, n = m["steps"]["step001"], m["n"]
    b = pkg.HipBackend(n, *tables, device=0)
    b.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
    b.set_sources(s["srcpos"], s["normflux"]); b.set_rank(0, 1)
    b.load(ndens=F(a["step001_ndens"]), xh=F(a["step001_xh_before"]))
    rep = b.evolve3d_native(s["dt"])
    after = b.photon_sums("xh_intermed", "xh_av")
    vol, dt = s["vol"], s["dt"]
    trec, tcol = after[2] * vol * dt, after[3] * vol * dt
    tion = trec + (rep.h0_before - after[0] * vol)
    expect = (tion - tcol) / rep.totalsrc
    assert rep.it_photcons[rep.niter - 1] == expect
    assert all(np.isfinite(rep.it_photcons[k]) and rep.it_photcons[k] != 0.0 for k in range(rep.niter))
    b.close()


def test_failing_allreduce_callback_surfaces_as_ecallback_and_the_context_survives(pkg, tables):
    """c2r_set_rank's collective returns non-zero (a dead peer, a failed ncclAllReduce): c2r_allreduce_rates and
    c2r_evolve3d report C2R_ECALLBACK (-4) with a message, the context stays usable -- a working callback afterwards
    completes the step -- and destroyable."""
    m, a = load_case("evolve32_std_bubbles")
    s, n = m["steps"]["step001"], m["n"]
    lib, ctx = _ctx_for(pkg, tables, s, n)
    nd = F(a["step001_ndens"]); xh = F(a["step001_xh_before"])
    AR = pkg._capi.ALLREDUCE_FN
    calls = []
    bad = AR(lambda user, buf, count, stream: calls.append(count) or 7)
    assert lib.c2r_set_rank(ctx, 0, 2, bad, None) == 0
    assert lib.c2r_upload(ctx, 0, nd.ctypes.data) == 0 and lib.c2r_upload(ctx, 1, xh.ctypes.data) == 0
    assert lib.c2r_allreduce_rates(ctx) == -4 and b"all-reduce callback failed" in lib.c2r_last_error(ctx)
    rep = pkg.Report()
    xh1 = xh.copy()
    assert lib.c2r_evolve3d(ctx, s["dt"], nd.ctypes.data, xh1.ctypes.data, None, None, None, C.byref(rep)) == -4
    assert calls and calls[0] == n ** 3
    # rank 0 of 2 with a collective that adds nothing (the peer's half is zero): the step now runs to the end
    good = AR(lambda user, buf, count, stream: 0)
    assert lib.c2r_set_rank(ctx, 0, 2, good, None) == 0
    xh2 = xh.copy()
    assert lib.c2r_evolve3d(ctx, s["dt"], nd.ctypes.data, xh2.ctypes.data, None, None, None, C.byref(rep)) == 0
    assert rep.niter >= 1 and np.all(np.isfinite(xh2))
    lib.c2r_destroy(ctx)


def test_set_option_is_the_only_way_to_the_schedule_switches(pkg, tables, monkeypatch):
    """c2r_set_option: known names are accepted (and show in c2r_info where it reports them), an unknown name is C2R_EINVAL with a
    message; the library itself ignores the variables the tests' conftest translates (a raw c2r_create under C2R_CHAINS=3 keeps
    the library's own rule)."""
    lib = pkg.load_library()
    monkeypatch.setenv("C2R_CHAINS", "3"); monkeypatch.setenv("C2R_GRAPH", "0")
    p = pkg.default_params(32)
    ctx = C.c_void_p()
    assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
    assert p.sweep_mode == 1 and b"sweep_mode fast" in lib.c2r_info(ctx)            # the default mode
    assert lib.c2r_set_option(ctx, b"no_such_switch", 1.0) == -1 and b"no_such_switch" in lib.c2r_last_error(ctx)
    for name, v in ((b"graph", 0.0), (b"chain_graph", 0.0), (b"chains", 3.0), (b"xcd_order", -1.0), (b"stream_hint", -1.0), (b"batch_cap", 0.0),
                    (b"sparse_fraction", 0.25), (b"exchange_overlap_min", 16.0)):
        assert lib.c2r_set_option(ctx, name, v) == 0, name
    lib.c2r_destroy(ctx)
    # 96 sources under the library's own rule are two chains whatever C2R_CHAINS says in the environment of a raw context
    monkeypatch.delenv("C2R_GRAPH")
    tp = pkg.TestProblem(32); s = tp.step(1)
    pos, nf = pkg.seeded_sources(32, 96, seed=1)
    ctx = C.c_void_p()
    assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
    thick, thin = (np.ascontiguousarray(t) for t in tables)
    assert lib.c2r_set_tables(ctx, thick.ctypes.data, thin.ctypes.data, thick.size) == 0
    dr = (C.c_double * 3)(s["dr1"], s["dr1"], s["dr1"])
    assert lib.c2r_set_step(ctx, C.byref(dr), s["vol"], s["coldensh_LLS"], 1.0, 1e4) == 0
    posc = np.ascontiguousarray(pos, dtype=np.int32); nfc = np.ascontiguousarray(nf)
    assert lib.c2r_set_sources(ctx, posc.ctypes.data, nfc.ctypes.data, 96) == 0
    assert b"; chains 2;" in lib.c2r_info(ctx)
    assert lib.c2r_set_option(ctx, b"chains", 3.0) == 0 and lib.c2r_set_sources(ctx, posc.ctypes.data, nfc.ctypes.data, 96) == 0
    assert b"; chains 3;" in lib.c2r_info(ctx)
    lib.c2r_destroy(ctx)


def test_info_reports_device_mode_and_rank_and_the_library_ignores_the_environment(pkg, tables, monkeypatch):
    """c2r_info: how the device was chosen, the sweep mode that RUNS, rank/nranks.  c2r_params.sweep_mode is the only
    switch of the mode: C2R_SWEEP_MODE in the environment is a host-side convention (HipBackend(fast=None), the Fortran shim),
    and an explicit choice is not overridden by it."""
    lib = pkg.load_library()
    for env, mode, want in (("1", 0, b"exact"), ("0", 1, b"fast")):
        monkeypatch.setenv("C2R_SWEEP_MODE", env)
        p = pkg.default_params(16); p.sweep_mode = mode
        ctx = C.c_void_p()
        assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
        info = lib.c2r_info(ctx)
        assert b"sweep_mode " + want in info and b"device 0 of" in info and b"(explicit)" in info and b"rank 0 of 1" in info
        lib.c2r_destroy(ctx)
    # the Python host: explicit beats the environment, None follows it
    monkeypatch.setenv("C2R_SWEEP_MODE", "1")
    b = pkg.HipBackend(16, *tables, device=0, fast=False)
    assert "sweep_mode exact" in b.info()
    b.close()
    b = pkg.HipBackend(16, *tables, device=0)
    assert "sweep_mode fast" in b.info()
    b.close()
    # C2R_DEVICE_AUTO: which launcher variable decided is part of the line; several ranks and none set is a WARNING
    for var in ("C2R_DEVICE", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "PMI_LOCAL_RANK", "SLURM_LOCALID"):
        monkeypatch.delenv(var, raising=False)
    p = pkg.default_params(16); p.device = -1
    ctx = C.c_void_p()
    assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
    assert b"no local-rank variable is set" in lib.c2r_info(ctx) and b"WARNING" not in lib.c2r_info(ctx)
    AR = pkg._capi.ALLREDUCE_FN
    cb = AR(lambda user, buf, count, stream: 0)
    assert lib.c2r_set_rank(ctx, 1, 4, cb, None) == 0
    assert b"WARNING: C2R_DEVICE_AUTO with nranks > 1" in lib.c2r_info(ctx) and b"rank 1 of 4" in lib.c2r_info(ctx)
    lib.c2r_destroy(ctx)
    monkeypatch.setenv("SLURM_LOCALID", "0")
    ctx = C.c_void_p()
    assert lib.c2r_create(C.byref(ctx), C.byref(p)) == 0
    assert b"SLURM_LOCALID=0" in lib.c2r_info(ctx)
    lib.c2r_destroy(ctx)


@pytest.mark.parametrize("fast", [False, True])
def test_photon_loss_is_bit_reproducible_run_to_run(pkg, tables, fast):
    """130 sources on a structured field: they retire at different sub-boxes, so the active count crosses 64 (where the
    last shell's loss partials change hands between k_loss_reduce and k_box_decide) at a sub-box that depends on the data --
    but not on host timing: the photon loss, the per-source sub-box counts and, with ordered rates, Gamma itself are the same
    bits in every run."""
    from tests.golden.inputs import bubble_xfield, density_factor
    n, S = 64, 130
    tp = pkg.TestProblem(n); s = tp.step(1)
    nd = (tp.fields(1)[0].reshape((n, n, n), order="F") * density_factor(n, 21)).astype(np.float32)
    pos, nf = pkg.seeded_sources(n, S, seed=77)
    xh = bubble_xfield(n, [tuple(int(v) for v in q) for q in pos[:40]], 9.0)
    runs = []
    for rep in range(3):
        b = pkg.HipBackend(n, *tables, device=0, deterministic=True, fast=fast)
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
        b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=F(nd), xh=F(xh)); b.begin_step(); b.zero_rates()
        loss, nbox, vis = b.pass_sources()
        runs.append((loss, nbox, vis, b.last_nbox().copy(), b.fetch("phih_grid")))
        b.close()
    assert len(set(int(v) for v in runs[0][3])) > 2                 # sources do retire at different sub-boxes
    for r in runs[1:]:
        assert r[0] == runs[0][0] and r[1:3] == runs[0][1:3]
        assert np.array_equal(r[3], runs[0][3]) and np.array_equal(r[4], runs[0][4])
