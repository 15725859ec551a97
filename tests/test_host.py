"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host
mirror's logic (source sharding, outer loop, multi-rank reduction over gloo) on a CPU test
double, and the harness scalars against the reference's recorded values."""
import os
import re
import subprocess
import sys
import numpy as np
import pytest
from tests._util import F, load_case, oracle_for, load_tables, relerr, GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


def test_library_exports_every_declared_symbol(pkg):
    """include/c2ray_hip.h is the contract: each function it declares must resolve in the .so,
    and the ctypes table must cover exactly that list."""
    hdr = open(os.path.join(ROOT, "include", "c2ray_hip.h")).read()
    declared = set(re.findall(r"\b(c2r_\w+)\s*\(", hdr)) - {"c2r_allreduce_fn"}
    from c2ray3dm_amd import _capi
    typed = {s[0] for s in _capi.SYMBOLS}
    assert declared == typed, (declared ^ typed)
    lib = pkg.load_library()
    for name in declared:
        assert hasattr(lib, name)
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.LIB_PATH]).decode()
    exported = set(re.findall(r"\bT (c2r_\w+)", out))
    assert declared <= exported


def test_rccl_binding_exports_its_header(pkg):
    """include/c2ray_rccl.h (optional RCCL all-reduce callback) against libc2ray_rccl.so."""
    hdr = open(os.path.join(ROOT, "include", "c2ray_rccl.h")).read()
    declared = set(re.findall(r"\b(c2r_rccl_\w+)\s*\(", hdr))
    assert declared == {"c2r_rccl_unique_id", "c2r_rccl_attach", "c2r_rccl_allreduce", "c2r_rccl_slab_chemistry", "c2r_rccl_detach"}
    lib = os.path.join(os.path.dirname(pkg.LIB_PATH), "libc2ray_rccl.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    assert declared <= set(re.findall(r"\bT (c2r_rccl_\w+)", out))
    # the core library itself must not depend on RCCL
    needed = subprocess.check_output(["readelf", "-d", pkg.LIB_PATH]).decode()
    assert "rccl" not in needed


def test_struct_layout_matches_header(pkg):
    """sizeof() of the two ABI structs as the C compiler sees them."""
    src = ('#include <stdio.h>\n#include "c2ray_hip.h"\nint main(){printf("%zu %zu %zu %zu\\n",sizeof(c2r_params),sizeof(c2r_report),'
           'sizeof(c2r_thermal_params),sizeof(c2r_sed_params));return 0;}\n')
    exe = "/tmp/c2r_sizeof"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    a, b, c, d = map(int, subprocess.check_output([exe]).split())
    import ctypes as C
    assert C.sizeof(pkg.Params) == a and C.sizeof(pkg.Report) == b
    assert C.sizeof(pkg._capi.ThermalParams) == c and C.sizeof(pkg.SedParams) == d


def test_default_thermal_params_are_the_reference_constants(pkg):
    """c2ray_parameters.f90:105-110, atomic.f90:23-25, radiation_photoionrates.F90:333, evolve_point.F90:387-388."""
    import ctypes as C
    t = pkg._capi.ThermalParams()
    assert pkg.load_library().c2r_default_thermal(C.byref(t)) == 0
    assert t.minitemp == 1.0 and t.relative_denergy == float(np.float32(0.1)) and t.thermal_max_steps == 10000
    assert t.gamma1 == 5.0 / 3.0 - 1.0 and t.tau_heat_limit == float(np.float32(1.0e-4))
    assert t.temp_conv_rel == 0.1 and t.temp_conv_abs == 100.0 and t.cool_points == 61 and t.cosmological == 1
    assert t.thermal_time_tol == float(np.float32(1e-6)) and t.thermal_rate_floor == 1e-50
    assert t.Omega0 == float(np.float32(0.27)) and abs(t.H0 / (float(np.float32(0.7)) * 100.0 * 1e5 / 3.08600011031262003e+24) - 1) < 1e-15


def test_default_params_are_the_reference_constants(pkg):
    p = pkg.default_params(32)
    assert tuple(p.mesh) == (32, 32, 32)
    assert p.subboxsize == 5 and p.max_subbox == 1000 and p.numtau == 2000
    assert p.sigma_HI == float(np.float32(6.30e-18)) and p.pi == float(np.float32(3.141592654))
    assert p.loss_fraction == 1e-2 and p.epsilon == 1e-14
    assert p.sweep_mode == 1          # C2R_SWEEP_FAST: the library (and drop-in) default since round 6; 0 = C2R_SWEEP_EXACT is the opt-in


def test_no_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.C2RayHipError):
        pkg.HipBackend(8, *load_tables())


def test_static_source_share(pkg):
    # master_slave.F90:85  do ns1=1+rank,NumSrc,npr
    assert pkg.static_source_share(10, 0, 3) == [0, 3, 6, 9]
    assert pkg.static_source_share(10, 2, 3) == [2, 5, 8]
    assert pkg.static_source_share(2, 3, 4) == []
    allsrc = sorted(sum((pkg.static_source_share(17, r, 5) for r in range(5)), []))
    assert allsrc == list(range(17))


def test_harness_scalars_match_reference_step1(pkg):
    """TestProblem restates the driver-side set-up; compare with what the reference fed evolve3D."""
    m, a = load_case("evolve32_onesrc")
    for tag, step in (("step001", 1), ("step002", 2), ("step003", 3), ("step012", 12)):
        s = m["steps"][tag]
        t = pkg.TestProblem(32).step(step)
        assert abs(t["dt"] / s["dt"] - 1) < 1e-7          # dt: the reference goes through z(t) and back
        for k in ("dr1", "vol", "coldensh_LLS", "zred"):
            assert abs(t[k] / s[k] - 1) < 1e-7, (tag, k, t[k], s[k])
        assert abs(t["ndens"] / float(a[tag + "_ndens"].flat[0]) - 1) < 1e-6


def test_seeded_sources_are_reproducible_and_distinct(pkg):
    p1, f1 = pkg.seeded_sources(64, 100)
    p2, f2 = pkg.seeded_sources(64, 100)
    assert np.array_equal(p1, p2) and np.array_equal(f1, f2)
    assert len({tuple(r) for r in p1}) == 100 and p1.min() >= 1 and p1.max() <= 64
    assert f1.min() >= 1e6 and f1.max() <= 1e9


def test_evolve_loop_on_cpu_double_matches_reference(pkg):
    """The host mirror's piecewise entries (Evolve.iteration / pass_all_sources / global_pass ...) driven through a whole step
    by the oracle-backed test double and its loop (tests/_cpu_backend.py evolve3d_piecewise; the product's one outer loop is the
    library's, which needs a GPU) reproduce the reference's iteration history exactly."""
    from tests._cpu_backend import OracleBackend, evolve3d_piecewise
    tables = load_tables()
    m, a = load_case("evolve32_std_bubbles")
    for tag, s in m["steps"].items():
        b = OracleBackend(oracle_for(s, tables, m["n"]), F(a[tag + "_ndens"]), F(a[tag + "_xh_before"]),
                          s["srcpos"], s["normflux"])
        r = evolve3d_piecewise(pkg.Evolve(b), 0.0, s["dt"], 0)
        assert r["niter"] == s["niter"] and r["converged"]
        assert [e["conv_flag"] for e in r["log"]] == s["log"]["nonconv"]
        assert np.array_equal(b.xh, F(a[tag + "_xh_after"]))
        assert r["sum_nbox_all"] == s["sum_nbox_all"] and r["photon_loss_all"] == s["photon_loss_all"]


@pytest.mark.parametrize("name", ["restart32_std_bubbles", "restart32_onesrc"])
def test_restart_from_iteration_dump_on_cpu_double(pkg, tmp_path, name):
    """restart=3: Evolve.start_from_dump reads iterdump.bin (the reference's record layout) and the step resumes exactly
    as the reference did from the same file (fixture generated by the reference reading OUR dump)."""
    from tests._cpu_backend import OracleBackend, evolve3d_piecewise
    tables = load_tables()
    m, a = load_case(name)
    b = OracleBackend(oracle_for(m, tables, m["n"]), F(a["ndens"]), F(a["xh_before"]), m["srcpos"], m["normflux"])
    pkg.fileio.write_iteration_dump(str(tmp_path / "iterdump.bin"), m["dump_niter"], m["dump_photon_loss_all"],
                                    a["dump_phih"], a["dump_xh_av"], a["dump_xh_intermed"])
    ev = pkg.Evolve(b)
    ev.dump_dir = str(tmp_path)
    r = evolve3d_piecewise(ev, 0.0, m["dt"], 3)
    assert r["converged"] and [e["conv_flag"] for e in r["log"]] == m["log"]["nonconv"]
    assert np.array_equal(b.xh, F(a["xh_after"])) and np.array_equal(b.phih_grid, F(a["phih_grid"]))


def test_iteration_dump_then_restart_reproduces_the_rest_of_the_step(pkg, tmp_path):
    """Dumps written by the loop itself (dump interval 0: after every pass) alternate between
    iterdump1.bin and iterdump2.bin (evolve.F90:296-301) and a restart from the last one finishes the step
    with the same answer as the reference's restart semantics predict (one extra global pass)."""
    from tests._cpu_backend import OracleBackend, evolve3d_piecewise
    tables = load_tables()
    m, a = load_case("restart32_onesrc")
    o = oracle_for(m, tables, m["n"])
    b = OracleBackend(o, F(a["ndens"]), F(a["xh_before"]), m["srcpos"], m["normflux"])
    ev = pkg.Evolve(b); ev.dump_dir = str(tmp_path); ev.dump_interval_s = 0.0
    r = evolve3d_piecewise(ev, 0.0, m["dt"], 0)
    assert os.path.exists(tmp_path / "iterdump1.bin") and os.path.exists(tmp_path / "iterdump2.bin")
    last = 2 if r["niter"] % 2 == 0 else 1
    niter, loss, phih, xav, xint = pkg.fileio.read_iteration_dump(str(tmp_path / ("iterdump%d.bin" % last)), m["n"])
    assert niter == r["niter"]
    # the dump holds the state between the last pass over the sources and its global pass: restarting
    # from it redoes that global pass and must land on the same converged state
    b2 = OracleBackend(o, F(a["ndens"]), F(a["xh_before"]), m["srcpos"], m["normflux"])
    ev2 = pkg.Evolve(b2); ev2.dump_dir = str(tmp_path); ev2.dump_interval_s = None
    r2 = evolve3d_piecewise(ev2, 0.0, m["dt"], last)
    assert r2["converged"]
    # not bitwise: after a restart the previous-sum variables are zero, so at least one more pass is
    # forced and the loop stops at a different point of the same 1e-4 convergence criterion
    assert abs(b2.xh.sum() / b.xh.sum() - 1) < 1e-3
    assert np.max(np.abs(b2.xh - b.xh)) < 1e-2


def test_balanced_source_shares(pkg):
    """LPT partition by the previous pass's sub-box volume: a partition, deterministic, and far more
    even than the static stride when a few sources dominate."""
    rng = np.random.default_rng(0)
    nbox = rng.integers(1, 27, 200)
    nbox[::8] = 26                                    # the static stride gives rank 0 all the expensive ones
    cost = pkg.box_cost(nbox, (256, 256, 256))
    assert cost[nbox == 26][0] == 256 ** 3 and pkg.box_cost(np.array([0, 1]), (32, 32, 32)).tolist() == [0, 11 ** 3]
    sh = pkg.balanced_source_shares(cost, 8)
    assert sorted(sum(sh, [])) == list(range(200))
    assert sh == pkg.balanced_source_shares(cost, 8)
    load = np.array([cost[s].sum() for s in sh], dtype=np.float64)
    static = np.array([cost[pkg.static_source_share(200, r, 8)].sum() for r in range(8)], dtype=np.float64)
    assert load.max() / load.mean() < 1.02 < static.max() / static.mean()


def test_library_partition_equals_python_partition(pkg):
    """c2r_balanced_shares -- the partition c2r_set_balance uses inside the library, so that hosts which only call
    c2r_evolve3d (the Fortran shim) are balanced too -- is the same function as evolve.balanced_source_shares
    (pure host code: needs no GPU)."""
    import ctypes as C
    lib = pkg.load_library()
    rng = np.random.default_rng(1)
    for trial in range(40):
        n, npr = int(rng.integers(0, 300)), int(rng.integers(1, 12))
        cost = (rng.integers(0, 1000, n) * rng.integers(0, 2, n)).astype(np.int64)      # many ties and zeros
        ref = pkg.balanced_source_shares(cost, npr)
        seen = []
        for r in range(npr):
            idx, m = np.zeros(max(n, 1), dtype=np.int32), C.c_int32()
            assert lib.c2r_balanced_shares(cost.ctypes.data, n, npr, r, idx.ctypes.data, C.byref(m)) == 0
            assert idx[:m.value].tolist() == ref[r]
            seen += idx[:m.value].tolist()
        assert sorted(seen) == list(range(n))
    assert lib.c2r_balanced_shares(None, 3, 2, 2, None, None) == -1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["static", "balance"])
def test_two_ranks_gloo_equals_one_rank(pkg, tmp_path, mode):
    """world_size=2 over gloo: sources sharded 1+rank,NumSrc,npr, Gamma/photon-loss/nbox summed with
    all_reduce, every rank runs the global pass.  Result must equal the single-rank run up to the
    re-association of the Gamma sum (the reference's MPI path has the same property)."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    out = tmp_path / "out.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29731", script, str(out), mode],
                          env=env, cwd=ROOT, timeout=280)
    got = np.load(out)
    from tests._cpu_backend import OracleBackend, evolve3d_piecewise
    tables = load_tables()
    m, a = load_case("evolve32_std_bubbles")
    s = m["steps"]["step001"]
    b = OracleBackend(oracle_for(s, tables, m["n"]), F(a["step001_ndens"]), F(a["step001_xh_before"]),
                      s["srcpos"], s["normflux"])
    r = evolve3d_piecewise(pkg.Evolve(b), 0.0, s["dt"], 0)
    assert int(got["niter"]) == r["niter"]
    assert int(got["sum_nbox_all"]) == r["sum_nbox_all"]
    assert abs(float(got["photon_loss_all"]) / r["photon_loss_all"] - 1) < 1e-14
    assert np.max(np.abs(got["xh"] - b.xh)) < 1e-12
    # Gamma behind an ionization front is exponentially sensitive to the column in front of it:
    # the 1e-16 re-association noise of the rank-wise sum grows to ~1e-10 over the iterations
    assert relerr(got["phih"], b.phih_grid, floor=1e-60) < 1e-8
    assert np.array_equal(got["xh_rank1"], got["xh"])       # ranks agree bit for bit


@pytest.mark.timeout(300)
def test_two_ranks_gloo_nonisothermal_step(pkg, tmp_path):
    """The non-isothermal step on two ranks over gloo (CPU test double): the heating rates are sharded with the sources and
    all-reduced next to Gamma (evolve.F90:604-609), the temperature evolution is replicated with the global pass, and
    Evolve.accept finalises the temperatures -- against the reference's fixture."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    out = tmp_path / "out.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29733")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29733", script, str(out), "thermal"],
                          env=env, cwd=ROOT, timeout=280)
    got = np.load(out)
    m, a = load_case("evolve32_thermal")
    s = m["steps"]["step001"]
    assert int(got["niter"]) == s["niter"] and list(got["conv"]) == s["log"]["nonconv"]
    assert int(got["sum_nbox_all"]) == s["sum_nbox_all"]
    assert np.max(np.abs(got["xh"] - F(a["step001_xh_after"]))) < 1e-12
    assert np.max(np.abs(got["temper"].astype(np.float64) / a["step001_temper_after"] - 1)) <= 1.5e-7
    assert np.array_equal(got["temper"][:, 0], got["temper"][:, 2])
    ref = F(a["step001_phiheat_grid"])
    assert np.array_equal(got["heat"] == 0, ref == 0) and relerr(got["heat"], ref, floor=1e-60) < 1e-8
    assert np.array_equal(got["xh_rank1"], got["xh"])


@pytest.mark.timeout(300)
def test_slab_collectives_contract_two_ranks_gloo(tmp_path):
    """evolve.slab_collectives (what Evolve(slab=True) registers for the slab chemistry) over gloo, two ranks, unequal slabs:
    reduce-scatter leaves the sum over ranks in the own slab, all-gather leaves every owner's bytes everywhere."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    out = tmp_path / "slab.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29735")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29735", script, str(out), "slab"],
                          env=env, cwd=ROOT, timeout=280)
    got = np.load(out)
    assert np.array_equal(got["own0"], 3.0 * np.arange(0, 6)) and np.array_equal(got["own1"], 3.0 * np.arange(6, 10))
    assert np.array_equal(got["bytes"], np.array([7] * 24 + [8] * 16, dtype=np.uint8))


def test_bench_plain_multi_gpu_form_is_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts its ranks as child processes (torch.distributed.run) instead of
    refusing; here, without a GPU, both ranks fail loudly (no CPU fallback) and the parent reports that with a non-zero
    exit code and no JSON line -- and it got there without importing torch itself."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mesh", "16", "--sources", "2",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "2-rank child run failed" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # (since round 6 the ranks' own preflight -- fewer visible devices than ranks -- stops them before anything else does)
    assert "GPU(s) visible" in r.stderr or "No HIP GPUs are available" in r.stderr or "no GPU visible" in r.stderr or "no HIP device" in r.stderr
