"""Multi-rank runs BELOW Python: tests/native/mgpu_harness (C++, no torch, no MPI) drives the C ABI with one thread and
one context per rank, the way an MPI build of the Fortran driver does with one process per rank -- c2r_set_rank with an
all-reduce callback, c2r_set_balance, c2r_evolve3d on host arrays (mpi.F90:83-160, master_slave.F90:74-96 and
:124-330, evolve.F90:577-616).  On the one-GPU box every rank's context sits on device 0 and the collective is a
rank-ordered sum staged through host memory; where at least two devices are visible the same run goes through
libc2ray_rccl.so (ncclAllReduce), otherwise that case is skipped."""
import os
import subprocess
import numpy as np
import pytest
from tests._util import F, load_case, tol

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]
HARNESS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "mgpu_harness")


def write_input(path, n, s, nd, xh):
    with open(path, "wb") as f:
        np.array([n, len(s["normflux"])], dtype=np.int32).tofile(f)
        np.array([s["dr1"], s["vol"], s["coldensh_LLS"], s["dt"]], dtype=np.float64).tofile(f)
        np.asarray(nd, dtype=np.float32).tofile(f)
        np.asarray(xh, dtype=np.float64).tofile(f)
        np.asarray(s["srcpos"], dtype=np.int32).tofile(f)
        np.asarray(s["normflux"], dtype=np.float64).tofile(f)


def read_output(path, n):
    with open(path, "rb") as f:
        niter, converged, rccl_ranks = np.fromfile(f, np.int32, 3)
        sum_nbox = int(np.fromfile(f, np.int64, 1)[0])
        conv = np.fromfile(f, np.int64, niter)
        loss = float(np.fromfile(f, np.float64, 1)[0])
        xh = np.fromfile(f, np.float64, n ** 3)
        phih = np.fromfile(f, np.float64, n ** 3)
    return dict(niter=int(niter), converged=int(converged), rccl_ranks=int(rccl_ranks), sum_nbox=sum_nbox, conv=list(conv), loss=loss, xh=xh, phih=phih)


def run(tmp_path, tag, nranks, coll, balance, slab=0, stats=None):
    out = str(tmp_path / ("out_%s.bin" % tag))
    p = subprocess.run([HARNESS, str(tmp_path / "in.bin"), out, str(nranks), coll, str(int(balance)), str(int(slab))],
                       capture_output=True, text=True, timeout=600)
    if p.returncode == 77:
        pytest.skip(p.stdout.strip())
    assert p.returncode == 0, p.stdout + p.stderr
    if stats is not None:        # rank 0's c2r_exchange_stats
        w = [l for l in p.stdout.split("\n") if l.startswith("exchange:")][0].split()
        stats.update(calls=int(w[2]), packed=int(w[4]), bytes_last=int(w[6]), bytes_total=int(w[8]))
    return out


@pytest.mark.parametrize("case", ["evolve32_std_bubbles", "evolve64_std_bubbles"])
def test_ranks_as_threads_match_the_reference_step(tmp_path, case):
    assert os.path.exists(HARNESS), "build it: make -C tests/native (or __graft_entry__.build())"
    m, a = load_case(case)
    n, s = m["n"], m["steps"]["step001"]
    write_input(str(tmp_path / "in.bin"), n, s, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    ref = F(a["step001_xh_after"])
    res = {}
    for tag, nranks, coll, bal in (("1", 1, "host", 0), ("2", 2, "host", 0), ("3b", 3, "host", 1), ("4b", 4, "host", 1)):
        r = read_output(run(tmp_path, tag, nranks, coll, bal), n)
        assert r["converged"] and r["niter"] == s["niter"], tag
        assert r["conv"] == s["log"]["nonconv"], tag
        assert r["sum_nbox"] == s["sum_nbox_all"], tag
        assert abs(r["loss"] - s["photon_loss_all"]) <= tol("loss") * abs(s["photon_loss_all"]) + 1e-300, tag
        assert np.max(np.abs(r["xh"] - ref)) < tol("x"), tag
        res[tag] = r
    for tag in ("2", "3b", "4b"):        # the partition changes only the order of the sums
        assert np.max(np.abs(res[tag]["xh"] - res["1"]["xh"])) < 1e-12
        nz = res["1"]["phih"] != 0
        assert np.array_equal(res[tag]["phih"] != 0, nz)
        assert np.max(np.abs(res[tag]["phih"][nz] / res["1"]["phih"][nz] - 1)) < 1e-10


@pytest.mark.parametrize("case", ["evolve32_std_bubbles", "evolve64_std_bubbles"])
def test_slab_chemistry_equals_the_replicated_global_pass(tmp_path, case, monkeypatch):
    """c2r_set_slab_chemistry (reduce-scatter of Gamma by z-slabs, evolve0D_global on the own slab, all-gather of
    xh_av / xh_intermed) against the all-reduce + replicated global pass of evolve.F90:548-555, :599: 2, 3 (32 planes do not
    divide by 3: uneven slabs) and 4 ranks as threads with host-staged collectives in rank order -- the same iteration
    history, sub-box counts and photon loss, xh and the final phih_grid BIT FOR BIT (the convergence sums are taken over the
    gathered arrays exactly as the replicated pass takes them)."""
    assert os.path.exists(HARNESS)
    monkeypatch.setenv("C2R_HARNESS_DETERMINISTIC", "1")     # ordered per-source sums instead of atomics: two RUNS are comparable bit for bit
    m, a = load_case(case)
    n, s = m["n"], m["steps"]["step001"]
    write_input(str(tmp_path / "in.bin"), n, s, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    for nranks, bal in ((2, 0), (3, 1), (4, 0)):
        rep = read_output(run(tmp_path, "rep%d" % nranks, nranks, "host", bal, 0), n)
        slb = read_output(run(tmp_path, "slab%d" % nranks, nranks, "host", bal, 1), n)
        assert slb["converged"] and slb["niter"] == rep["niter"] == s["niter"]
        assert slb["conv"] == rep["conv"] == s["log"]["nonconv"]
        assert slb["sum_nbox"] == rep["sum_nbox"] and slb["loss"] == rep["loss"]
        assert np.array_equal(slb["xh"], rep["xh"]), nranks
        assert np.array_equal(slb["phih"], rep["phih"]), nranks


# a cold one-source start (41 iterations inside sub-box 1: 4 % of the mesh travels); ten sources in bubbles whose boxes overlap and
# add up to more than the mesh (packed all the same, by a fraction no run would use: every overlapping cell travels several times)
@pytest.mark.parametrize("case,fraction", [("evolve32_onesrc", "0.5"), ("evolve64_std_bubbles", "50")])
def test_sparse_exchange_equals_the_full_all_reduce(tmp_path, case, fraction, monkeypatch):
    """c2r_allreduce_rates packs the sources' final sub-boxes while they are a small part of the mesh (evolve.F90:599 reduces
    all of phih_grid whatever it holds).  Same step with C2R_SPARSE_EXCHANGE=0 (always the whole grid): 2, 3 and 4 ranks as
    threads with the rank-ordered host-staged sum -- iteration history, sub-box counts, photon loss, xh and phih_grid BIT FOR
    BIT; the packed calls moved fewer bytes."""
    assert os.path.exists(HARNESS)
    monkeypatch.setenv("C2R_HARNESS_DETERMINISTIC", "1")     # ordered per-source sums: two RUNS are comparable bit for bit
    m, a = load_case(case)
    n, s = m["n"], m["steps"]["step001"]
    write_input(str(tmp_path / "in.bin"), n, s, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    for nranks, bal in ((2, 0), (3, 1), (4, 0)):
        monkeypatch.setenv("C2R_SPARSE_EXCHANGE", "0")
        st_full, st_sp = {}, {}
        full = read_output(run(tmp_path, "full%d" % nranks, nranks, "host", bal, 0, st_full), n)
        monkeypatch.setenv("C2R_SPARSE_EXCHANGE", "1")
        monkeypatch.setenv("C2R_SPARSE_FRACTION", fraction)
        sp = read_output(run(tmp_path, "sparse%d" % nranks, nranks, "host", bal, 0, st_sp), n)
        assert sp["converged"] and sp["niter"] == full["niter"] == s["niter"]
        assert sp["conv"] == full["conv"] == s["log"]["nonconv"]
        assert sp["sum_nbox"] == full["sum_nbox"] and sp["loss"] == full["loss"]
        assert np.array_equal(sp["xh"], full["xh"]), nranks
        assert np.array_equal(sp["phih"], full["phih"]), nranks
        assert st_full["packed"] == 0 and st_full["calls"] == st_sp["calls"] == s["niter"]
        assert st_sp["packed"] >= 1 and (fraction != "0.5" or st_sp["bytes_total"] < 0.1 * st_full["bytes_total"])


def test_sparse_unpack_agrees_on_every_rank_when_the_collective_sums_in_an_offset_dependent_order(tmp_path, monkeypatch):
    """A cell of several overlapping sub-boxes travels once per box; a ring / tree all-reduce (RCCL) sums each copy in an order
    that depends on where in the buffer it lies, so the copies can come back differing in the last bit.  The write-back takes
    the maximum of the copies with a 64-bit atomic (k_pack_boxes), whichever lands last: phih_grid -- and with it conv_flag
    and the moment a rank leaves the evolve3D loop -- stays the same on every rank.  Harness collective: element i summed
    starting at rank i mod nranks (C2R_HARNESS_ROTATE_SUM); three and four ranks, ten sources whose boxes overlap and are
    packed although they add up to more than the mesh; the harness itself fails when the ranks' xh / phih_grid bits differ."""
    assert os.path.exists(HARNESS)
    m, a = load_case("evolve64_std_bubbles")
    n, s = m["n"], m["steps"]["step001"]
    write_input(str(tmp_path / "in.bin"), n, s, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    monkeypatch.setenv("C2R_HARNESS_ROTATE_SUM", "1")
    monkeypatch.setenv("C2R_SPARSE_FRACTION", "50")
    for nranks in (3, 4):
        st = {}
        r = read_output(run(tmp_path, "rot%d" % nranks, nranks, "host", 1, 0, st), n)     # (returns non-zero unless the replicas are identical)
        assert st["packed"] >= 1
        assert r["converged"] and r["niter"] == s["niter"] and r["conv"] == s["log"]["nonconv"]
        assert np.max(np.abs(r["xh"] - F(a["step001_xh_after"]))) < tol("x")


@pytest.mark.parametrize("case", ["evolve32_std_bubbles", "evolve64_std_bubbles"])
def test_exchange_overlapped_with_the_sweep_equals_the_plain_exchange(tmp_path, case, monkeypatch):
    """c2r_set_exchange_overlap (here through C2R_EXCHANGE_OVERLAP=1, and the 64-sources-per-rank threshold lowered for the
    ten-source fixtures): every pass as two halves of a rank's sources into two pairs of accumulators, the first half's
    all-reduce handed to the callback on a second stream before the second half is swept, the two reduced halves added.
    Against the plain pass + all-reduce of the whole grid (C2R_SPARSE_EXCHANGE=0 in both): same iteration history, sub-box
    counts, photon loss; xh and phih_grid to the re-association of the sums."""
    assert os.path.exists(HARNESS)
    m, a = load_case(case)
    n, s = m["n"], m["steps"]["step001"]
    write_input(str(tmp_path / "in.bin"), n, s, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    monkeypatch.setenv("C2R_SPARSE_EXCHANGE", "0")
    monkeypatch.setenv("C2R_EXCHANGE_OVERLAP_MIN", "1")
    for nranks, bal in ((2, 0), (3, 1), (4, 0)):
        monkeypatch.setenv("C2R_EXCHANGE_OVERLAP", "0")
        st0, st1 = {}, {}
        plain = read_output(run(tmp_path, "plain%d" % nranks, nranks, "host", bal, 0, st0), n)
        monkeypatch.setenv("C2R_EXCHANGE_OVERLAP", "1")
        out = run(tmp_path, "ovl%d" % nranks, nranks, "host", bal, 0, st1)
        ov = read_output(out, n)
        assert ov["converged"] and ov["niter"] == plain["niter"] == s["niter"]
        assert ov["conv"] == plain["conv"] == s["log"]["nonconv"]
        assert ov["sum_nbox"] == plain["sum_nbox"]
        assert abs(ov["loss"] - plain["loss"]) <= 1e-13 * abs(plain["loss"])
        assert np.max(np.abs(ov["xh"] - plain["xh"])) < 1e-12
        nz = plain["phih"] != 0
        assert np.array_equal(ov["phih"] != 0, nz) and np.max(np.abs(ov["phih"][nz] / plain["phih"][nz] - 1)) < 1e-10     # (the last pass starts from an xh_av that differs by 1e-13: n_HI = (1 - x) n amplifies it)
        # every pass exchanged two halves of a grid instead of one grid
        assert st1["calls"] == st0["calls"] == s["niter"] and st1["bytes_total"] == 2 * st0["bytes_total"]


def test_two_ranks_over_rccl(tmp_path):
    """ncclAllReduce of phih_grid between two devices of the node (libc2ray_rccl.so); skipped on a one-GPU box."""
    assert os.path.exists(HARNESS)
    m, a = load_case("evolve32_std_bubbles")
    n, s = m["n"], m["steps"]["step001"]
    write_input(str(tmp_path / "in.bin"), n, s, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    r = read_output(run(tmp_path, "rccl", 2, "rccl", 1), n)
    assert r["rccl_ranks"] == 2 and r["niter"] == s["niter"] and r["conv"] == s["log"]["nonconv"]
    assert np.max(np.abs(r["xh"] - F(a["step001_xh_after"]))) < tol("x")
    # the same over grouped ncclReduce / ncclBroadcast (slab chemistry)
    r2 = read_output(run(tmp_path, "rccl_slab", 2, "rccl", 1, 1), n)
    assert r2["niter"] == r["niter"] and r2["conv"] == r["conv"] and np.max(np.abs(r2["xh"] - r["xh"])) < 1e-12


def test_allfrac_ranks_slabs_and_packed_exchange(tmp_path, monkeypatch):
    """A driver built with -DALLFRAC over several ranks (the harness hands c2r_evolve3d its (mesh,0:1) arrays, c2r_params.allfrac):
    the step of the -DALLFRAC reference fixture on 1 rank, on 2 and 3 ranks with the all-reduce, and with slab chemistry (the stored
    neutral halves of xh_av / xh_intermed are gathered with the ionized ones) -- iteration history of the reference, both halves of
    xh within tol("x") of it, and with ordered rates the same bits with slab chemistry as with the replicated pass."""
    assert os.path.exists(HARNESS)
    monkeypatch.setenv("C2R_HARNESS_ALLFRAC", "1")
    monkeypatch.setenv("C2R_HARNESS_DETERMINISTIC", "1")
    m, a = load_case("evolve32_allfrac")
    n, s = m["n"], m["steps"]["step001"]
    write_input(str(tmp_path / "in.bin"), n, s, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    F(a["step001_xh_before0"]).tofile(str(tmp_path / "in.bin.x0"))

    def read_allfrac(path):
        r = read_output(path, n)
        with open(path, "rb") as f:
            f.seek(-8 * n ** 3, 2)
            r["xh_neutral"] = np.fromfile(f, np.float64, n ** 3)
        return r
    res = {}
    for tag, nranks, slab in (("1", 1, 0), ("2", 2, 0), ("3", 3, 0), ("2s", 2, 1), ("3s", 3, 1)):
        r = read_allfrac(run(tmp_path, tag, nranks, "host", 0, slab))
        assert r["converged"] and r["niter"] == s["niter"] and r["conv"] == s["log"]["nonconv"], tag
        assert r["sum_nbox"] == s["sum_nbox_all"], tag
        assert np.max(np.abs(r["xh"] - F(a["step001_xh_after"]))) < tol("x"), tag
        assert np.max(np.abs(r["xh_neutral"] - F(a["step001_xh_after0"]))) < tol("x"), tag
        res[tag] = r
    for tag in ("2", "3"):        # slab chemistry against the replicated pass at the same rank count: the same bits
        for k in ("xh", "xh_neutral", "phih"):
            assert np.array_equal(res[tag + "s"][k], res[tag][k]), (tag, k)
        # ... and against one rank: the partition changes only the association of the sums over ranks
        assert np.max(np.abs(res[tag]["xh"] - res["1"]["xh"])) < 1e-12 and np.max(np.abs(res[tag]["xh_neutral"] - res["1"]["xh_neutral"])) < 1e-12


@pytest.mark.parametrize("chunk", ["1", "3", "64"])
def test_sources_handed_out_on_request(tmp_path, chunk, monkeypatch):
    """c2r_set_source_queue -- do_grid_master / do_grid_slave of master_slave.F90:124-330: every rank asks a first-come-first-served
    queue (two atomics in the harness; MPI_Send / MPI_Recv with rank 0 in the reference) for `chunk` more sources whenever it has
    swept what it had.  Whoever sweeps what: the step's iteration history, sub-box sums and photon loss are the reference's, xh
    within tol("x"); against the static rule xh to the association of the sums over ranks; every source is taken exactly once per
    pass (chunk 1 = the reference's one source per request; chunk 64 > the list: one rank takes everything)."""
    assert os.path.exists(HARNESS)
    m, a = load_case("evolve32_std_bubbles")
    n, s = m["n"], m["steps"]["step001"]
    write_input(str(tmp_path / "in.bin"), n, s, F(a["step001_ndens"]), F(a["step001_xh_before"]))
    static = read_output(run(tmp_path, "static", 3, "host", 0), n)
    monkeypatch.setenv("C2R_HARNESS_QUEUE", chunk)
    for nranks in (1, 3, 4):
        out = str(tmp_path / ("out_q%d.bin" % nranks))
        p = subprocess.run([HARNESS, str(tmp_path / "in.bin"), out, str(nranks), "host", "0", "0"], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        taken = [int(v) for v in [l for l in p.stdout.split("\n") if l.startswith("queue:")][0].split(":")[-1].split()]
        r = read_output(out, n)
        assert r["converged"] and r["niter"] == s["niter"] and r["conv"] == s["log"]["nonconv"], nranks
        assert len(taken) == nranks and sum(taken) == len(s["normflux"]) * r["niter"], (taken, r["niter"])
        assert r["sum_nbox"] == s["sum_nbox_all"]
        assert abs(r["loss"] - s["photon_loss_all"]) <= tol("loss") * abs(s["photon_loss_all"]) + 1e-300
        assert np.max(np.abs(r["xh"] - F(a["step001_xh_after"]))) < tol("x")
        assert np.max(np.abs(r["xh"] - static["xh"])) < 1e-12
        assert "replicas identical: yes" in p.stdout
