"""GPU tests of the Fortran call surface piece by piece: oracle/ref_driver.F90 -- the program that
produced the fixtures by calling the REFERENCE's do_source / evolve3D -- linked instead with the
shim's modules `evolve_source` and `evolve` (c2-ray3dm_amd/fortran/evolve_hip.F90 + libc2ray_hip.so;
oracle/ref_build.sh: ref_driver_hip).  Same inputs, same dumps, compared with the fixtures:
  * do_source(dt,ns,niter) per source: phih_grid accumulation, coldensh_out of one source,
    photon_loss(1), sum_nbox                                       (evolve_source.F90:58-221)
  * do_grid(dt,niter) for all sources at once and evolve0D_global over the mesh
                                                                   (master_slave.F90:53, evolve_point.F90:305)
  * evolve0D(dt,rtpos,ns,niter) and evolve0D_global(dt,pos,conv_flag) cell by cell   (evolve_point.F90:83, :305)
  * evolve3D(time,dt,0) in builds with type_of_LLS=2,3 / type_of_clumping=5 (the shim forwards
    LLS_grid / R_max_LLS / clumping_grid)
  * evolve3D(time,dt,3): the shim's start_from_dump reads iterdump.bin (evolve.F90:328-426)
The binaries contain compiled reference objects and live in the git-ignored oracle/_ref/."""
import json
import os
import shutil
import sys
import tempfile
import numpy as np
import pytest
from tests._util import GOLDEN, relerr

sys.path.insert(0, GOLDEN)
import inputs as gi      # noqa: E402  (tests/golden/inputs.py: seeded inputs + run-directory writer)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sweep_mode")]     # once per sweep mode (C2R_SWEEP_MODE reaches child processes too)


def exe(n, variant=None):
    return os.path.join(gi.REF, "N%d%s" % (n, "_" + variant if variant else ""), "hip", "ref_driver_hip")


def need(n, variant=None):
    if not os.path.exists(exe(n, variant)):
        pytest.skip("ref_driver_hip not built (needs the reference: oracle/ref_build.sh %d%s)"
                    % (n, ":" + variant if variant else ""))


@pytest.fixture
def rundir():
    d = tempfile.mkdtemp(prefix="c2r_pieces_")
    yield d
    shutil.rmtree(d, ignore_errors=True)


BUBBLES = [(18, 18, 18), (20, 10, 10), (6, 6, 18)]


def test_fortran_do_source_matches_reference_sweep(rundir):
    need(32)
    m = json.load(open(os.path.join(GOLDEN, "sweep32_bubbles.json")))
    a = np.load(os.path.join(GOLDEN, "sweep32_bubbles.npz"))
    d = gi.run_driver(32, gi.SRC_STD, {"mode": "'sweep'", "ns_dump": m["ns_dump"]}, dens=gi.density_factor(32, 5),
                      xfield=gi.bubble_xfield(32, BUBBLES, 7.0), hip=True, d=rundir)
    assert np.array_equal(gi.rd(d, "step001_ndens.f32", 32, np.float32), a["ndens"])      # same inputs
    assert np.array_equal(gi.rd(d, "step001_xh_before.f64", 32), a["xh"])
    kv = gi.read_kv(d + "/dump/step001_sweep.txt")
    assert kv["sum_nbox"] == m["sum_nbox"]
    assert abs(kv["photon_loss"] - m["photon_loss"]) <= 1e-10 * abs(m["photon_loss"])
    cd = gi.rd(d, "step001_coldensh_out.f64", 32)
    assert np.array_equal(cd == 0.0, a["coldensh_out"] == 0.0)                            # same cells reached
    assert relerr(cd, a["coldensh_out"]) < 1e-11
    assert relerr(gi.rd(d, "step001_phih_grid.f64", 32), a["phih"]) < 1e-9


def test_fortran_do_grid_and_evolve0d_global_all_match_the_reference_modules(rundir):
    """The rest of the reference's call surface on this path: master_slave_processing::do_grid(dt,niter) (master_slave.F90:53,
    one GPU pass for all sources instead of NumSrc do_source calls) and evolve_point (local_chemistry, and evolve0D_global for
    the whole mesh: evolve0D_global_all) -- the driver's 'grid' mode through the shim's modules against the same mode run
    with the reference's."""
    need(32)
    m = json.load(open(os.path.join(GOLDEN, "grid32_bubbles.json")))
    a = np.load(os.path.join(GOLDEN, "grid32_bubbles.npz"))
    d = gi.run_driver(32, gi.SRC_STD, {"mode": "'grid'"}, dens=gi.density_factor(32, 5), xfield=gi.bubble_xfield(32, BUBBLES, 7.0),
                      hip=True, d=rundir)
    assert np.array_equal(gi.rd(d, "step001_ndens.f32", 32, np.float32), a["ndens"])      # same inputs
    assert np.array_equal(gi.rd(d, "step001_xh_before.f64", 32), a["xh"])
    kv = dict(line.split() for line in open(d + "/dump/step001_grid.txt"))
    assert int(kv["sum_nbox"]) == m["sum_nbox"] and int(kv["conv_flag"]) == m["conv_flag"] and kv["local_chemistry"] == "F"
    assert abs(float(kv["photon_loss"]) - m["photon_loss"]) <= 1e-10 * abs(m["photon_loss"])
    ph = gi.rd(d, "step001_phih_grid.f64", 32)
    assert np.array_equal(ph == 0.0, a["phih"] == 0.0) and relerr(ph, a["phih"]) < 1e-9
    assert np.max(np.abs(gi.rd(d, "step001_xh_av.f64", 32) - a["xh_av"])) < 1e-9
    assert np.max(np.abs(gi.rd(d, "step001_xh_intermed.f64", 32) - a["xh_intermed"])) < 1e-9


def test_fortran_evolve0d_and_evolve0d_global_cell_by_cell_match_the_reference(rundir):
    """The per-cell call surface as the reference's own loops use it (driver mode 'cells': the SAME driver code against the
    reference's evolve_point and against the shim's): evolve0D(dt,rtpos,ns,niter) for every cell of sub-boxes 1 and 2 of source 1
    in shell order (10 592 calls: the 1 331 cells of sub-box 1 are offered again and return at once, evolve_point.F90:125), then
    evolve0D_global(dt,pos,conv_flag) for the 9 261 cells of the box.  coldensh_out bit for bit the Fortran's in BOTH sweep
    modes of the context (the per-cell entry always takes cinterp in the reference's operation order)."""
    need(32)
    m = json.load(open(os.path.join(GOLDEN, "cells32_bubbles.json")))
    a = np.load(os.path.join(GOLDEN, "cells32_bubbles.npz"))
    d = gi.run_driver(32, gi.SRC_STD, {"mode": "'cells'"}, dens=gi.density_factor(32, 5), xfield=gi.bubble_xfield(32, BUBBLES, 7.0),
                      hip=True, d=rundir)
    assert np.array_equal(gi.rd(d, "step001_ndens.f32", 32, np.float32), a["ndens"])      # same inputs
    assert np.array_equal(gi.rd(d, "step001_xh_before.f64", 32), a["xh"])
    kv = dict(line.split() for line in open(d + "/dump/step001_cells.txt"))
    assert int(kv["evolve0D_calls"]) == m["evolve0D_calls"] == 10592 and int(kv["conv_flag"]) == m["conv_flag"]
    assert abs(float(kv["photon_loss_src"]) - m["photon_loss_src"]) <= 1e-10 * abs(m["photon_loss_src"])
    box = tuple(slice(l, h) for l, h in zip(m["box_lo"], m["box_hi"]))
    cd = gi.rd(d, "step001_coldensh_out.f64", 32)
    assert np.count_nonzero(cd) == np.count_nonzero(cd[box]) == np.count_nonzero(a["coldensh_out_box"]) == 21 ** 3
    assert np.array_equal(cd[box], a["coldensh_out_box"])                                 # bit for bit
    ph = gi.rd(d, "step001_phih_grid.f64", 32)
    assert np.count_nonzero(ph) == np.count_nonzero(ph[box])
    assert np.array_equal(ph[box] == 0.0, a["phih_grid_box"] == 0.0) and relerr(ph[box], a["phih_grid_box"]) < 1e-9
    assert np.max(np.abs(gi.rd(d, "step001_xh_av.f64", 32)[box] - a["xh_av_box"])) < 1e-9
    assert np.max(np.abs(gi.rd(d, "step001_xh_intermed.f64", 32)[box] - a["xh_intermed_box"])) < 1e-9


@pytest.mark.parametrize("variant,name", [("lls2", "evolve32_lls2"), ("lls3", "evolve32_lls3"),
                                          ("clump5", "evolve32_clump5")])
def test_fortran_evolve3d_physics_variants(rundir, variant, name):
    need(32, variant)
    m = json.load(open(os.path.join(GOLDEN, name + ".json")))["steps"]["step001"]
    a = np.load(os.path.join(GOLDEN, name + ".npz"))
    nml = {"mode": "'evolve'", "nsteps": 1, "dump_first": 1, "dump_last": 1}
    extra = {}
    if "lls_grid" in a.files:
        extra["lls.f32"] = lambda p: a["lls_grid"].T.tofile(p); nml["lls_file"] = "'lls.f32'"
    if "clump_grid" in a.files:
        extra["clump.f32"] = lambda p: a["clump_grid"].T.tofile(p); nml["clump_file"] = "'clump.f32'"
    d = gi.run_driver(32, gi.SRC_STD, nml, dens=gi.density_factor(32, 11), xfield=gi.bubble_xfield(32, BUBBLES, 7.0),
                      variant=variant, extra_files=extra, hip=True, d=rundir)
    assert np.array_equal(gi.rd(d, "step001_xh_before.f64", 32), a["step001_xh_before"])
    log = gi.parse_log(d + "/results/C2Ray.log")[0]
    assert log["nonconv"] == m["log"]["nonconv"]                    # same iteration history
    assert log["test1"] == m["log"]["test1"][1:]                    # (the reference also logs the test before iteration 1)
    kv = gi.read_kv(d + "/dump/step001_out.txt")
    assert kv["sum_nbox_all"] == m["sum_nbox_all"]
    assert abs(kv["photon_loss_all"] - m["photon_loss_all"]) <= 1e-10 * abs(m["photon_loss_all"]) + 1e-300
    assert np.max(np.abs(gi.rd(d, "step001_xh_after.f64", 32) - a["step001_xh_after"])) < 1e-9
    assert np.max(np.abs(gi.rd(d, "step001_xh_av.f64", 32) - a["step001_xh_av"])) < 1e-9
    assert relerr(gi.rd(d, "step001_phih_grid.f64", 32), a["step001_phih_grid"]) < 1e-9


@pytest.mark.parametrize("name,sources,dens_seed,bubbles", [
    ("restart32_onesrc", gi.SRC_ONE, 12, None), ("restart32_std_bubbles", gi.SRC_STD, 11, 6.0)])
def test_fortran_evolve3d_restart_from_iteration_dump(rundir, name, sources, dens_seed, bubbles):
    need(32)
    m = json.load(open(os.path.join(GOLDEN, name + ".json")))
    a = np.load(os.path.join(GOLDEN, name + ".npz"))

    def w_dump(p):      # Fortran sequential unformatted: niter | photon_loss_all(1) | phih | xh_av | xh_intermed
        with open(p, "wb") as f:
            for rec in (np.array([m["dump_niter"]], np.int32), np.array([m["dump_photon_loss_all"]], np.float64),
                        a["dump_phih"].ravel(order="F"), a["dump_xh_av"].ravel(order="F"),
                        a["dump_xh_intermed"].ravel(order="F")):
                b = rec.tobytes()
                f.write(np.array([len(b)], np.int32).tobytes() + b + np.array([len(b)], np.int32).tobytes())

    x = gi.bubble_xfield(32, BUBBLES, bubbles) if bubbles else None
    d = gi.run_driver(32, sources, {"mode": "'restart'", "nsteps": 1, "dump_first": 1, "dump_last": 1},
                      dens=gi.density_factor(32, dens_seed), xfield=x, extra_files={"iterdump.bin": w_dump},
                      hip=True, d=rundir)
    log = gi.parse_log(d + "/results/C2Ray.log")[0]
    assert len(log["nonconv"]) == m["niter_after_restart"]
    assert log["nonconv"] == m["log"]["nonconv"]
    assert np.max(np.abs(gi.rd(d, "step001_xh_after.f64", 32) - a["xh_after"])) < 1e-9
    assert relerr(gi.rd(d, "step001_phih_grid.f64", 32), a["phih_grid"]) < 1e-9


# ---- non-isothermal build of the driver (isothermal=.false.; oracle/ref_build.sh 32:thermal) ----------------------
# The shim reads tables/corocool.tab as setup_cool does, hands the heating tables rad_ini built to the library, and
# passes phiheat_grid / temperature_grid (an array of type(temperature_states)) through c2r_evolve3d_thermal.
T_RTOL = 1.5e-7        # one unit in the last place of the f32 temperature_grid (see tests/test_gpu_thermal.py)


def test_fortran_nonisothermal_evolve3d_three_steps(rundir):
    need(32, "thermal")
    m = json.load(open(os.path.join(GOLDEN, "evolve32_thermal.json")))["steps"]
    a = np.load(os.path.join(GOLDEN, "evolve32_thermal.npz"))
    d = gi.run_driver(32, gi.SRC_STD, {"mode": "'evolve'", "nsteps": 3, "dump_first": 1, "dump_last": 3},
                      dens=gi.density_factor(32, 11), xfield=gi.bubble_xfield(32, BUBBLES, 6.0), variant="thermal",
                      tfield=gi.temperature_field(32, 5), hip=True, d=rundir)
    assert "heating and cooling on the device" in open(d + "/results/C2Ray.log").read()
    log = gi.parse_log(d + "/results/C2Ray.log")
    for k, tag in ((0, "step001"), (2, "step003")):
        s = m[tag]
        assert log[k]["nonconv"] == s["log"]["nonconv"], tag                     # same iteration history
        kv = gi.read_kv("%s/dump/%s_out.txt" % (d, tag))
        assert kv["sum_nbox_all"] == s["sum_nbox_all"]
        assert np.max(np.abs(gi.rd(d, tag + "_xh_after.f64", 32) - a[tag + "_xh_after"])) < 1e-9
        tg = np.fromfile("%s/dump/%s_temper_after.f32" % (d, tag), dtype=np.float32).reshape(-1, 3)
        ref = a[tag + "_temper_after"]
        assert np.max(np.abs(tg.astype(np.float64) / ref - 1)) <= T_RTOL, tag
        assert np.count_nonzero(tg != ref) <= 1e-3 * tg.size
        assert np.array_equal(tg[:, 0], tg[:, 2])                                # set_final_temperature_point
        heat, rheat = gi.rd(d, tag + "_phiheat_grid.f64", 32), a[tag + "_phiheat_grid"]
        assert np.array_equal(heat == 0, rheat == 0)
        assert relerr(heat, rheat) < 1e-8
        for q in ("totrec", "totcollisions"):
            assert abs(kv[q] / s[q] - 1) < 1e-9
    if True:    # the temperatures the second step started from are the first step's results
        t1 = np.fromfile(d + "/dump/step001_temper_after.f32", dtype=np.float32)
        t2 = np.fromfile(d + "/dump/step002_temper_before.f32", dtype=np.float32)
        assert np.array_equal(t1, t2)


def test_fortran_do_source_nonisothermal_sweep(rundir):
    """do_source of the shim in a non-isothermal build: the source's heating rates are added to the host phiheat_grid."""
    need(32, "thermal")
    m = json.load(open(os.path.join(GOLDEN, "sweep32_thermal.json")))
    a = np.load(os.path.join(GOLDEN, "sweep32_thermal.npz"))
    d = gi.run_driver(32, gi.SRC_STD, {"mode": "'sweep'", "ns_dump": m["ns_dump"]}, dens=gi.density_factor(32, 5),
                      xfield=gi.bubble_xfield(32, BUBBLES, 7.0), variant="thermal", hip=True, d=rundir)
    kv = gi.read_kv(d + "/dump/step001_sweep.txt")
    assert kv["sum_nbox"] == m["sum_nbox"]
    assert relerr(gi.rd(d, "step001_phih_grid.f64", 32), a["phih"]) < 1e-8
    heat = gi.rd(d, "step001_phiheat_grid.f64", 32)
    assert np.array_equal(heat == 0, a["phiheat"] == 0) and heat.any()
    assert relerr(heat, a["phiheat"]) < 1e-8
