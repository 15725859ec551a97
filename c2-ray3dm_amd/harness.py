"""Stand-alone driver of the reference's test problem on the HIP path: the loop of C2Ray.F90:267-427
(redshift slices -> time steps -> evolve3D -> outputs) with the set-up of nbody_test.F90,
cosmology.F90, LLS.F90 and sourceprops.F90 restated in testproblem.py, reading the reference's
source-list format and writing its xfrac3D / IonRates3D files (fileio.py).

    python -m ...  is not needed: call run_test_problem() (tests/test_gpu_harness.py does).
"""
import os
import numpy as np

from .evolve import Evolve, HipBackend
from .testproblem import TestProblem, STEPS_PER_SLICE, XH_INITIAL
from . import fileio
from ._capi import build_tables, build_heat_tables

INITIAL_TEMPERATURE = 1e4        # c2ray_parameters.f90:112 initial_temperature (material.F90:84)


def run_test_problem(mesh, source_file, results_dir, nslices=14, device=0, comm=None, native_loop=True,
                     log=None, cooling_table=None):
    """Runs nslices x 10 time steps from z=9 (inputs/input_example_test: no restart, UV model 7,
    10 steps and 1 output per slice).  Returns the per-step evolve3D reports.
    cooling_table: path of a tables/corocool.tab -> the non-isothermal run (isothermal=.false.): heating and cooling,
    temperature_grid starts at initial_temperature, Temper3D / HeatRates3D outputs next to the others."""
    os.makedirs(results_dir, exist_ok=True)
    tp = TestProblem(mesh)
    thick, thin, _ = build_tables()                        # rad_ini
    srcpos, normflux = fileio.read_sources(source_file)    # source_properties, Test model
    b = HipBackend(mesh, thick, thin, device=device)
    b.set_sources(srcpos, normflux)
    thermal = cooling_table is not None
    if thermal:
        hk, hn = build_heat_tables()                       # rad_ini, make_heat_tables_HI
        lt, ll = fileio.read_cooling_table(cooling_table)  # setup_cool (C2Ray.F90:143)
        b.set_thermal(hk, hn, lt, ll)
        b.load(temperature_grid=np.full(mesh ** 3, INITIAL_TEMPERATURE, dtype=np.float32))   # temperature_array_init
    ev = Evolve(b, comm=comm)
    rank = comm.get_rank() if comm is not None else 0
    ncell = mesh ** 3
    b.load(xh=np.full(ncell, XH_INITIAL))                  # ionfractions_module.F90:49
    reports = []
    counts = fileio.PhotonCounts(results_dir) if rank == 0 else None

    def write_outputs(zred, time):                         # output.F90:175: streams 2, 3 and the photon statistics
        if rank != 0:
            return
        fileio.write_xfrac3D(results_dir, zred, b.fetch("xh"), mesh)
        fileio.write_IonRates3D(results_dir, zred, b.fetch("phih_grid"), mesh)
        if thermal:                                        # output.F90:314-329, :367-378
            fileio.write_Temper3D(results_dir, zred, b.fetch("temperature_grid"), mesh)
            fileio.write_HeatRates3D(results_dir, zred, b.fetch("phiheat_grid"), mesh)
        h0, h1, _, _ = b.photon_sums("xh", "xh")           # sum n(1-x), sum n x  (output.F90:573-581)
        counts.write(zred, time, h1 * b.vol, float(b.xh.sum()) / ncell, h1 / (h0 + h1))

    s0 = tp.at_time(0.0)
    b.set_step(s0["dr1"], s0["vol"], s0["coldensh_LLS"])
    b.load(ndens=np.full(ncell, s0["ndens"], dtype=np.float32))
    write_outputs(tp.zred_at(0.0), 0.0)                    # C2Ray.F90:343: output at sim_time = 0
    step = 0
    for nz in range(nslices):
        for _ in range(STEPS_PER_SLICE):
            step += 1
            s = tp.step(step)
            b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], s["clumping"], s["temper"])
            b.load(ndens=np.full(ncell, s["ndens"], dtype=np.float32))      # cosmo_evol rescales ndens
            if thermal:
                b.set_redshift(s["zred"])                                   # cosmology's zred after redshift_evol (C2Ray.F90:368)
            if native_loop and comm is None:
                rep = b.evolve3d_native(s["dt"])
                reports.append(dict(niter=rep.niter, converged=bool(rep.converged), photcons=rep.photcons))
                phot = {k: getattr(rep, k) for k in ("total_ion", "totcollisions", "totrec", "dh0", "totalsrc")}
                loss_all = rep.photon_loss_all
            else:
                r = ev.evolve3D((step - 1) * tp.dt, s["dt"], 0)
                reports.append(dict(niter=r["niter"], converged=r["converged"],
                                    photcons=r["photon_statistics"].get("photcons")))
                phot, loss_all = r["photon_statistics"], r["photon_loss_all"]
            if counts is not None:
                counts.update(phot, loss_all, s["dt"])
            if log:
                log("step %d niter %d" % (step, reports[-1]["niter"]))
        # C2Ray.F90:393-398: output at the end of the slice; the driver has rescaled ndens and vol to that
        # redshift by then (C2Ray.F90:416-419), which enters the total number of ions
        z_out = tp.zred_at(step * tp.dt)
        s_end = tp.at_time(step * tp.dt)
        b.set_step(s_end["dr1"], s_end["vol"], s_end["coldensh_LLS"])
        b.load(ndens=np.full(ncell, s_end["ndens"], dtype=np.float32))
        write_outputs(z_out, step * tp.dt)
    if counts is not None:
        counts.close()
    b.close()
    return reports
