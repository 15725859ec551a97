#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference and amdflang):
    oracle/ref_build.sh 32 33 64          # compile the reference per mesh size
    python tests/golden/make_golden.py     # run it, collect inputs + outputs

Every fixture is DATA: inputs handed to the reference's hot path and the values the
compiled reference produced (raw arrays + scalars parsed from its own log).  The driver
that calls the reference routines is oracle/ref_driver.F90 (our code).  Fixtures are always
generated with the SERIAL build: the reference's OpenMP path has a data race on
photon_loss_src_thread and is not deterministic.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, HERE)
from inputs import ANSWERS, SRC_ONE, SRC_STD, density_factor, bubble_xfield, run_driver, \
    read_kv, rd, parse_log, cooling_table, temperature_field, temperature_field_cold   # noqa: E402


def case_tables():
    d = run_driver(32, SRC_ONE, {"mode": "'tables'"})
    thick = np.fromfile(d + "/dump/thick_table.f64")
    thin = np.fromfile(d + "/dump/thin_table.f64")
    assert thick.size == 2001 and thin.size == 2001
    np.savez_compressed(os.path.join(HERE, "tables.npz"), thick=thick, thin=thin)
    print("tables: thick[0]=%.17g thin[0]=%.17g" % (thick[0], thin[0]))


def case_tables_variant(name, variant):
    """The rate tables of the reference rebuilt with another SED / opacity switch (ref_build.sh 32:pl, 32:grey):
    stellar_SED_type=2 (power law, radiation_tables.F90:455-466 PL_SED) and grey=.true. (:338-357)."""
    d = run_driver(32, SRC_ONE, {"mode": "'tables'"}, variant=variant)
    thick = np.fromfile(d + "/dump/thick_table.f64")
    thin = np.fromfile(d + "/dump/thin_table.f64")
    assert thick.size == 2001 and thin.size == 2001
    np.savez_compressed(os.path.join(HERE, name + ".npz"), thick=thick, thin=thin)
    print("%s: thick[0]=%.17g thin[0]=%.17g" % (name, thick[0], thin[0]))


def xray_files(xray):
    """run_driver arguments that hand the fixture driver the X-ray tables (ref_driver.F90: namelist xray_tables)."""
    if xray is None:
        return {}, {}
    def w(p):        # thick, thin[, heat thick, heat thin]
        with open(p, "wb") as f:
            for t in xray:
                np.ascontiguousarray(t, dtype=np.float64).tofile(f)
    return {"xray.f64": w}, {"xray_tables": "'xray.f64'"}


def case_point():
    n = 32
    rng = np.random.default_rng(20261003)
    cd = 10.0 ** rng.uniform(15.0, 21.0, size=(n, n, n))
    # photoion rows: tau in {0, <1e-20, thin/thick boundary, 1, 1e4, >1e4} x cell widths
    sig = 6.29999986469627735e-18
    tin = np.array([0.0, 1e-25, 1e-20, 3e-9, 1e-7, 1e-3, 0.5, 1.0, 37.0, 9.9e3, 1e4, 3e4]) / sig
    dt_ = np.array([0.0, 1e-9, 0.9e-7, 1.00000001168609742e-07, 1.1e-7, 1e-3, 0.3, 5.0, 2e4]) / sig
    rows = [(a, a + b, 10.0 ** rng.uniform(70, 76)) for a in tin for b in dt_]
    photo = np.array(rows, dtype=np.float64)
    # doric rows: (dt, temp0, rhe, rhh, x1_old, xav1, phih)
    rows = []
    for phih in (0.0, 1e-20, 1e-16, 1e-12, 1e-9, 1e-6):
        for dt in (3.15576e13, 3.15576e9, 1.0e2):
            for x1 in (2e-4, 0.3, 0.999, 1.0 - 1e-14):
                nh = 10.0 ** rng.uniform(-5, -2)
                xav = min(1.0, x1 * (1.0 + 0.1 * rng.uniform()))
                rows.append((dt, 1e4, nh * (xav + 7.09999994796817191e-07), nh, x1, xav, phih))
    dor = np.array(rows, dtype=np.float64)
    out = {"coldens": cd, "photo_in": photo, "doric_in": dor}
    for tag, (sp, r) in {"a": ((16, 17, 15), 4), "b": ((2, 31, 1), 3)}.items():
        def w_cd(p): cd.T.tofile(p)
        def w_ci(p): open(p, "w").write("%d %d %d %d\n" % (sp[0], sp[1], sp[2], r))
        def w_ph(p):
            with open(p, "wb") as f:
                np.int32(len(photo)).tofile(f); photo.tofile(f)
        def w_do(p):
            with open(p, "wb") as f:
                np.int32(len(dor)).tofile(f); dor.tofile(f)
        d = run_driver(n, SRC_ONE, {"mode": "'point'"},
                       extra_files={"point_coldens.f64": w_cd, "point_cinterp.txt": w_ci,
                                    "point_photo.f64": w_ph, "point_doric.f64": w_do})
        ci = np.fromfile(d + "/dump/point_cinterp_out.f64").reshape(-1, 2)
        out["cinterp_src_" + tag] = np.array(sp + (r,), dtype=np.int32)
        out["cinterp_out_" + tag] = ci
        ph = np.fromfile(d + "/dump/point_photo_out.f64")
        out["photo_normflux"] = ph[0]
        out["photo_out"] = ph[1:].reshape(-1, 3)
        out["doric_out"] = np.fromfile(d + "/dump/point_doric_out.f64").reshape(-1, 4)
    np.savez_compressed(os.path.join(HERE, "point.npz"), **out)
    print("point: cinterp %d+%d cells, photo %d rows, doric %d rows" %
          (len(out["cinterp_out_a"]), len(out["cinterp_out_b"]), len(photo), len(dor)))


def case_thermal_tables_and_points():
    """Reference rebuilt with isothermal=.false. (ref_build.sh 32:thermal), run with the synthetic cooling table:
    the heating tables (radiation_tables.F90:455-543), and heat_lookuptable / coolin / thermal on tabulated arguments."""
    text, lt, ll = cooling_table()
    d = run_driver(32, SRC_ONE, {"mode": "'tables'"}, variant="thermal")
    out = {"thick": np.fromfile(d + "/dump/thick_table.f64"), "thin": np.fromfile(d + "/dump/thin_table.f64"),
           "heat_thick": np.fromfile(d + "/dump/heat_thick_table.f64"), "heat_thin": np.fromfile(d + "/dump/heat_thin_table.f64"),
           "cool_logT": lt, "cool_logL": ll}
    iso = np.load(os.path.join(HERE, "tables.npz"))
    assert np.array_equal(out["thick"], iso["thick"]) and np.array_equal(out["thin"], iso["thin"])
    assert out["heat_thick"].size == 2001
    np.savez_compressed(os.path.join(HERE, "tables_thermal.npz"), **out)
    pt = np.load(os.path.join(HERE, "point.npz"))
    photo = pt["photo_in"]
    rng = np.random.default_rng(20261004)
    cool = np.array([(10.0 ** rng.uniform(-5, -1), 10.0 ** rng.uniform(-6, -1), T)
                     for T in list(10.0 ** rng.uniform(0.5, 7.5, 60)) + [10.0, 12.589254117941675, 1e7, 9.99, 1.0000001e7]])
    rows = []
    for T0 in (0.5, 1.0, 50.0, 1e3, 1e4, 3e4, 1e6):
        for heat in (0.0, 1e-30, 1e-26, 1e-24):
            for dt in (3.15576e13, 3.15576e9):
                for x in (2e-4, 0.5, 0.9995):
                    nh = 10.0 ** rng.uniform(-5, -2)
                    xav = min(1.0, x * (1.0 + 0.2 * rng.uniform()))
                    xnew = min(1.0, xav * (1.0 + 0.2 * rng.uniform()))
                    rows.append((dt, T0, nh * (xav + 7.09999994796817191e-07), nh, x, xav, xnew, heat))
    th = np.array(rows)
    def w(arr):
        def f(p):
            with open(p, "wb") as fh:
                np.int32(len(arr)).tofile(fh); arr.tofile(fh)
        return f
    cd = pt["coldens"]
    dor = pt["doric_in"]
    d = run_driver(32, SRC_ONE, {"mode": "'point'"}, variant="thermal",
                   extra_files={"point_coldens.f64": lambda p: cd.T.tofile(p),
                                "point_cinterp.txt": lambda p: open(p, "w").write("16 17 15 1\n"),
                                "point_photo.f64": w(photo), "point_doric.f64": w(dor),
                                "point_cool.f64": w(cool), "point_thermal.f64": w(th)})
    ph = np.fromfile(d + "/dump/point_photo_out.f64")
    assert np.array_equal(ph[1:].reshape(-1, 3), pt["photo_out"])        # the photo rates do not depend on the switch
    tho = np.fromfile(d + "/dump/point_thermal_out.f64")
    np.savez_compressed(os.path.join(HERE, "point_thermal.npz"), photo_in=photo, photo_normflux=ph[0],
                        heat_out=np.fromfile(d + "/dump/point_heat_out.f64"),
                        cool_in=cool, cool_out=np.fromfile(d + "/dump/point_cool_out.f64"),
                        thermal_in=th, thermal_zred=tho[0], thermal_out=tho[1:].reshape(-1, 2))
    print("thermal: heat_thick[0]=%.17g heat_thin[0]=%.17g, %d heat rows, %d cool rows, %d thermal rows (zred %.6f)" %
          (out["heat_thick"][0], out["heat_thin"][0], len(photo), len(cool), len(th), tho[0]))


def case_thermal_points_steep():
    """coolin and thermal of the reference (isothermal=.false. build) with the SECOND synthetic cooling table
    (inputs.cooling_table("steep")): rows above, inside and below the table's range, and thermal rows that start at or below
    minitemp, that relax in a few sub-steps, and that are driven to minitemp and leave through the 10 000 sub-step cap."""
    text, lt, ll = cooling_table("steep")
    pt = np.load(os.path.join(HERE, "point.npz"))
    rng = np.random.default_rng(20261005)
    cool = np.array([(10.0 ** rng.uniform(-5, -1), 10.0 ** rng.uniform(-6, -1), T)
                     for T in list(10.0 ** rng.uniform(0.0, 7.5, 80)) + [1.0, 2.0, 9.99, 10.0, 12.589254117941675, 1e7, 1.0000001e7]])
    rows = []
    for T0 in (0.5, 1.0, 1.5, 3.0, 20.0, 60.0, 1e3, 1e4, 3e4, 1e5):
        for heat in (0.0, 1e-28, 1e-25):
            for dt in (3.15576e13, 3.15576e10):
                for x in (2e-4, 0.3, 0.9995):
                    nh = 10.0 ** rng.uniform(-4, -2)
                    xav = min(1.0, x * (1.0 + 0.2 * rng.uniform()))
                    xnew = min(1.0, xav * (1.0 + 0.2 * rng.uniform()))
                    rows.append((dt, T0, nh * (xav + 7.09999994796817191e-07), nh, x, xav, xnew, heat))
    th = np.array(rows)
    def w(arr):
        def f(p):
            with open(p, "wb") as fh:
                np.int32(len(arr)).tofile(fh); arr.tofile(fh)
        return f
    d = run_driver(32, SRC_ONE, {"mode": "'point'"}, variant="thermal", cooling="steep",
                   extra_files={"point_coldens.f64": lambda p: pt["coldens"].T.tofile(p),
                                "point_cinterp.txt": lambda p: open(p, "w").write("16 17 15 1\n"),
                                "point_photo.f64": w(pt["photo_in"]), "point_doric.f64": w(pt["doric_in"]),
                                "point_cool.f64": w(cool), "point_thermal.f64": w(th)})
    tho = np.fromfile(d + "/dump/point_thermal_out.f64")
    np.savez_compressed(os.path.join(HERE, "point_thermal_steep.npz"), cool_logT=lt, cool_logL=ll,
                        cool_in=cool, cool_out=np.fromfile(d + "/dump/point_cool_out.f64"),
                        thermal_in=th, thermal_zred=tho[0], thermal_out=tho[1:].reshape(-1, 2))
    out = tho[1:].reshape(-1, 2)
    print("thermal (steep curve): %d cool rows, %d thermal rows, %d untouched, %d ending at gamma1*minitemp" %
          (len(cool), len(th), int(np.sum(out[:, 0] == -1.0)), int(np.sum(np.abs(out[:, 0] - 2.0 / 3.0) < 1e-9))))


def case_evolve(name, n, sources, nsteps, dump, dens_seed=None, xfield=None, keep=("xh_after", "phih_grid", "xh_av"),
                variant=None, lls_grid=None, clump_grid=None, tfield=None, cooling="primordial", dens_sigma=0.6, xray=None, x0field=None):
    dens = density_factor(n, dens_seed, dens_sigma) if dens_seed is not None else None
    nml = {"mode": "'evolve'", "nsteps": nsteps, "dump_first": dump[0], "dump_last": dump[-1]}
    extra, xn = xray_files(xray)
    nml.update(xn)
    if x0field is not None:            # -DALLFRAC builds: the stored neutral fraction the run starts from (ref_driver.F90 x0_file)
        extra["x0.f64"] = lambda p: x0field.T.tofile(p); nml["x0_file"] = "'x0.f64'"
    if lls_grid is not None:
        extra["lls.f32"] = lambda p: lls_grid.astype(np.float32).T.tofile(p); nml["lls_file"] = "'lls.f32'"
    if clump_grid is not None:
        extra["clump.f32"] = lambda p: clump_grid.astype(np.float32).T.tofile(p); nml["clump_file"] = "'clump.f32'"
    d = run_driver(n, sources, nml, dens=dens, xfield=xfield, variant=variant, extra_files=extra, tfield=tfield, cooling=cooling)
    log = parse_log(d + "/results/C2Ray.log")
    arrays, meta = {}, {"n": n, "steps": {}}
    for s in dump:
        tag = "step%03d" % s
        kv = read_kv("%s/dump/%s_in.txt" % (d, tag))
        kv.update(read_kv("%s/dump/%s_out.txt" % (d, tag)))
        kv["log"] = log[s - 1]
        kv["niter"] = len(log[s - 1]["nonconv"])
        meta["steps"][tag] = kv
        arrays[tag + "_xh_before"] = rd(d, tag + "_xh_before.f64", n)
        if variant == "allfrac":
            arrays[tag + "_xh_before0"] = rd(d, tag + "_xh_before0.f64", n)
        arrays[tag + "_ndens"] = rd(d, tag + "_ndens.f32", n, np.float32)
        for k in keep:
            arrays[tag + "_" + k] = rd(d, "%s_%s.f64" % (tag, k), n)
        if variant == "thermal":       # temperature_grid as it lies in memory: cell-major (current, average, intermed) f32
            for k in ("temper_before", "temper_after"):
                arrays[tag + "_" + k] = np.fromfile("%s/dump/%s_%s.f32" % (d, tag, k), dtype=np.float32).reshape(-1, 3)
            arrays[tag + "_phiheat_grid"] = rd(d, tag + "_phiheat_grid.f64", n)
    if lls_grid is not None: arrays["lls_grid"] = lls_grid.astype(np.float32)
    if clump_grid is not None: arrays["clump_grid"] = clump_grid.astype(np.float32)
    if xray is not None: arrays["xray_thick"], arrays["xray_thin"] = xray[:2]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print(name, {t: m["niter"] for t, m in meta["steps"].items()})


def case_sweep(name, n, sources, x_init=None, dens_seed=None, xfield=None, ns_dump=1, full=True, variant=None, xray=None, x0field=None):
    dens = density_factor(n, dens_seed) if dens_seed is not None else None
    nml = {"mode": "'sweep'", "ns_dump": ns_dump}
    if x_init is not None:
        nml["x_init"] = "%.17g" % x_init
    extra, xn = xray_files(xray)
    nml.update(xn)
    if x0field is not None:
        extra["x0.f64"] = lambda p: x0field.T.tofile(p); nml["x0_file"] = "'x0.f64'"
    d = run_driver(n, sources, nml, dens=dens, xfield=xfield, variant=variant, extra_files=extra)
    tag = "step001"
    kv = read_kv("%s/dump/%s_in.txt" % (d, tag))
    kv.update(read_kv("%s/dump/%s_sweep.txt" % (d, tag)))
    kv["ns_dump"] = ns_dump
    kv.pop("seconds_per_pass", None)
    phih = rd(d, tag + "_phih_grid.f64", n)
    cdo = rd(d, tag + "_coldensh_out.f64", n)
    arrays = {"xh": rd(d, tag + "_xh_before.f64", n), "ndens": rd(d, tag + "_ndens.f32", n, np.float32)}
    if variant == "allfrac":
        arrays["xh0"] = rd(d, tag + "_xh_before0.f64", n)
    if variant in ("thermal", "xraythermal"):
        arrays["phiheat"] = rd(d, tag + "_phiheat_grid.f64", n)
    if xray is not None:
        arrays["xray_thick"], arrays["xray_thin"] = xray[:2]
        if len(xray) == 4:
            arrays["xray_heat_thick"], arrays["xray_heat_thin"] = xray[2:]
    if full:
        arrays.update(phih=phih, coldensh_out=cdo)
    else:   # large grids: three orthogonal planes through source ns_dump + checksums
        s = [(p - 1) % n for p in kv["srcpos"][ns_dump - 1]]
        arrays.update(phih_px=phih[s[0]], phih_py=phih[:, s[1]], phih_pz=phih[:, :, s[2]],
                      cd_px=cdo[s[0]], cd_py=cdo[:, s[1]], cd_pz=cdo[:, :, s[2]])
        # uniform inputs need not be stored whole
        for k in ("xh", "ndens"):
            if np.all(arrays[k] == arrays[k].flat[0]):
                arrays[k] = arrays[k].flat[0:1].copy()
    kv["phih_sum"] = float(np.sum(phih, dtype=np.longdouble))
    kv["phih_max"] = float(phih.max())
    kv["phih_nonzero"] = int(np.count_nonzero(phih))
    kv["cd_sum"] = float(np.sum(cdo, dtype=np.longdouble))
    kv["cd_nonzero"] = int(np.count_nonzero(cdo))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    json.dump({"n": n, **kv}, open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print(name, "sum_nbox", kv["sum_nbox"], "loss", kv["photon_loss"], "phih_nonzero", kv["phih_nonzero"])


def case_grid(name, n, sources, dens_seed, xfield, variant=None, tfield=None):
    """master_slave_processing::do_grid for all sources, then evolve_point::evolve0D_global over the mesh (driver mode
    'grid'): the two modules' public routines as the reference's pass_all_sources / global_pass call them."""
    d = run_driver(n, sources, {"mode": "'grid'"}, dens=density_factor(n, dens_seed), xfield=xfield, variant=variant, tfield=tfield)
    kv = read_kv(d + "/dump/step001_in.txt")
    for line in open(d + "/dump/step001_grid.txt"):
        k, v = line.split()
        kv[k] = (v == "T") if k == "local_chemistry" else (float(v) if "E" in v else int(v))
    arrays = {"xh": rd(d, "step001_xh_before.f64", n), "ndens": rd(d, "step001_ndens.f32", n, np.float32),
              "phih": rd(d, "step001_phih_grid.f64", n), "xh_av": rd(d, "step001_xh_av.f64", n),
              "xh_intermed": rd(d, "step001_xh_intermed.f64", n)}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    json.dump({"n": n, **kv}, open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print(name, "sum_nbox", kv["sum_nbox"], "conv_flag", kv["conv_flag"], "loss", kv["photon_loss"])


def case_cells(name, n, sources, dens_seed, xfield):
    """The per-cell call surface as such (driver mode 'cells'): source 1 traced through sub-boxes 1 and 2 by calling
    evolve_point::evolve0D cell by cell in shell order, then evolve_point::evolve0D_global for every cell of that box."""
    d = run_driver(n, sources, {"mode": "'cells'"}, dens=density_factor(n, dens_seed), xfield=xfield)
    kv = read_kv(d + "/dump/step001_in.txt")
    for line in open(d + "/dump/step001_cells.txt"):
        k, v = line.split()
        kv[k] = float(v) if "E" in v else int(v)
    cd = rd(d, "step001_coldensh_out.f64", n)
    nz = np.nonzero(cd)
    lo, hi = [int(a.min()) for a in nz], [int(a.max()) + 1 for a in nz]
    box = tuple(slice(l, h) for l, h in zip(lo, hi))          # the traced box does not wrap for this source: store it alone
    assert np.count_nonzero(cd) == np.count_nonzero(cd[box])
    arrays = {"xh": rd(d, "step001_xh_before.f64", n), "ndens": rd(d, "step001_ndens.f32", n, np.float32)}
    for tag in ("coldensh_out", "phih_grid", "xh_av", "xh_intermed"):
        arrays[tag + "_box"] = rd(d, "step001_%s.f64" % tag, n)[box]
    kv["box_lo"], kv["box_hi"] = lo, hi
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    json.dump({"n": n, **kv}, open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print(name, "calls", kv["evolve0D_calls"], "conv_flag", kv["conv_flag"], "loss", kv["photon_loss_src"])


def read_sm3d(path, dtype):
    """Fortran sequential records: int32 12 | 3 x int32 | int32 12 | int32 nbytes | data | int32 nbytes"""
    raw = open(path, "rb").read()
    n = np.frombuffer(raw, dtype=np.int32, count=3, offset=4)
    nb = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=20)[0])
    data = np.frombuffer(raw, dtype=dtype, count=nb // np.dtype(dtype).itemsize, offset=24)
    return data.reshape(tuple(int(v) for v in n), order="F")


def case_refrun(name, n, sources, variant=None):
    """The reference's OWN program (C2Ray.F90, all 14 slices x 10 steps) on its test problem:
    the outputs a user of the reference sees.  Used by the drop-in integration test.
    variant "thermal": the non-isothermal build, with the synthetic cooling table in ./tables/."""
    d = "/tmp/c2ray_golden_refrun"
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d + "/results")
    if variant == "thermal":
        os.makedirs(d + "/tables")
        open(d + "/tables/corocool.tab", "w").write(cooling_table()[0])
    open(d + "/answers", "w").write(ANSWERS)
    with open(d + "/test_sources.dat", "w") as f:
        f.write("%d\n" % len(sources))
        for (i, j, k, flux) in sources:
            f.write("%d %d %d %.17e 0.0\n" % (i, j, k, flux))
    exe = os.path.join(REF, "N%d%s" % (n, "_" + variant if variant else ""), "serial", "c2ray_test")
    subprocess.check_call([exe, "answers"], cwd=d, stdout=subprocess.DEVNULL)
    outs = sorted(f for f in os.listdir(d + "/results") if f.startswith("xfrac3D_"))
    nonconv = [int(l.split(":")[1]) for l in open(d + "/results/C2Ray.log") if "Number of non-converged points:" in l]
    keep = [outs[0], outs[len(outs) // 2], outs[-1]]
    arrays = {"xfrac_" + f[len("xfrac3D_"):-4]: read_sm3d(d + "/results/" + f, np.float64) for f in keep}
    arrays.update({"ionrates_" + f[len("xfrac3D_"):-4]:
                   read_sm3d(d + "/results/IonRates3D_" + f[len("xfrac3D_"):], np.float32) for f in keep[:1]})
    if variant == "thermal":
        arrays.update({"temper_" + f[len("xfrac3D_"):-4]:
                       read_sm3d(d + "/results/Temper3D_" + f[len("xfrac3D_"):], np.float32) for f in keep})
        arrays.update({"heatrates_" + f[len("xfrac3D_"):-4]:
                       read_sm3d(d + "/results/HeatRates3D_" + f[len("xfrac3D_"):], np.float32) for f in keep[:1]})
    import hashlib
    sha = {f: hashlib.sha256(open(d + "/results/" + f, "rb").read()).hexdigest()
           for f in keep + ["IonRates3D_" + keep[0][len("xfrac3D_"):]]}
    zs = [float(l.split()[2]) for l in open(d + "/results/C2Ray.log") if l.strip().startswith("Doing redshift:")]
    def numeric_rows(path):
        rows = []
        for line in open(path):
            try:
                rows.append([float(v) for v in line.split()])
            except ValueError:
                pass
        return rows
    counts = {"PhotonCounts": numeric_rows(d + "/results/PhotonCounts.out"),
              "PhotonCounts2": numeric_rows(d + "/results/PhotonCounts2.out")}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    json.dump({"n": n, "outputs": outs, "kept": keep, "nonconv": nonconv, "sha256": sha, "slice_redshifts": zs, "photon_counts": counts, "total_outer_iterations": len(nonconv),
               "sources": [list(s) for s in sources], "answers": ANSWERS},
              open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print(name, "outputs", len(outs), "outer iterations", len(nonconv))


def case_evolve_planes(name, n, sources, dens_seed=None, xfield=None, store_inputs=True, xfield_recipe=None):
    """One evolve3D step on a large mesh: three orthogonal planes through source 1 of xh_after and
    phih_grid, checksums and the iteration history."""
    dens = density_factor(n, dens_seed) if dens_seed is not None else None
    d = run_driver(n, sources, {"mode": "'evolve'", "nsteps": 1, "dump_first": 1, "dump_last": 1}, dens=dens, xfield=xfield)
    log = parse_log(d + "/results/C2Ray.log")
    kv = read_kv(d + "/dump/step001_in.txt"); kv.update(read_kv(d + "/dump/step001_out.txt"))
    kv["log"] = log[0]; kv["niter"] = len(log[0]["nonconv"])
    xa = rd(d, "step001_xh_after.f64", n); ph = rd(d, "step001_phih_grid.f64", n)
    s0 = [(p - 1) % n for p in kv["srcpos"][0]]
    arrays = {"xh_px": xa[s0[0]], "xh_py": xa[:, s0[1]], "xh_pz": xa[:, :, s0[2]],
              "phih_px": ph[s0[0]], "phih_py": ph[:, s0[1]], "phih_pz": ph[:, :, s0[2]]}
    nd = rd(d, "step001_ndens.f32", n, np.float32); xb = rd(d, "step001_xh_before.f64", n)
    arrays["ndens"] = nd.flat[0:1].copy() if np.all(nd == nd.flat[0]) else nd
    if np.all(xb == xb.flat[0]):
        arrays["xh_before"] = xb.flat[0:1].copy()
    elif store_inputs:
        arrays["xh_before"] = xb
    else:       # too large to commit: the test regenerates the field from its recipe and checks these
        kv["xh_before_recipe"] = xfield_recipe
        kv["xh_before_sum"] = float(np.sum(xb, dtype=np.longdouble))
        import hashlib
        kv["xh_before_sha256"] = hashlib.sha256(np.ascontiguousarray(xb.ravel(order="F")).tobytes()).hexdigest()
    kv.update(xh_sum=float(np.sum(xa, dtype=np.longdouble)), xh_min=float(xa.min()), xh_max=float(xa.max()),
              phih_sum=float(np.sum(ph, dtype=np.longdouble)), phih_max=float(ph.max()),
              phih_nonzero=int(np.count_nonzero(ph)))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    json.dump({"n": n, **kv}, open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print(name, "niter", kv["niter"], "phih_nonzero", kv["phih_nonzero"])


def case_restart(name, n, sources, dens_seed, xfield, k_iter=2):
    """evolve3D(restart=3): the reference resumes a time step from an iteration dump
    (start_from_dump, evolve.F90:328).  The dump is a genuine mid-iteration state -- k_iter outer
    iterations of the (pinned) oracle on the same inputs -- written by OUR writer
    (fileio.write_iteration_dump) and read by the REFERENCE's reader."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    from oracle.oracle import Oracle
    fio = g.load_package().fileio
    dens = density_factor(n, dens_seed)
    # a plain run first, only to learn the step's scalars and arrays
    d = run_driver(n, sources, {"mode": "'evolve'", "nsteps": 1, "dump_first": 1, "dump_last": 1}, dens=dens, xfield=xfield)
    kv = read_kv(d + "/dump/step001_in.txt")
    nd = rd(d, "step001_ndens.f32", n, np.float32); xh0 = rd(d, "step001_xh_before.f64", n)
    t = np.load(os.path.join(HERE, "tables.npz"))
    o = Oracle(n, (kv["dr1"], kv["dr2"], kv["dr3"]), kv["vol"], kv["coldensh_LLS"], t["thick"], t["thin"], kv["clumping"])
    F = lambda a: np.asfortranarray(a).ravel(order="F").copy()
    ndf, xh = F(nd), F(xh0)
    xav, xint, phih, loss = xh.copy(), xh.copy(), np.zeros(n ** 3), 0.0
    for _ in range(k_iter):
        phih[:] = 0.0
        loss, nb, vis = o.pass_sources(ndf, xav, phih, kv["srcpos"], kv["normflux"])
        o.global_pass(kv["dt"], ndf, xh, xav, xint, phih)
    def w_dump(p): fio.write_iteration_dump(p, k_iter, loss, phih, xav, xint, mesh=n)
    d = run_driver(n, sources, {"mode": "'restart'", "nsteps": 1, "dump_first": 1, "dump_last": 1}, dens=dens,
                   xfield=xfield, extra_files={"iterdump.bin": w_dump})
    log = parse_log(d + "/results/C2Ray.log")
    kv = read_kv(d + "/dump/step001_in.txt"); kv.update(read_kv(d + "/dump/step001_out.txt"))
    kv["log"] = log[0]; kv["niter_after_restart"] = len(log[0]["nonconv"]); kv["dump_niter"] = k_iter
    kv["dump_photon_loss_all"] = loss
    np.savez_compressed(os.path.join(HERE, name + ".npz"), ndens=nd, xh_before=xh0,
                        dump_phih=phih.reshape((n, n, n), order="F"), dump_xh_av=xav.reshape((n, n, n), order="F"),
                        dump_xh_intermed=xint.reshape((n, n, n), order="F"),
                        xh_after=rd(d, "step001_xh_after.f64", n), phih_grid=rd(d, "step001_phih_grid.f64", n),
                        xh_av=rd(d, "step001_xh_av.f64", n))
    json.dump({"n": n, **kv}, open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print(name, "global passes after restart:", kv["niter_after_restart"], "nonconv", log[0]["nonconv"])


def main():
    which = set(sys.argv[1:])
    def want(k): return not which or k in which
    if want("tables"): case_tables()
    if want("point"): case_point()
    # config[0] of BASELINE.json: the reference's own test problem, 32^3, one source, cold start
    if want("evolve32"): case_evolve("evolve32_onesrc", 32, SRC_ONE, 12, [1, 2, 3, 12])
    # pre-ionised gas, the 10-source list (positions wrap periodically): sub-box growth to the limit
    if want("refrun32"): case_refrun("refrun32_onesrc", 32, SRC_ONE)
    # ten sources (positions wrap at 32^3): overlapping regions, conv_criterion = 3, sub-boxes to the limit
    if want("refrun32std"): case_refrun("refrun32_std", 32, SRC_STD)
    if want("sweep32"): case_sweep("sweep32_std_x999", 32, SRC_STD, x_init=0.999)
    if want("sweep33"): case_sweep("sweep33_std_x999", 33, SRC_STD, x_init=0.999, dens_seed=33)
    # perturbed density + ionized bubbles: exercises max_coldensh stop, thin/thick cells, clipping
    if want("sweep32b"):
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 7.0)
        case_sweep("sweep32_bubbles", 32, SRC_STD, dens_seed=5, xfield=x, ns_dump=5)
    if want("evolve32b"):
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 6.0)
        case_evolve("evolve32_std_bubbles", 32, SRC_STD, 3, [1, 3], dens_seed=11, xfield=x)
    if want("restart32"):
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 6.0)
        case_restart("restart32_std_bubbles", 32, SRC_STD, 11, x)
        # one source: conv_criterion = 0, so only Test 2 can end the step and real iterations follow
        case_restart("restart32_onesrc", 32, SRC_ONE, 12, None, k_iter=3)
    # non-default physics switches (reference rebuilt with the one parameter changed, ref_build.sh N:variant)
    if want("variants"):
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 7.0)
        rng = np.random.default_rng(77)
        lls = (7.0e16 * 10.0 ** rng.uniform(-1.0, 1.0, (32, 32, 32)))
        case_evolve("evolve32_lls2", 32, SRC_STD, 1, [1], dens_seed=11, xfield=x, variant="lls2", lls_grid=lls)
        case_evolve("evolve32_lls3", 32, SRC_STD, 1, [1], dens_seed=11, xfield=x, variant="lls3")
        clump = 1.0 + 9.0 * rng.random((32, 32, 32)) ** 3
        case_evolve("evolve32_clump5", 32, SRC_STD, 1, [1], dens_seed=11, xfield=x, variant="clump5", clump_grid=clump)
    # other SEDs / opacities of the table builder, and the second source type of photoion_rates (round 5)
    if want("seds"):
        case_tables_variant("tables_pl", "pl")
        case_tables_variant("tables_grey", "grey")
    if want("xray"):
        # The X-ray tables are inputs of the rate path (the reference's own fill integrates an array it never sets): the
        # fixture runs use the reference's POWER-LAW tables (tables_pl: photon index 3 between the HI and HeII edges).
        pl = np.load(os.path.join(HERE, "tables_pl.npz"))
        xr = (pl["thick"], pl["thin"])
        # column 5 of the list = X-ray photon rate / S_star_xray (= 1): four sources with an X-ray component of the order
        # of their stellar one (NormFlux ~ 1e7), one dominated by it, five without
        src = [SRC_STD[0] + (3e6,), SRC_STD[1] + (0.0,), SRC_STD[2] + (2e7,), SRC_STD[3] + (0.0,), SRC_STD[4] + (5e8,),
               SRC_STD[5] + (0.0,), SRC_STD[6] + (1e6,), SRC_STD[7] + (0.0,), SRC_STD[8] + (4e8,), SRC_STD[9] + (0.0,)]
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 7.0)
        case_sweep("sweep32_xray", 32, src, dens_seed=5, xfield=x, ns_dump=5, variant="xray", xray=xr)
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 6.0)
        case_evolve("evolve32_xray", 32, src, 1, [1], dens_seed=11, xfield=x, variant="xray", xray=xr)
    if want("xraythermal"):
        # both switches: use_xray_SED=.true. and isothermal=.false. (ref_build.sh 32:xraythermal): the X-ray type also heats
        # (heat_lookuptable "P", radiation_photoionrates.F90:165-171).  Its four tables are inputs: the power-law photo tables of
        # the reference (tables_pl) and the power-law heating tables of our builder (c2r_build_heat_tables, C2R_SED_POWER_LAW).
        sys.path.insert(0, ROOT)
        import ctypes as C
        import __graft_entry__ as g
        pkg = g.load_package()
        sed = pkg.SedParams(); pkg.load_library().c2r_default_sed_power_law(C.byref(sed))
        hk, hn = pkg._capi.build_heat_tables(sed)
        pl = np.load(os.path.join(HERE, "tables_pl.npz"))
        xr = (pl["thick"], pl["thin"], hk, hn)
        src = [SRC_STD[0] + (3e6,), SRC_STD[1] + (0.0,), SRC_STD[2] + (2e7,), SRC_STD[3] + (0.0,), SRC_STD[4] + (5e8,),
               SRC_STD[5] + (0.0,), SRC_STD[6] + (1e6,), SRC_STD[7] + (0.0,), SRC_STD[8] + (4e8,), SRC_STD[9] + (0.0,)]
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 7.0)
        case_sweep("sweep32_xraythermal", 32, src, dens_seed=5, xfield=x, ns_dump=5, variant="xraythermal", xray=xr)
    # the reference compiled with -DALLFRAC (ref_build.sh 32:allfrac): both fractions stored.  The runs start from a neutral
    # fraction that is NOT 1 - x (2e-3 of noise on it): a path that derived it from x would not reproduce these rates.  After
    # its first global pass the code keeps the two consistent itself (doric).  A second sweep case stores a ZERO neutral
    # fraction in one cell in fifty, which evolve0D raises to epsilon (evolve_point.F90:131-132): such cells are all but
    # transparent, their own rates ill-conditioned differences (in the reference too) -- a sweep fixture only.
    if want("allfrac"):
        rng = np.random.default_rng(606)
        x = bubble_xfield(32, [(16, 16, 16), (5, 27, 9), (24, 8, 20)], 7.0)
        x0 = (1.0 - x) * (1.0 + 2e-3 * rng.standard_normal(x.shape))
        x0z = x0.copy()
        x0z[rng.random(x.shape) < 0.02] = 0.0
        case_sweep("sweep32_allfrac", 32, SRC_STD, dens_seed=5, xfield=x, ns_dump=5, variant="allfrac", x0field=x0)
        case_sweep("sweep32_allfrac_zeros", 32, SRC_STD, dens_seed=5, xfield=x, ns_dump=5, variant="allfrac", x0field=x0z)
        case_evolve("evolve32_allfrac", 32, SRC_STD, 2, [1, 2], dens_seed=11, xfield=x, variant="allfrac", x0field=x0,
                    keep=("xh_after", "xh_after0", "phih_grid", "xh_av", "xh_av0", "xh_intermed", "xh_intermed0"))
    # non-isothermal run (isothermal=.false.), with the synthetic cooling table of inputs.cooling_table()
    if want("thermal"):
        case_thermal_tables_and_points()
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 7.0)
        case_sweep("sweep32_thermal", 32, SRC_STD, dens_seed=5, xfield=x, ns_dump=5, variant="thermal")
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 6.0)
        case_evolve("evolve32_thermal", 32, SRC_STD, 3, [1, 3], dens_seed=11, xfield=x, variant="thermal",
                    tfield=temperature_field(32, 5))
    # the same build with the SECOND synthetic cooling table (steep CIE-like rise, cold-gas coolant): cells that start at or
    # below minitemp, cold dense cells pinned at minitemp until the sub-step cap, hot cells on the steep part of the curve
    if want("thermal2"):
        case_thermal_points_steep()
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 6.0)
        case_evolve("evolve32_thermal_steep", 32, SRC_STD, 1, [1], dens_seed=12, xfield=x, variant="thermal",
                    tfield=temperature_field_cold(32, 6), cooling="steep", dens_sigma=1.0, keep=("xh_after", "phih_grid"))
    if want("refrun32thermal"):
        case_refrun("refrun32_thermal", 32, SRC_STD, variant="thermal")
    # the module surface beside evolve3D / do_source: do_grid (master_slave.F90:53) and evolve0D_global over the mesh
    if want("grid32"):
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 7.0)
        case_grid("grid32_bubbles", 32, SRC_STD, 5, x)
    # ... and its per-cell routines as the reference's own loops call them: evolve0D cell by cell, evolve0D_global cell by cell
    if want("cells32"):
        x = bubble_xfield(32, [(18, 18, 18), (20, 10, 10), (6, 6, 18)], 7.0)
        case_cells("cells32_bubbles", 32, SRC_STD, 5, x)
    if want("sweep64"):
        x = bubble_xfield(64, [(50, 50, 50), (20, 10, 10), (6, 8, 50), (20, 10, 26)], 14.0)
        srcs = SRC_STD[:8] + [(72, 72, 50, 1e58), (20, 10, 90, 1e54)]
        case_sweep("sweep64_bubbles", 64, srcs, dens_seed=64, xfield=x, ns_dump=9, full=False)
    # BASELINE.json grid sizes, straight from the reference (planes + checksums)
    if want("big"):
        case_sweep("sweep128_std_x999", 128, SRC_STD, x_init=0.999, full=False)
        case_evolve_planes("evolve128_std", 128, SRC_STD)
        case_sweep("sweep256_3src_x999", 256, [(200, 30, 77, 1e56), (5, 250, 130, 3e55), (128, 128, 128, 1e57)],
                   x_init=0.999, full=False, ns_dump=3)
    # BASELINE.json configs[1]: 128^3, ONE source (inputs/test_sources_onesrc.dat): the single-source sweep out to the
    # limits, and a whole evolve3D step from a field with a 30-cell ionized bubble around the source (conv_criterion = 0:
    # only Test 2 ends the step, so several outer iterations with growing sub-boxes follow -- the launch-bound regime the
    # library replays as a hipGraph)
    if want("onesrc128"):
        case_sweep("sweep128_onesrc_x999", 128, SRC_ONE, x_init=0.999, full=False)
        case_evolve_planes("evolve128_onesrc_bubble", 128, SRC_ONE, xfield=bubble_xfield(128, [(50, 50, 50)], 30.0), store_inputs=False,
                           xfield_recipe="inputs.bubble_xfield(128, [(50, 50, 50)], 30.0)")
    # BASELINE.json configs[2]: 256^3 x 100 sources as WHOLE evolve3D steps: the cold
    # start (x = 2e-4) and a late field (ionized bubbles of 14 cells around every source, sub-boxes grow)
    if want("evolve256"):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        pos, nf = g.load_package().seeded_sources(256, 100)
        srcs = [(int(p[0]), int(p[1]), int(p[2]), float(f) * 1.00000000000000004e+48) for p, f in zip(pos, nf)]
        only = os.environ.get("C2R_GOLDEN_ONLY", "")
        if only in ("", "cold"):
            case_evolve_planes("evolve256_100src_cold", 256, srcs)
        if only in ("", "late"):
            x = bubble_xfield(256, [tuple(int(v) for v in p) for p in pos], 14.0)
            case_evolve_planes("evolve256_100src_late", 256, srcs, xfield=x, store_inputs=False,
                               xfield_recipe="inputs.bubble_xfield(256, seeded_sources(256,100) positions, 14.0)")
    if want("evolve64"):
        x = bubble_xfield(64, [(50, 50, 50), (20, 10, 10), (6, 8, 50), (20, 10, 26)], 12.0)
        case_evolve("evolve64_std_bubbles", 64, SRC_STD, 1, [1], dens_seed=65, xfield=x,
                    keep=("xh_after",))


if __name__ == "__main__":
    main()
