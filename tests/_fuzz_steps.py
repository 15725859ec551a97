"""Randomized WHOLE time steps (evolve3D: sweep + all-reduce-free single rank + global pass, to convergence) on small
non-cubic meshes against the oracle's evolve3d: outer-iteration count, the sequence of non-converged-cell counts,
sub-box counts and the ionized fractions.  One case in three uses a clumping grid, one in three a non-default LLS type.
One case in three is a non-isothermal step (heating and cooling, temperatures 30 K .. 1e5 K).
Shared by tests/test_gpu_fuzz.py (asserting) and by hand:  python tests/_fuzz_steps.py [cases] [seed0] [exact|fast]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_step_case(seed, pkg):
    rng = np.random.default_rng(seed)
    s = pkg.TestProblem(32).step(1)
    mesh = tuple(int(v) for v in rng.integers(6, 25, 3))
    ncell = mesh[0] * mesh[1] * mesh[2]
    dr = tuple(float(s["dr1"] * 10.0 ** rng.uniform(-0.2, 0.4) * f) for f in rng.uniform(0.8, 1.25, 3))
    nd = (s["ndens"] * np.exp(0.6 * rng.standard_normal(ncell))).astype(np.float32)
    mode = seed % 3
    if mode == 0:            # cold start
        xh = np.full(ncell, 2e-4)
    elif mode == 1:          # partly ionized, structured
        xh = np.clip(10.0 ** rng.uniform(-3.5, 0, ncell) * 0.999, 1e-6, 0.999)
    else:                    # highly ionized
        xh = 1.0 - 10.0 ** rng.uniform(-5, -2.5, ncell)
    nsrc = int(rng.integers(1, 9))
    pos = np.stack([rng.integers(1, mesh[d] + 1, nsrc) for d in range(3)], axis=1).astype(np.int32)
    nf = 10.0 ** rng.uniform(5, 9, nsrc)
    clump = (1.0 + 9.0 * rng.random(ncell) ** 3).astype(np.float32) if rng.random() < 0.34 else None
    lls_type = int(rng.choice([1, 1, 2, 3]))
    lls = s["coldensh_LLS"] * 10.0 ** rng.uniform(-1, 1)
    lls_grid = (lls * 10.0 ** rng.uniform(-1.0, 1.0, ncell)).astype(np.float32) if lls_type == 2 else None
    r_max = float(dr[0] * rng.uniform(2.0, 0.7 * max(mesh))) if lls_type == 3 else 0.0
    dt = s["dt"] * 10.0 ** rng.uniform(-1.0, 0.3)
    # one case in three is a non-isothermal step (isothermal=.false.): log-uniform temperatures 30 K .. 1e5 K
    temper = None
    if rng.random() < 0.34:
        t = (10.0 ** rng.uniform(1.5, 5.0, ncell)).astype(np.float32)
        temper = np.ascontiguousarray(np.repeat(t[:, None], 3, axis=1))
    return dict(mesh=mesh, dr=dr, vol=dr[0] * dr[1] * dr[2], nd=nd, xh=xh, pos=pos, nf=nf, lls=lls, lls_type=lls_type,
                lls_grid=lls_grid, r_max=r_max, clump=clump, dt=dt, temper=temper, zred=float(rng.uniform(6.0, 12.0)))


def run_step_case(seed, pkg, tables, fast):
    from oracle.oracle import Oracle
    c = make_step_case(seed, pkg)
    o = Oracle(c["mesh"], c["dr"], c["vol"], c["lls"], *tables, lls_type=c["lls_type"], R_max_LLS=c["r_max"],
               lls_grid=c["lls_grid"], clump_grid=c["clump"])
    oxh = c["xh"].copy()
    otg = None
    if c["temper"] is not None:
        from tests._util import load_thermal_tables
        tt = load_thermal_tables()
        otg = c["temper"].copy()
        o.enable_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"], c["zred"], otg)
    orep, oxav, oxint, ophih = o.evolve3d(c["dt"], c["nd"], oxh, c["pos"], c["nf"])
    b = pkg.HipBackend(c["mesh"], *tables, device=0, fast=fast)
    b.set_step(c["dr"], c["vol"], c["lls"], 1.0)
    if c["lls_type"] != 1:
        b.set_lls(c["lls_type"], c["lls_grid"], c["r_max"])
    if c["clump"] is not None:
        b.set_clumping_grid(c["clump"])
    b.set_sources(c["pos"], c["nf"]); b.set_rank(0, 1); b.load(ndens=c["nd"], xh=c["xh"])
    dtemp = 0.0
    if c["temper"] is not None:
        b.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
        b.set_redshift(c["zred"])
        b.load(temperature_grid=c["temper"])
    rep = b.evolve3d_native(c["dt"])
    xh = b.fetch("xh")
    if c["temper"] is not None:
        dtemp = float(np.max(np.abs(b.fetch("temperature_grid").astype(np.float64) / otg - 1)))
    b.close()
    return dict(mesh=c["mesh"], nsrc=len(c["nf"]), niter=(rep.niter, orep.niter), converged=(rep.converged, orep.converged),
                conv=(list(rep.it_conv_flag[:rep.niter]), list(orep.it_conv_flag[:orep.niter])),
                nbox=(rep.sum_nbox_all, orep.sum_nbox_all), dx=float(np.max(np.abs(xh - oxh))), dtemp=dtemp,
                variant="lls%d%s%s" % (c["lls_type"], " clump" if c["clump"] is not None else "", " thermal" if c["temper"] is not None else ""))


if __name__ == "__main__":
    import __graft_entry__ as g
    from tests._util import load_tables
    pkg = g.load_package()
    tables = load_tables()
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    fast = len(sys.argv) > 3 and sys.argv[3] == "fast"
    bad = 0
    worst = worst_t = 0.0
    for k in range(ncase):
        r = run_step_case(seed0 + k, pkg, tables, fast)
        ok = r["niter"][0] == r["niter"][1] and r["conv"][0] == r["conv"][1] and r["nbox"][0] == r["nbox"][1] and r["converged"][0] == r["converged"][1]
        bad += not ok
        worst = max(worst, r["dx"]); worst_t = max(worst_t, r["dtemp"])
        print("case %3d %-20s mesh %-12s nsrc %d  niter %s  nbox %s  integers %s  dx %.1e  dT/T %.1e" % (k, r["variant"], r["mesh"], r["nsrc"], r["niter"], r["nbox"], ok, r["dx"], r["dtemp"]), flush=True)
    print("STEP FUZZ (%s): %d cases, %d with differing integers, worst dx %.2e, worst dT/T %.2e" % ("fast" if fast else "exact", ncase, bad, worst, worst_t))
