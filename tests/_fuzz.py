"""Randomized GPU-vs-oracle cases shared by tests/test_gpu_fuzz.py (asserting, 25 cases per mode) and
tests/fuzz_gpu.py (the long run by hand).  Random non-cubic meshes (every extent 3..44, odd and even), 1..12
sources anywhere (also outside [1,N]), rates over 6 decades, density and ionization fields with structure,
(one case in seven: 64 - 300 or 769 - 900 sources -- chains in flight, the plane-ordered block mapping),
both fully and barely ionized gas, so that sub-boxes end anywhere between the first and the clipped last; the
row-group tiling of k_sweep_shell (three rows per thread, groups per sign class) meets every remainder; one case in
two also draws a non-default switch: type_of_LLS 2 or 3, source-ordered (deterministic) Gamma accumulation, one
source per batch; three in ten run in a non-isothermal context and compare the heating rates too (asserted here, within
the Gamma tolerance with the heating weight W_heat); a quarter carry the second (X-ray) source type with the reference's
power-law tables and a random NormFlux_xray per source -- with its heating tables where the context heats (round 5)."""
import numpy as np


def make_case(seed, pkg):
    rng = np.random.default_rng(seed)
    s = pkg.TestProblem(32).step(1)
    mesh = tuple(int(v) for v in rng.integers(3, 45, 3))
    ncell = mesh[0] * mesh[1] * mesh[2]
    scale = 10.0 ** rng.uniform(-0.3, 0.6)
    dr = tuple(float(s["dr1"] * scale * f) for f in rng.uniform(0.7, 1.4, 3))
    nd = (s["ndens"] * np.exp(0.7 * rng.standard_normal(ncell))).astype(np.float32)
    if seed % 3 == 0:       # mostly neutral gas: everything ends in the first sub-boxes
        lo = rng.choice([-5.0, -2.0, -0.5])
        xh = np.clip(10.0 ** rng.uniform(lo, 0, ncell) * 0.99999, 1e-7, 0.99999)
    else:                   # highly ionized with neutral clumps: rays run to the trace limits
        xh = 1.0 - 10.0 ** rng.uniform(-6.5, -3.0, ncell)
        clumps = rng.random(ncell) < 0.02
        xh[clumps] = 10.0 ** rng.uniform(-4, -0.3, int(clumps.sum()))
        nd = (nd * 10.0 ** rng.uniform(-1.5, 0.0)).astype(np.float32)
    nsrc = int(rng.integers(1, 13))
    pos = np.stack([rng.integers(-3, mesh[d] + 5, nsrc) for d in range(3)], axis=1).astype(np.int32)
    nf = 10.0 ** rng.uniform(4, 10, nsrc)
    if rng.random() < 0.2:
        nf[rng.integers(0, nsrc)] = 0.0
    lls = s["coldensh_LLS"] * 10.0 ** rng.uniform(-1, 1)
    k = int(rng.integers(0, nsrc))          # the source whose column densities are compared
    # the switches beside the shipped configuration (drawn last: the fields above are those of the earlier runs):
    # type_of_LLS 2 (per-cell column) / 3 (hard barrier), source-ordered Gamma accumulation, several source batches
    lls_type = int(rng.choice([1, 1, 2, 3]))
    lls_grid = (lls * 10.0 ** rng.uniform(-1.0, 1.0, ncell)).astype(np.float32) if lls_type == 2 else None
    r_max = float(dr[0] * rng.uniform(2.0, 0.7 * max(mesh))) if lls_type == 3 else 0.0
    deterministic = bool(rng.random() < 0.25)
    scratch = int(rng.choice([0, 0, 1]))     # 1 byte: one source per batch
    heating = bool(rng.random() < 0.3)       # a non-isothermal context: the sweep also accumulates the heating rates
    # (drawn last again, round 5) the second source type of photoion_rates: NormFlux_xray per source, some of them zero
    xray = bool(rng.random() < 0.25)         # (with heating: the X-ray type's heating tables too)
    nfx = 10.0 ** rng.uniform(3, 9, nsrc) * (rng.random(nsrc) < 0.7)
    # (drawn last, late round 5) MANY sources on the same fields: 64 - 300 run as chains in flight (sweep.hip run_chains), more than
    # 768 as one chain whose far shells use the XCD-aware, plane-ordered block mapping (k_sweep_shell_xcd) -- the schedules of the
    # headline workload, which 1 - 12 sources never reach
    r = rng.random()
    if r < 0.14:
        nsrc = int(rng.integers(64, 301)) if r < 0.08 else int(rng.integers(769, 901))
        pos = np.stack([rng.integers(-3, mesh[d] + 5, nsrc) for d in range(3)], axis=1).astype(np.int32)
        nf = 10.0 ** rng.uniform(4, 10, nsrc)
        nf[rng.random(nsrc) < 0.03] = 0.0
        nfx = 10.0 ** rng.uniform(3, 9, nsrc) * (rng.random(nsrc) < 0.7)
    return dict(xray=xray, nfx=nfx, heating=heating, mesh=mesh, dr=dr, vol=dr[0] * dr[1] * dr[2], nd=nd, xh=xh, pos=pos, nf=nf, lls=lls, k=k,
                lls_type=lls_type, lls_grid=lls_grid, r_max=r_max, deterministic=deterministic, scratch=scratch)


def run_case(seed, pkg, tables, fast):
    """One pass of all sources on the GPU and in the oracle.  Returns the comparison metrics; integer
    results and zero patterns are asserted here (they are exact in both sweep modes)."""
    from oracle.oracle import Oracle
    c = make_case(seed, pkg)
    mesh, ncell = c["mesh"], c["nd"].size
    o = Oracle(mesh, c["dr"], c["vol"], c["lls"], *tables, lls_type=c["lls_type"], R_max_LLS=c["r_max"], lls_grid=c["lls_grid"])
    w = o.enable_tolerance_weight()
    wh = None
    if c["heating"]:
        from tests._util import load_thermal_tables
        tt = load_thermal_tables()
        o.enable_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"], 9.0, np.zeros((ncell, 3), dtype=np.float32))
        wh = o.enable_heat_tolerance_weight()
    if c["xray"]:
        import os
        from tests._util import GOLDEN
        pl = np.load(os.path.join(GOLDEN, "tables_pl.npz"))
        o.enable_xray(pl["thick"], pl["thin"], c["nfx"])
        if c["heating"]:
            import ctypes as C
            sed = pkg.SedParams(); pkg.load_library().c2r_default_sed_power_law(C.byref(sed))
            xhk, xhn = pkg._capi.build_heat_tables(sed)
            o.enable_xray_heat(xhk, xhn)
    phih_o = np.zeros(ncell)
    oloss, onb, ovis = o.pass_sources(c["nd"], c["xh"], phih_o, c["pos"], c["nf"])
    w = w.copy()
    heat_o = o.phiheat.copy() if c["heating"] else None
    wh = wh.copy() if c["heating"] else None
    b = pkg.HipBackend(mesh, *tables, device=0, fast=fast, deterministic=c["deterministic"], scratch_bytes=c["scratch"])
    b.set_step(c["dr"], c["vol"], c["lls"], 1.0)
    if c["lls_type"] != 1:
        b.set_lls(c["lls_type"], c["lls_grid"], c["r_max"])
    b.set_sources(c["pos"], c["nf"]); b.set_rank(0, 1); b.load(ndens=c["nd"], xh=c["xh"])
    if c["heating"]:
        b.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
    if c["xray"]:
        b.set_xray(pl["thick"], pl["thin"], c["nfx"])
        if c["heating"]:
            b.set_xray_heat(xhk, xhn)
    b.begin_step(); b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    phih = b.fetch("phih_grid")
    heat_w = 0.0
    if c["heating"]:
        heat = b.fetch("phiheat_grid")
        assert np.array_equal(heat == 0, heat_o == 0), (seed, mesh)
        hz = heat_o != 0
        from tests._util import TOL
        t = TOL["fast" if fast else "exact"]
        excess = np.abs(heat - heat_o) - (t["gamma_rtol"] * heat_o + t["gamma_wtol"] * wh)
        assert excess.max() <= 0, (seed, mesh, "heating rate out of tolerance", float(excess.max()))
        heat_w = float(np.max(np.abs(heat - heat_o)[hz] / wh[hz])) if hz.any() else 0.0
    assert (nbox, vis) == (onb, ovis), (seed, mesh, nbox, onb, vis, ovis)
    assert np.array_equal(phih == 0, phih_o == 0), (seed, mesh)
    k = c["k"]
    nb1, l1, v1, cd = b.do_source(k + 1, want_coldens=True)
    nbo, lo1, vo, cdo = o.do_source(c["nd"], c["xh"], np.zeros(ncell), c["pos"][k], c["nf"][k], c["nfx"][k] if c["xray"] else 0.0)
    b.close()
    assert nb1 == nbo and v1 == vo, (seed, mesh, k)
    assert np.array_equal(cd == 0, cdo == 0), (seed, mesh, k)
    d = np.abs(phih - phih_o)
    nz = phih_o != 0
    return dict(mesh=mesh, nsrc=len(c["nf"]), nbox=nbox, visited=vis, variant="lls%d%s%s%s" % (c["lls_type"], " det" if c["deterministic"] else "", " batches" if c["scratch"] else "", " heat" if c["heating"] else "") + (" xray" if c["xray"] else ""), heat_w=heat_w,
                loss=abs(loss - oloss) / max(abs(oloss), 1e-300),
                cd=float(np.max(np.abs(cd - cdo) / np.maximum(cdo, 1e-300))),
                gamma_rel=float(np.max(d[nz] / phih_o[nz])) if nz.any() else 0.0,
                gamma_w=float(np.max(d[nz] / w[nz])) if nz.any() else 0.0,        # |dGamma| / W, see oracle_cfg.tolw
                dgamma=d, gamma_ref=phih_o, w=w)
