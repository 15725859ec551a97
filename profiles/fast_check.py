#!/usr/bin/env python3
"""Error survey of the two sweep modes (C2R_SWEEP_EXACT / C2R_SWEEP_FAST) against the reference fixtures
and the oracle: prints one line per case and mode with the worst relative error of column densities,
rates, photon loss and the integer results.  Exploratory companion of tests/test_gpu_parity.py (the
asserted tolerances live there); run on the GPU box:  python profiles/fast_check.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g                                   # noqa: E402
from tests._util import F, load_case, load_tables, oracle_for, expand, relerr   # noqa: E402

pkg = g.load_package()
tables = load_tables()


def backend(m, n, nd, xh, fast):
    b = pkg.HipBackend(n, *tables, device=0, fast=fast)
    b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
    b.set_sources(m["srcpos"], m["normflux"]); b.set_rank(0, 1); b.load(ndens=nd, xh=xh)
    return b


def sweep_case(name):
    m, a = load_case(name)
    n = m["n"]
    nd, xh = F(expand(a["ndens"], n)), F(expand(a["xh"], n))
    res = {}
    for fast in (False, True):
        b = backend(m, n, nd, xh, fast)
        b.begin_step(); b.zero_rates()
        loss, nbox, vis = b.pass_sources()
        phih = b.fetch("phih_grid")
        b.zero_rates()
        nb1, l1, v1, cd = b.do_source(m["ns_dump"], want_coldens=True)
        res[fast] = (loss, nbox, phih, cd)
        line = "%-22s %-5s nbox %s(ref %s) loss %.2e" % (name, "fast" if fast else "exact", nbox, m["sum_nbox"],
                                                          abs(loss / m["photon_loss"] - 1) if m["photon_loss"] else 0.0)
        if "phih" in a:
            ref, cref = F(a["phih"]), F(a["coldensh_out"])
            line += "  Gamma %.2e  cd %.2e  zero-pattern %s" % (relerr(phih, ref, floor=1e-60), relerr(cd, cref),
                                                                 np.array_equal(phih == 0, ref == 0) and np.array_equal(cd == 0, cref == 0))
        print(line, flush=True)
        b.close()
    e, f = res[False], res[True]
    nz = e[2] != 0
    print("%-22s fast-vs-exact: Gamma %.2e (median %.1e)  cd %.2e  loss %.2e" %
          (name, relerr(f[2], e[2], floor=1e-60), float(np.median(np.abs(f[2][nz] / e[2][nz] - 1))) if nz.any() else 0,
           relerr(f[3], e[3]), abs(f[0] / e[0] - 1) if e[0] else 0.0), flush=True)


def evolve_case(name):
    m, a = load_case(name)
    n = m["n"]
    for tag, s in m["steps"].items():
        for fast in (False, True):
            b = backend(s, n, F(a[tag + "_ndens"]), F(a[tag + "_xh_before"]), fast)
            rep = b.evolve3d_native(s["dt"])
            ok = rep.niter == s["niter"] and list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"] and rep.sum_nbox_all == s["sum_nbox_all"]
            dx = float(np.max(np.abs(b.fetch("xh") - F(a[tag + "_xh_after"]))))
            line = "%-22s %s %-5s integers %s  niter %d  dx %.2e" % (name, tag, "fast" if fast else "exact", ok, rep.niter, dx)
            if tag + "_phih_grid" in a:
                line += "  Gamma %.2e" % relerr(b.fetch("phih_grid"), F(a[tag + "_phih_grid"]), floor=1e-60)
            print(line, flush=True)
            b.close()


def random_case(n, nsrc, seed):
    rng = np.random.default_rng(seed)
    s = pkg.TestProblem(n).step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(n ** 3))).astype(np.float32)
    xh = np.clip(10.0 ** rng.uniform(-4, 0, n ** 3) * 0.9999, 1e-6, 0.9999)
    x3 = xh.reshape((n, n, n), order="F"); c = max(1, n // 8)
    x3[:] = np.repeat(np.repeat(np.repeat(x3[::c, ::c, ::c], c, 0), c, 1), c, 2)[:n, :n, :n]
    pos, nf = pkg.seeded_sources(n, nsrc, seed=seed)
    o = oracle_for(s, tables, n)
    phih_o = np.zeros(o.ncell)
    oloss, onb, ovis = o.pass_sources(nd, F(x3), phih_o, pos, nf)
    for fast in (False, True):
        b = backend(dict(s, srcpos=pos, normflux=nf), n, nd, F(x3), fast)
        b.begin_step(); b.zero_rates()
        loss, nbox, vis = b.pass_sources()
        phih = b.fetch("phih_grid")
        err = np.abs(phih - phih_o) / np.maximum(np.abs(phih_o), 1e-60)
        print("random n=%d S=%d %-5s nbox %s  loss %.2e  Gamma max %.2e  99.9%% %.2e  median %.2e" %
              (n, nsrc, "fast" if fast else "exact", (nbox, vis) == (onb, ovis), abs(loss / oloss - 1) if oloss else 0,
               err.max(), np.quantile(err[phih_o != 0], 0.999), np.median(err[phih_o != 0])), flush=True)
        b.close()


if __name__ == "__main__":
    for nm in ("sweep32_std_x999", "sweep33_std_x999", "sweep32_bubbles"):
        sweep_case(nm)
    for nm in ("evolve32_onesrc", "evolve32_std_bubbles", "evolve64_std_bubbles"):
        evolve_case(nm)
    for args in ((24, 7, 1), (40, 20, 2), (48, 33, 3), (64, 40, 5)):
        random_case(*args)
