import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
from tests._util import F, load_case, load_tables, load_thermal_tables, thermal_oracle_for
pkg = g.load_package()
TAB = load_tables(); TT = load_thermal_tables()
def backend(m, n, fast):
    b = pkg.HipBackend(n, *TAB, device=0, fast=fast)
    b.set_step((m["dr1"], m["dr2"], m["dr3"]), m["vol"], m["coldensh_LLS"], m["clumping"])
    b.set_thermal(TT["heat_thick"], TT["heat_thin"], TT["cool_logT"], TT["cool_logL"])
    b.set_redshift(m["zred"])
    b.set_sources(m["srcpos"], m["normflux"]); b.set_rank(0, 1)
    return b
for fast in (False, True):
    m, a = load_case("sweep32_thermal"); n = m["n"]
    b = backend(m, n, fast)
    b.load(ndens=F(a["ndens"]), xh=F(a["xh"]), temperature_grid=np.full(n**3, 1e4, dtype=np.float32))
    b.begin_step(); b.zero_rates()
    loss, nb, vis = b.pass_sources()
    ph = b.fetch("phih_grid"); he = b.fetch("phiheat_grid")
    rp, rh = F(a["phih"]), F(a["phiheat"])
    nz = rh > 0
    print("fast" if fast else "exact", "sweep: nbox", nb, m["sum_nbox"], "zero pattern", np.array_equal(he == 0, rh == 0),
          "max rel heat", np.max(np.abs(he[nz] / rh[nz] - 1)), "max rel gamma", np.max(np.abs(ph[nz] / rp[nz] - 1)),
          "heat abs/max", np.max(np.abs(he - rh)) / rh.max())
    b.close()
    m, a = load_case("evolve32_thermal")
    for tag, s in m["steps"].items():
        b = backend(s, n, fast)
        b.load(ndens=F(a[tag + "_ndens"]), xh=F(a[tag + "_xh_before"]), temperature_grid=a[tag + "_temper_before"])
        rep = b.evolve3d_native(s["dt"])
        xh = b.fetch("xh"); tg = b.fetch("temperature_grid"); he = b.fetch("phiheat_grid")
        ref_t = a[tag + "_temper_after"]; rh = F(a[tag + "_phiheat_grid"])
        nz = rh > 0
        print(" ", tag, "niter", rep.niter, s["niter"], "nonconv", list(rep.it_conv_flag[:rep.niter]) == s["log"]["nonconv"], list(rep.it_conv_flag[:rep.niter]), s["log"]["nonconv"],
              "dx", np.max(np.abs(xh - F(a[tag + "_xh_after"]))), "dT rel", np.max(np.abs(tg / ref_t - 1)), "n(T differs)", np.count_nonzero(tg != ref_t),
              "heat rel", np.max(np.abs(he[nz] / rh[nz] - 1)), "totrec", rep.totrec / s["totrec"] - 1)
        b.close()
