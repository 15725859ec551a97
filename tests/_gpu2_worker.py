"""Worker of tests/test_gpu_two_ranks.py: two ranks, both on cuda:0 (the GPU box has one GPU), gloo
for the collective (RCCL refuses two ranks on one device; the data path is otherwise the product's:
HipBackend + Evolve with comm=torch.distributed, and the C++ loop with the all-reduce callback)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                        # noqa: E402
import torch.distributed as dist   # noqa: E402
import __graft_entry__ as g        # noqa: E402
from tests._util import F, load_case, load_tables   # noqa: E402


def main():
    dist.init_process_group("gloo")
    pkg = g.load_package()
    tables = load_tables()
    m, a = load_case("evolve32_std_bubbles")
    s = m["steps"]["step001"]
    out = {}
    for mode in ("python", "native", "balanced", "native_slab"):
        b = pkg.HipBackend(m["n"], *tables, device=0)
        b.set_step((s["dr1"], s["dr2"], s["dr3"]), s["vol"], s["coldensh_LLS"], s["clumping"])
        b.set_sources(s["srcpos"], s["normflux"])
        b.load(ndens=F(a["step001_ndens"]), xh=F(a["step001_xh_before"]))
        ev = pkg.Evolve(b, comm=dist, balance=(mode == "balanced"), slab=(mode == "native_slab"))   # sets rank/size and the callbacks in the context
        if mode in ("python", "balanced"):
            r = ev.evolve3D(0.0, s["dt"], 0)
            niter, nbox, loss, conv = r["niter"], r["sum_nbox_all"], r["photon_loss_all"], [e["conv_flag"] for e in r["log"]]
        else:
            rep = b.evolve3d_native(s["dt"])
            niter, nbox, loss, conv = rep.niter, rep.sum_nbox_all, rep.photon_loss_all, list(rep.it_conv_flag[:rep.niter])
        out[mode] = dict(niter=niter, nbox=nbox, loss=loss, conv=np.array(conv), xh=b.fetch("xh"), phih=b.fetch("phih_grid"))
        b.close()
    # non-isothermal step (isothermal=.false.): the heating rates are sharded and all-reduced like Gamma
    # (evolve.F90:604-609), the temperature evolution is replicated with the global pass
    from tests._util import load_thermal_tables
    tt = load_thermal_tables()
    mt, at = load_case("evolve32_thermal")
    st = mt["steps"]["step001"]
    for mode in ("thermal_python", "thermal_native", "thermal_native_slab", "thermal_native_det", "thermal_native_slab_det"):
        b = pkg.HipBackend(mt["n"], *tables, device=0, deterministic=mode.endswith("_det"))
        b.set_step((st["dr1"], st["dr2"], st["dr3"]), st["vol"], st["coldensh_LLS"], st["clumping"])
        b.set_thermal(tt["heat_thick"], tt["heat_thin"], tt["cool_logT"], tt["cool_logL"])
        b.set_redshift(st["zred"])
        b.set_sources(st["srcpos"], st["normflux"])
        b.load(ndens=F(at["step001_ndens"]), xh=F(at["step001_xh_before"]), temperature_grid=at["step001_temper_before"])
        ev = pkg.Evolve(b, comm=dist, slab="slab" in mode)
        if mode == "thermal_python":
            r = ev.evolve3D(0.0, st["dt"], 0)
            niter, conv = r["niter"], [e["conv_flag"] for e in r["log"]]
        else:
            rep = b.evolve3d_native(st["dt"])
            niter, conv = rep.niter, list(rep.it_conv_flag[:rep.niter])
        out[mode] = dict(niter=niter, conv=np.array(conv), xh=b.fetch("xh"), heat=b.fetch("phiheat_grid"),
                         temper=b.fetch("temperature_grid"), phih=b.fetch("phih_grid"), xh_av=b.fetch("xh_av"))
        b.close()
    if dist.get_rank() == 0:
        flat = {"%s_%s" % (k, kk): v for k, d in out.items() for kk, v in d.items()}
        np.savez(sys.argv[1], **flat)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
