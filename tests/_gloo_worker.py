"""Worker of tests/test_host.py::test_two_ranks_gloo_equals_one_rank (one process per rank)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist   # noqa: E402
import __graft_entry__ as g        # noqa: E402
from tests._util import F, load_case, oracle_for, load_tables, thermal_oracle_for   # noqa: E402
from tests._cpu_backend import OracleBackend                   # noqa: E402


def slab_contract(pkg, out):
    """The collectives Evolve(slab=True) hands to c2r_set_slab_chemistry (evolve.slab_collectives), on CPU tensors: unequal
    slabs, in place; after the reduce-scatter the OWN slab holds the sum over ranks, after the all-gather every slab is its
    owner's bytes everywhere."""
    import torch
    rank, npr = dist.get_rank(), dist.get_world_size()
    rs, ag = pkg.evolve.slab_collectives(dist)
    offs, cnts = [0, 6], [6, 4]
    t = torch.arange(10, dtype=torch.float64) * (rank + 1)          # rank 0: k, rank 1: 2k -> sum 3k
    rs(t, offs, cnts)
    own = t[offs[rank]:offs[rank] + cnts[rank]].clone()
    u = torch.full((40,), 7 + rank, dtype=torch.uint8)              # every rank's own byte slab is valid, the rest garbage
    boffs, bcnts = [0, 24], [24, 16]
    ag(u, boffs, bcnts)
    gathered = [torch.zeros(6, dtype=torch.float64) for _ in range(npr)]
    pad = torch.zeros(6, dtype=torch.float64); pad[:own.numel()] = own
    dist.all_gather(gathered, pad)
    if rank == 0:
        np.savez(out, own0=gathered[0].numpy(), own1=gathered[1].numpy()[:4], bytes=u.numpy())
    dist.barrier()
    dist.destroy_process_group()


def main():
    dist.init_process_group("gloo")
    pkg = g.load_package()
    if len(sys.argv) > 2 and sys.argv[2] == "slab":
        return slab_contract(pkg, sys.argv[1])
    tables = load_tables()
    m, a = load_case("evolve32_std_bubbles")
    s = m["steps"]["step001"]
    b = OracleBackend(oracle_for(s, tables, m["n"]), F(a["step001_ndens"]), F(a["step001_xh_before"]),
                      s["srcpos"], s["normflux"])
    balance = len(sys.argv) > 2 and sys.argv[2] == "balance"
    extra = {}
    if len(sys.argv) > 2 and sys.argv[2] == "thermal":      # non-isothermal step: the heating rates are all-reduced too
        m, a = load_case("evolve32_thermal")
        s = m["steps"]["step001"]
        tg = np.ascontiguousarray(a["step001_temper_before"]).copy()
        b = OracleBackend(thermal_oracle_for(s, tables, tg, m["n"]), F(a["step001_ndens"]), F(a["step001_xh_before"]),
                          s["srcpos"], s["normflux"])
        extra = dict(temper=tg, heat=b.o.phiheat)
    from tests._cpu_backend import evolve3d_piecewise       # (the double has no native loop: its own driver over Evolve's piecewise entries)
    r = evolve3d_piecewise(pkg.Evolve(b, comm=dist, balance=balance), 0.0, s["dt"], 0)
    import torch
    mine = torch.from_numpy(b.xh.copy())
    gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    if dist.get_rank() == 0:
        np.savez(sys.argv[1], niter=r["niter"], sum_nbox_all=r["sum_nbox_all"],
                 photon_loss_all=r["photon_loss_all"], xh=b.xh, phih=b.phih_grid,
                 xh_rank1=gathered[1].numpy(), conv=np.array([e["conv_flag"] for e in r["log"]]), **extra)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
