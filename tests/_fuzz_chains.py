"""Randomized cases for the replayed chains (csrc/sweep.hip run_chains), shared by tests/test_gpu_fuzz.py (a short cut, asserting)
and tests/fuzz_chains_gpu.py (the long run by hand): a random non-cubic mesh, 64 - 400 sources, SIX outer iterations through
c2r_iterate with the field changed between some of them (traces grow, shrink or stay) -- once with the chains' passes replayed
as captured launch sequences behind a device-gated tail (the default), once driven launch by launch (chain_graph = 0).  Every
iteration: photon loss (bits), sub-box sum, visited pairs, non-converged count equal; the first pass's integers equal the
oracle's."""
import numpy as np


def run_chain_case(seed, pkg, tables, fast, check_oracle=True):
    rng = np.random.default_rng(1000 + seed)
    s = pkg.TestProblem(32).step(1)
    mesh = tuple(int(v) for v in rng.integers(14, 41, 3))
    ncell = mesh[0] * mesh[1] * mesh[2]
    dr = tuple(float(s["dr1"] * 10.0 ** rng.uniform(-0.2, 0.5) * f) for f in rng.uniform(0.8, 1.3, 3))
    nd = (s["ndens"] * np.exp(0.6 * rng.standard_normal(ncell)) * 10.0 ** rng.uniform(-1.0, 0.0)).astype(np.float32)
    nsrc = int(rng.integers(64, 401))
    pos = np.stack([rng.integers(-2, mesh[d] + 4, nsrc) for d in range(3)], axis=1).astype(np.int32)
    nf = 10.0 ** rng.uniform(4, 10, nsrc)
    nf[rng.random(nsrc) < 0.03] = 0.0
    lls = s["coldensh_LLS"] * 10.0 ** rng.uniform(-1, 1)

    def field(kind):
        if kind == 0:       # highly ionized with clumps: long traces
            x = 1.0 - 10.0 ** rng.uniform(-6.0, -3.0, ncell)
            c = rng.random(ncell) < 0.02
            x[c] = 10.0 ** rng.uniform(-4, -0.3, int(c.sum()))
            return x
        if kind == 1:       # mostly neutral: everything ends in the first sub-boxes
            return np.clip(10.0 ** rng.uniform(-3.0, 0, ncell) * 0.99999, 1e-7, 0.99999)
        return 1.0 - 10.0 ** rng.uniform(-4.0, -1.5, ncell)      # in between
    kinds = [int(rng.integers(0, 3))]
    for _ in range(5):
        kinds.append(kinds[-1] if rng.random() < 0.55 else int(rng.integers(0, 3)))
    fields, last = [], None
    for k, kind in enumerate(kinds):
        if k == 0 or kind != kinds[k - 1]:
            last = field(kind)
        fields.append(last)
    chains = int(rng.choice([0, 0, 2, 3, 4]))
    hist = {}
    for cg in (0, 1):
        opts = {"chain_graph": cg}
        if chains:
            opts["chains"] = chains
        b = pkg.HipBackend(mesh, *tables, device=0, fast=fast, options=opts)
        b.set_step(dr, dr[0] * dr[1] * dr[2], lls, 1.0)
        b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=fields[0]); b.begin_step()
        out = []
        for x in fields:
            b.load(xh_av=x, xh_intermed=fields[0])
            out.append(tuple(b.iterate(s["dt"])[:4]))
        hist[cg] = (out, b.info())
        b.close()
    assert hist[0][0] == hist[1][0], (seed, mesh, nsrc, chains, kinds, hist[0][0], hist[1][0], hist[1][1])
    if check_oracle:
        from oracle.oracle import Oracle
        o = Oracle(mesh, dr, dr[0] * dr[1] * dr[2], lls, *tables)
        g = np.zeros(ncell)
        oloss, onb, ovis = o.pass_sources(nd, fields[0], g, pos, nf)
        assert (onb, ovis) == hist[1][0][0][1:3], (seed, mesh, nsrc)
        assert abs(hist[1][0][0][0] - oloss) <= 1e-10 * abs(oloss) + 1e-300
    import re
    m = re.search(r"chain passes replayed (\d+) \(halted (\d+)\), launch by launch (\d+), iterations with a device-gated tail (\d+)", hist[1][1])
    return dict(mesh=mesh, nsrc=nsrc, chains=chains, kinds=kinds, counts=tuple(int(v) for v in m.groups()) if m else None)
