#!/usr/bin/env python3
"""Per-step times of one GPU's share of the 8-GPU bench (125 of the 1000 sources, static share of rank 0) -- the probe behind
bench.py's `one_gpu_share_of_8` leg: how many steps the leg needs before the numbers settle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
n, S = 256, 1000
tp = pkg.TestProblem(n); s = tp.step(1)
nd, xh = tp.fields(1, 0.999)
srcpos, normflux = pkg.seeded_sources(n, S)
thick, thin, _ = pkg.build_tables()
for label, idx in (("share of rank 0", pkg.static_source_share(S, 0, 8)), ("seeded 125", None)):
    if idx is None:
        pos, nf = pkg.seeded_sources(n, 125)
    else:
        pos, nf = srcpos[idx], normflux[idx]
    b = pkg.HipBackend(n, thick, thin, device=0, fast=True)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], s["clumping"], s["temper"])
    b.set_sources(pos, nf); b.load(ndens=nd, xh=xh)
    ev = pkg.Evolve(b); b.begin_step()
    ts = []
    for k in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev.iteration(k, s["dt"])
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(label, b.info().split("; ")[-3], "ms per step:", " ".join("%.2f" % t for t in ts), " sum_nbox", ev.sum_nbox_all, flush=True)
    b.close()
