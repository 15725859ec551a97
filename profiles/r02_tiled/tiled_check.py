#!/usr/bin/env python3
"""The tiled sub-box path (C2R_TILED=1: a sub-box in three launches with the shells in LDS) against the per-shell launches,
fast mode, same inputs: sub-box counts and visited cells equal, photon loss and rates equal to the order of the sums."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from tests._util import load_tables
pkg = g.load_package()
tables = load_tables()


def run(n, S, seed, x, tiled):
    os.environ["C2R_TILED"] = "1" if tiled else "0"
    rng = np.random.default_rng(seed)
    tp = pkg.TestProblem(n); s = tp.step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(n ** 3))).astype(np.float32)
    xh = np.clip(x * (1.0 - 1e-3 * rng.random(n ** 3)), 1e-6, 1 - 1e-9)
    pos, nf = pkg.seeded_sources(n, S, seed=seed)
    b = pkg.HipBackend(n, *tables, device=0, fast=True)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh); b.begin_step(); b.zero_rates()
    out = b.pass_sources()
    res = (out, b.fetch("phih_grid"), b.last_nbox().copy())
    b.close()
    return res


for (n, S, seed, x) in ((64, 5, 1, 0.9995), (64, 40, 2, 0.999), (96, 12, 3, 0.9995), (128, 9, 4, 0.9999), (48, 7, 5, 0.99)):
    a = run(n, S, seed, x, False); t = run(n, S, seed, x, True)
    nz = a[1] != 0
    rel = np.abs(t[1][nz] / a[1][nz] - 1)
    print("n=%d S=%d: nbox %s %s  visited %s  loss rel %.2e  Gamma zero-pattern %s  max rel %.2e  median %.1e" %
          (n, S, a[0][1], np.array_equal(a[2], t[2]), a[0][2] == t[0][2], abs(t[0][0] / a[0][0] - 1) if a[0][0] else 0.0,
           np.array_equal(a[1] == 0, t[1] == 0), rel.max() if rel.size else 0, np.median(rel) if rel.size else 0), flush=True)
