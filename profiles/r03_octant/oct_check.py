#!/usr/bin/env python3
"""The strip path (C2R_OCTANT=n: whole sub-boxes of the first n face pairs in k_sweep_strip_fast, shells through LDS)
against the per-shell launches, fast mode, same inputs: sub-box counts and visited cells equal, photon loss equal to the
order of the block sums, rates equal to the order of the atomic adds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from tests._util import load_tables
pkg = g.load_package()
tables = load_tables()
LEVEL = sys.argv[1] if len(sys.argv) > 1 else "1"


def run(n, S, seed, x, strip):
    os.environ["C2R_OCTANT"] = strip
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


bad = 0
for (n, S, seed, x) in ((64, 5, 1, 0.9995), (64, 40, 2, 0.999), (96, 12, 3, 0.9995), (128, 9, 4, 0.9999), (48, 7, 5, 0.99), (130, 3, 6, 0.99995)):
    a = run(n, S, seed, x, "0"); t = run(n, S, seed, x, LEVEL)
    nz = a[1] != 0
    rel = np.abs(t[1][nz] / a[1][nz] - 1)
    ok = (np.array_equal(a[2], t[2]) and a[0][2] == t[0][2] and np.array_equal(a[1] == 0, t[1] == 0) and
          (rel.max() if rel.size else 0) < 1e-13 and (abs(t[0][0] / a[0][0] - 1) if a[0][0] else 0.0) < 1e-13)
    bad += 0 if ok else 1
    print("n=%d S=%d: nbox %s %s  visited %s  loss rel %.2e  Gamma zero-pattern %s  max rel %.2e  median %.1e  %s" %
          (n, S, a[0][1], np.array_equal(a[2], t[2]), a[0][2] == t[0][2], abs(t[0][0] / a[0][0] - 1) if a[0][0] else 0.0,
           np.array_equal(a[1] == 0, t[1] == 0), rel.max() if rel.size else 0, np.median(rel) if rel.size else 0,
           "ok" if ok else "MISMATCH"), flush=True)
sys.exit(1 if bad else 0)
