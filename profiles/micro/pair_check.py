#!/usr/bin/env python3
"""The look-ahead pairs (k_sweep_pair_fast: two shells per launch with few sources, C2R_PAIR_SHELLS=1, the default) against
one launch per shell (C2R_PAIR_SHELLS=0), fast mode, same inputs: sub-box counts, visited cells and photon loss equal,
rates equal bit for bit in deterministic-rates mode and to the order of the atomic adds otherwise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
from tests._util import load_tables
pkg = g.load_package()
tables = load_tables()
LEVEL = "1"


def run(n, S, seed, x, strip, det=False):
    os.environ["C2R_PAIR_SHELLS"] = strip
    rng = np.random.default_rng(seed)
    tp = pkg.TestProblem(n); s = tp.step(1)
    nd = (s["ndens"] * np.exp(0.5 * rng.standard_normal(n ** 3))).astype(np.float32)
    xh = np.clip(x * (1.0 - 1e-3 * rng.random(n ** 3)), 1e-6, 1 - 1e-9)
    pos, nf = pkg.seeded_sources(n, S, seed=seed)
    b = pkg.HipBackend(n, *tables, device=0, fast=True, deterministic=det)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh); b.begin_step(); b.zero_rates()
    out = b.pass_sources()
    res = (out, b.fetch("phih_grid"), b.last_nbox().copy())
    b.close()
    return res


bad = 0
for (n, S, seed, x, det) in ((64, 5, 1, 0.9995, False), (64, 30, 2, 0.999, False), (96, 12, 3, 0.9995, True), (128, 9, 4, 0.9999, False),
                            (48, 7, 5, 0.99, True), (130, 3, 6, 0.99995, True), (128, 1, 7, 0.9995, True)):
    a = run(n, S, seed, x, "0", det); t = run(n, S, seed, x, LEVEL, det)
    nz = a[1] != 0
    rel = np.abs(t[1][nz] / a[1][nz] - 1)
    ok = (np.array_equal(a[2], t[2]) and a[0][2] == t[0][2] and np.array_equal(a[1] == 0, t[1] == 0) and
          (rel.max() if rel.size else 0) < 1e-13 and t[0][0] == a[0][0] and (not det or np.array_equal(a[1], t[1])))
    bad += 0 if ok else 1
    print("n=%d S=%d det=%d: nbox %s %s  visited %s  loss rel %.2e  Gamma zero-pattern %s  max rel %.2e  median %.1e  %s" %
          (n, S, det, a[0][1], np.array_equal(a[2], t[2]), a[0][2] == t[0][2], abs(t[0][0] / a[0][0] - 1) if a[0][0] else 0.0,
           np.array_equal(a[1] == 0, t[1] == 0), rel.max() if rel.size else 0, np.median(rel) if rel.size else 0,
           "ok" if ok else "MISMATCH"), flush=True)
sys.exit(1 if bad else 0)
