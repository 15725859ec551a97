#!/usr/bin/env python3
"""Pass-only timing on a FROZEN state for ablation / variant builds of the library (C2RAY_HIP_LIB selects the build):
256^3, 1000 seeded sources, uniform x (default 0.9995: every source traces to the limits), no global pass between
the passes, so every variant sees the same state whatever it does to the rates.  Prints ms per pass, visited pairs
per pass and ps per visited pair -- variants that change the photon loss change the sub-box counts, so compare the
last column.     python profiles/micro/ablate.py [passes] [x] [sources]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
npass = int(sys.argv[1]) if len(sys.argv) > 1 else 3
x = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9995
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
n = 256
tp = pkg.TestProblem(n); s = tp.step(1)
nd, xh = tp.fields(1, x)
pos, nf = pkg.seeded_sources(n, S)
thick, thin, _ = pkg.build_tables()
b = pkg.HipBackend(n, thick, thin, device=0, fast=os.environ.get("ABL_MODE", "fast") == "fast")
b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], s["clumping"], s["temper"])
b.set_sources(pos, nf); b.set_rank(0, 1); b.load(ndens=nd, xh=xh); b.begin_step()
import torch
b.zero_rates(); b.pass_sources(); torch.cuda.synchronize()
t0 = time.perf_counter(); vis = 0
for _ in range(npass):
    b.zero_rates(); l, nb, v = b.pass_sources(); vis += v
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%-12s %8.2f ms/pass  visited/pass %.4e  mean sub-boxes %.2f  %7.2f ps/visit  (%.3e visits/s)" %
      (os.path.basename(os.environ.get("C2RAY_HIP_LIB", "base")).replace("libc2ray_hip_", "").replace(".so", ""),
       1e3 * dt / npass, vis / npass, nb / S, 1e12 * dt / vis, vis / dt))
