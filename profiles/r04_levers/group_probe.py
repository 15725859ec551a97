#!/usr/bin/env python3
"""Does a small group of sources trace the SAME field faster per source (shell planes resident in the 256 MB MALL between the
launch that writes them and the one that reads them)?  One relaxed field from the 1000-source bench state; passes over the first
S sources of the list with per-launch event timing; per-source time of the per-shell launches (sub-boxes 3..26)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
n = 256
tp = pkg.TestProblem(n); s = tp.step(1)
nd, xh = tp.fields(1, 0.999)
pos, nf = pkg.seeded_sources(n, 1000)
thick, thin, _ = pkg.build_tables()
os.environ["C2R_PAIR_SHELLS"] = "0"; os.environ["C2R_GRAPH"] = "0"
b = pkg.HipBackend(n, thick, thin, device=0, fast=True)
b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], s["clumping"], s["temper"])
b.set_sources(pos, nf); b.load(ndens=nd, xh=xh); b.begin_step()
for _ in range(2):
    b.zero_rates(); b.pass_sources(); b.global_pass(s["dt"])
for S in (1000, 250, 96, 48, 32, 16):
    b.set_sources(pos[:S], nf[:S])
    b.zero_rates(); b.pass_sources()             # warm
    b.profile(1); b.zero_rates(); loss, nbox, vis = b.pass_sources(); p = b.profile_read(); b.profile(0)
    print(json.dumps({"S": S, "mean_nbox": nbox / S, "sweep_ms": p["sweep_ms"], "launches": p["sweep_launches"],
                      "ms_per_source": p["sweep_ms"] / S}))
b.close()
