#!/usr/bin/env python3
"""Create / use / destroy contexts whose passes replay captured chain sequences, many times over: resident memory, open file
descriptors, threads and free HBM must stay flat (a leak of graph, stream or event objects would show here before it shows as a
late failure of a long test session)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
from tests._util import load_tables
pkg, tables = g.load_package(), load_tables()
n, S = 32, 130
tp = pkg.TestProblem(n); s = tp.step(1)
nd, xh = tp.fields(1, 0.999)
pos, nf = pkg.seeded_sources(n, S, seed=5)


def status():
    kv = dict(l.split(":", 1) for l in open("/proc/self/status") if ":" in l)
    free, tot = torch.cuda.mem_get_info()
    return int(kv["VmRSS"].split()[0]) // 1024, len(os.listdir("/proc/self/fd")), int(kv["Threads"]), free // (1 << 20)


for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 600):
    b = pkg.HipBackend(n, *tables, device=0, options={"chains": 2 + it % 3})
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], 1.0)
    b.set_sources(pos, nf); b.load(ndens=nd, xh=xh); b.begin_step()
    for k in range(4):
        b.iterate(s["dt"])
    b.close()
    if it % 100 == 0 or it == 599:
        print(it, "RSS MB %d  fds %d  threads %d  free HBM MB %d" % status(), flush=True)
