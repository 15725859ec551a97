#!/usr/bin/env python3
"""By hand on the GPU box:  python tests/fuzz_chains_gpu.py [first=0] [count=100] [fast=1]  -- tests/_fuzz_chains.py cases, a line each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
from tests._util import load_tables
from tests._fuzz_chains import run_chain_case

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fast = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
pkg, tables = g.load_package(), load_tables()
tot = [0, 0, 0, 0]
for seed in range(first, first + count):
    r = run_chain_case(seed, pkg, tables, fast, check_oracle=(seed % 4 == 0))
    if r["counts"]:
        tot = [a + b for a, b in zip(tot, r["counts"])]
    print(seed, r["mesh"], r["nsrc"], "chains", r["chains"], "fields", r["kinds"], "replayed/halted/eager/tails", r["counts"], flush=True)
print("all equal over %d cases; chain passes replayed %d (halted %d), launch by launch %d, gated tails %d" % (count, *tot))
