#!/usr/bin/env python3
"""Long randomized GPU-vs-oracle run by hand on a GPU box (tests/test_gpu_fuzz.py asserts a 25-case cut of it):
    python tests/fuzz_gpu.py [cases] [seed0] [exact|fast]
Prints per case the worst relative error of the rates, the worst |dGamma|/W (the tolerance weight of
tests/_util.gamma_ok) and the worst column-density error, and the overall maxima."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    fast = (len(sys.argv) > 3 and sys.argv[3] == "fast") or (len(sys.argv) <= 3 and os.environ.get("C2R_SWEEP_MODE") == "1")
    import __graft_entry__ as g
    from tests._util import load_tables, gamma_ok
    from tests._fuzz import run_case
    pkg = g.load_package()
    tables = load_tables()
    worst = {"gamma_rel": 0.0, "gamma_w": 0.0, "cd": 0.0, "loss": 0.0, "heat_w": 0.0}      # heat_w: |d heat| / W_heat in the cases with heating
    for case in range(ncase):
        r = run_case(seed0 + case, pkg, tables, fast)
        if os.environ.get("C2R_FUZZ_CALIBRATE") != "1":        # calibration runs only print
            assert gamma_ok(r["dgamma"], r["gamma_ref"], r["w"], fast), (case, r["mesh"], r["gamma_rel"], r["gamma_w"])
        for k in worst:
            worst[k] = max(worst[k], r[k])
        print("case %3d %-16s mesh %-14s nsrc %3d  sum_nbox %5d  visited %8d  dGamma/Gamma %.1e  dGamma/W %.1e  dcd %.1e  dloss %.1e" %
              (case, r["variant"], r["mesh"], r["nsrc"], r["nbox"], r["visited"], r["gamma_rel"], r["gamma_w"], r["cd"], r["loss"]), flush=True)
    print("FUZZ OK (%s): %d cases, worst" % ("fast" if fast else "exact", ncase), worst)


if __name__ == "__main__":
    main()
