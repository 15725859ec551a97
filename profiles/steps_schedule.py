#!/usr/bin/env python3
"""Whole time steps of the reference test problem from the cold start (x = 2e-4), as SURVEY.md s8d asks
(steps 1, 5 and 14 of the schedule: neutral -> overlapping ionized regions): 256^3, S seeded sources, one
MI355X, c2r_evolve3d_dev (the native outer loop).  Prints one JSON line per step and a summary.

usage: python profiles/steps_schedule.py [--mesh 256] [--sources 1000] [--steps 14]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--sources", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=14)
    ap.add_argument("--share-of", type=int, default=1, help="sweep only rank 0's static share of the sources among this many ranks (what one GPU of a multi-GPU run does)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="c2r_set_option (include/c2ray_hip.h)")
    a = ap.parse_args()
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    n, S = a.mesh, a.sources
    tp = pkg.TestProblem(n)
    nd, xh = tp.fields(1)
    srcpos, normflux = pkg.seeded_sources(n, S)
    if a.share_of > 1:
        sh = pkg.static_source_share(S, 0, a.share_of)
        srcpos, normflux = srcpos[sh], normflux[sh]
        S = len(normflux)
    thick, thin, _ = pkg.build_tables()
    b = pkg.HipBackend(n, thick, thin, device=0, options={kv.split("=", 1)[0]: float(kv.split("=", 1)[1]) for kv in a.option})
    b.set_sources(srcpos, normflux)
    b.load(ndens=nd, xh=xh)
    rows = []
    for step in range(1, a.steps + 1):
        s = tp.step(step)
        nds, _ = tp.fields(step)
        b.load(ndens=nds)                                   # the driver rescales ndens every step (cosmology.F90:186)
        b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], s["clumping"], s["temper"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rep = b.evolve3d_native(s["dt"])
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        row = {"step": step, "outer_iterations": rep.niter, "converged": rep.converged, "wall_s": wall,
               "visited": int(rep.visited), "mean_subboxes_last": rep.sum_nbox_all / S,
               "seconds_sweep": rep.seconds_sweep, "seconds_chem": rep.seconds_chem,
               "mean_x": float(b.fetch("xh").mean()), "photcons": rep.photcons, "info": b.info().split("; chains ")[1].split("; exchanges")[0] + "; " + b.info().split("; ")[-1]}
        rows.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps({"mesh": n, "sources": S, "total_wall_s": sum(r["wall_s"] for r in rows),
                      "total_visited": sum(r["visited"] for r in rows),
                      "total_outer_iterations": sum(r["outer_iterations"] for r in rows)}))
    b.close()


if __name__ == "__main__":
    main()
