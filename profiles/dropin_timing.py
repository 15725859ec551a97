#!/usr/bin/env python3
"""Wall clock of the reference's OWN program (C2Ray.F90 + every set-up module, unmodified) per evolve3D time step, with
its evolve modules (a) as the reference has them -- serial and OpenMP builds, oracle/_ref/N<mesh>/{serial,omp}/c2ray_test --
and (b) replaced by the HIP drop-in -- oracle/_ref/N<mesh>/hip/c2ray_test_hip = C2Ray.F90 + evolve_hip.F90 + libc2ray_hip.so.
This is the boundary north_star names ("drops in behind the existing Fortran driver") and the only wall clock a C2-Ray
user sees.

Per time step: the reference legs from results/Timings.log ("Time before starting iteration" .. "Time after iteration",
evolve.F90:166, :272; 0.1 s resolution), the HIP leg from the shim's own log line (system_clock, nanoseconds):
    c2ray_hip: evolve3D seconds [iterations|total set-up upload iterations download library-rest host-rest]

usage: python profiles/dropin_timing.py --cases 128x1,128x10,256x100@13 --legs hip-exact,hip-fast[,serial,omp] [--out DIR]
  a case is <mesh>x<sources>[@<first slice>]: the test problem has 14 redshift slices (nbody_test.F90:222), run with ONE
  time step per slice (answers "n n <first> 7 1 1"); @13 runs slices 13 and 14 only (two steps).
Prints one JSON object (all cases and legs) on the last line.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC_ONE = [(50, 50, 50, 1e57)]                                  # inputs/test_sources_onesrc.dat
SRC_STD = [(50, 50, 50, 1e55), (51, 50, 50, 1e55), (52, 50, 50, 1e55), (53, 50, 50, 1e55),
           (20, 10, 10, 1e57), (70, 70, 50, 1e55), (72, 70, 50, 1e55), (70, 72, 50, 1e55),
           (72, 72, 50, 1e56), (20, 10, 90, 1e54)]              # inputs/test_sources_standard.dat


def sources_for(mesh, nsrc):
    if nsrc == 1:
        return SRC_ONE
    if nsrc == 10:
        return SRC_STD
    import __graft_entry__ as g
    pkg = g.load_package()
    pos, nf = pkg.seeded_sources(mesh, nsrc)
    return [(int(p[0]), int(p[1]), int(p[2]), float(f) * pkg.testproblem.S_STAR) for p, f in zip(pos, nf)]


def read_sm3d(path, dtype):
    """xfrac3D / IonRates3D files (read_sm3d.f90:63-103): int32 12 | n1 n2 n3 | 12 | nbytes | data | nbytes."""
    import numpy as np
    raw = open(path, "rb").read()
    nb = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=20)[0])
    return np.frombuffer(raw, dtype=dtype, count=nb // np.dtype(dtype).itemsize, offset=24)


def run_leg(mesh, srcs, first_slice, leg, timeout, keep_outputs=None):
    sub, prog = {"serial": ("serial", "c2ray_test"), "omp": ("omp", "c2ray_test"),
                 "hip-hoststats": ("hip_hoststats", "c2ray_test_hip")}.get(leg, ("hip", "c2ray_test_hip"))
    exe = os.path.join(ROOT, "oracle", "_ref", "N%d" % mesh, sub, prog)
    if not os.path.exists(exe):
        return {"leg": leg, "skipped": "not built: " + os.path.relpath(exe, ROOT)}
    d = tempfile.mkdtemp(prefix="c2r_dropin_t_")
    try:
        os.makedirs(d + "/results")
        open(d + "/answers", "w").write("n\nn\n%d\n7\n1\n1\n" % first_slice)
        with open(d + "/test_sources.dat", "w") as f:
            f.write("%d\n" % len(srcs))
            for (i, j, k, flux) in srcs:
                f.write("%d %d %d %.17e 0.0\n" % (i, j, k, flux))
        env = dict(os.environ, OMP_NUM_THREADS=str(min(8, os.cpu_count() or 1) if leg == "omp" else 1))
        if leg.startswith("hip"):
            env["C2R_SWEEP_MODE"] = "1" if leg == "hip-fast" else "0"
        t0 = time.perf_counter()
        try:
            rc = subprocess.run([exe, "answers"], cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                timeout=timeout).returncode
        except subprocess.TimeoutExpired:
            rc = "timeout after %d s" % timeout
        wall = time.perf_counter() - t0
        out = {"leg": leg, "exit": rc, "process_wall_s": wall, "threads": int(env["OMP_NUM_THREADS"])}
        log = open(d + "/results/C2Ray.log").read() if os.path.exists(d + "/results/C2Ray.log") else ""
        out["outer_iterations"] = len(re.findall(r"Number of non-converged points:", log))
        tl = open(d + "/results/Timings.log").read() if os.path.exists(d + "/results/Timings.log") else ""
        starts = [float(x) for x in re.findall(r"Time before starting iteration:\s*([\d.]+)", tl)]
        # the last "Time after iteration" line before the next start closes a step
        steps = []
        for m in re.finditer(r"Time before starting iteration:\s*([\d.]+)((?:\s*Time after iteration\s*\d+\s*:\s*[\d.]+)+)", tl):
            ends = [float(x) for x in re.findall(r":\s*([\d.]+)", m.group(2))]
            steps.append(ends[-1] - float(m.group(1)))
        out["steps"] = len(starts)
        out["evolve3d_s_per_step_timings_log"] = steps
        out["evolve3d_s_timings_log"] = sum(steps)
        if leg.startswith("hip"):
            rows = []
            for m in re.finditer(r"c2ray_hip: evolve3D seconds \[[^\]]*\]:\s*(\d+)((?:\s+[-+.\dEe]+){7})", log):
                v = [float(x) for x in m.group(2).split()]
                rows.append({"iterations": int(m.group(1)), "total": v[0], "setup": v[1], "upload": v[2], "iterate": v[3],
                             "download": v[4], "library_rest": v[5], "host_rest": v[6]})
            out["split_per_step"] = rows
            if rows:
                tot = {k: sum(r[k] for r in rows) for k in rows[0]}
                out["split_total"] = tot
                out["evolve3d_s"] = tot["total"]
                # what c2r_evolve3d_dev would contain: the library call without its two groups of copies
                out["fraction_inside_evolve3d_dev"] = (tot["iterate"] + tot["library_rest"]) / tot["total"] if tot["total"] > 0 else None
                # first step apart: it carries one-off set-up (context, page-locking, code object)
                if len(rows) > 1:
                    rest = {k: sum(r[k] for r in rows[1:]) for k in rows[0]}
                    out["split_total_without_first_step"] = rest
                    out["fraction_inside_evolve3d_dev_without_first_step"] = (rest["iterate"] + rest["library_rest"]) / rest["total"]
            m = re.search(r"c2ray_hip: sweep mode\s+(\S+)", log)
            out["sweep_mode"] = m.group(1) if m else None
        else:
            out["evolve3d_s"] = sum(steps)
        if keep_outputs is not None:        # the ionized-fraction outputs of the run, for --compare
            import numpy as np
            for f in sorted(os.listdir(d + "/results")):
                if f.startswith("xfrac3D_"):
                    keep_outputs[f] = np.array(read_sm3d(d + "/results/" + f, np.float64))
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="128x1,128x10")
    ap.add_argument("--legs", default="hip-exact,hip-fast")
    ap.add_argument("--timeout", type=int, default=1800)
    ap.add_argument("--out", default=None, help="also write the JSON there")
    ap.add_argument("--compare", action="store_true",
                    help="keep every leg's xfrac3D outputs and report, per leg, the largest |difference| from the FIRST leg's over all of them")
    a = ap.parse_args()
    res = {"host_cores": os.cpu_count(), "cases": []}
    for case in a.cases.split(","):
        m = re.fullmatch(r"(\d+)x(\d+)(?:@(\d+))?", case)
        mesh, nsrc, first = int(m.group(1)), int(m.group(2)), int(m.group(3) or 1)
        srcs = sources_for(mesh, nsrc)
        row = {"case": case, "mesh": mesh, "sources": nsrc, "first_slice": first, "steps_expected": 15 - first, "legs": []}
        ref_out = None
        for leg in a.legs.split(","):
            outs = {} if a.compare else None
            r = run_leg(mesh, srcs, first, leg, a.timeout, outs)
            if a.compare:
                if ref_out is None:
                    ref_out = outs
                    r["outputs"] = sorted(outs)
                else:
                    import numpy as np
                    r["same_output_files_as_first_leg"] = sorted(outs) == sorted(ref_out)
                    r["xfrac_max_abs_diff_vs_first_leg"] = max([float(np.max(np.abs(outs[k] - ref_out[k]))) for k in outs if k in ref_out] or [None])
            row["legs"].append(r)
            sys.stderr.write("%s %s: %s\n" % (case, leg, json.dumps({k: v for k, v in r.items() if k not in ("split_per_step", "evolve3d_s_per_step_timings_log")})))
        res["cases"].append(row)
    txt = json.dumps(res)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
