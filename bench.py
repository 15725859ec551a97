#!/usr/bin/env python3
"""Benchmark of the C2-Ray evolve hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2]/[3], SURVEY.md s8d state B): 256^3 mesh of the reference's test
problem (uniform mean IGM at z=9, pre-ionised to x=0.999 so every source traces out to its
photon-loss limit), S=1000 seeded sources (positions uniform, rates log-uniform 1e54..1e57 /s).
A "step" is ONE outer iteration of evolve3D (evolve.F90:170-272): zero the rates, sweep every
source (sharded over the GPUs, static 1+rank,NumSrc,npr), all-reduce Gamma over RCCL, global
chemistry pass.  The total source count is fixed, so --gpus N is STRONG scaling.

metric = cells-traced/s = N^3 x S x K / wall  (BASELINE.json).  Inputs are resident in HBM before the
timed region.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
SWEEP_BYTES_PER_VISIT = 28       # ndens 4 + xh_av 8 + phih_grid 16 (RMW)
CHEM_BYTES_PER_CELL = 44


def _run_reference(exe, mesh, srcpos, normflux, xfield, threads, box_cost):
    """do_source over the given sources with the compiled reference (oracle/_ref ref_driver, mode 'sweep') on the
    state `xfield`; returns seconds, per-source sub-box counts and the (cell, source) pairs actually visited."""
    from c2ray3dm_amd.testproblem import write_source_file
    d = tempfile.mkdtemp(prefix="c2r_cpu_")
    try:
        os.makedirs(d + "/results"); os.makedirs(d + "/dump")
        open(d + "/answers", "w").write("n\nn\n1\n7\n10\n1\n")
        write_source_file(d + "/test_sources.dat", srcpos, normflux)
        xfield.tofile(d + "/x.f64")                       # flat Fortran order = the stream the driver reads
        open(d + "/driver.nml", "w").write("&ctl mode='sweep', x_file='x.f64', ns_dump=0, nrep=1 /\n")
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        subprocess.check_call([exe, "answers"], cwd=d, env=env, stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=900)
        kv = dict(l.split() for l in open(d + "/dump/step001_sweep.txt"))
        nbox = np.loadtxt(d + "/dump/step001_nbox.txt", dtype=np.int64, ndmin=1)[:len(normflux)]
        return float(kv["seconds_per_pass"]), nbox, int(np.sum(box_cost(nbox, (mesh,) * 3)))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(mesh, srcpos, normflux, xfield, gpu_nbox, box_cost, nd=None):
    """The reference's CPU path timed on this box's host cores, on a bounded sample of the SAME work the GPU
    steps do: the first few sources of the bench's list, on the same mesh, on the relaxed xh_av field the timed GPU
    steps start from (handed to the reference through its driver's x_file).  Two legs of the compiled Fortran
    (oracle/_ref): the SERIAL build (per-thread number; deterministic, its sub-box counts must equal the GPU's) and
    the OpenMP build with min(8, cores) threads (the reference's scheme is at most 8-way: octants; its
    photon-loss accumulation races, SURVEY.md s5, so its sub-box counts may come out smaller).  `value` is the
    nominal metric N^3 x sources / time of the OpenMP leg; the visited pairs are reported beside it.  Falls back to
    the serial C oracle ("port") where the reference binaries are absent."""
    ncores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "N%d" % mesh)
    per_visit = 2.0e-7                                    # ~0.2 us per visited pair and thread
    full = float(mesh) ** 3
    uniform = nd is None or bool(np.all(nd == nd.flat[0]))       # the reference driver builds the test problem's uniform density itself
    if uniform and os.path.exists(os.path.join(ref, "omp", "ref_driver")) and os.path.exists(os.path.join(ref, "serial", "ref_driver")):
        try:
            threads = min(8, ncores)
            n1 = int(max(1, min(len(normflux), round(8.0 / (full * per_visit)))))               # ~8 s serial
            s_sec, s_nbox, s_vis = _run_reference(os.path.join(ref, "serial", "ref_driver"), mesh, srcpos[:n1], normflux[:n1], xfield, 1, box_cost)
            # the OpenMP leg is sized from the serial leg's measured rate for ~12 s (its speed-up on 8 threads is small:
            # the octant scheme is memory-bound and synchronises per plane)
            nt = int(max(1, min(len(normflux), round(12.0 * 1.2 * (s_vis / s_sec) / full))))
            o_sec, o_nbox, o_vis = _run_reference(os.path.join(ref, "omp", "ref_driver"), mesh, srcpos[:nt], normflux[:nt], xfield, threads, box_cost)
            return {"value": full * nt / o_sec, "unit": "cells-traced/s", "cores": threads, "host_cores": ncores, "kind": "reference",
                    "openmp_speedup_over_serial": (o_vis / o_sec) / (s_vis / s_sec),
                    "seconds": o_sec, "sources": nt, "visited": o_vis, "visited_per_s": o_vis / o_sec,
                    "sum_nbox": int(o_nbox.sum()), "gpu_sum_nbox_same_sources": int(np.sum(gpu_nbox[:nt])),
                    "serial": {"value": full * n1 / s_sec, "cores": 1, "seconds": s_sec, "sources": n1, "visited": s_vis,
                               "visited_per_s": s_vis / s_sec, "sum_nbox": int(s_nbox.sum()),
                               "sub_boxes_equal_gpu": bool(np.array_equal(s_nbox, gpu_nbox[:n1]))},
                    "sample": "compiled Fortran reference (oracle/_ref, amdflang -O2), do_source over the first %d (OpenMP build, "
                              "%d threads) / %d (serial build) of the bench's sources on the same %d^3 mesh and on the relaxed "
                              "xh_av field the timed GPU steps start from; `value` = nominal N^3 x sources / s of the OpenMP leg, "
                              "`visited` = (cell, source) pairs it actually traced; `cores` = OpenMP threads used (the reference's scheme "
                              "is at most 8-way), `host_cores` = what this host has" % (nt, threads, n1, mesh)}
        except Exception as exc:          # fall through to the port
            sys.stderr.write("cpu_baseline: reference run failed (%r), using the C port\n" % (exc,))
    from oracle.oracle import Oracle
    from tests._util import load_tables
    from c2ray3dm_amd.testproblem import TestProblem
    tp = TestProblem(mesh); s = tp.step(1)
    if nd is None:
        nd, _ = tp.fields(1)
    o = Oracle(mesh, s["dr1"], s["vol"], s["coldensh_LLS"], *load_tables())
    nsamp = int(max(1, min(len(normflux), round(20.0 / (full * per_visit)))))
    phih = np.zeros(o.ncell)
    t0 = time.perf_counter()
    loss, nb, vis = o.pass_sources(nd, xfield, phih, srcpos[:nsamp], normflux[:nsamp])
    sec = time.perf_counter() - t0
    return {"value": full * nsamp / sec, "unit": "cells-traced/s", "cores": 1, "host_cores": ncores, "kind": "port",
            "seconds": sec, "sources": nsamp, "visited": int(vis), "visited_per_s": vis / sec, "sum_nbox": int(nb),
            "sample": "serial C oracle, pass over the first %d of the bench's sources on the same %d^3 mesh and on the "
                      "relaxed xh_av field the timed GPU steps start from" % (nsamp, mesh)}


def mix_ceiling_this_box(device, mask=31):
    """Visits per second the memory system of THIS box sustains for the sweep kernel's four streams with no arithmetic at
    all (profiles/micro/trafficmix.hip as a library: one n_HI load from a pseudo-random run of a 134 MB grid, the previous
    shell's plane rows, one plane store, one f64 atomic per visit; runs start anywhere, as the kernel's do; mask 63: the same
    with every n_HI load an L2 hit -- the plane-ordered block mapping of the far shells serves three quarters of them
    from the L2s, the kernel's real mix lies between the two).  Taken in the bench process, after the timed region; None where the
    library is not built."""
    import ctypes
    path = os.path.join(ROOT, "profiles", "micro", "libtrafficmix.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.c2r_micro_traffic_mix.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    v = ctypes.c_double(0.0)
    return v.value if lib.c2r_micro_traffic_mix(int(device), int(mask), ctypes.byref(v)) == 0 else None


def parity_check(pkg, mesh, step, nd, xh_init, xfield, srcpos, normflux, tables, device, fast, nsub=64):
    """The second half of BASELINE.json's metric -- "xh L-inf error vs Fortran ref" -- for THIS workload, outside the timed
    region: one pass over `nsub` sources drawn evenly ACROSS the bench's list plus one global pass (evolve0D_global over the mesh,
    evolve_point.F90:305-406) on the relaxed field the timed steps start from, on the GPU and in the oracle (the pinned C
    restatement of the reference, the checker).  Reports max |dxh| over the mesh, the worst rate error in units of the
    tolerance weight W (tests/_util.py: |dGamma| <= rtol Gamma + wtol W) and whether the integer results agree."""
    from oracle.oracle import Oracle
    from tests._util import oracle_pass, gamma_plain_rel, GAMMA_PLAIN_RTOL, GAMMA_PLAIN_FLOOR     # (the oracle's pass with its source chunks in threads)
    thick, thin = tables
    o = Oracle(mesh, step["dr1"], step["vol"], step["coldensh_LLS"], thick, thin)
    sel = np.unique(np.linspace(0, len(normflux) - 1, min(nsub, len(normflux))).astype(np.int64))
    pos, nf = srcpos[sel], normflux[sel]
    t0 = time.perf_counter()
    oloss, onb, ovis, phih_o, w = oracle_pass(o, nd, xfield, pos, nf)
    xav, xint = xfield.copy(), xh_init.copy()
    oconv = o.global_pass(step["dt"], nd, xh_init, xav, xint, phih_o)
    sec = time.perf_counter() - t0
    b = pkg.HipBackend(mesh, thick, thin, device=device, fast=fast)
    b.set_step(step["dr1"], step["vol"], step["coldensh_LLS"], step["clumping"], step["temper"])
    b.set_sources(pos, nf)
    b.load(ndens=nd, xh=xh_init)
    b.begin_step()
    b.load(xh_av=xfield)
    b.zero_rates()
    loss, nbox, vis = b.pass_sources()
    phih = b.fetch("phih_grid")
    conv, _ = b.global_pass(step["dt"])
    dx = float(np.max(np.abs(b.fetch("xh_intermed") - xint)))
    dxav = float(np.max(np.abs(b.fetch("xh_av") - xav)))
    b.close()
    live = w > 0
    plain = gamma_plain_rel(phih - phih_o, phih_o, w)
    return {"xh_linf": max(dx, dxav), "gamma_max_over_W": float(np.max(np.abs(phih - phih_o)[live] / w[live])),
            "gamma_max_rel": float(np.max(np.abs(phih - phih_o)[live] / np.maximum(phih_o[live], 1e-300))),
            # the plain bound the tests assert (tests/_util.py): relative error wherever a cell's own rate is >= 1e-6 of the rate passing through it
            "gamma_max_rel_significant_cells": plain, "gamma_plain_bound": GAMMA_PLAIN_RTOL, "gamma_plain_floor_over_W": GAMMA_PLAIN_FLOOR,
            "gamma_plain_ok": bool(plain <= GAMMA_PLAIN_RTOL),
            "source_indices": [int(sel[0]), int(sel[1]) if len(sel) > 1 else None, "...", int(sel[-1])],
            "photon_loss_rel": abs(loss - oloss) / abs(oloss) if oloss else 0.0,
            "integers_equal": bool((nbox, vis, conv) == (onb, ovis, oconv)), "sum_nbox": int(nbox), "nonconverged_cells": int(conv),
            "sources": int(len(nf)), "checker": "oracle", "oracle_seconds": sec,
            "what": "one pass over %d sources drawn evenly across the list + one global pass on the relaxed xh_av field of the timed steps, GPU vs the "
                    "pinned C restatement of the reference (oracle/c2ray_oracle.c); xh_linf = max |xh_intermed, xh_av difference| "
                    "(north_star: 1e-5), gamma_max_over_W in units of the tolerance weight (stated bound 2e-14)" % len(nf)}


def timed_leg(pkg, torch, n, S, nd, xh, srcpos, normflux, step, tables, device, fast, options, steps, warm=1, prof_mode=1,
              deterministic=False):
    """A short run of the same kind as the headline -- its own context; one relaxing iteration, `warm` warm-up steps, `steps`
    timed ones (set_rates_to_zero + pass over all sources + global pass each) -- with the sweep launches timed by HIP events on the
    context's stream (c2r_profile, as the headline's roofline is): ms per step, the nominal metric and the roofline of the shell
    kernel (28 algorithmic bytes per visited pair outside the fused first sub-boxes / launch time)."""
    thick, thin = tables
    b = pkg.HipBackend(n, thick, thin, device=device, deterministic=deterministic, fast=fast, options=options)
    b.set_step(step["dr1"], step["vol"], step["coldensh_LLS"], step["clumping"], step["temper"])
    b.set_sources(srcpos, normflux)
    b.load(ndens=nd, xh=xh)
    ev = pkg.Evolve(b)
    b.begin_step()
    for k in range(-1, warm):
        ev.iteration(k, step["dt"])
    torch.cuda.synchronize()
    b.profile(prof_mode)
    ev.visited = 0
    fused = 0.0
    t0 = time.perf_counter()
    for k in range(steps):
        ev.iteration(k, step["dt"])
        if prof_mode != 0 and options.get("fuse_small", 1.0) != 0.0:
            fused += float(np.sum(pkg.box_cost(np.minimum(b.last_nbox(), 2), (n, n, n))))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = b.profile_read()
    out = {"mesh": n, "sources": int(len(normflux)), "sweep_mode": "fast" if fast else "exact", "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "value": float(n) ** 3 * len(normflux) * steps / dt, "unit": "cells-traced/s", "sum_nbox_last_step": int(ev.sum_nbox_all),
           "visited_per_step": float(ev.visited) / steps, "schedule": b.info().split("; chains ")[1].split("; exchanges")[0]}
    if prof_mode != 0 and prof["sweep_ms"] > 0:
        vis = max(0.0, float(ev.visited) - fused)
        ach = SWEEP_BYTES_PER_VISIT * vis / (prof["sweep_ms"] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_sweep_shell_fast" if fast else "k_sweep_shell", "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": prof["sweep_ms"] / max(1, prof["sweep_launches"]),
                           "launches": prof["sweep_launches"],
                           "timing": {1: "HIP events around every shell launch", 2: "HIP events around every sub-box (chained passes: around the round)"}[prof_mode]}
    b.close()
    return out


def share_leg(pkg, torch, n, share, nd, xh, srcpos, normflux, step, tables, device, fast, options, eighth_ms):
    """`share` (indices into the bench's source list: what ONE of eight GPUs sweeps) on its own context: five relaxing / warm-up
    steps (the first pass is driven launch by launch, the second captures the chains' launch sequences), then ten steps timed one
    by one -- the median, next to an eighth of the headline step; and the same once more with the chains driven launch by launch
    (option chain_graph = 0: a host round trip per sub-box and chain, the schedule of round 5) -- what the replay buys on THIS box."""
    thick, thin = tables

    def run(opts):
        b = pkg.HipBackend(n, thick, thin, device=device, fast=fast, options=opts)
        b.set_step(step["dr1"], step["vol"], step["coldensh_LLS"], step["clumping"], step["temper"])
        b.set_sources(srcpos[share], normflux[share])
        b.load(ndens=nd, xh=xh)
        ev = pkg.Evolve(b)
        b.begin_step()
        t_steps = []
        for k in range(-5, 10):
            torch.cuda.synchronize(); t4 = time.perf_counter()
            ev.iteration(k, step["dt"])
            torch.cuda.synchronize()
            if k >= 0:
                t_steps.append(1e3 * (time.perf_counter() - t4))
        r = (float(np.median(t_steps)), [min(t_steps), max(t_steps)], b.info().split("; chains ")[1].split("; exchanges")[0], int(ev.sum_nbox_all))
        b.close()
        return r
    ms, mm, sched, nbox = run(options)
    out = {"sources": int(len(share)), "ms_per_step": ms, "ms_per_step_min_max": mm, "eighth_of_headline_ms": eighth_ms,
           "ratio": ms / eighth_ms, "schedule": sched, "sum_nbox_last_step": nbox}
    if "chain_graph" not in options:
        ms0, mm0, sched0, nbox0 = run(dict(options, chain_graph=0.0))
        out["launch_by_launch"] = {"ms_per_step": ms0, "ms_per_step_min_max": mm0, "ratio": ms0 / eighth_ms, "schedule": sched0,
                                   "sum_nbox_last_step": nbox0}
    return out


def self_launch(ngpus):
    """Start `ngpus` ranks of this script under torch.distributed.run as a child process and relay its output."""
    import socket
    with socket.socket() as sk:                 # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    out = proc.stdout.splitlines()
    lines = [l for l in out if l.startswith("{")]
    for l in out:                               # anything else the ranks printed
        if not l.startswith("{"):
            print(l)
    if proc.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank child run failed (exit code %d)\n" % (ngpus, proc.returncode))
        return proc.returncode or 1
    print(lines[-1])                            # ONE JSON line, rank 0's
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mesh", type=int, default=256)
    ap.add_argument("--sources", type=int, default=1000)
    ap.add_argument("--x-init", type=float, default=0.999)
    ap.add_argument("--density", choices=["uniform", "lognormal"], default="uniform",
                    help="lognormal: sigma_ln=1, mean 1 times the same mean density (SURVEY.md s8d, config 5)")
    ap.add_argument("--sweep-mode", choices=["exact", "fast"], default=os.environ.get("C2R_BENCH_SWEEP_MODE", "fast"),
                    help="c2r_params.sweep_mode: exact = the reference's f64 operation order (column densities bit-identical "
                         "to the Fortran), fast = re-associated arithmetic within the stated tolerance (include/c2ray_hip.h)")
    ap.add_argument("--density-file", default=None,
                    help="coarsened cubep3m density file (<z>n_all.dat: 3 x int32 + N^3 float32 stream, nbody_cubep3m.F90:87-107) "
                         "instead of the synthetic field; scaled as scale_density does (density_unit grid, --n-box fine cells per side)")
    ap.add_argument("--n-box", type=int, default=13824, help="fine N-body cells per side of the density file's simulation (nbody_cubep3m.F90:9)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin-leg", action="store_true",
                    help="skip the informational run of the reference driver linked with the Fortran drop-in (128^3, one source, 14 steps)")
    ap.add_argument("--no-mix-ceiling", action="store_true",
                    help="skip the in-process run of the memory-only traffic mix (profiles/micro/libtrafficmix.so, 6.7 GB of HBM, ~1 s)")
    ap.add_argument("--no-other-mode", action="store_true",
                    help="skip the short leg that times the same steps in the OTHER sweep mode (reported as `other_sweep_mode`)")
    ap.add_argument("--no-small-leg", action="store_true",
                    help="skip the short leg that times BASELINE configs[1] (128^3, one source: the launch-bound regime; reported "
                         "as `configs1_128_1src`; only the default workload runs it)")
    ap.add_argument("--no-configs-leg", action="store_true",
                    help="skip the short legs that time BASELINE configs[2] (256^3 x 100 sources) and one GPU's share of configs[4] "
                         "(504^3, 1250 of 10 000 sources, log-normal density; needs ~45 GB of HBM and ~10 s)")
    ap.add_argument("--balance", action="store_true",
                    help="cost-balanced source shares (by the previous pass) instead of the static stride")
    ap.add_argument("--overlap-exchange", action="store_true",
                    help="several ranks: the all-reduce of the first half of a rank's sources travels while the second half is swept "
                         "(c2r_set_exchange_overlap; off by default until an N > 1 RCCL run has measured it)")
    ap.add_argument("--deterministic", action="store_true",
                    help="per-source Gamma grids reduced in source order instead of f64 atomics")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="c2r_set_option on every context of the run (include/c2ray_hip.h has the table; A/B measurements), e.g. --option chains=3")
    ap.add_argument("--thermal", action="store_true",
                    help="time the non-isothermal variant (isothermal=.false.: heating rates in the sweep, thermal.f90 in the "
                         "global pass; synthetic cooling table, T = 1e4 K start).  Not the headline configuration; no CPU baseline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  Nothing here has touched
        # the GPU (no torch import, no HIP call), the ranks are CHILD processes (torch.distributed.run, one per GPU,
        # rendezvous on 127.0.0.1), rank 0's JSON line is relayed, and a failing rank fails the run.
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    pkg = g.load_package()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                     # started by a launcher with another rank count: the launcher wins
        args.gpus = world
    one_gpu_test = os.environ.get("C2R_BENCH_TEST_ONE_GPU") == "1"    # CI only: every rank on cuda:0, gloo
    if world > 1 and not one_gpu_test:
        # preflight, before any rendezvous (so that a bad launch FAILS, on every rank by itself, instead of hanging in a collective):
        # one process per GPU needs as many visible devices as there are ranks on this node (device_count() does not initialise the GPU)
        ndev = torch.cuda.device_count()
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if ndev < local_world or local_rank >= ndev:
            sys.stderr.write("bench.py: rank %d (local rank %d): %d rank(s) on this node but %d GPU(s) visible -- one process per GPU "
                             "needs one device each (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)\n" % (rank, local_rank, local_world, ndev))
            sys.exit(4)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if os.environ.get("C2R_BENCH_TEST_FAIL_RANK") == str(rank):       # CI only: a rank that dies must fail the whole run
        sys.stderr.write("bench.py: rank %d fails on purpose (C2R_BENCH_TEST_FAIL_RANK)\n" % rank)
        os._exit(3)
    n, S = args.mesh, args.sources
    options = {kv.split("=", 1)[0]: float(kv.split("=", 1)[1]) for kv in args.option}
    tp = pkg.TestProblem(n)
    s = tp.step(1)
    nd, xh = tp.fields(1, args.x_init)
    if args.density == "lognormal":
        rng = np.random.default_rng(20261003)
        nd = (nd * np.exp(rng.standard_normal(nd.size, dtype=np.float32) - 0.5)).astype(np.float32)
    if args.density_file:
        raw = pkg.fileio.read_density(args.density_file, mesh=n)               # density_module.F90:203-243
        nd = pkg.fileio.scale_density(raw, s["zred"], n, args.n_box).ravel(order="F")
        # cosmo_evol (cosmology.F90:186): the slice's comoving-at-z density is rescaled to the mid-step redshift -- here
        # the file is taken to hold the slice of the step's own redshift
    srcpos, normflux = pkg.seeded_sources(n, S)
    thick, thin, _ = pkg.build_tables()          # rad_ini on the host (c2r_build_tables)
    b = pkg.HipBackend(n, thick, thin, device=local_rank, deterministic=args.deterministic, fast=args.sweep_mode == "fast", options=options)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], s["clumping"], s["temper"])
    b.set_sources(srcpos, normflux)
    b.load(ndens=nd, xh=xh)
    if args.thermal:
        hk, hn = pkg._capi.build_heat_tables()
        _, lt, ll = pkg.testproblem.synthetic_cooling_table()
        b.set_thermal(hk, hn, lt, ll)
        b.set_redshift(s["zred"])
        b.load(temperature_grid=np.full(n ** 3, 1e4, dtype=np.float32))
        args.no_cpu_baseline = True
    bytes_per_visit = SWEEP_BYTES_PER_VISIT + (16 if args.thermal else 0)      # + phiheat_grid read-modify-write
    ev = pkg.Evolve(b, comm=dist if world > 1 else None, balance=args.balance)
    rank_devices = None
    if world > 1:
        # which physical device every rank's context resolved to: two ranks on one GPU would "scale" by time-slicing it -- fail loudly
        import socket
        pr = torch.cuda.get_device_properties(local_rank)
        ident = str(getattr(pr, "uuid", "")) or ":".join(str(getattr(pr, k, "?")) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        mine = {"rank": rank, "host": socket.gethostname(), "local_rank": local_rank, "c2r_device": int(b.get_device()), "device_id": ident,
                "info": b.info().split(";")[0]}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
        seen = {}
        for d in rank_devices:
            seen.setdefault((d["host"], d["device_id"], d["c2r_device"]), []).append(d["rank"])
        shared = [r for r in seen.values() if len(r) > 1]
        if shared and not one_gpu_test:
            sys.stderr.write("bench.py: ranks %s resolved to the SAME device: one process per GPU is the contract (rank %d: %s)\n" % (shared, rank, mine["info"]))
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(5)
        if dist.get_world_size() != world:
            sys.stderr.write("bench.py: the communicator has %d ranks, the launcher announced %d\n" % (dist.get_world_size(), world))
            sys.exit(6)
    if args.overlap_exchange and world > 1:
        b.set_exchange_overlap(True)
    b.begin_step()

    def one_step(k):
        # one rank: c2r_iterate (the three steps in one call; few sources: one replayed graph and one host wait)
        return ev.iteration(k, s["dt"])

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # input preparation, not a warm-up: one outer iteration relaxes the freshly pre-ionised state (the very
    # first pass over a uniform x stops at a smaller sub-box than all later ones), so that warm-up and
    # timed steps -- and a profiler that sees both -- do the same work
    one_step(-1)
    t_w = time.perf_counter()
    for k in range(args.warmup):
        one_step(k)
    sync()
    warm_ms = 1e3 * (time.perf_counter() - t_w) / max(1, args.warmup)      # (0 warm-up steps: 0 ms -- treated as a long step)
    # roofline timing: per-launch events on one GPU (the measurement the contract asks for); with several
    # GPUs the launches are 1/N as long and the per-launch barrier packets would cost ~3 %, so one event pair
    # per sub-box is used there (C2R_BENCH_PROFILE overrides: 0, 1, 2)
    xh_state = b.fetch("xh_av") if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None   # what the timed steps start from
    # ... and with few sources a launch is a few microseconds, the event pairs around each would cost more than the
    # launches themselves (128^3 x 1 source: 0.73 ms per step with them, 0.46 without), and they keep the library from
    # replaying the pass as a hipGraph: one pair per sub-box there
    prof_mode = int(os.environ.get("C2R_BENCH_PROFILE", "1" if (world == 1 and S >= 64) else "2"))
    if prof_mode == 2 and S < 64 and "C2R_BENCH_PROFILE" not in os.environ:
        prof_mode = 0
    # ... and where a whole step takes a few milliseconds the launches are tens of microseconds: an event pair around each
    # costs a quarter of a cold 256^3 x 1000 step (1.20 ms with them, 1.03 per sub-box, 0.95 without): per sub-box there
    if prof_mode == 1 and args.warmup > 0 and warm_ms < 5.0 and "C2R_BENCH_PROFILE" not in os.environ:
        prof_mode = 2
    b.profile(prof_mode)
    visited_before = ev.visited
    ev.visited = 0
    ev.phase_seconds = {"sweep": 0.0, "exchange": 0.0, "chem": 0.0}
    xchg0 = b.exchange_stats()
    nbox_hist = []
    t0 = time.perf_counter()
    fused_visited = 0.0        # pairs traced by k_sweep_box_fused (sub-boxes 1 and 2), this rank, timed steps
    nbox_first = None
    for k in range(args.steps):
        one_step(k)
        nbox_hist.append(ev.sum_nbox_all)
        if k == 0 and world == 1:
            nbox_first = b.last_nbox().astype(np.int64)        # per-source sub-box counts of the first timed pass
        # (only where launches are timed: with few sources a step is a few hundred microseconds and this bookkeeping would be
        # a tenth of it)
        if prof_mode != 0 and options.get("fuse_small", 1.0) != 0.0:
            fused_visited += float(np.sum(pkg.box_cost(np.minimum(b.last_nbox(), 2), (n, n, n))))
    sync()
    dt_wall = time.perf_counter() - t0
    prof = b.profile_read()
    xchg1 = b.exchange_stats()
    # checksums of the state the timed steps leave (the same on every rank: Gamma is all-reduced, the global pass replicated)
    check = {"phih_grid_sum": float(b.phih_grid.sum(dtype=torch.float64)), "xh_intermed_sum": float(b.xh_intermed.sum(dtype=torch.float64)),
             "xh_av_sum": float(b.xh_av.sum(dtype=torch.float64)), "sum_nbox_last_step": int(nbox_hist[-1]) if nbox_hist else 0}
    # max over ranks of the wall time; sum over ranks of the visited pairs
    stats = torch.tensor([dt_wall, float(ev.visited), prof["sweep_ms"], float(prof["sweep_launches"])],
                         dtype=torch.float64, device=b.device)
    shares = None
    phases = None
    if world > 1:
        # where each rank's wall time went (Evolve.iteration), min and max over the ranks: a rank that waits for a slower one
        # shows a long exchange; and what the exchange would cost over xGMI as one ring (2 (N-1)/N x bytes at 153 GB/s per link)
        ph = torch.tensor([ev.phase_seconds["sweep"], ev.phase_seconds["exchange"], ev.phase_seconds["chem"]], dtype=torch.float64, device=b.device)
        pmin = ph.clone(); dist.all_reduce(pmin, op=dist.ReduceOp.MIN)
        pmax = ph.clone(); dist.all_reduce(pmax, op=dist.ReduceOp.MAX)
        phases = {k: {"min_s_per_step": float(pmin[i]) / max(1, args.steps), "max_s_per_step": float(pmax[i]) / max(1, args.steps)}
                  for i, k in enumerate(("sweep", "exchange", "chem"))}
        # the source shares of the last pass (static stride or the library's LPT partition): they must partition the list
        mine = [int(i) for i in b.local_sources()]
        shares = [None] * world
        dist.all_gather_object(shares, mine)
        tmax = stats.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_wall = float(tmax[0]); visited_all = float(tsum[1])
    else:
        visited_all = float(stats[1])

    if rank == 0:
        value = float(n) ** 3 * S * args.steps / dt_wall
        vis_rank = float(ev.visited)
        launches = max(1, prof["sweep_launches"])
        sweep_s = prof["sweep_ms"] * 1e-3
        # the timed launches are those of k_sweep_shell; the first two sub-boxes of every source (21^3 cells)
        # run in k_sweep_box_fused and are left out of both the bytes and the time
        vis_rank = max(0.0, vis_rank - fused_visited)
        # prof_mode 0 (few sources: launches of a few microseconds, replayed as a hipGraph): no kernel was timed, so the
        # roofline object carries nulls instead of numbers derived from the whole step's wall time
        timed = prof_mode != 0 and sweep_s > 0
        achieved = bytes_per_visit * vis_rank / sweep_s / 1e9 if timed else None
        traffic, traffic_note, mix_ceiling, mix_ceiling_l2 = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "TRAFFIC.json")
        if os.path.exists(tpath) and timed:      # PMC counters of the same command, from the latest committed profile
            tj = json.load(open(tpath))
            traffic = (tj["fetch_corrected_bytes_per_visit"] + tj["write_bytes_per_visit"]) * vis_rank / launches
            traffic_note = "%s (profiled commit %s)" % (tj["source"], tj.get("commit", "?"))
        # the memory-only ceiling of the kernel's traffic mix, measured on THIS box, now (outside the timed region)
        if timed and world == 1 and not args.thermal and not args.no_mix_ceiling:
            torch.cuda.synchronize()
            mix_ceiling = mix_ceiling_this_box(local_rank)
            mix_ceiling_l2 = mix_ceiling_this_box(local_rank, 63)
        out = {
            "metric": "cells-traced/sec (grid^3 x sources / wallclock) on 256^3",
            "value": value, "unit": "cells-traced/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt_wall / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d^3 mesh, %d sources (seeded), reference test problem at z=9 pre-ionised to "
                                   "x=%.3f, %s density, one evolve3D outer iteration per step (sweep all sources + "
                                   "all-reduce + global chemistry pass)" % (n, S, args.x_init, "cubep3m-file" if args.density_file else args.density),
                       "mesh": n, "sources": S, "sweep_mode": args.sweep_mode, "isothermal": not args.thermal, "gamma_accumulation": "ordered" if args.deterministic else "atomic", "sources_per_gpu": len(pkg.static_source_share(S, 0, world)),
                       "parallelism": "sources sharded over %d GPU(s), RCCL all-reduce of Gamma" % world,
                       "ranks": dist.get_world_size() if world > 1 else 1,
                       # first contact with a real multi-GPU node: the communicator's own rank count and the device every rank's context
                       # runs on (checked above: distinct devices, or the run has failed) -- one_gpu_test: every rank on cuda:0 on purpose
                       "communicator_ranks": dist.get_world_size() if world > 1 else 1, "rank_devices": rank_devices,
                       "ranks_on_distinct_devices": (len({(d["host"], d["device_id"], d["c2r_device"]) for d in rank_devices}) == world) if rank_devices else True,
                       # evolve.F90:599 through c2r_allreduce_rates: the whole grid, or the sources' packed sub-boxes while those are few
                       "gamma_exchange": {"calls": xchg1["calls"] - xchg0["calls"], "packed_calls": xchg1["sparse_calls"] - xchg0["sparse_calls"],
                                          "bytes_per_step": (xchg1["bytes_total"] - xchg0["bytes_total"]) / max(1, args.steps),
                                          "full_grid_bytes": 8 * n ** 3,
                                          # one ring over xGMI: 2 (N-1)/N x bytes per link at ~153 GB/s (MI355X_MICROARCH.md); RCCL may do better
                                          "ring_over_xgmi_s_per_step": 2.0 * (world - 1) / world * (xchg1["bytes_total"] - xchg0["bytes_total"]) / max(1, args.steps) / 153e9}
                       if world > 1 else None,
                       "source_share_sizes": [len(x) for x in shares] if shares else [S],
                       "shares_partition_sources": (sorted(i for x in shares for i in x) == list(range(S))) if shares else True,
                       "rank_phases": phases, "exchange_overlapped_with_sweep": bool(args.overlap_exchange and world > 1),
                       "collective": None if world == 1 else ("gloo (C2R_BENCH_TEST_ONE_GPU)" if one_gpu_test else "nccl (RCCL)"),
                       "visited_cell_sources_per_step": visited_all / args.steps,
                       "visited_per_s": visited_all / dt_wall,
                       "visited_cell_sources_whole_run_rank0": float(ev.visited + visited_before),
                       "mean_subboxes_per_source": [x / S for x in nbox_hist]},
            "check": check,
            "roofline": {"bound": "hbm", "kernel": "k_sweep_shell_fast" if args.sweep_mode == "fast" else "k_sweep_shell",
                         # the same shell kernel under two block mappings: the plain (tile, face, source) grid and, for the far shells of many
                         # sources, the XCD-aware plane-ordered one; a rocprofv3 --stats table lists them under two names, the launch
                         # average below is over both (profiles/summarize.py merges them the same way)
                         "kernel_names_in_a_trace": ["k_sweep_shell_fast", "k_sweep_shell_xcd<..., FAST = true>"] if args.sweep_mode == "fast" else ["k_sweep_shell", "k_sweep_shell_xcd<..., FAST = false>"],
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if timed else None,
                         "traffic": traffic, "traffic_source": traffic_note,
                         "traffic_measured_in_this_run": False,     # bytes per visit from the committed PMC profile (builder's box) x this run's visits per launch
                         # informational: the kernel's visits/s against what the memory system sustains for the same four
                         # streams with no arithmetic at all, measured on THIS box after the timed region (mix_ceiling_this_box, profiles/micro/trafficmix.hip)
                         "mix_ceiling_this_box": mix_ceiling,
                         "frac_of_memory_only_mix": (vis_rank / sweep_s / mix_ceiling) if (mix_ceiling and timed and not args.thermal) else None,
                         # ... and with every n_HI load an L2 hit: the far shells' plane-ordered block mapping serves ~3/4 of
                         # them from the L2s, so the ceiling of the kernel's real mix lies between the two
                         "mix_ceiling_nhi_in_l2_this_box": mix_ceiling_l2,
                         "mix_ceiling_note": "the first micro-benchmark (every n_HI load from HBM) is what the kernel's traffic looked like BEFORE the plane-ordered "
                                             "mapping: the kernel now reaches ~1.0 of it while also doing 170 VALU instructions per visit, so it is not a ceiling of "
                                             "today's kernel; the second (n_HI in the L2s) is an upper bound of today's mix -- the kernel's real mix lies between",
                         "frac_of_memory_only_mix_nhi_in_l2": (vis_rank / sweep_s / mix_ceiling_l2) if (mix_ceiling_l2 and timed and not args.thermal) else None,
                         "algorithmic_bytes_per_launch": bytes_per_visit * vis_rank / launches if timed else None, "algorithmic_bytes_per_visit": bytes_per_visit,
                         "avg_launch_ms": prof["sweep_ms"] / launches if timed else None, "launches": prof["sweep_launches"],
                         "timing": {0: "off (few sources: launches of a few microseconds inside a hipGraph): no kernel timing, roofline fields are null", 1: "HIP events around every k_sweep_shell launch",
                                    2: "HIP events around every sub-box (5 launches + the small kernels between them)"}[prof_mode],
                         "chem_kernel_ms_per_launch": prof["chem_ms"] / max(1, prof["chem_launches"]),
                         "chem_achieved_GBs": (CHEM_BYTES_PER_CELL * float(n) ** 3 * prof["chem_launches"] /
                                               (prof["chem_ms"] * 1e-3) / 1e9) if prof["chem_ms"] > 0 else 0.0,
                         "note": "achieved = 28 algorithmic B per visited (cell, source) / launch time; the traffic actually leaving the L2s is ~32 B per visit "
                                 "(shell planes make a round trip through HBM between launches; n_HI is shared in the L2s by the sources of a mesh "
                                 "plane since round 5: ~2 of its 8 B) and the Gamma atomics cost the memory side a read and a write each: ~41 B per "
                                 "visit at DRAM level, ~4.9 TB/s of the ~6.3 TB/s achievable; the memory-side f64 atomics run at 72 % of their chip-wide "
                                 "rate; see DESIGN.md s3e, s5"},
        }
        if world == 1 and not args.no_other_mode and not args.thermal:
            # the same steps in the OTHER sweep mode (`value` above is the mode named in config.sweep_mode: fast, the library and
            # drop-in default; exact is the opt-in with column densities bit-identical to the Fortran), outside the timed region of
            # the headline, its shell kernel timed the same way
            b.close()
            other = "exact" if args.sweep_mode == "fast" else "fast"
            out["other_sweep_mode"] = timed_leg(pkg, torch, n, S, nd, xh, srcpos, normflux, s, (thick, thin), local_rank, other == "fast", options,
                                                min(args.steps, 3), deterministic=args.deterministic, prof_mode=prof_mode)
        if world == 1 and not args.no_small_leg and (n, S) == (256, 1000) and not args.thermal and not args.density_file:
            # BASELINE configs[1] -- 128^3, one source, the reference's own CPU-runnable case -- is bound by the number of
            # dependent launches, not by bytes: informational, outside the timed region of the headline
            b.close()
            n3 = 128
            tp3 = pkg.TestProblem(n3); s3 = tp3.step(1)
            nd3, xh3 = tp3.fields(1, args.x_init)
            pos3, nf3 = pkg.seeded_sources(n3, 1)
            out["configs1_128_1src"] = {"workload": "128^3 mesh, 1 source, same test problem and pre-ionisation, one outer iteration per step"}
            for mode in (args.sweep_mode, "exact" if args.sweep_mode == "fast" else "fast"):
                b3 = pkg.HipBackend(n3, thick, thin, device=local_rank, fast=mode == "fast")
                b3.set_step(s3["dr1"], s3["vol"], s3["coldensh_LLS"], s3["clumping"], s3["temper"])
                b3.set_sources(pos3, nf3)
                b3.load(ndens=nd3, xh=xh3)
                ev3 = pkg.Evolve(b3)
                b3.begin_step()
                k3 = 200
                for k in range(-10, k3):
                    if k == 0:
                        torch.cuda.synchronize(); t3 = time.perf_counter()
                    ev3.iteration(k, s3["dt"])
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t3
                out["configs1_128_1src"][mode] = {"steps": k3, "ms_per_step": 1e3 * dt3 / k3, "value": float(n3) ** 3 * k3 / dt3,
                                                  "unit": "cells-traced/s", "sum_nbox_last_step": int(ev3.sum_nbox_all)}
                b3.close()
        if world == 1 and not args.no_small_leg and (n, S) == (256, 1000) and not args.thermal and not args.density_file:
            # What one GPU's share of the 8-GPU strong-scaling run costs, on THIS box: 125 of the 1000 sources (the static share of
            # rank 0), same field, next to an eighth of the headline step -- the ratio bounds the 8-GPU speed-up before the exchange
            # (DESIGN.md s3d, s6).  Informational, outside the timed region.  From its second pass on each chain of the share replays
            # its launch sequence as one hipGraph (no host round trip per sub-box).
            b.close()
            out["one_gpu_share_of_8"] = share_leg(pkg, torch, n, pkg.static_source_share(S, 0, 8), nd, xh, srcpos, normflux, s, (thick, thin), local_rank,
                                                  args.sweep_mode == "fast", options, 1e3 * dt_wall / args.steps / 8.0)
            out["one_gpu_share_of_8"]["note"] = "its own field relaxes from x = %.3f with 125 sources only: sub-box counts as in the headline" % args.x_init
        if world == 1 and not args.no_configs_leg and (n, S) == (256, 1000) and not args.thermal and not args.density_file:
            # The other BASELINE configs at HEAD, bounded (outside the timed region): configs[2] = 256^3 x 100 sources, and one GPU's
            # share of configs[4] = 504^3, 1250 of 10 000 sources, log-normal density, everything resident in HBM -- each with its
            # shell kernel's roofline from the same HIP-event path as the headline's.
            b.close()
            pos100, nf100 = pkg.seeded_sources(n, 100)
            out["configs"] = {"256_100src": timed_leg(pkg, torch, n, 100, nd, xh, pos100, nf100, s, (thick, thin), local_rank, args.sweep_mode == "fast", options, 5, warm=2, prof_mode=2)}
            out["configs"]["256_100src"]["what"] = "BASELINE configs[2]: 256^3, 100 seeded sources, same field and pre-ionisation; chained pass (2 chains), timed around the round"
            try:
                n5, S5 = 504, 10000
                tp5 = pkg.TestProblem(n5); s5 = tp5.step(1)
                nd5, xh5 = tp5.fields(1, args.x_init)
                rng = np.random.default_rng(20261003)
                nd5 = (nd5 * np.exp(rng.standard_normal(nd5.size, dtype=np.float32) - 0.5)).astype(np.float32)
                pos5, nf5 = pkg.seeded_sources(n5, S5)
                sh5 = pkg.static_source_share(S5, 0, 8)
                leg = timed_leg(pkg, torch, n5, len(sh5), nd5, xh5, pos5[sh5], nf5[sh5], s5, (thick, thin), local_rank, args.sweep_mode == "fast", options, 2, warm=1, prof_mode=1)
                leg["what"] = ("one GPU's share of BASELINE configs[4]: 504^3, log-normal density (sigma_ln = 1), the 1250 sources rank 0 of 8 sweeps of the "
                               "10 000 seeded ones, all in flight at once (30 GB of shell planes); `value` counts these 1250 sources")
                out["configs"]["504_share_of_8"] = leg
                # (the review's name for this leg; the 10 000-source step it is an eighth of takes ~11 s and is not timed in the default run:
                # `bench.py --mesh 504 --sources 10000 --density lognormal --no-cpu-baseline` does)
                out["share_504"] = {"see": "configs.504_share_of_8", "ms_per_step": leg["ms_per_step"], "frac": leg.get("roofline", {}).get("frac")}
                del nd5, xh5
            except Exception as exc:                    # (e.g. a GPU shared with another job: not enough free HBM)
                out["configs"]["504_share_of_8"] = {"skipped": repr(exc)}
        if world == 1 and not args.no_dropin_leg and (n, S) == (256, 1000) and not args.thermal and not args.density_file:
            # The boundary north_star names, end to end: the reference's OWN program (C2Ray.F90 and every set-up module,
            # unmodified) with its evolve modules replaced by the Fortran shim + this library, on its own test problem
            # (128^3, the one-source list, 14 time steps) -- wall clock per evolve3D call as the shim logs it.  Informational,
            # outside the timed region; needs the drop-in program (oracle/_ref, built where the reference is present).
            b.close()
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("dropin_timing", os.path.join(ROOT, "profiles", "dropin_timing.py"))
                dt_mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(dt_mod)
                leg = dt_mod.run_leg(128, dt_mod.SRC_ONE, 1, "hip-fast" if args.sweep_mode == "fast" else "hip-exact", 600)
                keep = ("skipped", "exit", "process_wall_s", "outer_iterations", "steps", "evolve3d_s", "split_total",
                        "split_total_without_first_step", "fraction_inside_evolve3d_dev_without_first_step", "sweep_mode")
                out["dropin_128_1src"] = dict({k: leg[k] for k in keep if k in leg},
                                              what="oracle/_ref/N128/hip/c2ray_test_hip: reference driver + evolve_hip.F90 + libc2ray_hip.so, "
                                                   "14 evolve3D calls; seconds summed over the calls (split: shim set-up, upload, outer "
                                                   "iterations, download, rest of the library call, logging/statistics after it); the "
                                                   "compiled reference takes 134 s for the same 14 calls (INTEGRATION.md)")
            except Exception as exc:
                out["dropin_128_1src"] = {"skipped": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, srcpos, normflux, xh_state, nbox_first, pkg.box_cost, nd=nd)
            # ... and, in the same CPU leg, the oracle as the CHECKER of this workload's results (never timed as the product)
            b.close()               # (idempotent: the other legs may have closed the context already)
            out["parity"] = parity_check(pkg, n, s, nd, xh, xh_state, srcpos, normflux, (thick, thin), local_rank, args.sweep_mode == "fast")
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    b.close()


if __name__ == "__main__":
    main()
