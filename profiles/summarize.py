#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel_stats.csv + one counter_collection.csv per --pmc pass) of a
bench.py run into traffic.json: per-kernel launch counts, average durations and HBM bytes.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, i.e. reports HALF the bytes of a
wide coalesced read, so the read side is doubled ("fetch_corrected"); WRITE_SIZE is taken as is.
The guide calibrates that factor for 16-B-per-lane streams; this kernel reads 8 B and 4 B per
lane, so both the raw and the corrected figure are kept.

usage: summarize.py <dir> [visited cell-sources in the pmc run]
<dir> holds kernel_stats.csv, pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv, optionally pmc_SQ.csv, and the bench.py
lines printed by the profiled runs: kt_bench.json (kernel-trace run), pmc_bench.json (a PMC run; supplies
the visited count when it is not given).  Also records whether rocprof's average k_sweep_shell duration
agrees with the HIP-event average bench.py measured in the same run.
"""
import collections
import csv
import json
import os
import sys


def kname(full):
    """'void c2r::k_sweep_shell<false, 1>(c2r::KParams, ...)' -> 'c2r::k_sweep_shell'.  The plane-ordered launches of a shell
    (k_sweep_shell_xcd<..., FAST>: the same cells and arithmetic under another block mapping, round 5) count as launches of the
    mode's shell kernel."""
    head = full.split("(")[0]
    k = head.split("<")[0].strip()
    k = k[5:] if k.startswith("void ") else k
    if k == "c2r::k_sweep_shell_xcd":
        args = head.split("<", 1)[1].rsplit(">", 1)[0] if "<" in head else ""
        return "c2r::k_sweep_shell_fast" if args.replace(" ", "").endswith("true") or not args else "c2r::k_sweep_shell"
    return k


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    d = sys.argv[1]
    visited = float(sys.argv[2]) if len(sys.argv) > 2 else None
    pb = os.path.join(d, "pmc_bench.json")
    if visited is None and os.path.exists(pb):
        visited = json.load(open(pb))["config"]["visited_cell_sources_whole_run_rank0"]
    out = {"kernels": {}}
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
        k = kname(r["Name"])
        if k.startswith("c2r::"):          # (several instantiations / block mappings of one kernel: merged)
            e = out["kernels"].setdefault(k, {"calls": 0, "avg_ns": 0.0, "total_ns": 0.0, "pct": 0.0})
            e["calls"] += int(r["Calls"]); e["total_ns"] += float(r["TotalDurationNs"]); e["pct"] += float(r["Percentage"])
            e["avg_ns"] = e["total_ns"] / e["calls"]
            if "k_sweep_shell_xcd" in r["Name"]:
                e["plane_ordered_calls"] = e.get("plane_ordered_calls", 0) + int(r["Calls"])
    f = per_kernel(os.path.join(d, "pmc_FETCH_SIZE.csv"))
    w = per_kernel(os.path.join(d, "pmc_WRITE_SIZE.csv"))
    for k in out["kernels"]:
        if k in f:
            n, kib = f[k]
            out["kernels"][k].update(pmc_launches=n, fetch_raw_bytes_per_launch=kib * 1024 / n,
                                     fetch_corrected_bytes_per_launch=2 * kib * 1024 / n)
        if k in w:
            n, kib = w[k]
            out["kernels"][k].update(write_bytes_per_launch=kib * 1024 / n)
    # the sweep kernel of the run: the tolerance-mode kernel when bench.py ran with --sweep-mode fast
    SWEEP = "c2r::k_sweep_shell_fast" if "c2r::k_sweep_shell_fast" in out["kernels"] else "c2r::k_sweep_shell"
    out["sweep_kernel"] = SWEEP
    sw = out["kernels"].get(SWEEP)
    if sw and visited and "pmc_launches" in sw:
        vis_per_launch = visited / sw["pmc_launches"]
        out["sweep_bytes_per_visit"] = {
            "fetch_raw": sw["fetch_raw_bytes_per_launch"] / vis_per_launch,
            "fetch_corrected": sw["fetch_corrected_bytes_per_launch"] / vis_per_launch,
            "write": sw["write_bytes_per_launch"] / vis_per_launch,
            "visited_in_pmc_run": visited}
    kb = os.path.join(d, "kt_bench.json")
    if sw and os.path.exists(kb):
        ev_ms = json.load(open(kb))["roofline"]["avg_launch_ms"]
        kbj = json.load(open(kb))
        out["agreement"] = {"rocprof_stats_avg_ms_all_launches": sw["avg_ns"] * 1e-6,
                            "hip_event_avg_ms_timed_steps": ev_ms,
                            "note": "--stats averages every launch of the process; the very first pass (input "
                                    "preparation, uniform x) stops 18 shells earlier than all later ones, so "
                                    "its launches are the smaller ones"}
        tr = os.path.join(d, "kernel_trace.csv")
        if os.path.exists(tr):     # like with like: the launches of the timed region are the last ones
            n = int(kbj["roofline"]["launches"])
            durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(tr))
                    if kname(r["Kernel_Name"]) == SWEEP]
            last = durs[-n:]
            out["agreement"].update(rocprof_trace_avg_ms_timed_launches=sum(last) / len(last) * 1e-6,
                                    timed_launches=n, ratio_trace_over_events=sum(last) / len(last) * 1e-6 / ev_ms)
    sq = os.path.join(d, "pmc_SQ.csv")
    if os.path.exists(sq):
        agg = collections.defaultdict(float)
        dur = {}
        for r in csv.DictReader(open(sq)):
            if kname(r["Kernel_Name"]) == SWEEP:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if agg.get("SQ_WAVES"):
            # SIMD-cycles available while the kernel ran: 256 CUs x 4 SIMDs at the measured engine clock
            # (pmc_GRBM.csv: GRBM_COUNT summed over the 8 XCDs / kernel time), else the nominal 2.4 GHz
            ghz = 2.4
            gr = os.path.join(d, "pmc_GRBM.csv")
            if os.path.exists(gr):
                cnt, gd = 0.0, {}
                for r in csv.DictReader(open(gr)):
                    if kname(r["Kernel_Name"]) == SWEEP and r["Counter_Name"] == "GRBM_COUNT":
                        cnt += float(r["Counter_Value"])
                        gd[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                if gd:
                    ghz = cnt / 8 / sum(gd.values())
            simd_cycles = sum(dur.values()) * ghz * 1024
            sq2 = os.path.join(d, "pmc_SQ2.csv")
            if os.path.exists(sq2):
                a2 = collections.defaultdict(float)
                for r in csv.DictReader(open(sq2)):
                    if kname(r["Kernel_Name"]) == SWEEP:
                        a2[r["Counter_Name"]] += float(r["Counter_Value"])
                if a2.get("SQ_WAVES"):
                    out["sweep_sq2_per_wave"] = {k: v / a2["SQ_WAVES"] for k, v in a2.items() if k != "SQ_WAVES"}
            out["sweep_sq"] = {"valu_insts_per_wave": agg["SQ_INSTS_VALU"] / agg["SQ_WAVES"],
                               "engine_clock_ghz_during_kernel": ghz,
                               "valu_busy_frac_of_simd_cycles": 4 * agg["SQ_ACTIVE_INST_VALU"] / simd_cycles,
                               "kernel_ns_in_this_pass": sum(dur.values()), "raw": dict(agg)}
    # every other counter pass (pmc_TCC*.csv, pmc_TCP*.csv): totals of the sweep kernel per visited pair
    extra = {}
    for fn in sorted(os.listdir(d)):
        if fn.startswith("pmc_TC") and fn.endswith(".csv"):
            for r in csv.DictReader(open(os.path.join(d, fn))):
                if kname(r["Kernel_Name"]) == SWEEP:
                    extra[r["Counter_Name"]] = extra.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if extra and visited:
        out["sweep_counters_per_visit"] = {k: v / visited for k, v in sorted(extra.items())}
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
