#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel_stats.csv + one counter_collection.csv per --pmc pass) of a
bench.py run into traffic.json: per-kernel launch counts, average durations and HBM bytes.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, i.e. reports HALF the bytes of a
wide coalesced read, so the read side is doubled ("fetch_corrected"); WRITE_SIZE is taken as is.
The guide calibrates that factor for 16-B-per-lane streams; this kernel reads 8 B and 4 B per
lane, so both the raw and the corrected figure are kept.

usage: summarize.py <dir with kernel_stats.csv pmc_FETCH_SIZE.csv pmc_WRITE_SIZE.csv> <visited cell-sources in the pmc run>
"""
import collections
import csv
import json
import os
import sys


def kname(full):
    """'void c2r::k_sweep_shell<false, 1>(c2r::KParams, ...)' -> 'c2r::k_sweep_shell'"""
    k = full.split("(")[0].split("<")[0].strip()
    return k[5:] if k.startswith("void ") else k


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
    out = {"kernels": {}}
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
        k = kname(r["Name"])
        if k.startswith("c2r::"):
            out["kernels"][k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                 "total_ns": float(r["TotalDurationNs"]), "pct": float(r["Percentage"])}
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
    sw = out["kernels"].get("c2r::k_sweep_shell")
    if sw and visited and "pmc_launches" in sw:
        vis_per_launch = visited / sw["pmc_launches"]
        out["sweep_bytes_per_visit"] = {
            "fetch_raw": sw["fetch_raw_bytes_per_launch"] / vis_per_launch,
            "fetch_corrected": sw["fetch_corrected_bytes_per_launch"] / vis_per_launch,
            "write": sw["write_bytes_per_launch"] / vis_per_launch,
            "visited_in_pmc_run": visited}
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
