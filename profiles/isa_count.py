#!/usr/bin/env python3
"""Static instruction mix of the sweep kernels from the gfx950 assembly hipcc emits.

    python profiles/isa_count.py [substring-of-mangled-name ...]

Compiles c2-ray3dm_amd/csrc/sweep.hip with the Makefile's flags to assembly (device only) and
counts, per kernel, VALU / f64 VALU / SALU / vector-memory / LDS instructions in the kernel body.
Static counts are an upper bound of what one wave executes (both sides of divergent branches are
counted); the dynamic count per wave is in profiles/*/pmc_SQ.csv.
"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "c2-ray3dm_amd", "csrc", "sweep.hip")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics".split()


def main():
    pats = sys.argv[1:] or ["k_sweep_shell_fastILb0ELi1ELb1", "k_sweep_shellILb0ELi1ELb1"]
    extra = os.environ.get("C2R_EXTRA_FLAGS", "").split()
    out = "/tmp/c2r_isa.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-S", "--cuda-device-only", "-o", out, SRC],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for n, (i, name) in enumerate(starts):
        if not any(p in name for p in pats):
            continue
        c = Counter()
        for l in lines[i + 1:]:
            t = l.strip()
            if t.startswith(".Lfunc_end"):
                break
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            c[t.split()[0]] += 1
        grp = lambda pred: sum(v for k, v in c.items() if pred(k))
        valu = grp(lambda k: k.startswith("v_"))
        f64 = grp(lambda k: k.startswith("v_") and "f64" in k)
        trans = grp(lambda k: k in ("v_rcp_f64_e32", "v_rsq_f64_e32", "v_rcp_f64_e64", "v_rsq_f64_e64"))
        print("%s\n  VALU %d (f64 %d, of them rcp/rsq %d)  SALU %d  SMEM %d  buffer/global %d  LDS %d" %
              (name, valu, f64, trans, grp(lambda k: k.startswith("s_") and not k.startswith("s_load") and not k.startswith("s_buffer")),
               grp(lambda k: k.startswith("s_load") or k.startswith("s_buffer")),
               grp(lambda k: k.startswith(("buffer_", "global_", "flat_"))), grp(lambda k: k.startswith("ds_"))))
        top = sorted(((v, k) for k, v in c.items() if k.startswith("v_")), reverse=True)[:24]
        print("  " + "  ".join("%s:%d" % (k, v) for v, k in top))


if __name__ == "__main__":
    main()
