#!/usr/bin/env python3
"""Instruction mix of the innermost (trip) loop of k_sweep_octant_fast and of k_sweep_shell_fast, from the gfx950 assembly:
basic blocks between the loop header of greatest depth and its back edge.   python profiles/r03_octant/isa_loop.py"""
import os, re, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "c2-ray3dm_amd", "csrc", "c2ray_hip.hip")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics".split()
out = "/tmp/c2r_isa.s"
subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + os.environ.get("C2R_EXTRA_FLAGS", "").split() + ["-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
def body(pat):
    i = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
    j = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(".Lfunc_end"))
    return lines[i + 1:j]
def mix(ls):
    c = Counter()
    for l in ls:
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"): continue
        c[t.split()[0]] += 1
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    return dict(VALU=g(lambda k: k.startswith("v_")), f64=g(lambda k: k.startswith("v_") and "f64" in k),
                lane=g(lambda k: k.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))),
                mov=g(lambda k: k.startswith("v_mov") or k.startswith("v_accvgpr")), cnd=g(lambda k: k.startswith("v_cndmask")),
                cmp=g(lambda k: k.startswith("v_cmp")), SALU=g(lambda k: k.startswith("s_") and not k.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop"))),
                SMEM=g(lambda k: k.startswith(("s_load", "s_buffer"))), VMEM=g(lambda k: k.startswith(("buffer_", "global_", "flat_", "scratch_"))),
                LDS=g(lambda k: k.startswith("ds_")), wait=c["s_waitcnt"]), c
b = body("k_sweep_octant_fastILi1ELb1E")
# the trip loop: from the deepest loop header to the last line that mentions it as parent / itself
depth = [(int(re.search(r"Depth=(\d+)", l).group(1)), i) for i, l in enumerate(b) if "Loop Header: Depth=" in l or "Inner Loop Header: Depth=" in l]
print("loop headers (depth, line):", depth)
for d, i in depth:
    lab = b[i].split(":")[0].strip()
    ends = [k for k, l in enumerate(b) if ("in Loop: Header=" + lab.lstrip(".L")) in l.replace(".L", "") or lab in l]
    j = max(ends)
    # extend to the end of that last block
    while j + 1 < len(b) and not b[j + 1].strip().startswith(".LBB"): j += 1
    m, c = mix(b[i:j + 1])
    print("loop %s depth %d lines %d..%d:" % (lab, d, i, j), m)
    if d == max(x for x, _ in depth) or "-v" in sys.argv:
        print("   ", "  ".join("%s:%d" % (k, v) for k, v in c.most_common(40)))
m, c = mix(body("k_sweep_shell_fastILb0ELi1ELb1ELb0E"))
print("k_sweep_shell_fast whole kernel:", m)
print("   ", "  ".join("%s:%d" % (k, v) for k, v in c.most_common(40)))
