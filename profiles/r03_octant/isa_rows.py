#!/usr/bin/env python3
"""Instruction mix of k_sweep_octant_fast's innermost loops from the gfx950 assembly: the row loop (one cell per
iteration) and the item prologue around it.   python profiles/r03_octant/isa_rows.py [-v]"""
import os, re, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "c2-ray3dm_amd", "csrc", "c2ray_hip.hip")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics".split()
out = "/tmp/c2r_isa.s"
subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + os.environ.get("C2R_EXTRA_FLAGS", "").split() + ["-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
s = open(out).read()
m = re.search(r'^_ZN3c2r19k_sweep_octant_fastILi1ELb1EEEvNS_7KParamsENS_7OctArgsE:(.*?)\.Lfunc_end', s, re.S | re.M)
b = m.group(1).split('\n')
def mix(ls):
    c = Counter()
    for l in ls:
        t = l.strip()
        if not t or t[0] in ';.' or t.endswith(':'): continue
        c[t.split()[0]] += 1
    return c
def show(name, a, e):
    c = mix(b[a:e])
    g = lambda f: sum(v for k, v in c.items() if f(k))
    print("%s [%d..%d]: VALU %d (f64 %d, lane %d, mov %d)  SALU %d  SMEM %d  VMEM %d  LDS %d  waitcnt %d" % (name, a, e,
          g(lambda k: k.startswith('v_')), g(lambda k: k.startswith('v_') and 'f64' in k), g(lambda k: 'lane' in k), g(lambda k: k.startswith('v_mov')),
          g(lambda k: k.startswith('s_') and not k.startswith(('s_load', 's_waitcnt', 's_nop'))), g(lambda k: k.startswith('s_load')),
          g(lambda k: k.startswith(('buffer_', 'global_'))), g(lambda k: k.startswith('ds_')), c['s_waitcnt']))
    if '-v' in sys.argv: print('   ', '  '.join('%s:%d' % (k, v) for k, v in c.most_common(50)))
hdr = [(int(re.search(r'Depth=(\d+)', l).group(1)), i) for i, l in enumerate(b) if 'Loop Header: Depth=' in l]
# the header comment sits on the label line itself (".LBBx_y: ; =>This Inner Loop Header") or on the line after it
d5 = [i for d, i in hdr if d == 5]; d4 = [i for d, i in hdr if d == 4]
at = [i for i, l in enumerate(b) if 'buffer_atomic_add_f64' in l or 'global_atomic_add_f64' in l]
row0 = d5[-1]; item0 = max(i for i in d4 if i < row0)
# end of the row loop: the backward branch to its header label
lab = [b[k].split(':')[0].strip() for k in range(max(0, row0 - 6), row0 + 1) if b[k].strip().startswith('.LBB')][-1]
rowe = max(i for i, l in enumerate(b) if re.search(r's_cbranch\w+\s+' + re.escape(lab) + r'\b', l) or re.search(r's_branch\s+' + re.escape(lab) + r'\b', l))
show("row loop", row0, rowe + 1)
show("item prologue", item0, row0)
print("atomics at", at)
for i in range(row0, rowe + 1):
    if any(k in b[i] for k in ('s_waitcnt', 'global_', 'buffer_', 'ds_', 's_cbranch', 's_barrier')): print(i, b[i].strip()[:90])
