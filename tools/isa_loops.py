#!/usr/bin/env python3
"""Static look at one kernel's ISA (an .s file made with hipcc --cuda-device-only -S): instruction counts of the whole
kernel and of every loop (backward branch).  usage: isa_loops.py file.s <substring of the mangled kernel name>"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().endswith(l.split(":")[0][:0] + l[l.index(":"):]))
fend = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:fend]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}


def kind(l):
    t = l.strip()
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        return None
    op = t.split()[0]
    for p, k in (("v_", "VALU"), ("s_", "SALU"), ("ds_", "LDS"), ("global_", "VMEM"), ("buffer_", "VMEM"), ("flat_", "VMEM"), ("scratch_", "VMEM")):
        if op.startswith(p):
            return k
    return "other"


def count(a, b):
    c = {}
    for l in body[a:b + 1]:
        k = kind(l)
        if k:
            c[k] = c.get(k, 0) + 1
    return c


print("kernel:", body[0].split(":")[0][:90], "lines", len(body))
print("static totals", count(0, len(body) - 1))
loops = []
for i, l in enumerate(body):
    m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))
for a, b in sorted(loops):
    print(f"loop @{a}-{b} ({b - a + 1} lines)", count(a, b))
