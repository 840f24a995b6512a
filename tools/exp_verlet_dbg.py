#!/usr/bin/env python3
"""Debug aid: the association after k iterations of ppcr_align, Verlet lists on against off, for k = 1, 2, ... (a fresh
source each time: exporting an association applies the pending move and ends the steady state).
usage: exp_verlet_dbg.py [n] [m] [inner] [key=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10
inner = int(sys.argv[3]) if len(sys.argv) > 3 else 1
opts = [kv.split("=") for kv in sys.argv[4:]]
src, tgt, _, _ = synth.make_pair(n, cfg=2, stride=3)
ctx = []
for verlet in (0, int(os.environ.get("VERLET", "1"))):
    c = _lib.Context(0)
    c.set_option("verlet", verlet)
    for k, v in opts:
        c.set_option(k, int(v))
    c.set_params(1.0, m, 5.0, 3)
    c.set_target(tgt)
    ctx.append(c)
bad = 0
for k in range(1, 13):
    res = []
    for c in ctx:
        c.set_source(src)
        h = c.align(k, cost_drop_thresh=0.0, inner_steps=inner)["history"][-1]
        info = c.debug_verlet()
        rp, col, d2 = c.get_association()
        res.append((rp, col, d2, h))
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    print(f"{k} iterations: same association = {same}, max |dT| = {np.max(np.abs(res[0][3] - res[1][3])):.3e}", info, flush=True)
    if not same and bad < 2:
        bad += 1
        cnt0, cnt1 = np.diff(res[0][0]), np.diff(res[1][0])
        rows = np.nonzero(cnt0 != cnt1)[0]
        print("   rows whose counts differ:", rows.size, rows[:10], cnt0[rows[:10]], cnt1[rows[:10]])
        shown = 0
        for r in range(len(cnt0)):
            a = res[0][1][res[0][0][r]:res[0][0][r + 1]]
            b = res[1][1][res[1][0][r]:res[1][0][r + 1]]
            if not np.array_equal(a, b):
                print("   row", r, "plain ", a, np.sqrt(res[0][2][res[0][0][r]:res[0][0][r + 1]]).round(4))
                print("   row", r, "verlet", b, np.sqrt(res[1][2][res[1][0][r]:res[1][0][r + 1]]).round(4))
                shown += 1
                if shown == 3:
                    break
