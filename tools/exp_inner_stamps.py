#!/usr/bin/env python3
"""Diagnostic: timeline of the first device-paced IRLS step inside inner_steps_kernel (wall-clock stamps, 10 ns ticks)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src, tgt, _, _ = synth.make_pair(n, cfg=3)
c = _lib.Context(0)
c.set_option("fold_stamps", 1)
c.set_params(1.0, 10, 5.0, 3); c.set_target(tgt); c.set_source(src)
L = _lib.load(); L.ppcr_debug_get_fold_stamps.argtypes = [C.c_void_p, C.c_void_p]
c.align(2, inner_steps=100, f_tol=10e-6)
rows = []
for it in range(24):
    r = c.align(1, cost_drop_thresh=-1.0, inner_steps=100, f_tol=10e-6)
    if int(r["inner_steps"][0]) != 2:
        continue
    out = (C.c_ulonglong * 8)()
    assert L.ppcr_debug_get_fold_stamps(c._h, out) == 0
    t = np.array(out[:8], dtype=np.float64) * 0.01
    # [7] first K23 wg starts, [6] first fold wg starts waiting, [0] solving fold wg past its wait, [1] folded, [2] ticket,
    # [3] sums read, [4] solved, [5] published
    rows.append([t[6] - t[7], t[0] - t[7], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[7]])
rows = np.array(rows)
print("us from the first K23 workgroup's start: fold wg dispatched, flags complete (+fence), +fold, +ticket, +sums, +solve, +publish, total")
print(np.round(np.median(rows, axis=0), 2), " n =", len(rows))
