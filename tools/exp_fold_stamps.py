#!/usr/bin/env python3
"""Diagnostic: where the fold-and-solve launch spends its time (wall-clock stamps left by the solve lane, 10 ns ticks)."""
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
rows = []
c.align(6, inner_steps=1)
for k in range(12):
    c.align(1, inner_steps=1)
    out = (C.c_ulonglong * 8)()
    assert L.ppcr_debug_get_fold_stamps(c._h, out) == 0
    t = np.array(out[:6], dtype=np.float64) * 0.01   # us
    rows.append(np.diff(t))
rows = np.array(rows)
print("us: entry->folded, folded->ticket, ticket->sums read, sums->solved, solved->published")
print(np.round(np.median(rows, axis=0), 2), " total", round(float(np.median(rows.sum(axis=1))), 2))
