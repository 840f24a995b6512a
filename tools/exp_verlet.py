#!/usr/bin/env python3
"""Verlet lists against the plain steady-state K1: iterations/s of ppcr_align at one size, per-kernel HIP-event times,
and the transforms of both (must agree to the last bit of the neighbour sets: compared via the final transform and the
exported association).  usage: exp_verlet.py [n] [key=value options ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
opts = [kv.split("=") for kv in sys.argv[2:]]
cfg = 3 if n >= 1_000_000 else 2
src, tgt, _, _ = synth.make_pair(n, cfg=cfg)
out = {}
for verlet in (0, 1):
    c = _lib.Context(0)
    c.set_option("verlet", verlet)
    for k, v in opts:
        c.set_option(k, int(v))
    c.set_params(1.0, 10, 5.0, 3)
    c.set_target(tgt)
    c.set_source(src)
    c.align(5, cost_drop_thresh=0.0, inner_steps=1, want_history=False)
    best = 0.0
    for rep in range(3):
        c.synchronize()
        t0 = time.perf_counter()
        c.align(40, cost_drop_thresh=0.0, inner_steps=1, want_history=False)
        c.synchronize()
        best = max(best, 40 / (time.perf_counter() - t0))
    c.profile_enable(True)
    r = c.align(20, cost_drop_thresh=0.0, inner_steps=1)
    stats = c.profile_get()
    c.profile_enable(False)
    if verlet:
        for k in range(4):
            c.align(1, cost_drop_thresh=0.0, inner_steps=1, want_history=False)
            print("   ", c.debug_verlet(), flush=True)
    rp, col, d2 = c.get_association()
    out[verlet] = (r["history"][-1], rp, col)
    print(f"verlet={verlet}: {best:9.1f} it/s;", "  ".join(f"{k} {1e3 * v['total_ms'] / v['launches']:.1f}us x{v['launches']}" for k, v in stats.items()), flush=True)
    # the cold start: a fresh source, six iterations the way the command line stops
    c.set_source(src)
    c.synchronize()
    t0 = time.perf_counter()
    c.align(6, cost_drop_thresh=0.0, inner_steps=100, f_tol=1e-5, want_history=False)
    c.synchronize()
    print(f"   six iterations from a fresh source, inner loop on: {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
    c.close()
print("same association:", np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2]),
      " max |dT|:", float(np.max(np.abs(out[0][0] - out[1][0]))))
