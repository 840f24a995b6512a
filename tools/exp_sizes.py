#!/usr/bin/env python3
"""Iterations/s over cloud sizes at the benchmark's density and parameters, full and partial overlap (gpurun)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib

rng = np.random.default_rng(10)
for n in (3_000, 10_000, 30_000, 100_000, 300_000, 1_000_000, 3_000_000):
    side = (n / 3.81) ** (1 / 3)
    tgt = rng.uniform(0, side, size=(n, 3)).astype(np.float32)
    base = (tgt[rng.permutation(n)] + rng.normal(0, 0.02, size=(n, 3))).astype(np.float32)
    row = []
    for label, shift in (("full overlap", 0.03), ("half overlap", 0.5 * side)):
        src = base + np.float32([shift, -0.02, 0.01])
        with _lib.Context(0) as c:
            c.set_params(1.0, 10, 5.0, 3)
            c.set_target(tgt)
            c.set_source(src)
            c.align(30, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
            c.synchronize()
            t0 = time.perf_counter()
            c.align(40, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
            c.synchronize()
            dt = time.perf_counter() - t0
            row.append(f"{label}: {40 / dt:8.0f} it/s = {40 * n / dt / 1e9:6.2f} G rows/s (lists {'on' if c.debug_verlet()['trusted'] else 'off'}, handed over {c.debug_host_figures()[7] / 40:.0f}/it)")
    print(f"n = {n:8d}: " + "; ".join(row), flush=True)
