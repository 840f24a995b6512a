#!/usr/bin/env python3
"""A coarse map of iterations/s over list width x points in the radius (uniform clouds, n points each) — to find holes, not to
tune.  usage: exp_sweep.py [n]  (gpurun)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
rng = np.random.default_rng(9)
print("n =", n, "| columns: max_neighbours; cells: it/s (reach, levels, lists, handed-over workgroups per iteration)")
for in_radius in (6, 16, 42, 100, 430):
    rho = in_radius / 4.18879
    side = (n / rho) ** (1 / 3)
    tgt = rng.uniform(0, side, size=(n, 3)).astype(np.float32)
    src = (tgt[rng.permutation(n)] + rng.normal(0, 0.02, size=(n, 3)) + [0.04, -0.03, 0.02]).astype(np.float32)
    row = []
    for m in (4, 5, 8, 10, 16, 20, 32):
        with _lib.Context(0) as c:
            c.set_params(1.0, m, 5.0, 3)
            c.set_target(tgt)
            c.set_source(src)
            c.align(30, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
            c.synchronize()
            t0 = time.perf_counter()
            c.align(40, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
            c.synchronize()
            dt = time.perf_counter() - t0
            row.append(f"m{m}: {40 / dt:7.0f} ({c.search_reach()},{c.debug_levels()['levels']},{'L' if c.debug_verlet()['trusted'] else '-'},{c.debug_host_figures()[7] / 40:.0f})")
    print(f"{in_radius:4d} in radius: " + "  ".join(row), flush=True)
