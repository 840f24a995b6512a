#!/usr/bin/env python3
"""One-pass against two-pass at densities between the benchmark's (16 points in the radius, max_neighbours 10) and the
command line's defaults (430 in the radius, 20 neighbours): where does the automatic choice of a two-pass search pay?  (gpurun)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib

rng = np.random.default_rng(5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for in_radius in (16, 28, 34, 38, 42, 50, 64):
    rho = in_radius / 4.18879
    side = (n / rho) ** (1 / 3)
    tgt = rng.uniform(0, side, size=(n, 3)).astype(np.float32)
    src = (tgt[rng.permutation(n)] + rng.normal(0, 0.02, size=(n, 3)) + [0.05, -0.03, 0.02]).astype(np.float32)
    row = []
    for label, opts in (("auto", {}), ("one-pass", {"two_pass": 0}), ("one-pass, no lists", {"two_pass": 0, "verlet": 0}), ("two-pass", {"two_pass": 1})):
        with _lib.Context(0) as c:
            for k, v in opts.items():
                c.set_option(k, v)
            c.set_params(1.0, 10, 5.0, 3)
            c.set_target(tgt)
            c.set_source(src)
            c.align(40, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
            c.synchronize()
            t0 = time.perf_counter()
            c.align(60, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
            c.synchronize()
            row.append(f"{label}: {60 / (time.perf_counter() - t0):8.0f} it/s (reach {c.search_reach()}, levels {c.debug_levels()['levels']}, lists {c.debug_verlet()['trusted']})")
    print(f"{in_radius:4d} points in the radius, n = {n}: " + "; ".join(row), flush=True)
