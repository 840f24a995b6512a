#!/usr/bin/env python3
"""Pairs per second end to end through ppcr_batch_run (host buffers in: uploads, grid build, source sort, 23 iterations) on
one GPU, by the number of registrations in flight; 64 pairs of 250k points (BASELINE configs[4]) and 64 of 100k."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

for n in (250_000, 100_000):
    pairs = [synth.make_pair(n, cfg=5, pair=p)[:2] for p in range(16)] * 4
    _lib.batch_run(pairs[:10], 1.0, 10, 5.0, n_iter=23, inner_steps=1, device_ids=(0,), lanes_per_device=8)
    for lanes in (1, 2, 3, 4, 6, 8):
        best = 0.0
        for rep in range(2):
            t0 = time.perf_counter()
            _lib.batch_run(pairs, 1.0, 10, 5.0, n_iter=23, inner_steps=1, device_ids=(0,), lanes_per_device=lanes)
            best = max(best, len(pairs) / (time.perf_counter() - t0))
        print(f"{n:7d} points, {lanes} in flight: {best:7.1f} pairs/s", flush=True)
