#!/usr/bin/env python3
"""Diagnostic: the scan's lane balance (see exp_lanebalance.py) on the non-uniform scenes under the multi-level search, and on
the uniform 200k cloud for comparison.  Run lengths are recorded with 7 bits (saturating at 127).  usage: ... [m]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10
L = _lib.load()
L.ppcr_debug_get_stamps_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for scene in ("lidar", "slab", "uniform"):
    if scene == "uniform":
        src, tgt, _, _ = synth.make_pair(200_000, cfg=2, stride=3)
    else:
        src, tgt, _, _ = synth.make_scene(scene, 200_000, stride=3)
    c = _lib.Context(0)
    c.set_params(3.0, m, 5.0, 3)
    c.set_target(tgt)
    c.set_source(src)
    c.align(6, inner_steps=1)
    c.synchronize()
    c.set_option("stamps", 1)
    c.iterate()
    c.synchronize()
    multi = c.debug_levels()["levels"] > 1
    nb = (len(src) + 255) // 256
    grid = 2 * ((nb + 7) // 8 * 8) if multi else nb + 128
    base = (grid * 4 + 64) * 8
    h = np.zeros(base + grid * 256, dtype=np.uint64)
    assert L.ppcr_debug_get_stamps_raw(c._h, h.ctypes.data, h.size) == 0
    pk = h[base:].reshape(grid, 256)
    runs = np.stack([(pk >> np.uint64(7 * k)) & np.uint64(127) for k in range(9)], axis=-1).astype(np.int64)
    live_wg = runs.reshape(grid, -1).sum(axis=1) > 0
    runs = runs[live_wg]
    trips = (runs + 1) // 2
    w = trips.reshape(-1, 4, 64, 9)
    now = w.max(axis=2).sum(axis=-1)
    ideal = w.sum(axis=(2, 3)) / 64.0
    q = runs.sum(axis=-1)
    print(f"{scene:8s} m={m} multi={multi}: workgroups {live_wg.sum()}  candidates/query mean {q[q > 0].mean():.1f} p50 {np.percentile(q[q > 0], 50):.0f} "
          f"p90 {np.percentile(q[q > 0], 90):.0f} p99 {np.percentile(q[q > 0], 99):.0f} | trips per wave now {now.mean():.1f} ideal {ideal.mean():.1f} "
          f"(utilisation {ideal.sum() / now.sum():.3f}) | saturated runs {(runs == 127).mean():.4f}")
    print("          rank maxima per wave " + " ".join(f"{v:.1f}" for v in w.max(axis=2).mean(axis=(0, 1))) + " | rank means "
          + " ".join(f"{v:.2f}" for v in w.mean(axis=(0, 1, 2))), flush=True)
    c.close()
