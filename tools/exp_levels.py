#!/usr/bin/env python3
"""Diagnostic: where the rows of a multi-level search go — per level of the grid the blocks that picked it, the blocks
that handed over (halo shape / size), the rows listed for nn_wide_kernel, the staged candidates — on the pinned
non-uniform scenes and the uniform 200k cloud at the command line's defaults (radius 3, 20 neighbours)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

opts = [kv.split("=") for kv in sys.argv[1:]]
for scene in ("lidar", "slab", "uniform"):
    if scene == "uniform":
        src, tgt, _, _ = synth.make_pair(200_000, cfg=2, stride=3)
    else:
        src, tgt, _, _ = synth.make_scene(scene, 200_000, stride=3)
    c = _lib.Context(0)
    for k, v in opts:
        c.set_option(k, int(v))
    c.set_params(3.0, 20, 5.0, 3)
    c.set_target(tgt)
    c.set_source(src)
    c.align(4, inner_steps=1)
    c.synchronize()
    c.set_option("level_stats", 1)
    t0 = time.perf_counter()
    c.align(10, cost_drop_thresh=-1.0, inner_steps=1)
    c.synchronize()
    dt = time.perf_counter() - t0
    st = c.debug_levels()
    print(f"{scene}: {10 / dt:7.0f} it/s  levels {st['levels']} base {st['base']}  short rows last {c.debug_short_rows()}")
    for l, lv in enumerate(st["per_level"]):
        b = max(lv["blocks"], 1)
        print(f"   level {l} r={lv['radius']:6.3f}: blocks/it {lv['blocks'] / 10:7.1f}  handed over shape {lv['handed_over_shape'] / 10:6.1f} size "
              f"{lv['handed_over_size'] / 10:6.1f}  short rows/it {lv['short_rows'] / 10:8.1f} of {lv['rows'] / 10:9.1f}  staged/block {lv['staged'] / b:7.1f}")
    c.profile_enable(True)
    c.align(5, cost_drop_thresh=-1.0, inner_steps=1)
    print("   ", {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in c.profile_get().items()}, flush=True)
    c.close()
