#!/usr/bin/env python3
"""Diagnostic: one-pass association at growing cell occupancy (uniform 200k cloud, radius = cell edge): K1 / cleanup
durations and hand-overs per iteration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth
src, tgt, _, _ = synth.make_pair(200_000, cfg=2, stride=3)
for (r, m) in ((1.0, 10), (1.25, 10), (1.5, 10), (1.5, 20), (1.75, 20), (2.0, 20)):
    c = _lib.Context(0)
    c.set_option("two_pass", 0)
    for kv in sys.argv[1:]:
        k, v = kv.split("="); c.set_option(k, int(v))
    c.set_params(r, m, 5.0, 3); c.set_target(tgt); c.set_source(src)
    c.align(3, inner_steps=1); c.synchronize()
    c.profile_enable(True); c.align(6, cost_drop_thresh=-1.0, inner_steps=1)
    prof = {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in c.profile_get().items()}
    print(f"r={r} m={m} q={3.8147 * r ** 3:5.1f}: handed over/it {c.debug_host_figures()[7] / 6:6.1f}  {prof}", flush=True)
    c.close()
