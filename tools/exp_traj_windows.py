#!/usr/bin/env python3
"""The benchmark's timed windows (fresh source, warm-up + 20 timed iterations, five windows) on the headline pair and on the
second trajectory (3 x motion, 3 x noise), with the workgroups handed over per window.
usage: exp_traj_windows.py [n] [key=value ...]   (PPCR_HIP_LIB selects a variant library)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
opts = [kv.split("=") for kv in sys.argv[2:]]
for scale in (1.0, 3.0):
    src, tgt, _, _ = synth.make_pair(n, cfg=3 if n >= 1_000_000 else 2, motion_scale=scale, noise_scale=scale)
    c = _lib.Context(0)
    for k, v in opts:
        c.set_option(k, int(v))
    c.set_params(1.0, 10, 5.0, 3)
    c.set_target(tgt)
    rates, handed = [], []
    for w in range(6):
        c.set_source(src)
        c.align(3, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
        c.synchronize()
        t0 = time.perf_counter()
        c.align(20, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
        c.synchronize()
        rates.append(20 / (time.perf_counter() - t0))
        handed.append(c.debug_host_figures()[7])
    print(f"n {n} trajectory x{scale:.0f}: window {np.median(rates[1:]):8.1f} it/s (min {min(rates[1:]):8.1f} max {max(rates[1:]):8.1f})  handed over per window {handed[1:]}", flush=True)
    c.close()
