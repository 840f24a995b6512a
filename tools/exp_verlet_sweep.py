#!/usr/bin/env python3
"""The benchmark's timed windows (fresh source, 5 untimed + 20 timed iterations of ppcr_align, one inner step) over a grid of
Verlet options: skin (how far beyond the cut-off bound a list reaches) x engage threshold.  Prints the median window rate,
the share of workgroups that searched and the rows rebuilt one by one per window.
usage: exp_verlet_sweep.py [n] [skins=500,250] [engages=350,75] [key=value ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
skins, engages, opts = [500], [-1], []
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "skins":
        skins = [int(x) for x in v.split(",")]
    elif k == "engages":
        engages = [int(x) for x in v.split(",")]
    else:
        opts.append((k, int(v)))
src, tgt, _, _ = synth.make_pair(n, cfg=3 if n >= 1_000_000 else (2 if n <= 100_000 else 5))
for skin in skins:
    for engage in engages:
        c = _lib.Context(0)
        c.set_option("verlet_skin", skin)
        c.set_option("verlet_engage", engage)
        for k, v in opts:
            c.set_option(k, v)
        c.set_params(1.0, 10, 5.0, 3)
        c.set_target(tgt)
        rates, shares, rows = [], [], []
        for w in range(5):
            c.set_source(src)
            c.align(5, cost_drop_thresh=0.0, inner_steps=1, want_history=False)
            c.synchronize()
            v0 = c.debug_verlet()
            t0 = time.perf_counter()
            c.align(20, cost_drop_thresh=0.0, inner_steps=1, want_history=False)
            c.synchronize()
            rates.append(20 / (time.perf_counter() - t0))
            v1 = c.debug_verlet()
            shares.append((v1["rebuilt"] - v0["rebuilt"]) / (20 * max(1, v1["workgroups"] - 128)))
            rows.append(v1["rows_rebuilt"] - v0["rows_rebuilt"])
        # converged: 60 more iterations, then 60 timed
        c.align(60, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
        c.synchronize()
        t0 = time.perf_counter()
        c.align(60, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
        c.synchronize()
        conv = 60 / (time.perf_counter() - t0)
        print(f"skin {skin:4d} engage {engage:4d}: window {np.median(rates):8.1f} it/s (min {min(rates):8.1f} max {max(rates):8.1f})  searched share {np.median(shares):.4f}  "
              f"rows rebuilt {int(np.median(rows)):7d}  converged {conv:8.1f} it/s  no-list {v1['rows_without_list']}", flush=True)
        c.close()
