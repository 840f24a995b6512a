#!/usr/bin/env python3
"""The command line's default shape (bench.py --config 8 / 9 / 10) through the benchmark's window schedule: iterations/s,
mean inner steps, rows left to the row-per-wave search, hand-overs, and the per-kernel HIP-event averages of a profiled
repetition.  usage: exp_cli_prof.py [configs=8,9,10] [key=value ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

cfgs, opts = [8, 9, 10], []
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "configs":
        cfgs = [int(x) for x in v.split(",")]
    else:
        opts.append((k, int(v)))
for cid in cfgs:
    cfg = synth.CONFIGS[cid]
    src, tgt, _, _ = synth.make_config(cid, pair=0)
    c = _lib.Context(0)
    for k, v in opts:
        c.set_option(k, v)
    c.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
    c.set_target(tgt)
    inner = int(cfg.get("inner_steps", 1))
    rates = []
    for w in range(4):
        c.set_source(src)
        c.align(5, cost_drop_thresh=-1.0, inner_steps=inner, want_history=False)
        c.synchronize()
        t0 = time.perf_counter()
        r = c.align(20, cost_drop_thresh=-1.0, inner_steps=inner)
        c.synchronize()
        rates.append(20 / (time.perf_counter() - t0))
    v = c.debug_verlet()
    c.profile_enable(True)
    c.set_source(src)
    c.align(5, cost_drop_thresh=-1.0, inner_steps=inner, want_history=False)
    base = {k: (x["total_ms"], x["launches"]) for k, x in c.profile_get().items()}
    c.align(20, cost_drop_thresh=-1.0, inner_steps=inner, want_history=False)
    prof = {}
    for k, x in c.profile_get().items():
        b = base.get(k, (0.0, 0))
        if x["launches"] > b[1]:
            prof[k] = f"{1e3 * (x['total_ms'] - b[0]) / 20:.1f}us/it x{(x['launches'] - b[1]) / 20:.1f}"
    c.profile_enable(False)
    print(f"config {cid}: window {np.median(rates):8.1f} it/s (min {min(rates):.0f} max {max(rates):.0f}) mean inner {np.mean(r['inner_steps']):.2f} "
          f"short rows {c.debug_short_rows()} handed/it {c.debug_host_figures()[7] / 20:.0f} reach {c.search_reach()} "
          f"lists trusted {int(v['trusted'])} rows {v['rows']} no-list {v['rows_without_list']} searched(last) {v['searched_last'][:3]}", flush=True)
    print("    " + "  ".join(f"{k} {x}" for k, x in prof.items()), flush=True)
    c.close()
