#!/usr/bin/env python3
"""The benchmark's timed windows in small: fresh source, `warm` untimed iterations, `steps` timed ones (ppcr_align, one
inner step), Verlet lists on and off, with the hand-over and rebuild counters.  usage: exp_verlet_bench.py [n] [key=value ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
opts = [kv.split("=") for kv in sys.argv[2:]]
src, tgt, _, _ = synth.make_pair(n, cfg=3 if n >= 1_000_000 else 2)
for verlet in (0, 1):
    c = _lib.Context(0)
    c.set_option("verlet", verlet)
    for k, v in opts:
        c.set_option(k, int(v))
    c.set_params(1.0, 10, 5.0, 3)
    c.set_target(tgt)
    rates = []
    for w in range(4):
        c.set_source(src)
        c.align(5, cost_drop_thresh=0.0, inner_steps=1, want_history=False)
        c.synchronize()
        before = c.debug_verlet()["rebuilt"] if verlet else 0
        t0 = time.perf_counter()
        r = c.align(20, cost_drop_thresh=0.0, inner_steps=1, want_history=False)
        c.synchronize()
        rates.append(20 / (time.perf_counter() - t0))
    info = c.debug_verlet() if verlet else {}
    host = c.debug_host_times() if hasattr(c, "debug_host_times") else None
    print(f"verlet={verlet}: windows {[round(x) for x in rates]} it/s; rebuilt workgroups in the last window: {info.get('rebuilt', 0) - before} of 20 x {info.get('workgroups', 0)}; "
          f"mean list {info.get('mean_list', 0):.2f}; rows without list {info.get('rows_without_list', 0)}; host {host}", flush=True)
    c.profile_enable(True)
    c.set_source(src)
    c.align(25, cost_drop_thresh=0.0, inner_steps=1, want_history=False)
    st = c.profile_get()
    print("   ", "  ".join(f"{k} {1e3 * v['total_ms'] / v['launches']:.1f}us x{v['launches']}" for k, v in st.items() if k.startswith("nn_")), flush=True)
    c.close()
