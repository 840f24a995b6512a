#!/usr/bin/env python3
"""Diagnostic: pairs per second end to end (host buffers in: upload, grid build, source sort, 23 iterations) on one GPU
with 1 / 2 / 4 pairs in flight — ppcr_batch_run, and the same from Python threads with the two-pass machinery's occupancy
measurement of every new target switched on and off."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

pairs = [synth.make_pair(250_000, cfg=5, pair=p)[:2] for p in range(16)]
_lib.batch_run(pairs[:2], 1.0, 10, 5.0, n_iter=23, inner_steps=1, device_ids=(0,), lanes_per_device=2)
for rep in range(2):
    for lanes in (1, 2, 4):
        t0 = time.perf_counter()
        _lib.batch_run(pairs, 1.0, 10, 5.0, n_iter=23, inner_steps=1, device_ids=(0,), lanes_per_device=lanes)
        print(f"ppcr_batch_run lanes {lanes}: {len(pairs) / (time.perf_counter() - t0):7.1f} pairs/s", flush=True)
for two_pass, merge in ((1, 1), (1, 0), (0, 1)):
    for lanes in (1, 2, 4):
        ctxs = [_lib.Context(0) for _ in range(lanes)]
        for c in ctxs:
            c.set_option("two_pass", two_pass)
            c.set_option("merge_fold", merge)
            c.set_params(1.0, 10, 5.0, 3)
        def work(k):
            for p in range(k, len(pairs), lanes):
                ctxs[k].set_target(pairs[p][1]); ctxs[k].set_source(pairs[p][0]); ctxs[k].align(23, want_history=False)
        for k in range(lanes):
            ctxs[k].set_target(pairs[k][1]); ctxs[k].set_source(pairs[k][0]); ctxs[k].align(3, want_history=False)
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(lanes)]
        [t.start() for t in th]; [t.join() for t in th]
        print(f"python threads two_pass {two_pass} merge_fold {merge} lanes {lanes}: {len(pairs) / (time.perf_counter() - t0):7.1f} pairs/s", flush=True)
        [c.close() for c in ctxs]
