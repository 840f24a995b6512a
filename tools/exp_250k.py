#!/usr/bin/env python3
"""Diagnostic: per-iteration time of ONE 250k pair, alone and with other (idle) handles alive on the device."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000
def mk(p):
    s, t, _, _ = synth.make_pair(n, cfg=5, pair=p)
    c = _lib.Context(0); c.set_params(1.0, 10, 5.0, 3); c.set_target(t); c.set_source(s)
    return c, s
def rate(c, s, k=20, reps=5):
    out = []
    for _ in range(reps):
        c.set_source(s); c.align(3, inner_steps=1); c.synchronize()
        t0 = time.perf_counter(); c.align(k, inner_steps=1); c.synchronize()
        out.append((time.perf_counter() - t0) / k * 1e6)
    return np.median(out)
c0, s0 = mk(0)
print("alone: %.1f us/iteration (20), %.1f (100)" % (rate(c0, s0), rate(c0, s0, 100)))
c0.profile_enable(True); c0.set_source(s0); c0.align(3, inner_steps=1); c0.profile_get(); c0.profile_enable(True); c0.align(20, inner_steps=1)
print({k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in c0.profile_get().items()}); c0.profile_enable(False)
others = [mk(p) for p in range(1, 16)]
print("with 15 idle handles: %.1f us/iteration" % rate(c0, s0))
