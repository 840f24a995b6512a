#!/usr/bin/env python3
"""Diagnostic: what one device-paced IRLS step costs.  Runs the reference's schedule (inner loop to function_tolerance) and
prints, per outer iteration, the inner step count next to the HIP-event duration of that iteration's inner_steps_kernel
launch (profiling on: one launch per profile_get), so that the launch cost at 0, 1, 2 ... device steps can be read off."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src, tgt, _, _ = synth.make_pair(n, cfg=3)
c = _lib.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("="); c.set_option(k, int(v))
c.set_params(1.0, 10, 5.0, 3); c.set_target(tgt); c.set_source(src)
c.align(2, inner_steps=100, f_tol=10e-6)
by_steps = {}
for it in range(24):
    c.profile_enable(True)
    r = c.align(1, cost_drop_thresh=-1.0, inner_steps=100, f_tol=10e-6)
    st = c.profile_get()
    c.profile_enable(False)
    k = int(r["inner_steps"][0])
    us = {name: v["total_ms"] * 1e3 / v["launches"] for name, v in st.items()}
    by_steps.setdefault(k, []).append(us)
for k in sorted(by_steps):
    rows = by_steps[k]
    names = sorted(rows[0])
    print(f"{k} inner step(s), {len(rows)} iterations:", "  ".join(f"{nm}={np.median([r.get(nm, 0) for r in rows]):.1f}us" for nm in names))
