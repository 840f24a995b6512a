#!/usr/bin/env python3
"""A sparse source against a dense target (scan-to-map: 100k source rows sampled from a 1M-point target): every block of
256 queries spans a halo no tile holds, so the tiled kernel hands everything over.  How fast is that path?  (gpurun)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

rng = np.random.default_rng(5)
nt = 1_000_000
side = (nt / 10.0) ** (1 / 3)
tgt = rng.uniform(0, side, size=(nt, 3)).astype(np.float32)
for ns in (100_000, 250_000, 1_000_000):
    src = (tgt[rng.permutation(nt)[:ns]] + rng.normal(0, 0.02, size=(ns, 3)) + [0.05, -0.03, 0.02]).astype(np.float32)
    with _lib.Context(0) as c:
        c.set_params(1.0, 10, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        c.profile_enable(True)
        c.align(30, cost_drop_thresh=-1.0, inner_steps=1)
        c.synchronize()
        st = c.profile_get()
        c.profile_enable(False)
        t0 = time.perf_counter()
        c.align(60, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
        c.synchronize()
        dt = time.perf_counter() - t0
        print(f"ns {ns}: {60 / dt:9.1f} it/s; handed over (30 its) {c.debug_host_figures()[7]:.0f}, short rows last {c.debug_short_rows()};",
              {k: (v['launches'], round(v['total_ms'] / max(1, v['launches']) * 1e3, 1)) for k, v in st.items() if v['launches'] and k.startswith(('nn_', 'accum', 'reduce'))})
