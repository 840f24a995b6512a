#!/usr/bin/env python3
"""Set-up cost of one pair (uploads + first association) with the early grid build on and off: wall times, median of 7."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

which = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = synth.CONFIGS[which]
src, tgt, _, _ = synth.make_config(which)
for eager in (1, 0, 1, 0):
    with _lib.Context(0) as sc:
        sc.set_option("eager_grid", eager)
        sc.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
        sc.set_target(tgt)
        sc.set_source(src)
        sc.associate()
        sc.synchronize()
        reps = []
        for _ in range(7):
            t0 = time.perf_counter()
            sc.set_target(tgt)
            t1 = time.perf_counter()
            sc.set_source(src)
            t2 = time.perf_counter()
            sc.associate()
            sc.synchronize()
            t3 = time.perf_counter()
            reps.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
        med = np.median(np.array(reps), axis=0) * 1e3
        n, nnz = sc.association_size()
        if eager and "-k" in sys.argv:
            sc.profile_enable(True)
            sc.set_target(tgt)
            sc.set_source(src)
            sc.associate()
            sc.synchronize()
            ks = sc.profile_get()
            sc.profile_enable(False)
            for name, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"]):
                print(f"    {name:28s} {v['total_ms']:8.3f} ms  x{v['launches']}")
        print(f"eager_grid {eager}: set_target {med[0]:.3f}  set_source {med[1]:.3f}  first associate {med[2]:.3f}  total {med[3]:.3f} ms   (rows {n}, pairs {nnz})")
