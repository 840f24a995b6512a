#!/usr/bin/env python3
"""Kernel-level experiments on the headline workload (run on the GPU box)."""
import argparse
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--m", type=int, nargs="+", default=[10])
ap.add_argument("--variants", type=int, nargs="+", default=[0])
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--dof", type=float, default=5.0)
ap.add_argument("--check", action="store_true")
ap.add_argument("--sort", type=int, default=1)
a = ap.parse_args()
src, tgt, _, _ = synth.make_pair(a.n, cfg=3)
for m in a.m:
    ref = None
    for var in a.variants:
        c = _lib.Context(0)
        c.set_option("nn_variant", var)
        c.set_option("sort_source", a.sort)
        c.set_params(1.0, m, a.dof, 3)
        c.set_target(tgt); c.set_source(src)
        c.associate(); c.synchronize()
        c.profile_enable(True)
        for _ in range(a.reps):
            c.associate()
            c.accumulate([1, 0, 0, 0], [0, 0, 0])
        st = c.profile_get()
        c.profile_enable(False)
        line = " ".join(f"{k}={v['total_ms']/v['launches']*1e3:.1f}us" for k, v in st.items())
        print(f"m={m} variant={var}: {line}", flush=True)
        if a.check:
            got = c.get_association()
            if ref is None:
                ref = got
            else:
                ok = all(np.array_equal(x, y) for x, y in zip(ref, got))
                print("   equal to first variant:", ok, flush=True)
        c.close()
