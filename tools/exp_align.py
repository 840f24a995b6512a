#!/usr/bin/env python3
"""Profiling workload for rocprofv3 (kernel stats and --pmc passes): the kernels exactly as the benchmark's timed
windows launch them — ppcr_align with one inner step (nn_fast_kernel<...,8>: K23 folded in, fold-and-solve merged into
the cleanup launch), then the reference's schedule (inner loop to function_tolerance: the same K1 + inner_steps_kernel),
then the host-paced ppcr_iterate chain (nn_fast_kernel<...,-2> + accumulate_ell_kernel + reduce_solve_kernel), so that
every instantiation has its own rows in one pass.  usage: exp_align.py [n] [dof=<v|inf>] [key=value options ...]
(dof: the weight model, hence which fused form of K1 is profiled: 5 -> <..., 8>, inf -> <..., 0>, 3 / 10 -> <..., -3>)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg = 3 if n >= 1_000_000 else 2
src, tgt, _, _ = synth.make_pair(n, cfg=cfg)
c = _lib.Context(0)
dof = 5.0
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "dof":
        dof = float(v)
    else:
        c.set_option(k, int(v))
c.set_params(1.0, 10, dof, 3)
c.set_target(tgt)
for rep in range(2):
    c.set_source(src)
    c.align(3, cost_drop_thresh=0.0, inner_steps=1, want_history=False)     # cold start
    c.align(25, cost_drop_thresh=0.0, inner_steps=1, want_history=False)    # the timed windows' schedule
c.set_source(src)
c.align(3, cost_drop_thresh=0.0, inner_steps=100, f_tol=10e-6, want_history=False)
r = c.align(20, cost_drop_thresh=0.0, inner_steps=100, f_tol=10e-6)         # the reference's schedule
print("inner steps:", list(r["inner_steps"]), flush=True)
c.set_source(src)
for k in range(12):
    c.iterate()                                                              # host-paced chain: K1 alone
c.synchronize()
c.close()
