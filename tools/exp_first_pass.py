#!/usr/bin/env python3
"""Diagnostic: the automatic first-pass radius of a two-pass search — how many times max_neighbours its sphere should
hold (first_pass_fill) and how full a cell may be (first_pass_occupancy) — on the three 200k clouds of exp_cli_shape.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib
import importlib.util
spec = importlib.util.spec_from_file_location("shape", os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp_cli_shape.py"))
src_text = open(spec.origin).read().split("for kind in (")[0]          # only the cloud generators of that script
ns = {"__file__": spec.origin}
exec(compile(src_text, spec.origin, "exec"), ns)
for kind in ("uniform", "scan", "slab"):
    src, tgt = ns["clouds"](kind)
    for m in (20, 10):
        for fill, occ in ((22, 110), (22, 85), (17, 85), (27, 110), (22, 140)):
            c = _lib.Context(0)
            c.set_option("first_pass_fill", fill)
            c.set_option("first_pass_occupancy", occ)
            c.set_params(3.0, m, 5.0, 3); c.set_target(tgt); c.set_source(src)
            c.align(3, inner_steps=1); c.synchronize()
            t0 = time.perf_counter(); c.align(15, cost_drop_thresh=-1.0, inner_steps=1); c.synchronize()
            dt = time.perf_counter() - t0
            c.profile_enable(True); c.align(5, cost_drop_thresh=-1.0, inner_steps=1)
            prof = {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in c.profile_get().items()}
            print(f"{kind} r=3 m={m} fill {fill / 10} cap {occ / 10}: {15 / dt:8.0f} it/s  handed over/it {c.debug_host_figures()[7] / 5:.0f} "
                  f"listed rows {c.debug_short_rows()}  K1 {prof.get('nn_fast_kernel')} wide {prof.get('nn_wide_kernel')}", flush=True)
            c.close()
