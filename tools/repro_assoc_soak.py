#!/usr/bin/env python3
"""Replays tests/test_gpu_parity.py::test_randomised_association_soak (same random draws) and stops at the first association that
is not the oracle's — or whose size is absurd (rows left marked unsearched) —, printing the trial's parameters.
usage: repro_assoc_soak.py <seed> [trials] [key=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402
from oracle import binding as po  # noqa: E402  (a diagnostic, like the tests: the oracle is the checker)

seed = int(sys.argv[1])
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 24
extra = [kv.split("=") for kv in sys.argv[3:]]
rng = np.random.default_rng(seed)
for trial in range(trials):
    nt = int(rng.integers(200, 30000))
    ns = int(rng.integers(1, 12000))
    ext = rng.uniform(2.0, 40.0, size=3) * rng.choice([1.0, 0.05], size=3, p=[0.8, 0.2])
    off = rng.uniform(-500, 500, size=3) * rng.choice([0.0, 1.0])
    tgt = (rng.uniform(0, 1, size=(nt, 3)) * ext + off).astype(np.float32)
    if trial % 3 == 0:
        tgt = (np.round(tgt * 4) / 4).astype(np.float32)
    pick = rng.integers(0, nt, size=ns)
    src = (tgt[pick] + rng.normal(0, rng.choice([0.0, 0.02, 0.5]), size=(ns, 3))).astype(np.float32)
    radius = float(rng.uniform(0.2, 3.0))
    m = int(rng.choice([1, 2, 5, 10, 12, 16, 20, 32]))
    xf = int(rng.choice([1, 2, 4, 8]))
    with _lib.Context(0) as c:
        c.set_option("grid_xf", xf)
        c.set_option("defer_moves", trial % 2)
        for k_, v_ in extra:
            c.set_option(k_, int(v_))
        c.set_params(radius, m, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        cur = src.copy()
        for step in range(4 + 2 * (trial % 2)):
            c.associate()
            rows, nnz = c.association_size()
            tag = f"trial {trial} step {step}: nt {nt} ns {ns} ext {np.round(ext, 2)} r {radius:.3f} m {m} xf {xf} defer {trial % 2} reach {c.search_reach()} levels {c.debug_levels()['levels']}"
            if nnz > rows * max(m, 1) or nnz < 0:
                print("ABSURD SIZE", nnz, tag, "short rows seen", c.debug_short_rows() if hasattr(c, "debug_short_rows") else "?")
                sys.exit(1)
            rp, col, d2 = c.get_association()
            orp, ocol, od2 = po.radius_search(cur, tgt, radius, m, method=1)
            if not (np.array_equal(rp, orp) and np.array_equal(col, ocol) and np.array_equal(d2, od2)):
                print("MISMATCH", tag)
                sys.exit(1)
            if step == 1 and orp[-1] > 0:
                c.accumulate(np.array([0.999, 0.01, -0.02, 0.015]), np.array([0.01, 0.02, -0.015]))
            T = np.eye(4)
            T[:3, :3] = synth.rodrigues(rng.normal(size=3), float(rng.choice([0.0, 0.003, 0.05])))
            T[:3, 3] = rng.normal(0, float(rng.choice([0.0, 0.01, 0.3])) * radius, size=3)
            c.apply_transform(T)
            po.transform_cloud(cur, T)
    print("ok trial", trial, flush=True)
