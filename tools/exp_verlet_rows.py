#!/usr/bin/env python3
"""Iteration by iteration through the benchmark's window (fresh source, ppcr_align one iteration at a time): the size of
the move, workgroups that searched, rows rebuilt one by one and the workgroups that did so, failing rows per workgroup
(1 / 2-4 / 5-16 / more), K1's HIP-event time.  usage: exp_verlet_rows.py [n] [iterations] [key=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
opts = [kv.split("=") for kv in sys.argv[3:]]
inner = 1
if n in (8, 9, 10):      # (a config of synth.CONFIGS instead of a size: the command line's default shapes)
    cfg = synth.CONFIGS[n]
    src, tgt, _, _ = synth.make_config(n, pair=0)
    prm = (cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
    inner = int(cfg.get("inner_steps", 1))
else:
    scale = 1.0
    for kv in list(opts):
        if kv[0] == "traj":          # traj=3: bench.py's second trajectory (3 x motion, 3 x noise)
            scale = float(kv[1])
            opts.remove(kv)
    src, tgt, _, _ = synth.make_pair(n, cfg=3 if n >= 1_000_000 else 2, motion_scale=scale, noise_scale=scale)
    prm = (1.0, 10, 5.0, 3)
c = _lib.Context(0)
for k, v in opts:
    c.set_option(k, int(v))
c.set_params(*prm)
c.set_target(tgt)
c.set_source(src)
c.profile_enable(True)
prev = c.debug_verlet()
prev_t = {}
for it in range(iters):
    rep = c.align_report(1, cost_drop_thresh=-1.0, inner_steps=inner, f_tol=10e-6)
    T = rep["iterations"][0]["T_step"]
    v = c.debug_verlet()
    st = c.profile_get()
    k1 = {k: (x["total_ms"], x["launches"]) for k, x in st.items() if k.startswith("nn_") or n in (8, 9, 10)}
    dt = {k: 1e3 * (k1[k][0] - prev_t.get(k, (0, 0))[0]) for k in k1 if k1[k][1] > prev_t.get(k, (0, 0))[1]}
    prev_t = k1
    hist = [a - b for a, b in zip(v["failing_rows_hist"], prev["failing_rows_hist"])]
    print(f"it {it:2d} |t| {np.linalg.norm(T[:, 3]):.5f} trusted {int(v['trusted'])} searched {v['rebuilt'] - prev['rebuilt']:5d} "
          f"rows {v['rows_rebuilt'] - prev['rows_rebuilt']:6d} in {v['workgroups_rebuilding_rows'] - prev['workgroups_rebuilding_rows']:5d} wgs  "
          f"failing-rows hist {hist}  no-list {v['rows_without_list']} short {c.debug_short_rows()} handed {int(c.debug_host_figures()[7])}  " + "  ".join(f"{k.replace('_kernel', '')} {x:.1f}us" for k, x in dt.items()), flush=True)
    prev = v
c.close()
