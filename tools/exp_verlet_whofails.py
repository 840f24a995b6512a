#!/usr/bin/env python3
"""Which rows' Verlet lists are about to fail, and why: after `iters` iterations of a fresh registration the per-row list
state (reach, path travelled, length) and cut-off state are read back and the completeness test of the NEXT launch is
replayed on the host.  usage: exp_verlet_whofails.py [n] [iters] [key=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 9
opts = [kv.split("=") for kv in sys.argv[3:]]
radius, inner = 1.0, 1
if n in (8, 9, 10):
    cfg = synth.CONFIGS[n]
    src, tgt, _, _ = synth.make_config(n, pair=0)
    prm = (cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
    radius, inner = cfg["radius"], int(cfg.get("inner_steps", 1))
    n = src.shape[0]
else:
    src, tgt, _, _ = synth.make_pair(n, cfg=3 if n >= 1_000_000 else 2)
    prm = (1.0, 10, 5.0, 3)
c = _lib.Context(0)
for k, v in opts:
    c.set_option(k, int(v))
c.set_params(*prm)
c.set_target(tgt)
c.set_source(src)
rep = c.align_report(iters, cost_drop_thresh=-1.0, inner_steps=inner, f_tol=10e-6)
T = rep["iterations"][-1]["T_step"]
moved = float(np.linalg.norm(T[:, 3]))           # (the pending move: roughly every row's next displacement)
v = c.debug_verlet()
print(v)
g2 = c.debug_read("vg2", np.float32, n)
acc = c.debug_read("vacc", np.float32, n)
vn = c.debug_read("vn", np.uint8, n)
dm2 = c.debug_read("dm2", np.uint32, n)
has = dm2 != 0xFFFFFFFF
bound = np.sqrt(dm2.view(np.float32).astype(np.float64)) + moved
t2 = bound * bound * 1.00001
r2f = radius * radius
need = np.where(has & (t2 < r2f), np.sqrt(np.where(has, t2, r2f)), radius)
reach = (need + acc + moved) * 1.0001
fail = ~(reach * reach < g2)
G = np.sqrt(g2.astype(np.float64))
room = G - need - acc
slots = int(vn.max())
print(f"pending move {moved:.5f}; rows that will fail next: {int(fail.sum())} of {n}; max list length {slots}")
for name, m in (("failing", fail), ("all", np.ones(n, bool))):
    print(f"  {name:8s}: full lists {float((vn[m] == slots).mean()):.3f}  no cut-off {float((~has[m]).mean()):.3f}  need==radius {float((need[m] >= radius).mean()):.3f}  "
          f"mean G {G[m].mean():.4f}  mean need {need[m].mean():.4f}  mean acc {acc[m].mean():.4f}  mean G - need(at the build ~ skin) {float((G[m] - need[m]).mean()):.4f}")
print("  room (G - need - acc) percentiles over all rows [0.1 1 5 25 50]:", np.percentile(room, [0.1, 1, 5, 25, 50]).round(4).tolist())
if fail.any():
    idx = np.flatnonzero(fail)[:12]
    for r in idx:
        print(f"    row {r}: G {G[r]:.4f} need {need[r]:.4f} acc {acc[r]:.5f} vn {vn[r]} dm2 {'none' if not has[r] else round(float(np.sqrt(dm2.view(np.float32)[r])), 4)}")
blk = fail.reshape(-1, 256) if n % 256 == 0 else fail[: n // 256 * 256].reshape(-1, 256)
per = blk.sum(axis=1)
print("  blocks with a failing row:", int((per > 0).sum()), "of", len(per), " rows per such block: mean", float(per[per > 0].mean()) if (per > 0).any() else 0.0)
lens = np.bincount(vn, minlength=slots + 1)
print("  list length histogram:", lens.tolist())
c.close()
