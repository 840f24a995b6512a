#!/usr/bin/env python3
"""How are the rows whose Verlet list runs out spread over waves and workgroups?  (gpurun; round-5 experiment)
For the bench's schedule (1M <-> 1M, a converged handle, the source uploaded again, k iterations): the test the kernel will
make at iteration k + 1 — (need + path) * 1.0001 squared against the list's reach — evaluated per row from the state after
iteration k and the move of iteration k + 1; then the share of rows / 64-row waves / 256-row workgroups that hold a failing row."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6, 10, 15, 20, 25]
src, tgt, _, _ = synth.make_pair(n, cfg=3 if n >= 500_000 else 2, stride=3)
ns = src.shape[0]


def state(iters):
    with _lib.Context(0) as c:
        c.set_params(1.0, 10, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        c.align(300, cost_drop_thresh=-1.0, inner_steps=1)      # settle
        c.set_source(src)                                        # the window's upload
        c.align(iters, cost_drop_thresh=-1.0, inner_steps=1)
        v = c.debug_verlet()
        return dict(vg2=c.debug_read("vg2", np.float32, ns), vacc=c.debug_read("vacc", np.float32, ns),
                    dm2=c.debug_read("dm2", np.uint32, ns), s=c.debug_read("src", np.float32, 4 * ns).reshape(ns, 4)[:, :3].copy(), v=v)


for k in ks:
    a, b = state(k), state(k + 1)
    moved = np.linalg.norm(b["s"].astype(np.float64) - a["s"], axis=1).astype(np.float32)
    has = a["dm2"] != 0xFFFFFFFF
    dm = np.sqrt(a["dm2"].view(np.float32), where=has, out=np.ones(ns, np.float32))
    need = np.where(has, np.minimum((dm + moved) * np.float32(1.000005), 1.0), 1.0).astype(np.float32)
    acc = a["vacc"] + moved
    reach = (need + acc) * np.float32(1.0001)
    ok = reach * reach < a["vg2"]
    fail = ~ok
    pad = (-ns) % 256
    f = np.concatenate([fail, np.zeros(pad, bool)])
    waves = f.reshape(-1, 64).any(axis=1)
    wgs = f.reshape(-1, 256).any(axis=1)
    per_wg = waves.reshape(-1, 4).sum(axis=1)
    room = np.sqrt(a["vg2"]) - need - a["vacc"]
    print(f"after {k} iterations (lists trusted: {a['v']['trusted']}, device says {b['v']['rebuilt'] - a['v']['rebuilt'] if False else '-'}): move of the next one mean {moved.mean():.2e} max {moved.max():.2e}; "
          f"failing rows {fail.mean() * 100:.3f} %, waves {waves.mean() * 100:.2f} %, workgroups {wgs.mean() * 100:.2f} %; "
          f"failing waves per failing workgroup {per_wg[wgs].mean() if wgs.any() else 0:.2f}; rows without a list {(a['vg2'] == 0).mean() * 100:.2f} %; "
          f"median room {np.median(room):.4f}, 1st percentile {np.percentile(room, 1):.4f}")
