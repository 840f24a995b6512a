#!/usr/bin/env python3
"""Debug aid: rows answered from their lists at iteration k (vacc > 0) against brute force at their position."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
src, tgt, _, _ = synth.make_pair(n, cfg=2, stride=3)
ns, nt = src.shape[0], tgt.shape[0]


def state(iters):
    c = _lib.Context(0)
    c.set_params(1.0, 10, 5.0, 3)
    c.set_target(tgt)
    c.set_source(src)
    c.align(iters, cost_drop_thresh=0.0, inner_steps=1)
    d = dict(vl=c.debug_read("vl", np.int32, 16 * ns).reshape(16, ns), vn=c.debug_read("vn", np.uint8, ns),
             vg2=c.debug_read("vg2", np.float32, ns), vacc=c.debug_read("vacc", np.float32, ns),
             dm2=c.debug_read("dm2", np.uint32, ns).view(np.float32),
             nbr=c.debug_read("nbr", np.int32, 10 * ns).reshape(10, ns), cnt=c.debug_read("cnt", np.int32, ns),
             s=c.debug_read("src", np.float32, 4 * ns).reshape(ns, 4)[:, :3].copy(),
             t=c.debug_read("tgt", np.float32, 4 * nt).reshape(nt, 4)[:, :3].copy())
    c.close()
    return d


a, b = state(k - 1), state(k)
t = b["t"]
moved = np.linalg.norm(b["s"] - a["s"], axis=1)
print("moved: mean", moved.mean(), "max", moved.max())
rows = np.nonzero(b["vacc"] > 0)[0]
print("rows answered from lists:", len(rows))
bad = 0
for r in rows:
    d = t - b["s"][r]
    d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
    order = np.lexsort((np.arange(nt), d2))[:12]
    true_dm = d2[order[9]]
    true_set = sorted(int(x) for x in order[:10] if d2[x] < np.float32(1.0))
    dev_set = sorted(b["nbr"][:b["cnt"][r], r].tolist())
    if true_set != dev_set:
        bad += 1
        if bad <= 4:
            da = t - a["s"][r]
            d2a = ((da[:, 0] * da[:, 0] + da[:, 1] * da[:, 1]) + da[:, 2] * da[:, 2]).astype(np.float32)
            lst = a["vl"][:a["vn"][r], r]
            print(f"row {r}: moved {moved[r]:.5f} vacc_prev {a['vacc'][r]:.5f} vacc_now {b['vacc'][r]:.5f} G {np.sqrt(a['vg2'][r]):.5f} G_now {np.sqrt(b['vg2'][r]):.5f} "
                  f"dm_prev {np.sqrt(a['dm2'][r]):.5f} dm_now(dev) {np.sqrt(b['dm2'][r]):.5f} dm_now(true) {np.sqrt(true_dm):.5f}")
            print("    device:", dev_set)
            print("    true top-12 at new position:", order.tolist(), np.sqrt(d2[order]).round(5).tolist())
            print("    list (built earlier):", sorted(lst.tolist()), " distances from previous position:", np.sqrt(d2a[sorted(lst.tolist())]).round(5).tolist())
            missing = [int(x) for x in order[:10] if x not in set(lst.tolist())]
            print("    missing from the list:", missing, "their distance from the previous position:", np.sqrt(d2a[missing]).round(5).tolist())
print("rows with a wrong m-th distance:", bad)
