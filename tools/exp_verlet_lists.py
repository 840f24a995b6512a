#!/usr/bin/env python3
"""Debug aid: after k iterations of ppcr_align, read the Verlet lists back and check them against brute force at the
source's position on the device: a row's list must hold exactly the targets with d2 <= vg2 of where the row was when the
list was built (rows with vacc = 0 were built in the last association).  usage: exp_verlet_lists.py [n] [k]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
src, tgt, _, _ = synth.make_pair(n, cfg=2, stride=3)
for k in [int(v) for v in sys.argv[2:]] or [2, 3]:
    c = _lib.Context(0)
    c.set_params(1.0, 10, 5.0, 3)
    c.set_target(tgt)
    c.set_source(src)
    c.align(k, cost_drop_thresh=0.0, inner_steps=1)
    ns, nt = src.shape[0], tgt.shape[0]
    vl = c.debug_read("vl", np.int32, 16 * ns).reshape(16, ns)
    vn = c.debug_read("vn", np.uint8, ns)
    vg2 = c.debug_read("vg2", np.float32, ns)
    vacc = c.debug_read("vacc", np.float32, ns)
    dm2 = c.debug_read("dm2", np.uint32, ns).view(np.float32)
    s = c.debug_read("src", np.float32, 4 * ns).reshape(ns, 4)[:, :3]
    t = c.debug_read("tgt", np.float32, 4 * nt).reshape(nt, 4)[:, :3]
    print(f"k={k}:", c.debug_verlet(), "vacc: zero rows", int((vacc == 0).sum()), "max", float(vacc.max()), flush=True)
    fresh = np.nonzero((vacc == 0) & (vg2 > 0))[0]
    bad = 0
    for r in fresh[:: max(1, len(fresh) // 3000)]:
        d = t - s[r]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        want = set(np.nonzero(d2.astype(np.float32).view(np.uint32) <= np.float32(vg2[r]).view(np.uint32))[0].tolist())
        got = set(vl[:vn[r], r].tolist())
        if want != got:
            bad += 1
            if bad <= 3:
                print("   row", r, "vg2", vg2[r], "dm", np.sqrt(dm2[r]), "list", sorted(got), "brute", sorted(want))
    print(f"   fresh lists checked: {len(fresh[:: max(1, len(fresh) // 3000)])}, wrong: {bad}", flush=True)
    c.close()
