#!/usr/bin/env python3
"""Diagnostic: per-phase cycle totals of nn_fast_kernel (STAMPS builds) on the non-uniform scenes, with the multi-level
search and without, next to the uniform 200k cloud.  usage: exp_stamps_scene.py [m]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10
names = ["prologue", "rowtable", "stage", "scan", "select", "emit", "k23", "-"]
L = _lib.load()
L.ppcr_debug_get_stamps.argtypes = [C.c_void_p, C.c_void_p]
for scene in ("lidar", "slab", "uniform"):
    if scene == "uniform":
        src, tgt, _, _ = synth.make_pair(200_000, cfg=2, stride=3)
    else:
        src, tgt, _, _ = synth.make_scene(scene, 200_000, stride=3)
    for levels in (1, 0):
        c = _lib.Context(0)
        c.set_option("levels", -1 if levels else 0)
        c.set_params(3.0, m, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        c.align(6, inner_steps=1)
        c.synchronize()
        c.set_option("stamps", 1)
        acc = np.zeros(8)
        for _ in range(4):
            c.iterate()
            c.synchronize()
            out = (C.c_ulonglong * 8)()
            assert L.ppcr_debug_get_stamps(c._h, out) == 0
            acc += np.array(list(out), dtype=np.float64)
        acc /= 4
        n_lv = c.debug_levels()["levels"]
        waves = (len(src) + 255) // 256 * 4
        print(f"{scene:8s} m={m} levels={n_lv}: " + " ".join(f"{names[k]}={acc[k] / waves:7.0f}" for k in range(6)) +
              f" | ticks per block-wave {acc[:6].sum() / waves:8.0f}  short rows {c.debug_short_rows()}", flush=True)
        c.close()
