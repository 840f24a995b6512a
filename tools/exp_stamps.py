#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of nn_fast_kernel (s_memtime stamps; never used in timed runs)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10
src, tgt, _, _ = synth.make_pair(n, cfg=3)
c = _lib.Context(0)
c.set_params(1.0, m, 5.0, 3); c.set_target(tgt); c.set_source(src)
c.associate(); c.synchronize()
c.set_option("stamps", 1)
L = _lib.load(); L.ppcr_debug_get_stamps.argtypes = [C.c_void_p, C.c_void_p]
names = ["bbox", "rowtable", "stage", "scan", "select", "emit+moments", "reduce/tail", "fallback"]
nw = (n + 255)//256*4
def show(tag):
    out = (C.c_ulonglong * 8)()
    assert L.ppcr_debug_get_stamps(c._h, out) == 0
    v = np.array(list(out), dtype=np.float64)
    tot = v[:8].sum()
    print(tag, " ".join(f"{names[k]}={v[k]/nw:.0f}" for k in range(8)), f"| ticks per wave: {tot/nw:.0f}")
def halo_stats(tag):
    """slots 6 / 7 of every wave's stamp record carry the block's halo size and shape (not cycle counts)"""
    nst = nw * 8
    h = np.zeros(nst, dtype=np.uint64)
    import ctypes
    L.ppcr_debug_get_stamps_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert L.ppcr_debug_get_stamps_raw(c._h, h.ctypes.data, nst) == 0
    h = h.reshape(-1, 8)[::4]          # one record per block (wave 0)
    tot, shape = h[:, 6].astype(np.int64), h[:, 7].astype(np.int64)
    q = np.percentile(tot, [50, 90, 99, 99.9, 100])
    ny, nz = shape >> 8, shape & 255
    print(tag, "halo candidates p50/p90/p99/p99.9/max:", " ".join(f"{v:.0f}" for v in q),
          f"| >1536: {(tot > 1536).mean():.4f} >1664: {(tot > 1664).mean():.4f} >1792: {(tot > 1792).mean():.4f} >2048: {(tot > 2048).mean():.4f} >2240: {(tot > 2240).mean():.5f}",
          f"| ny max {ny.max()} nz max {nz.max()} ny>8: {(ny > 8).mean():.4f} nz>8: {(nz > 8).mean():.4f} both: {((ny > 8) & (nz > 8)).mean():.5f}")
c.associate(); c.synchronize(); show("fresh    "); halo_stats("fresh    ")
for k in range(8):
    c.iterate(); c.synchronize()
    if k in (0, 3, 7):
        show(f"iterate {k}")
        halo_stats(f"iterate {k}")
