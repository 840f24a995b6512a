#!/usr/bin/env python3
"""Diagnostic: how much of the scan's trip count is lost to uneven run lengths inside a wave, and what the
alternatives discussed in DESIGN.md §4 could recover at best.

The STAMPS instantiation of nn_fast_kernel records every lane's nine sorted run lengths.  A run of L candidates takes
ceil(L / 2) trips; the wave pays, per rank, the maximum over its 64 lanes.  Reported per association:
  now        sum over ranks of (max over lanes)         — what the kernel does
  per-lane   max over lanes of (sum over ranks)         — one flattened loop per lane (run switches not priced)
  by-work    'now' after dealing the block's 256 queries to its four waves by descending total work
  ideal      total trips / 64                           — perfect balance across the wave
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
src, tgt, _, _ = synth.make_pair(n, cfg=cfg)
c = _lib.Context(0)
c.set_params(1.0, 10, 5.0, 3); c.set_target(tgt); c.set_source(src)
L = _lib.load()
L.ppcr_debug_get_stamps_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
nb = (n + 255) // 256
kMaxSplit = 64


def lane_records(grid_blocks):
    base = (grid_blocks * 4 + 64) * 8
    h = np.zeros(base + grid_blocks * 256, dtype=np.uint64)
    assert L.ppcr_debug_get_stamps_raw(c._h, h.ctypes.data, h.size) == 0
    pk = h[base:].reshape(grid_blocks, 256)
    runs = np.stack([(pk >> np.uint64(7 * k)) & np.uint64(127) for k in range(9)], axis=-1).astype(np.int64)
    return runs  # [block, lane, rank]


def report(tag, runs):
    trips = (runs + 1) // 2                                  # [B, 256, 9]
    w = trips.reshape(-1, 4, 64, 9)                          # waves
    now = w.max(axis=2).sum(axis=-1)                         # [B, 4]
    flat = w.sum(axis=-1).max(axis=2)
    ideal = w.sum(axis=(2, 3)) / 64.0
    tot = trips.sum(axis=-1)                                 # [B, 256]
    order = np.argsort(-tot, axis=1, kind="stable")
    dealt = np.take_along_axis(trips, order[:, :, None], axis=1).reshape(-1, 4, 64, 9)
    bywork = dealt.max(axis=2).sum(axis=-1)
    # by-work, and the lanes of each wave keep their ranks but the wave walks ranks of equal index together (same as now)
    # ranks walked in pairs (longest with shortest) as ONE loop each, a branch-free switch in the body
    pairs = [(0, 8), (1, 7), (2, 6), (3, 5)]
    paired = sum((w[..., a] + w[..., b]).max(axis=2) for a, b in pairs) + w[..., 4].max(axis=2)
    halves = (w[..., 0:9:2].sum(axis=-1)).max(axis=2) + (w[..., 1:9:2].sum(axis=-1)).max(axis=2)   # two flattened loops
    busy = now.sum()
    cand = runs.sum() / max((runs.sum(axis=-1) > 0).sum(), 1)
    print(f"{tag}: candidates/query {cand:.1f} | trips per wave: now {now.mean():.1f}  per-lane {flat.mean():.1f}  "
          f"by-work {bywork.mean():.1f}  ideal {ideal.mean():.1f} | lane utilisation of the scan now {ideal.sum() / busy:.3f}, "
          f"per-lane {ideal.sum() / flat.sum():.3f}, by-work {ideal.sum() / bywork.sum():.3f}")
    print(f"{tag}: paired ranks (0+8, 1+7, 2+6, 3+5, 4) {paired.mean():.1f} trips, utilisation {ideal.sum() / paired.sum():.3f}; "
          f"even / odd ranks as two loops {halves.mean():.1f} trips")
    live = (runs > 0).sum(axis=-1)
    print(f"{tag}: live runs per query {live[live > 0].mean():.2f}; rank maxima per wave "
          + " ".join(f"{v:.1f}" for v in w.max(axis=2).mean(axis=(0, 1))) + " | rank means "
          + " ".join(f"{v:.2f}" for v in w.mean(axis=(0, 1, 2))))


c.associate(); c.synchronize()
for k in range(6):
    c.iterate(); c.synchronize()
c.set_option("stamps", 1)
c.iterate(); c.synchronize()
report("steady (moving source)", lane_records(nb + kMaxSplit))
c.associate(); c.synchronize()
report("unmoved source        ", lane_records(nb + kMaxSplit))
