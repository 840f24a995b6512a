#!/usr/bin/env python3
"""Diagnostic: the command line's default shape (radius 3, max_neighbours 20, inner loop to function_tolerance) on a
non-uniform 200k cloud and on the uniform benchmark cloud: iterations/s and the per-kernel durations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

def clouds(kind):
    rng = np.random.default_rng(33)
    if kind == "slab":
        n = 200_000
        x = rng.beta(2.0, 5.0, size=n) * 160.0
        base = np.stack([x, rng.uniform(0, 80, n), rng.uniform(0, 12, n)], axis=1)
        blobs = np.concatenate([c + rng.normal(0, 1.5, size=(6000, 3)) for c in rng.uniform([20, 10, 2], [140, 70, 10], size=(5, 3))])
        tgt = np.concatenate([base[: n - len(blobs)], blobs]).astype(np.float32)
        Rg = synth.rodrigues([0.1, 0.3, 1.0], 0.01)
        src = ((tgt[rng.permutation(n)].astype(np.float64) - [0.3, -0.2, 0.1]) @ Rg + rng.normal(0, 0.02, size=(n, 3))).astype(np.float32)
        return src, tgt
    if kind == "scan":
        # a LiDAR-like scene: ground plane, four walls and a few boxes seen from a sensor at the origin — surfaces, with
        # the sampling density falling off with the square of the range
        n = 200_000
        az = rng.uniform(0, 2 * np.pi, n)
        el = np.radians(rng.uniform(-25, 3, n))
        d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
        rng_hit = np.full(n, 80.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(d[:, 2] < 0, -1.8 / d[:, 2], np.inf)                      # ground 1.8 m below the sensor
            rng_hit = np.minimum(rng_hit, tg)
            for nx, ny, off in ((1, 0, 30.0), (-1, 0, 22.0), (0, 1, 14.0), (0, -1, 40.0)):   # walls
                den = d[:, 0] * nx + d[:, 1] * ny
                tw = np.where(den > 1e-6, off / den, np.inf)
                rng_hit = np.minimum(rng_hit, tw)
        tgt = (d * rng_hit[:, None] + rng.normal(0, 0.02, size=(n, 3))).astype(np.float32)
        Rg = synth.rodrigues([0.0, 0.05, 1.0], 0.01)
        src = ((tgt[rng.permutation(n)].astype(np.float64) - [0.3, -0.2, 0.02]) @ Rg + rng.normal(0, 0.02, size=(n, 3))).astype(np.float32)
        return src, tgt
    s, t, _, _ = synth.make_pair(200_000, cfg=2, stride=3)
    return s, t

for kind in ("scan", "slab", "uniform"):
    src, tgt = clouds(kind)
    for (r, m, inner) in ((3.0, 20, 100), (3.0, 20, 1), (3.0, 10, 1), (1.0, 20, 1)):
        c = _lib.Context(0)
        c.set_params(r, m, 5.0, 3); c.set_target(tgt); c.set_source(src)
        c.align(3, inner_steps=inner, f_tol=10e-6); c.synchronize()
        t0 = time.perf_counter(); res = c.align(15, cost_drop_thresh=-1.0, inner_steps=inner, f_tol=10e-6); c.synchronize()
        dt = time.perf_counter() - t0
        c.profile_enable(True); c.align(5, cost_drop_thresh=-1.0, inner_steps=inner, f_tol=10e-6)
        prof = {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in c.profile_get().items()}
        nnz = c.association_size()[1]
        handed = c.debug_host_figures()[7] / 5
        short = c.debug_short_rows()
        print(f"{kind} r={r} m={m} inner<={inner}: {15 / dt:8.0f} it/s ({dt / 15 * 1e6:7.1f} us/iteration) nnz/row {nnz / len(src):5.1f} "
              f"mean inner {np.mean(res['inner_steps']):.1f} handed over/it {handed:.0f} short rows {short}  {prof}", flush=True)
        c.close()

# forced reach at the CLI default shape on the uniform cloud: which first-pass cell size is best
src, tgt = clouds("uniform")
for reach in (2, 3, 4):
    c = _lib.Context(0)
    c.set_option("two_pass", reach)
    c.set_params(3.0, 20, 5.0, 3); c.set_target(tgt); c.set_source(src)
    c.align(3, inner_steps=1); c.synchronize()
    t0 = time.perf_counter(); c.align(15, cost_drop_thresh=-1.0, inner_steps=1); c.synchronize()
    dt = time.perf_counter() - t0
    c.profile_enable(True); c.align(5, cost_drop_thresh=-1.0, inner_steps=1)
    prof = {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in c.profile_get().items()}
    print(f"uniform r=3 m=20 forced reach {reach}: {15 / dt:8.0f} it/s  handed over/it {c.debug_host_figures()[7] / 5:.0f}  {prof}", flush=True)
    c.close()
