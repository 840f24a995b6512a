import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import binding as po
from probabilistic_point_clouds_registration_amd import _lib, synth
rng = np.random.default_rng(5)
# batch stress: 64 pairs of different sizes, several lane counts, results must be bit-identical across lane counts
pairs = [synth.make_pair(int(rng.integers(2000, 40000)), cfg=5, pair=p, stride=3)[:2] for p in range(64)]
base = None
for lanes in (1, 4, 8, 16):
    t0 = time.perf_counter()
    T, done = _lib.batch_run(pairs, radius=1.0, max_neighbours=10, dof=5.0, n_iter=8, device_ids=(0,), lanes_per_device=lanes)
    dt = time.perf_counter() - t0
    if base is None: base = T
    print("lanes", lanes, "time", round(dt, 3), "identical", np.array_equal(T, base), "done", set(done.tolist()), flush=True)
# voxel + 1-NN random soak
bad = 0
for k in range(40):
    n = int(rng.integers(1, 60000)); leaf = float(rng.uniform(0.05, 5.0))
    a = (rng.normal(0, rng.uniform(0.5, 30), size=(n, 3)) + rng.uniform(-100, 100, 3)).astype(np.float32)
    if k % 5 == 0: a[rng.integers(0, n, size=max(1, n // 50))] = np.nan
    g, o = _lib.voxel_filter(a, leaf), po.voxel_filter(a, leaf)
    if g.shape != o.shape or not np.array_equal(g, o, equal_nan=True): bad += 1; print("voxel mismatch", k, n, leaf)
    if k % 2 == 0 and n < 20000:
        fin = a[np.isfinite(a).all(1)]
        if len(fin) > 1:
            q = (fin[rng.integers(0, len(fin), size=2000)] + rng.normal(0, 1.0, size=(2000, 3))).astype(np.float32)
            if not np.array_equal(_lib.nearest_sq_distances(q, fin), po.nearest_sq_distances(q, fin)): bad += 1; print("nn1 mismatch", k)
print("mismatches:", bad)
