import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import binding as po
from probabilistic_point_clouds_registration_amd import _lib, synth
n = int(sys.argv[1])
src, tgt, _, _ = synth.make_pair(n, cfg=3)
with _lib.Context(0) as c:
    c.set_params(1.0, 10, 5.0, 3); c.set_target(tgt); c.set_source(src)
    t0 = time.perf_counter(); res = c.align(6, cost_drop_thresh=0.0, inner_steps=1); dt = time.perf_counter() - t0
    print(n, "align 6 it:", dt * 1e3, "ms")
    cur = c.get_source()
    c.associate()
    rp, col, d2 = c.get_association()
    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(n, size=800, replace=False))
    orp, ocol, od2 = po.radius_search(cur[pick], tgt, 1.0, 10, method=0)
    bad = 0
    for j, i in enumerate(pick):
        if not (np.array_equal(col[rp[i]:rp[i + 1]], ocol[orp[j]:orp[j + 1]]) and np.array_equal(d2[rp[i]:rp[i + 1]], od2[orp[j]:orp[j + 1]])):
            bad += 1
    print("sampled rows mismatching:", bad, "nnz/n", col.size / n)
    t0 = time.perf_counter(); c.align(20, cost_drop_thresh=0.0, inner_steps=1, want_history=False); dt = time.perf_counter() - t0
    print("it/s:", 20 / dt)
