"""Is the split of rows between the tiled multi-level K1 and nn_wide_kernel the same in two runs of one registration?
(gpurun; round-5 experiment, docs/experiments.md: it is NOT on scenes with split blocks — the two halves of a split block
write one feedback word — while the association and every transform are: whoever answers a row finds the same neighbours,
and K23 runs over the finished association in a fixed order.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from probabilistic_point_clouds_registration_amd import _lib, synth


def run(src, tgt, m, inner, n_it=6):
    with _lib.Context(0) as c:
        c.set_option("level_stats", 1)
        c.set_params(3.0, m, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        res = c.align(n_it, cost_drop_thresh=-1.0, inner_steps=inner, f_tol=10e-6)
        return np.array(res["history"]), c.debug_levels(), c.debug_short_rows(), c.debug_host_figures()[7]


for name, m, inner in (("slab", 20, 100), ("slab", 10, 1), ("lidar", 20, 1)):
    src, tgt, _, _ = synth.make_scene(name, 100_000, stride=3)
    runs = [run(src, tgt, m, inner) for _ in range(3)]
    print(name, "m", m, "inner", inner, "history identical:", all(np.array_equal(runs[0][0], r[0]) for r in runs[1:]),
          "level counters identical:", all(runs[0][1] == r[1] for r in runs[1:]), "short", [r[2] for r in runs], "handed", [r[3] for r in runs])
