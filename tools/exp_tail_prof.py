import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from probabilistic_point_clouds_registration_amd import _lib, synth
n = int(sys.argv[1]); ft = int(sys.argv[2])
src, tgt, _, _ = synth.make_pair(n, cfg=3 if n >= 1_000_000 else 2)
c = _lib.Context(0)
c.set_option("fold_tail", ft)
c.set_params(1.0, 10, 5.0, 3)
c.set_target(tgt)
c.set_source(src)
c.align(30, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
c.synchronize()
t0 = time.perf_counter()
c.align(200, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
c.synchronize()
print("it/s", 200 / (time.perf_counter() - t0), c.debug_host_figures())
