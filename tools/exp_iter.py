#!/usr/bin/env python3
"""Per-iteration kernel times inside the align loop vs repeated association of a fixed source."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src, tgt, _, _ = synth.make_pair(n, cfg=3)
c = _lib.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split('=')
    c.set_option(k, int(v))
c.set_params(1.0, 10, 5.0, 3); c.set_target(tgt); c.set_source(src)
def show(tag):
    st = c.profile_get(); c.profile_enable(True)
    print(tag, " ".join(f"{k}={v['total_ms']/v['launches']*1e3:.1f}us" for k, v in st.items()), flush=True)
c.associate(); c.synchronize(); c.profile_enable(True)
for k in range(1):
    c.associate(); show(f"assoc-only {k}:")
for k in range(12):
    T, cost, st = c.iterate()
    if k < 8 or k == 11: show(f"iterate {k} (|t|={np.linalg.norm(T[:,3]):.4f}):")
    else: c.profile_get(); c.profile_enable(True)
