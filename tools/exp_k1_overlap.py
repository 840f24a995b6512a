#!/usr/bin/env python3
"""Dev experiment: does K1 alone fill the GPU?  N handles each run associate() in a loop from their own thread."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctxs = []
for p in range(3):
    s, t, _, _ = synth.make_pair(n, cfg=3, pair=p)
    c = _lib.Context(0)
    c.set_params(1.0, 10, 5.0, 3)
    c.set_target(t)
    c.set_source(s)
    c.align(3, want_history=False)
    ctxs.append(c)


def loop(c, k):
    for _ in range(k):
        c.associate()
    c.synchronize()


for rep in range(3):
    for nthreads in (1, 2, 3):
        for c in ctxs:
            c.synchronize()
        th = [threading.Thread(target=loop, args=(ctxs[i], 30)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {nthreads} concurrent K1 loops: {dt / 30 * 1e6:.0f} us per round, {dt / 30 / nthreads * 1e6:.0f} us per K1", flush=True)
