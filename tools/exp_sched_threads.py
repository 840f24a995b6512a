#!/usr/bin/env python3
"""Dev experiment: is the one-thread scheduler of ppcr_align_many the limit for many small pairs?  Splits the resident
pairs over T Python threads (ctypes releases the GIL), each calling ppcr_align_many with lanes / T pairs in flight."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctxs = []
for p in range(npairs):
    s, t, _, _ = synth.make_pair(n, cfg=5, pair=p)
    c = _lib.Context(0)
    c.set_params(1.0, 10, 5.0, 3); c.set_target(t); c.set_source(s)
    c.align(3, want_history=False)
    ctxs.append(c)
for threads, lanes in ((1, 8), (2, 4), (2, 8), (4, 2), (4, 4), (1, 8)):
    best = 0
    for rep in range(3):
        parts = [ctxs[k::threads] for k in range(threads)]
        ths = [threading.Thread(target=_lib.align_many, args=(part, 20), kwargs={"lanes": lanes}) for part in parts]
        t0 = time.perf_counter()
        for th in ths: th.start()
        for th in ths: th.join()
        best = max(best, npairs * 20 / (time.perf_counter() - t0))
    print(f"threads={threads} lanes/thread={lanes}: {best:8.0f} it/s aggregate (best of 3)", flush=True)
