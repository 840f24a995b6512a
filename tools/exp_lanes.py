#!/usr/bin/env python3
"""Dev experiment: per-kernel durations and wall time when several resident pairs are aligned concurrently."""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from probabilistic_point_clouds_registration_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
opts = [o.split("=") for o in sys.argv[5:]]
opts0 = list(opts)
opts = [o for o in opts if o[0] != "noprof"]
if any(k == "torch" for k, _ in opts):          # same process state as bench.py
    import torch
    torch.cuda.synchronize()
    opts = [o for o in opts if o[0] != "torch"]
ctxs = []
for p in range(npairs):
    s, t, _, _ = synth.make_pair(n, cfg=3 if n >= 1000000 else 5, pair=p)
    c = _lib.Context(0)
    for k, v in opts:
        c.set_option(k, int(v))
    c.set_params(1.0, 10, 5.0, 3)
    c.set_target(t)
    c.set_source(s)
    c.align(3, want_history=False)
    ctxs.append(c)
for rep in range(reps):
    for prof in ((0,) if any(k == "noprof" for k, _ in opts0) else (0, 1)):
        for c in ctxs:
            c.profile_enable(bool(prof))
        t0 = time.perf_counter()
        _lib.align_many(ctxs, 20, lanes=lanes)
        dt = time.perf_counter() - t0
        line = f"rep {rep} prof={prof} lanes={lanes}: {npairs * 20 / dt:8.1f} it/s aggregate"
        import ctypes as C
        for c in ctxs:
            out = (C.c_double * 8)()
            c._L.ppcr_debug_get_host_times.argtypes = [C.c_void_p, C.c_void_p]
            c._L.ppcr_debug_get_host_times(c._h, out)
            line += f" [it={out[0]:.0f} wait={out[1]*1e3:.1f}ms max={out[2]*1e3:.2f} busy={out[3]*1e3:.1f}ms max={out[4]*1e3:.2f} assoc={out[5]*1e3:.2f} red={out[6]*1e3:.2f} acc={out[7]*1e3:.2f}]"
        if prof:
            st = ctxs[0].profile_get()
            line += "  " + " ".join(f"{k.split('_kernel')[0]}={v['total_ms'] / v['launches'] * 1e3:.0f}us" for k, v in st.items())
        print(line, flush=True)
