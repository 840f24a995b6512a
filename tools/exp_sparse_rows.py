import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from probabilistic_point_clouds_registration_amd import _lib, synth
for cid in (9, 10, 8):
    cfg = synth.CONFIGS[cid]
    src, tgt, _, _ = synth.make_config(cid, pair=0)
    c = _lib.Context(0)
    c.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
    c.set_target(tgt); c.set_source(src)
    c.align(12, cost_drop_thresh=-1.0, inner_steps=1, want_history=False)
    c.synchronize()
    n = len(src)
    cnt = np.zeros(n, dtype=np.int32)
    L = _lib.load()
    L.ppcr_debug_read_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    assert L.ppcr_debug_read_buffer(c._h, b"cnt", cnt.ctypes.data, cnt.nbytes) == 0
    m = cfg["max_neighbours"]
    print("config", cid, "rows", n, "rows with fewer than m", int((cnt < m).sum()), "zero", int((cnt == 0).sum()), "hist", np.bincount(np.clip(cnt, 0, m))[:m:4])
