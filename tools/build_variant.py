#!/usr/bin/env python3
"""Dev aid for A/B kernel experiments on the GPU box: link a VARIANT of libppcr_hip.so whose K1 translation unit for one
list width (default 10) is compiled with extra -D flags; every other object is the regular build's.
    tools/build_variant.py <name> [-DFOO=1 ...] [--width 10]
writes probabilistic_point_clouds_registration_amd/_variants/libppcr_hip_<name>.so (git-ignored, travels with gpurun);
select it with PPCR_HIP_LIB=<path> (read by _lib.py: experiments only — the product loads libppcr_hip.so)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import build  # noqa: E402

name = sys.argv[1]
args = sys.argv[2:]
width = 10
if "--width" in args:
    k = args.index("--width")
    width = int(args[k + 1])
    del args[k:k + 2]
build.build()
vdir = os.path.join(build.PKG, "_variants")
os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, f"tile_m{width}_{name}.o")
subprocess.check_call([build.hipcc()] + build.flags() + ["-c", f"-DPPCR_TILE_M={width}"] + args + [build.TILE_TU, "-o", obj])
objs = [os.path.join(build.OBJ, "ppcr_hip.o"), os.path.join(build.OBJ, "ppcr_comm.o")] + [
    obj if m == width else os.path.join(build.OBJ, f"ppcr_nn_tile_m{m}.o") for m in build.TILE_WIDTHS]
out = os.path.join(vdir, f"libppcr_hip_{name}.so")
subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
print(out)
