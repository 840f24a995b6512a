#!/usr/bin/env python3
"""Dev aid: a FULL variant of libppcr_hip.so — every translation unit compiled with extra -D flags (for knobs the host side
sees too, e.g. -DPPCR_VERLET_SLOTS=24).  usage: tools/build_full_variant.py <name> [-DFOO=1 ...]
writes probabilistic_point_clouds_registration_amd/_variants/libppcr_hip_<name>.so; select it with PPCR_HIP_LIB=<path>."""
import concurrent.futures
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilistic_point_clouds_registration_amd import build  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
vdir = os.path.join(build.PKG, "_variants")
odir = os.path.join(vdir, "obj_" + name)
os.makedirs(odir, exist_ok=True)
jobs = []
for obj, cmd in build._jobs():
    o = os.path.join(odir, os.path.basename(obj))
    jobs.append((o, cmd[:-2] + extra + ["-o", o]))


def run(job):
    subprocess.check_call(job[1])
    return job[0]


with concurrent.futures.ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
    objs = list(pool.map(run, jobs))
out = os.path.join(vdir, f"libppcr_hip_{name}.so")
subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
print(out)
