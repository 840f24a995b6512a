"""Build the HIP C-ABI library (libppcr_hip.so) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: neighbour membership is
decided by an uncontracted float d^2 (see csrc/ppcr_kernels.hip.h).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libppcr_hip.so")
SOURCES = [os.path.join(CSRC, "ppcr_hip.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "ppcr_kernels.hip.h"), os.path.join(CSRC, "ppcr_host_math.hpp"),
                  os.path.join(ROOT, "include", "ppcr.h")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")
    return exe


def flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
            "-Wall", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + flags() + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
