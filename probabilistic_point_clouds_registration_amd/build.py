"""Build the HIP C-ABI library (libppcr_hip.so) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  -ffp-contract=off is load-bearing: neighbour membership is
decided by an uncontracted float d^2 (see csrc/ppcr_device.hip.h).
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libppcr_hip.so")
MAIN_TU = os.path.join(CSRC, "ppcr_hip.hip")
COMM_TU = os.path.join(CSRC, "ppcr_comm.hip")   # the native RCCL gather (host code only; RCCL bound at run time)
TILE_TU = os.path.join(CSRC, "ppcr_nn_tile.hip")
TILE_WIDTHS = (10, 4, 5, 8, 16, 20, 32)   # K1's compiled-in list widths, one object each (the default first)
SOURCES = [MAIN_TU, TILE_TU, COMM_TU]
DEPS = SOURCES + [os.path.join(CSRC, h) for h in ("ppcr_device.hip.h", "ppcr_kernels.hip.h", "ppcr_nn_tile.hip.h",
                                                  "ppcr_nn_tile_launch.hip.h", "ppcr_host_math.hpp", "ppcr_pool.hpp",
                                                  # the C-ABI unit ppcr_hip.hip in reading order
                                                  "ppcr_hip_setup.inc", "ppcr_hip_iteration.inc", "ppcr_hip_api.inc",
                                                  "ppcr_hip_align.inc", "ppcr_hip_extras.inc", "ppcr_hip_batch.inc")] + [
    os.path.join(ROOT, "include", "ppcr.h")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")
    return exe


def flags():
    # (PPCR_EXTRA_FLAGS: experiment builds only, e.g. tools/build_variant.py --all -DPPCR_VERLET_SLOTS=24)
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
            "-Wall", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + os.environ.get("PPCR_EXTRA_FLAGS", "").split()


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def _jobs():
    """(object, compile command) for every translation unit: the C-ABI unit and K1 once per list width."""
    cc = [hipcc()] + flags() + ["-c"]
    jobs = [(os.path.join(OBJ, "ppcr_hip.o"), cc + [MAIN_TU]), (os.path.join(OBJ, "ppcr_comm.o"), cc + [COMM_TU])]
    for m in TILE_WIDTHS:
        jobs.append((os.path.join(OBJ, "ppcr_nn_tile_m%d.o" % m), cc + ["-DPPCR_TILE_M=%d" % m, TILE_TU]))
    return [(obj, cmd + ["-o", obj]) for obj, cmd in jobs]


def build(force=False, verbose=False):
    """Compile the translation units in parallel (one hipcc per unit, as many at a time as there are cores) and link."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    jobs = _jobs()

    def compile_one(job):
        if verbose:
            print(" ".join(job[1]), file=sys.stderr)
        subprocess.check_call(job[1])
        return job[0]

    workers = max(1, min(len(jobs), os.cpu_count() or 1))
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        objects = list(pool.map(compile_one, jobs))
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objects + ["-ldl"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    return LIB


CPP = os.path.join(CSRC, "cpp")
CLI = os.path.join(PKG, "probabilistic_point_cloud_registration")   # the reference's executable name
CPP_TEST = os.path.join(PKG, "ppcr_cpp_api_test")
CPP_SOURCES = [os.path.join(CPP, "src", "prob_point_cloud_registration.cc"), os.path.join(CPP, "src", "pcd_io.cc")]


def build_host_programs(force=False, verbose=False):
    """C++ host layer above the C ABI: the CLI and the C++ API test program (plain g++, links libppcr_hip.so)."""
    build()
    common = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(CPP, "include")]
    link = ["-L", PKG, "-lppcr_hip", "-Wl,-rpath,$ORIGIN"]
    jobs = [(CLI, CPP_SOURCES + [os.path.join(CPP, "src", "prob_point_cloud_registration_ex.cc")]),
            (CPP_TEST, CPP_SOURCES + [os.path.join(ROOT, "tests", "cpp", "test_api.cc")])]
    for out, srcs in jobs:
        hdrs = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(CPP, "include")) for f in fs]
        stale = force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in srcs + hdrs + [LIB])
        if stale:
            cmd = common + srcs + link + ["-o", out]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    return CLI, CPP_TEST


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host_programs(force="--force" in sys.argv, verbose=True))
