"""CPU-side sanitizer runs (the reference has no sanitizer configuration; GPU ASan is not available on the
pool, so the host-side code is what can be checked): the oracle under AddressSanitizer + UBSan on its own golden tests,
and the PCD / LZF reader of the command line (csrc/cpp/src/pcd_io.cc) under both over a corpus of malformed files.
No GPU, nothing from /root/reference."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from test_cpp_cli import lzf_compress, write_pcd, write_pcd_compressed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "probabilistic_point_clouds_registration_amd", "csrc", "cpp")


def _gcc_lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


needs_asan = pytest.mark.skipif(shutil.which("gcc") is None or _gcc_lib("libasan.so") is None,
                                reason="gcc's libasan is not installed")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


@needs_asan
def test_oracle_under_asan_and_ubsan():
    """`make -C oracle asan`, then the oracle's own golden-vector tests in a child interpreter that loads THAT build
    (PPCR_ORACLE_LIB) with libasan preloaded: every radius search, weight update, solve and align of those tests runs
    instrumented; a report aborts the child."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    lib = os.path.join(ROOT, "oracle", "libppcr_oracle_asan.so")
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms, "the sanitizer build is not instrumented"
    env = dict(os.environ, LD_PRELOAD=_gcc_lib("libasan.so"), PPCR_ORACLE_LIB=lib, **SAN_ENV)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_vs_golden.py")],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert " passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def _header(fields="x y z", sizes="4 4 4", types="F F F", counts="1 1 1", width="5", height="1", points="5", data="ascii"):
    return (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS {fields}\nSIZE {sizes}\nTYPE {types}\n"
            f"COUNT {counts}\nWIDTH {width}\nHEIGHT {height}\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {points}\nDATA {data}\n").encode()


def _corpus(d):
    """valid files of the three encodings, then truncations, byte flips, hostile headers and hostile LZF streams"""
    rng = np.random.default_rng(20260404)
    pts = rng.normal(size=(300, 3)).astype(np.float32)
    valid = []
    for name, kw in (("ascii", dict(binary=False)), ("binary", dict(binary=True))):
        p = os.path.join(d, f"valid_{name}.pcd")
        write_pcd(p, pts, **kw)
        valid.append(p)
    for extra in (False, True):
        p = os.path.join(d, f"valid_compressed_{int(extra)}.pcd")
        write_pcd_compressed(p, pts, extra_field=extra)
        valid.append(p)
    files = list(valid)

    def emit(name, blob):
        p = os.path.join(d, name)
        open(p, "wb").write(blob)
        files.append(p)

    for v in valid:
        blob = open(v, "rb").read()
        stem = os.path.basename(v)[:-4]
        for k in range(1, 16):                                   # truncations, header and payload
            emit(f"{stem}_cut{k}.pcd", blob[:len(blob) * k // 16])
        for k in range(60):                                      # byte flips anywhere (header digits, sizes, LZF control bytes)
            b = bytearray(blob)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            emit(f"{stem}_flip{k}.pcd", bytes(b))
    row = b"1 2 3\n" * 5
    hostile = {
        "neg_points": _header(points="-5") + row,
        "huge_points": _header(width="99999999999", points="99999999999999999") + row,
        "overflow_wh": _header(width="4294967296", height="4294967296", points="-1").replace(b"POINTS -1\n", b"") + row,
        "zero_size": _header(sizes="0 4 4") + row,
        "neg_count": _header(counts="-1 1 1") + row,
        "giant_count": _header(counts="1048576 1048576 1048576", data="binary") + b"\0" * 64,
        "no_xyz": _header(fields="a b c") + row,
        "double_xyz": _header(sizes="8 8 8", types="F F F") + row,
        "int_xyz": _header(types="I I I") + row,
        "short_rows": _header() + b"1 2\n" * 5,
        "garbage_numbers": _header(width="abc", points="xyz") + row,
        "empty_tokens": b"FIELDS\nSIZE\nTYPE\nCOUNT\nWIDTH\nDATA ascii\n",
        "only_data": b"DATA binary\n" + b"\0" * 100,
        "no_data_line": _header().replace(b"DATA ascii\n", b""),
        "unknown_mode": _header(data="ebcdic") + row,
        "empty": b"",
        "many_fields": _header(fields=" ".join(["x", "y", "z"] + [f"f{k}" for k in range(400)]), sizes=" ".join(["4"] * 3 + ["8"] * 400),
                               types=" ".join(["F"] * 403), counts=" ".join(["1048576"] * 403), data="binary") + b"\0" * 4096,
        "nan_ascii": _header() + b"nan NaN inf\n1e999 -1e999 0x10\n" + b"1 2 3\n" * 3,
        "crlf": _header().replace(b"\n", b"\r\n") + row.replace(b"\n", b"\r\n"),
    }
    raw = np.ascontiguousarray(pts[:5].T).tobytes()              # struct of arrays, 5 points
    comp = lzf_compress(raw)
    hz = _header(data="binary_compressed")
    hostile.update({
        "lzf_ok": hz + struct.pack("<II", len(comp), len(raw)) + comp,
        "lzf_wrong_raw_size": hz + struct.pack("<II", len(comp), len(raw) + 4) + comp,
        "lzf_huge_comp_size": hz + struct.pack("<II", 0xFFFFFFF0, len(raw)) + comp,
        "lzf_zero_comp": hz + struct.pack("<II", 0, len(raw)),
        "lzf_backref_before_start": hz + struct.pack("<II", 3, len(raw)) + bytes([0xE0, 0xFF, 0xFF]),
        "lzf_literal_overrun": hz + struct.pack("<II", 2, len(raw)) + bytes([31, 1]),
        "lzf_long_match_overrun": hz + struct.pack("<II", 5, len(raw)) + bytes([0, 7, 0xE0, 0xFF, 0x00]),
        "lzf_truncated_sizes": hz + b"\x01\x02\x03",
    })
    for name, blob in hostile.items():
        emit(name + ".pcd", blob)
    return valid, files


@needs_asan
def test_pcd_reader_under_asan_and_ubsan(tmp_path):
    """The CLI's PCD / LZF reader, instrumented, over ~330 files: the valid ones must load and survive a binary round
    trip bit for bit; every malformed one must be REFUSED OR LOADED, never crash, overflow or allocate from a hostile
    header (the child aborts on the first sanitizer report)."""
    exe = str(tmp_path / "pcd_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-I", os.path.join(CPP, "include"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(CPP, "src", "pcd_io.cc"), os.path.join(ROOT, "tests", "cpp", "pcd_fuzz.cc"), "-o", exe])
    d = str(tmp_path / "corpus")
    os.makedirs(d)
    valid, files = _corpus(d)
    # an allocation above 1 GiB is a failure too (a hostile header must be refused before anything is sized by it)
    env = dict(os.environ, ASAN_OPTIONS=SAN_ENV["ASAN_OPTIONS"] + ":max_allocation_size_mb=1024:allocator_may_return_null=0",
               UBSAN_OPTIONS=SAN_ENV["UBSAN_OPTIONS"])
    r = subprocess.run([exe] + files, capture_output=True, text=True, errors="replace", env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    got = {ln.split(" ", 2)[2]: (int(ln.split()[0]), int(ln.split()[1])) for ln in r.stdout.splitlines() if ln[:1] in "0-"}
    assert len(got) == len(files)
    for v in valid:
        assert got[v] == (0, 300), (v, got[v])
    assert got[os.path.join(d, "lzf_ok.pcd")] == (0, 5)
    assert got[os.path.join(d, "neg_points.pcd")] == (0, 5)    # a negative POINTS falls back to WIDTH x HEIGHT
    for name in ("huge_points", "zero_size", "neg_count", "giant_count", "no_xyz", "double_xyz", "int_xyz", "short_rows",
                 "garbage_numbers", "empty_tokens", "only_data", "no_data_line", "unknown_mode", "empty", "many_fields",
                 "lzf_wrong_raw_size", "lzf_huge_comp_size", "lzf_zero_comp", "lzf_backref_before_start", "lzf_literal_overrun",
                 "lzf_long_match_overrun", "lzf_truncated_sizes"):
        assert got[os.path.join(d, name + ".pcd")][0] == -1, name
    assert got[os.path.join(d, "crlf.pcd")] == (0, 5)
    refused = sum(1 for rc, _ in got.values() if rc != 0)
    assert refused > 100   # most truncations and many flips must be refused, none may crash


def _tsan_ok():
    return shutil.which("g++") is not None and _gcc_lib("libtsan.so") is not None


@needs_asan
def test_device_pool_suballocator_under_asan_ubsan_and_tsan(tmp_path):
    """csrc/ppcr_pool.hpp — the sub-allocator every handle's device buffers are cut from — with a host stand-in for the
    driver: randomised alloc / free traffic (overlap shows as a wrong byte, a block outside its slab as a sanitizer
    report), coalescing back to whole slabs, trimming, a full device; then four threads on one pool under TSan (the
    preparing threads of ppcr_batch_run share the pool with their callers)."""
    src = os.path.join(ROOT, "tests", "cpp", "test_pool.cc")
    common = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
              "-I", os.path.join(ROOT, "probabilistic_point_clouds_registration_amd", "csrc"), src, "-pthread"]
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path / "pool_asan")
    subprocess.check_call(common + ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=600)
    assert r.returncode == 0 and "pool test: 0 failed" in r.stdout, (r.stdout + r.stderr)[-3000:]
    if _tsan_ok():
        exe = str(tmp_path / "pool_tsan")
        subprocess.check_call(common + ["-fsanitize=thread", "-DPOOL_TEST_SMALL_SLABS", "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"), timeout=600)
        assert r.returncode == 0 and "pool test: 0 failed" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-3000:]


@needs_asan
def test_batch_scheduler_threads_under_asan_and_tsan(tmp_path):
    """csrc/ppcr_batch_sched.hpp — the threads, hand-over queues and error latch of ppcr_batch_run, the very template the
    library instantiates with its handles — with a host stand-in for handles and registrations (tests/cpp/test_sched.cc):
    every pair prepared / run / retired exactly once on a handle nobody else holds, bounded handles and lanes, every handle
    released once, errors from each operation reported and the share wound down; four devices' shares side by side on one
    error latch.  Under ASan + UBSan, then under TSan."""
    src = os.path.join(ROOT, "tests", "cpp", "test_sched.cc")
    common = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-I",
              os.path.join(ROOT, "probabilistic_point_clouds_registration_amd", "csrc"), src, "-pthread"]
    exe = str(tmp_path / "sched_asan")
    subprocess.check_call(common + ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=600)
    assert r.returncode == 0 and "sched test: 0 failed" in r.stdout, (r.stdout + r.stderr)[-3000:]
    if _tsan_ok():
        exe = str(tmp_path / "sched_tsan")
        subprocess.check_call(common + ["-fsanitize=thread", "-o", exe])
        for _ in range(3):  # (thread interleavings differ from run to run)
            r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"), timeout=600)
            assert r.returncode == 0 and "sched test: 0 failed" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-3000:]

