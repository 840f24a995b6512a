"""ctypes binding of the C oracle (oracle/ppcr_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPCR_ORACLE_LIB: another build of the same source (the sanitizer build `make -C oracle asan`, tests/test_sanitizers.py)
_LIB_PATH = os.environ.get("PPCR_ORACLE_LIB") or os.path.join(_HERE, "libppcr_oracle.so")
NSUMS = 19

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with gcc (no reference sources involved)."""
    if os.environ.get("PPCR_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "ppcr_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.po_radius_search.restype = C.c_int64
        L.po_radius_search.argtypes = [_f32p, C.c_int64, C.c_int, _f32p, C.c_int64, C.c_int,
                                       C.c_double, C.c_int, C.c_int, C.c_int, _i32p, C.c_void_p,
                                       C.c_void_p, C.c_int64]
        L.po_squared_errors.restype = None
        L.po_squared_errors.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _i32p, _i32p, C.c_int64,
                                        _f64p, _f64p, _f64p]
        L.po_update_weights.restype = None
        L.po_update_weights.argtypes = [_i32p, C.c_int64, _f64p, C.c_double, C.c_int, _f64p]
        L.po_accumulate.restype = None
        L.po_accumulate.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _i32p, _i32p, C.c_int64, _f64p,
                                    _f64p, C.c_double, C.c_int, _f64p, C.c_int, _f64p]
        L.po_kabsch.restype = C.c_int
        L.po_kabsch.argtypes = [_f64p, _f64p, _f64p, _f64p]
        L.po_cost_from_sums.restype = C.c_double
        L.po_cost_from_sums.argtypes = [_f64p, _f64p, _f64p, _f64p]
        L.po_transform_cloud.restype = None
        L.po_transform_cloud.argtypes = [_f32p, C.c_int64, C.c_int, _f64p, C.c_int]
        L.po_solve.restype = C.c_int
        L.po_solve.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _i32p, _i32p, C.c_int64, _f64p, _f64p,
                               C.c_double, C.c_int, _f64p, C.c_int, C.c_double, C.c_int, _f64p,
                               _f64p, _f64p, C.POINTER(C.c_int)]
        L.po_align.restype = C.c_int
        L.po_align.argtypes = [_f32p, C.c_int64, C.c_int, _f32p, C.c_int64, C.c_int, C.c_double,
                               C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, _f64p,
                               _f64p, C.c_int, C.c_double, C.c_int, C.c_int, _f64p, _f64p, _i32p,
                               C.c_void_p]
        L.po_quat_to_R.restype = None
        L.po_quat_to_R.argtypes = [_f64p, _f64p]
        L.po_R_to_quat.restype = None
        L.po_R_to_quat.argtypes = [_f64p, _f64p]
        L.po_calculate_mse.restype = C.c_double
        L.po_calculate_mse.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_int64]
        L.po_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] in (3, 4)
    return a, int(a.shape[1])


def radius_search(src, tgt, radius, max_nn, method=1, threads=0, want_d2=True):
    """-> (row_ptr[ns+1], col[nnz], d2[nnz]) — CSR with ascending columns."""
    src, ss = _cloud(src)
    tgt, ts = _cloud(tgt)
    ns, nt = src.shape[0], tgt.shape[0]
    unbounded = max_nn <= 0 or max_nn >= nt
    cap = max(1, ns * (64 if unbounded else max_nn))
    row_ptr = np.zeros(ns + 1, dtype=np.int32)
    while True:
        col = np.zeros(cap, dtype=np.int32)
        d2 = np.zeros(cap, dtype=np.float32) if want_d2 else None
        nnz = lib().po_radius_search(src, ns, ss, tgt, nt, ts, float(radius), int(max_nn), int(method),
                                     int(threads), row_ptr, col.ctypes.data,
                                     d2.ctypes.data if want_d2 else None, cap)
        if nnz >= 0:
            break
        if nnz == -(2 ** 63):
            raise MemoryError("oracle radius_search allocation failure")
        cap = -nnz - 1
    return row_ptr, col[:nnz].copy(), (d2[:nnz].copy() if want_d2 else None)


def squared_errors(src, tgt, row_ptr, col, q, t):
    src, ss = _cloud(src)
    tgt, ts = _cloud(tgt)
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    col = np.ascontiguousarray(col, np.int32)
    s = np.zeros(max(1, col.shape[0]), dtype=np.float64)
    lib().po_squared_errors(src, ss, tgt, ts, row_ptr, _pad(col), src.shape[0],
                            np.asarray(q, np.float64), np.asarray(t, np.float64), s)
    return s[:col.shape[0]]


def _pad(a):
    return a if a.shape[0] > 0 else np.zeros(1, dtype=a.dtype)


def update_weights(row_ptr, s, v, dim):
    row_ptr = np.ascontiguousarray(row_ptr, np.int32)
    s = np.ascontiguousarray(s, np.float64)
    w = np.zeros(max(1, s.shape[0]), dtype=np.float64)
    lib().po_update_weights(row_ptr, row_ptr.shape[0] - 1, _pad(s), float(v), int(dim), w)
    return w[:s.shape[0]]


def accumulate(src, tgt, row_ptr, col, q, t, v, dim, origin, threads=0):
    src, ss = _cloud(src)
    tgt, ts = _cloud(tgt)
    sums = np.zeros(NSUMS, dtype=np.float64)
    lib().po_accumulate(src, ss, tgt, ts, np.ascontiguousarray(row_ptr, np.int32),
                        _pad(np.ascontiguousarray(col, np.int32)), src.shape[0],
                        np.asarray(q, np.float64), np.asarray(t, np.float64), float(v), int(dim),
                        np.asarray(origin, np.float64), int(threads), sums)
    return sums


def kabsch(sums, origin):
    R = np.zeros(9)
    t = np.zeros(3)
    rc = lib().po_kabsch(np.ascontiguousarray(sums, np.float64), np.asarray(origin, np.float64), R, t)
    return R.reshape(3, 3), t, rc


def cost_from_sums(sums, origin, R, t):
    return lib().po_cost_from_sums(np.ascontiguousarray(sums, np.float64),
                                   np.asarray(origin, np.float64),
                                   np.ascontiguousarray(R, np.float64).reshape(9),
                                   np.asarray(t, np.float64))


def transform_cloud(xyz, T, threads=0):
    """In place, like pcl::transformPointCloud(cloud, cloud, T). T: 3x4 or 4x4."""
    assert xyz.dtype == np.float32 and xyz.flags.c_contiguous
    T = np.ascontiguousarray(np.asarray(T, np.float64)[:3, :4]).reshape(12)
    lib().po_transform_cloud(xyz, xyz.shape[0], xyz.shape[1], T, int(threads))
    return xyz


def solve(src, tgt, row_ptr, col, v, dim, origin, q0=(1, 0, 0, 0), t0=(0, 0, 0), max_steps=100,
          f_tol=1e-5, threads=0):
    """-> (R 3x3, t, (initial_cost, final_cost), steps)"""
    src, ss = _cloud(src)
    tgt, ts = _cloud(tgt)
    R = np.zeros(9)
    t = np.zeros(3)
    cost = np.zeros(2)
    steps = C.c_int(0)
    lib().po_solve(src, ss, tgt, ts, np.ascontiguousarray(row_ptr, np.int32),
                   _pad(np.ascontiguousarray(col, np.int32)), src.shape[0],
                   np.asarray(q0, np.float64), np.asarray(t0, np.float64), float(v), int(dim),
                   np.asarray(origin, np.float64), int(max_steps), float(f_tol), int(threads), R, t,
                   cost, C.byref(steps))
    return R.reshape(3, 3), t, cost, steps.value


def align(src, tgt, radius, max_nn, dof, n_iter, cost_drop_thresh=0.0, n_cost_drop_it=5, dim=3,
          q0=(1, 0, 0, 0), t0=(0, 0, 0), inner_max_steps=1, f_tol=1e-5, nn_method=1, threads=0,
          return_source=False):
    """Full outer loop. -> dict(history [k,3,4], costs [k,2], inner_steps [k], n_iter, source)"""
    src, ss = _cloud(src)
    tgt, ts = _cloud(tgt)
    hist = np.zeros(max(1, n_iter) * 12)
    costs = np.zeros(max(1, n_iter) * 2)
    steps = np.zeros(max(1, n_iter), dtype=np.int32)
    out = np.zeros_like(src) if return_source else None
    k = lib().po_align(src, src.shape[0], ss, tgt, tgt.shape[0], ts, float(radius), int(max_nn),
                       float(dof), int(dim), int(n_iter), float(cost_drop_thresh),
                       float(n_cost_drop_it), np.asarray(q0, np.float64), np.asarray(t0, np.float64),
                       int(inner_max_steps), float(f_tol), int(nn_method), int(threads), hist, costs,
                       steps, out.ctypes.data if return_source else None)
    return dict(history=hist[:k * 12].reshape(k, 3, 4).copy(), costs=costs[:2 * k].reshape(k, 2).copy(),
                inner_steps=steps[:k].copy(), n_iter=k, source=out)


def quat_to_R(q):
    R = np.zeros(9)
    lib().po_quat_to_R(np.asarray(q, np.float64), R)
    return R.reshape(3, 3)


def R_to_quat(R):
    q = np.zeros(4)
    lib().po_R_to_quat(np.ascontiguousarray(R, np.float64).reshape(9), q)
    return q


def calculate_mse(a, b):
    a, sa = _cloud(a)
    b, sb = _cloud(b)
    return lib().po_calculate_mse(a, sa, b, sb, a.shape[0])


def voxel_filter(cloud, leaf):
    """pcl::VoxelGrid centroid down-sampling (restated) -> float32 [k, 3]"""
    a, sa = _cloud(cloud)
    out = np.zeros((max(a.shape[0], 1), 3), dtype=np.float32)
    L = lib()
    L.po_voxel_filter.restype = C.c_int64
    L.po_voxel_filter.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_int]
    k = L.po_voxel_filter(a.ctypes.data, sa, a.shape[0], float(leaf), out.ctypes.data, 3)
    if k < 0:
        raise ValueError("bad arguments")
    return out[:k].copy()


def nearest_sq_distances(q, t, threads=0):
    """exact 1-NN squared distances (brute force) -> float32 [nq]"""
    q, sq = _cloud(q)
    t, st = _cloud(t)
    out = np.zeros(q.shape[0], dtype=np.float32)
    L = lib()
    L.po_nearest_sq_distances.restype = None
    L.po_nearest_sq_distances.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int]
    L.po_nearest_sq_distances(q.ctypes.data, sq, q.shape[0], t.ctypes.data, st, t.shape[0], out.ctypes.data, int(threads))
    return out


def num_threads():
    return lib().po_num_threads()
