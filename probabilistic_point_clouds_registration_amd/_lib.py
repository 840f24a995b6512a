"""ctypes binding of the C ABI declared in include/ppcr.h (libppcr_hip.so, built by build.py).

The HIP library is the ONLY compute path: if it is missing or no GPU is visible, loading or
context creation raises — there is no CPU fallback in this package.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# PPCR_HIP_LIB: another BUILD of the same HIP library (tools/build_variant.py: A/B kernel experiments); unset = the product
LIB_PATH = os.environ.get("PPCR_HIP_LIB") or os.path.join(PKG, "libppcr_hip.so")
NSUMS = 19

# every symbol include/ppcr.h declares (tests check that the built library exports them all)
SYMBOLS = [
    "ppcr_abi_version", "ppcr_device_count", "ppcr_create", "ppcr_destroy", "ppcr_last_error",
    "ppcr_set_params", "ppcr_set_target", "ppcr_set_source", "ppcr_set_target_device",
    "ppcr_set_source_device", "ppcr_associate", "ppcr_association_size", "ppcr_get_association",
    "ppcr_set_association", "ppcr_weights", "ppcr_update_weights", "ppcr_accumulate", "ppcr_get_origin",
    "ppcr_solve_moments", "ppcr_cost_from_moments", "ppcr_solve", "ppcr_apply_transform",
    "ppcr_iterate", "ppcr_align", "ppcr_get_source", "ppcr_synchronize", "ppcr_profile_enable",
    "ppcr_profile_get", "ppcr_set_option", "ppcr_batch_run", "ppcr_align_many", "ppcr_set_companion",
    "ppcr_get_companion", "ppcr_set_ground_truth", "ppcr_mse_ground_truth", "ppcr_mse_previous", "ppcr_voxel_filter",
    "ppcr_nearest_sq_distances", "ppcr_stop_rule_check", "ppcr_align_report", "ppcr_batch_release",
    "ppcr_memory_stats", "ppcr_memory_trim",
    "ppcr_comm_get_id", "ppcr_comm_create", "ppcr_comm_destroy", "ppcr_gather_transforms", "ppcr_comm_last_error",
]


class StopRule(C.Structure):
    """ppcr_stop_rule: the hasConverged() state machine shared by ppcr_align and the C++ class."""
    _fields_ = [("iteration", C.c_int32), ("idle", C.c_int32), ("cost_drop", C.c_double)]

    def check(self, n_iter, cost_drop_thresh, n_cost_drop_it):
        """0 = continue, 1 = iteration cap reached, 2 = cost drop below the threshold for too long."""
        L = load()
        L.ppcr_stop_rule_check.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        return int(L.ppcr_stop_rule_check(C.byref(self), int(n_iter), float(cost_drop_thresh), float(n_cost_drop_it)))


class IterationInfo(C.Structure):
    """ppcr_iteration_info: what ppcr_align_report hands its callback once per outer iteration."""
    _fields_ = [("iteration", C.c_int32), ("inner_steps", C.c_int32), ("cost", C.c_double * 2),
                ("T_step", C.c_double * 12), ("T_cum", C.c_double * 12), ("mse_truth", C.c_double),
                ("moved", C.c_double)]


ITERATION_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(IterationInfo))
REPORT_TRUTH, REPORT_MOVED = 1, 2


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double)]


class Pair(C.Structure):
    _fields_ = [("source", C.c_void_p), ("n_source", C.c_int64), ("source_stride_bytes", C.c_int64),
                ("target", C.c_void_p), ("n_target", C.c_int64), ("target_stride_bytes", C.c_int64)]


class BatchOptions(C.Structure):
    _fields_ = [("radius", C.c_double), ("dof", C.c_double), ("cost_drop_thresh", C.c_double),
                ("n_cost_drop_it", C.c_double), ("f_tol", C.c_double), ("q0", C.c_double * 4), ("t0", C.c_double * 3),
                ("max_neighbours", C.c_int32), ("dim", C.c_int32), ("n_iter", C.c_int32), ("inner_steps", C.c_int32)]


class PpcrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ppcr error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libppcr_hip.so; raises ImportError loudly when the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m probabilistic_point_clouds_registration_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
    L.ppcr_abi_version.restype = i32
    L.ppcr_device_count.argtypes = [C.POINTER(i32)]
    L.ppcr_create.argtypes = [i32, C.POINTER(vp)]
    L.ppcr_destroy.argtypes = [vp]
    L.ppcr_last_error.argtypes = [vp]
    L.ppcr_last_error.restype = C.c_char_p
    L.ppcr_set_params.argtypes = [vp, dbl, i32, dbl, i32]
    for f in (L.ppcr_set_target, L.ppcr_set_source, L.ppcr_set_target_device, L.ppcr_set_source_device):
        f.argtypes = [vp, vp, i64, i64]
    L.ppcr_associate.argtypes = [vp]
    L.ppcr_association_size.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.ppcr_get_association.argtypes = [vp, vp, vp, vp]
    L.ppcr_set_association.argtypes = [vp, vp, vp, i64]
    L.ppcr_weights.argtypes = [vp, vp, vp, vp, vp]
    L.ppcr_accumulate.argtypes = [vp, vp, vp, vp]
    L.ppcr_update_weights.argtypes = [i32, vp, i64, vp, dbl, i32, vp]
    L.ppcr_get_origin.argtypes = [vp, vp]
    L.ppcr_solve_moments.argtypes = [vp, vp, vp, vp]
    L.ppcr_cost_from_moments.argtypes = [vp, vp, vp, vp]
    L.ppcr_cost_from_moments.restype = dbl
    L.ppcr_solve.argtypes = [vp, vp, vp, i32, dbl, vp, vp, C.POINTER(i32)]
    L.ppcr_apply_transform.argtypes = [vp, vp]
    L.ppcr_iterate.argtypes = [vp, vp, vp, i32, dbl, vp, vp, C.POINTER(i32)]
    L.ppcr_align.argtypes = [vp, i32, dbl, dbl, vp, vp, i32, dbl, vp, vp, vp, C.POINTER(i32)]
    L.ppcr_align_report.argtypes = [vp, i32, dbl, dbl, vp, vp, i32, dbl, vp, i32, ITERATION_FN, vp, vp, C.POINTER(i32)]
    L.ppcr_get_source.argtypes = [vp, vp, i64]
    L.ppcr_synchronize.argtypes = [vp]
    L.ppcr_profile_enable.argtypes = [vp, i32]
    L.ppcr_profile_get.argtypes = [vp, C.POINTER(KernelStat), i32, C.POINTER(i32)]
    L.ppcr_set_option.argtypes = [vp, C.c_char_p, i32]
    for f in (L.ppcr_set_companion, L.ppcr_set_ground_truth):
        f.argtypes = [vp, vp, i64, i64]
    L.ppcr_get_companion.argtypes = [vp, vp, i64]
    L.ppcr_mse_ground_truth.argtypes = [vp, C.POINTER(dbl)]
    L.ppcr_mse_previous.argtypes = [vp, C.POINTER(dbl)]
    L.ppcr_voxel_filter.argtypes = [i32, vp, i64, i64, C.c_float, vp, i64, C.POINTER(i64)]
    L.ppcr_nearest_sq_distances.argtypes = [i32, vp, i64, i64, vp, i64, i64, vp]
    L.ppcr_batch_run.argtypes = [C.POINTER(Pair), i64, C.POINTER(BatchOptions), C.POINTER(i32), i32, i32, vp, vp,
                                 C.c_char_p, i64]
    L.ppcr_align_many.argtypes = [C.POINTER(vp), i32, i32, i32, dbl, dbl, vp, vp, i32, dbl, vp, vp]
    L.ppcr_memory_stats.argtypes = [i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ppcr_memory_trim.argtypes = [i32]
    L.ppcr_comm_get_id.argtypes = [vp]
    L.ppcr_comm_create.argtypes = [i32, i32, i32, vp, C.POINTER(vp)]
    L.ppcr_comm_destroy.argtypes = [vp]
    L.ppcr_gather_transforms.argtypes = [vp, vp, i64, vp]
    L.ppcr_comm_last_error.restype = C.c_char_p
    for name in SYMBOLS:
        f = getattr(L, name)
        if name not in ("ppcr_last_error", "ppcr_cost_from_moments", "ppcr_comm_last_error"):
            f.restype = i32
    _lib = L
    return L


def device_count():
    n = C.c_int(0)
    load().ppcr_device_count(C.byref(n))
    return n.value


def _f64(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    assert a.shape[0] == n
    return a


class Context:
    """One device-resident source/target pair (thin, 1:1 over the C ABI)."""

    def __init__(self, device_id=0):
        self._L = load()
        h = C.c_void_p()
        rc = self._L.ppcr_create(int(device_id), C.byref(h))
        if rc != 0:
            raise PpcrError(rc, self._L.ppcr_last_error(None).decode())
        self._h = h
        self.ns = 0
        self.nt = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.ppcr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise PpcrError(rc, self._L.ppcr_last_error(self._h).decode())

    # -- parameters and clouds
    def set_params(self, radius, max_neighbours, dof=5.0, dim=3):
        self._ck(self._L.ppcr_set_params(self._h, float(radius), int(max_neighbours), float(dof), int(dim)))

    def set_option(self, key, value):
        self._ck(self._L.ppcr_set_option(self._h, key.encode(), int(value)))

    @staticmethod
    def _cloud(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] not in (3, 4):
            raise ValueError("cloud must be [n,3] or [n,4] float32")
        return a

    def set_target(self, xyz):
        a = self._cloud(xyz)
        self._ck(self._L.ppcr_set_target(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4))
        self.nt = a.shape[0]

    def set_source(self, xyz):
        a = self._cloud(xyz)
        self._ck(self._L.ppcr_set_source(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4))
        self.ns = a.shape[0]

    def set_target_device(self, ptr, n, stride_bytes):
        self._ck(self._L.ppcr_set_target_device(self._h, C.c_void_p(int(ptr)), int(n), int(stride_bytes)))
        self.nt = int(n)

    def set_source_device(self, ptr, n, stride_bytes):
        self._ck(self._L.ppcr_set_source_device(self._h, C.c_void_p(int(ptr)), int(n), int(stride_bytes)))
        self.ns = int(n)

    # -- association
    def associate(self):
        self._ck(self._L.ppcr_associate(self._h))

    def association_size(self):
        r, z = C.c_int64(0), C.c_int64(0)
        self._ck(self._L.ppcr_association_size(self._h, C.byref(r), C.byref(z)))
        return r.value, z.value

    def get_association(self, want_d2=True):
        rows, nnz = self.association_size()
        rp = np.zeros(rows + 1, dtype=np.int32)
        col = np.zeros(max(nnz, 1), dtype=np.int32)
        d2 = np.zeros(max(nnz, 1), dtype=np.float32) if want_d2 else None
        self._ck(self._L.ppcr_get_association(self._h, rp.ctypes.data, col.ctypes.data,
                                              d2.ctypes.data if want_d2 else None))
        return rp, col[:nnz], (d2[:nnz] if want_d2 else None)

    def set_association(self, row_ptr, col):
        rp = np.ascontiguousarray(row_ptr, dtype=np.int32)
        cl = np.ascontiguousarray(col, dtype=np.int32)
        self._ck(self._L.ppcr_set_association(self._h, rp.ctypes.data, cl.ctypes.data if cl.size else None,
                                              rp.shape[0] - 1))

    # -- weights / moments / solve
    def weights(self, q, t):
        _, nnz = self.association_size()
        w = np.zeros(max(nnz, 1))
        s = np.zeros(max(nnz, 1))
        q, t = _f64(q, 4), _f64(t, 3)
        self._ck(self._L.ppcr_weights(self._h, q.ctypes.data, t.ctypes.data, w.ctypes.data, s.ctypes.data))
        return w[:nnz], s[:nnz]

    def accumulate(self, q, t):
        sums = np.zeros(NSUMS)
        q, t = _f64(q, 4), _f64(t, 3)
        self._ck(self._L.ppcr_accumulate(self._h, q.ctypes.data, t.ctypes.data, sums.ctypes.data))
        return sums

    def origin(self):
        o = np.zeros(3)
        self._ck(self._L.ppcr_get_origin(self._h, o.ctypes.data))
        return o

    def solve(self, q0=(1, 0, 0, 0), t0=(0, 0, 0), max_steps=100, f_tol=1e-5):
        T, cost, st = np.zeros(12), np.zeros(2), C.c_int(0)
        q0, t0 = _f64(q0, 4), _f64(t0, 3)
        self._ck(self._L.ppcr_solve(self._h, q0.ctypes.data, t0.ctypes.data, int(max_steps), float(f_tol),
                                    T.ctypes.data, cost.ctypes.data, C.byref(st)))
        return T.reshape(3, 4), cost, st.value

    def apply_transform(self, T):
        T = _f64(np.asarray(T, dtype=np.float64)[:3, :4], 12)
        self._ck(self._L.ppcr_apply_transform(self._h, T.ctypes.data))

    def iterate(self, q0=(1, 0, 0, 0), t0=(0, 0, 0), inner_steps=1, f_tol=1e-5):
        T, cost, st = np.zeros(12), np.zeros(2), C.c_int(0)
        q0, t0 = _f64(q0, 4), _f64(t0, 3)
        self._ck(self._L.ppcr_iterate(self._h, q0.ctypes.data, t0.ctypes.data, int(inner_steps), float(f_tol),
                                      T.ctypes.data, cost.ctypes.data, C.byref(st)))
        return T.reshape(3, 4), cost, st.value

    def align(self, n_iter, cost_drop_thresh=0.0, n_cost_drop_it=5, q0=(1, 0, 0, 0), t0=(0, 0, 0),
              inner_steps=1, f_tol=1e-5, want_history=True):
        """n_iter < 0 = no iteration cap (the reference's meaning): only with want_history=False and
        cost_drop_thresh > 0 — the per-iteration arrays could not be sized and the loop would never end."""
        if int(n_iter) < 0 and want_history:
            raise ValueError("align(n_iter < 0) has no iteration cap: pass want_history=False (and cost_drop_thresh > 0)")
        k = max(int(n_iter), 1)
        hist = np.zeros(k * 12) if want_history else None
        costs = np.zeros(k * 2) if want_history else None
        steps = np.zeros(k, dtype=np.int32) if want_history else None
        done = C.c_int(0)
        q0, t0 = _f64(q0, 4), _f64(t0, 3)
        self._ck(self._L.ppcr_align(self._h, int(n_iter), float(cost_drop_thresh), float(n_cost_drop_it),
                                    q0.ctypes.data, t0.ctypes.data, int(inner_steps), float(f_tol),
                                    hist.ctypes.data if want_history else None,
                                    costs.ctypes.data if want_history else None,
                                    steps.ctypes.data if want_history else None, C.byref(done)))
        n = done.value
        if not want_history:
            return dict(n_iter=n)
        return dict(n_iter=n, history=hist[:12 * n].reshape(n, 3, 4).copy(), costs=costs[:2 * n].reshape(n, 2).copy(),
                    inner_steps=steps[:n].copy())

    def align_report(self, n_iter, cost_drop_thresh=0.0, n_cost_drop_it=5, q0=(1, 0, 0, 0), t0=(0, 0, 0),
                     inner_steps=1, f_tol=1e-5, report_truth=False, report_moved=False, rule=None, on_iteration=None):
        """ppcr_align_report: the device-paced loop with the per-iteration reports of the reference's align()
        (cc:114-129) delivered through a callback.  Returns the iterations as a list of dicts (+ the final cumulative
        transform and the stop rule's state); on_iteration(dict), when given, is also called as they arrive."""
        rows = []

        def _cb(_user, info_p):
            i = info_p.contents
            row = dict(iteration=i.iteration, inner_steps=i.inner_steps, cost=(i.cost[0], i.cost[1]),
                       T_step=np.array(i.T_step[:]).reshape(3, 4), T_cum=np.array(i.T_cum[:]).reshape(3, 4),
                       mse_truth=i.mse_truth, moved=i.moved)
            rows.append(row)
            if on_iteration is not None:
                on_iteration(row)

        cb = ITERATION_FN(_cb)
        rule = rule if rule is not None else StopRule(0, 0, 0.0)
        flags = (REPORT_TRUTH if report_truth else 0) | (REPORT_MOVED if report_moved else 0)
        T, done = np.zeros(12), C.c_int(0)
        q0, t0 = _f64(q0, 4), _f64(t0, 3)
        self._ck(self._L.ppcr_align_report(self._h, int(n_iter), float(cost_drop_thresh), float(n_cost_drop_it),
                                           q0.ctypes.data, t0.ctypes.data, int(inner_steps), float(f_tol),
                                           C.byref(rule), flags, cb, None, T.ctypes.data, C.byref(done)))
        return dict(n_iter=done.value, iterations=rows, T_final=T.reshape(3, 4), rule=rule)

    def get_source(self, stride=3):
        out = np.zeros((self.ns, stride), dtype=np.float32)
        self._ck(self._L.ppcr_get_source(self._h, out.ctypes.data, stride * 4))
        return out

    # -- reporting clouds (full-resolution companion, ground truth, previous-iteration snapshot)
    def set_companion(self, cloud):
        a = _cloud(cloud)
        self._keep_companion = a
        self._ck(self._L.ppcr_set_companion(self._h, a.ctypes.data if a.size else None, a.shape[0], a.shape[1] * 4))
        self.n_companion = a.shape[0]

    def get_companion(self, stride=3):
        out = np.zeros((self.n_companion, stride), dtype=np.float32)
        self._ck(self._L.ppcr_get_companion(self._h, out.ctypes.data, stride * 4))
        return out

    def set_ground_truth(self, cloud):
        a = _cloud(cloud)
        self._ck(self._L.ppcr_set_ground_truth(self._h, a.ctypes.data if a.size else None, a.shape[0], a.shape[1] * 4))

    def mse_ground_truth(self):
        v = C.c_double(0)
        self._ck(self._L.ppcr_mse_ground_truth(self._h, C.byref(v)))
        return v.value

    def mse_previous(self):
        v = C.c_double(0)
        self._ck(self._L.ppcr_mse_previous(self._h, C.byref(v)))
        return v.value

    def debug_host_figures(self):
        """ppcr_debug_get_host_times (diagnostic, not part of ppcr.h): 8 doubles about the last ppcr_align on this
        handle; [7] = workgroups its associations handed over to the cleanup kernel, summed over the iterations."""
        out = (C.c_double * 8)()
        f = self._L.ppcr_debug_get_host_times
        f.argtypes = [C.c_void_p, C.c_void_p]
        f.restype = C.c_int
        self._ck(f(self._h, out))
        return list(out)

    def search_reach(self):
        """ppcr_debug_get_search (diagnostic): cells per radius of the grid in use — 1: one-pass search, > 1: two passes."""
        out = (C.c_double * 2)()
        f = self._L.ppcr_debug_get_search
        f.argtypes = [C.c_void_p, C.c_void_p]
        f.restype = C.c_int
        self._ck(f(self._h, out))
        return int(out[0])

    def debug_levels(self):
        """ppcr_debug_get_levels (diagnostic; option level_stats): per level of a multi-level search
        {radius, blocks, handed_over_shape, handed_over_size, short_rows, staged, rows}, cumulative."""
        buf = (C.c_uint * 128)()
        f = self._L.ppcr_debug_get_levels
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        f.restype = C.c_int
        self._ck(f(self._h, buf, 128))
        n, base = int(buf[0]), int(buf[1])
        keys = ("blocks", "handed_over_shape", "handed_over_size", "short_rows", "staged", "rows")
        return dict(levels=n, base=base, per_level=[dict(radius=buf[2 + 7 * l] / 1000.0, **{k: int(buf[3 + 7 * l + j]) for j, k in enumerate(keys)})
                                                   for l in range(n)])

    def debug_short_rows(self):
        """ppcr_debug_get_short_rows (diagnostic): rows the first pass of the most recent two-pass association left short."""
        out = C.c_uint(0)
        f = self._L.ppcr_debug_get_short_rows
        f.argtypes = [C.c_void_p, C.c_void_p]
        f.restype = C.c_int
        self._ck(f(self._h, C.byref(out)))
        return int(out.value)

    def debug_verlet(self):
        """ppcr_debug_get_verlet (diagnostic): the steady-state Verlet lists as the last association left them."""
        buf = (C.c_longlong * 20)()
        f = self._L.ppcr_debug_get_verlet
        f.argtypes = [C.c_void_p, C.c_void_p]
        f.restype = C.c_int
        self._ck(f(self._h, buf))
        return dict(workgroups=int(buf[0]), rebuilt=int(buf[1]), rows=int(buf[2]), rows_without_list=int(buf[3]),
                    mean_list=(buf[4] / max(1, buf[2] - buf[3])), trusted=bool(buf[5]),
                    ordered=bool(buf[6]), launches=int(buf[7]), rows_rebuilt=int(buf[8]), workgroups_rebuilding_rows=int(buf[9]),
                    searched_last=[int(buf[10 + j]) for j in range(min(6, int(buf[7])))],
                    failing_rows_hist=[int(buf[16 + j]) for j in range(4)])

    def debug_read(self, name, dtype, count):
        """ppcr_debug_read_buffer (diagnostic): a raw copy of a device buffer of the handle."""
        out = np.zeros(count, dtype=dtype)
        f = self._L.ppcr_debug_read_buffer
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
        f.restype = C.c_int
        self._ck(f(self._h, name.encode(), out.ctypes.data, out.nbytes))
        return out

    def synchronize(self):
        self._ck(self._L.ppcr_synchronize(self._h))

    # -- profiling
    def profile_enable(self, on=True):
        self._ck(self._L.ppcr_profile_enable(self._h, 1 if on else 0))

    def profile_get(self):
        arr = (KernelStat * 32)()
        n = C.c_int(0)
        self._ck(self._L.ppcr_profile_get(self._h, arr, 32, C.byref(n)))
        return {arr[i].name.decode(): dict(launches=arr[i].launches, total_ms=arr[i].total_ms)
                for i in range(min(n.value, 32))}


def update_weights(row_ptr, sq_errors, dof, dim, device_id=0):
    """ProbabilisticWeights(dof, dim, .).updateWeights on caller-supplied squared errors (CSR order)."""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int32)
    s = np.ascontiguousarray(sq_errors, dtype=np.float64)
    w = np.zeros(max(1, s.shape[0]))
    L = load()
    rc = L.ppcr_update_weights(int(device_id), rp.ctypes.data, rp.shape[0] - 1, s.ctypes.data if s.size else None,
                               float(dof), int(dim), w.ctypes.data)
    if rc != 0:
        raise PpcrError(rc, L.ppcr_last_error(None).decode())
    return w[:s.shape[0]]


def solve_moments(sums, origin):
    R, t = np.zeros(9), np.zeros(3)
    s, o = _f64(sums, NSUMS), _f64(origin, 3)
    rc = load().ppcr_solve_moments(s.ctypes.data, o.ctypes.data, R.ctypes.data, t.ctypes.data)
    return R.reshape(3, 3), t, rc


def cost_from_moments(sums, origin, R, t):
    s, o, R, t = _f64(sums, NSUMS), _f64(origin, 3), _f64(R, 9), _f64(t, 3)
    return load().ppcr_cost_from_moments(s.ctypes.data, o.ctypes.data, R.ctypes.data, t.ctypes.data)


def _cloud(a):
    a = np.asarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] not in (3, 4):
        raise ValueError("clouds are float32 [n,3] or [n,4]")
    return np.ascontiguousarray(a)


def batch_run(pairs, radius, max_neighbours, dof=5.0, n_iter=20, cost_drop_thresh=0.0, n_cost_drop_it=5,
              inner_steps=1, f_tol=1e-5, q0=(1, 0, 0, 0), t0=(0, 0, 0), device_ids=(0,), lanes_per_device=2, dim=3):
    """ppcr_batch_run: pairs = [(src, tgt), ...] host arrays -> ([n,3,4] final transforms, [n] iterations done)."""
    L = load()
    n = len(pairs)
    keep = [(_cloud(s), _cloud(t)) for s, t in pairs]
    arr = (Pair * max(n, 1))()
    for k, (s, t) in enumerate(keep):
        arr[k] = Pair(s.ctypes.data, s.shape[0], s.shape[1] * 4, t.ctypes.data, t.shape[0], t.shape[1] * 4)
    opt = BatchOptions(float(radius), float(dof), float(cost_drop_thresh), float(n_cost_drop_it), float(f_tol),
                       (C.c_double * 4)(*[float(v) for v in q0]), (C.c_double * 3)(*[float(v) for v in t0]),
                       int(max_neighbours), int(dim), int(n_iter), int(inner_steps))
    devs = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
    T = np.zeros((n, 3, 4))
    done = np.zeros(n, dtype=np.int32)
    err = C.create_string_buffer(512)
    rc = L.ppcr_batch_run(arr, n, C.byref(opt), devs, len(device_ids), int(lanes_per_device), T.ctypes.data,
                          done.ctypes.data, err, 512)
    if rc != 0:
        raise PpcrError(rc, err.value.decode())
    return T, done


def memory_stats(device_id=0):
    """ppcr_memory_stats: the device pool the handles' buffers are cut from — bytes held from the driver, bytes in use,
    hipMalloc calls made so far."""
    r, u, n = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    rc = load().ppcr_memory_stats(int(device_id), C.byref(r), C.byref(u), C.byref(n))
    if rc != 0:
        raise PpcrError(rc, load().ppcr_last_error(None).decode())
    return {"reserved_bytes": r.value, "in_use_bytes": u.value, "driver_allocs": n.value}


def memory_trim(device_id=0):
    """ppcr_memory_trim: unused slabs (and the pinned blocks of destroyed handles) back to the driver."""
    rc = load().ppcr_memory_trim(int(device_id))
    if rc != 0:
        raise PpcrError(rc, load().ppcr_last_error(None).decode())


def batch_release():
    """ppcr_batch_release: frees the handles ppcr_batch_run keeps between calls."""
    rc = load().ppcr_batch_release()
    if rc != 0:
        raise PpcrError(rc, "ppcr_batch_release failed")


def align_many(ctxs, n_iter, lanes=2, cost_drop_thresh=0.0, n_cost_drop_it=5, q0=(1, 0, 0, 0), t0=(0, 0, 0),
               inner_steps=1, f_tol=1e-5):
    """ppcr_align_many over resident handles -> ([n,3,4] final transforms, [n] iterations done)."""
    L = load()
    n = len(ctxs)
    hs = (C.c_void_p * max(n, 1))(*[c._h for c in ctxs])
    T = np.zeros((n, 3, 4))
    done = np.zeros(n, dtype=np.int32)
    q0, t0 = _f64(q0, 4), _f64(t0, 3)
    rc = L.ppcr_align_many(hs, n, int(lanes), int(n_iter), float(cost_drop_thresh), float(n_cost_drop_it),
                           q0.ctypes.data, t0.ctypes.data, int(inner_steps), float(f_tol), T.ctypes.data,
                           done.ctypes.data)
    if rc != 0:
        msgs = [L.ppcr_last_error(c._h).decode() for c in ctxs]
        raise PpcrError(rc, next((m for m in msgs if m), "ppcr_align_many failed"))
    return T, done


class Comm:
    """ppcr_comm: the native RCCL communicator of the one-process-per-GPU deployment (ppcr_gather_transforms).
    Rank 0 draws the id (Comm.new_id()) and hands its 128 bytes to the other ranks; creation is collective."""

    @staticmethod
    def new_id():
        buf = (C.c_ubyte * 128)()
        rc = load().ppcr_comm_get_id(buf)
        if rc != 0:
            raise PpcrError(rc, load().ppcr_comm_last_error().decode())
        return bytes(buf)

    def __init__(self, device_id, rank, world, comm_id):
        self._L = load()
        h = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(comm_id)
        rc = self._L.ppcr_comm_create(int(device_id), int(rank), int(world), buf, C.byref(h))
        if rc != 0:
            raise PpcrError(rc, self._L.ppcr_comm_last_error().decode())
        self._h, self.rank, self.world = h, int(rank), int(world)

    def gather_transforms(self, T_local, n_pairs):
        """T_local: this rank's pairs (p % world == rank, ascending) as [k, 3, 4] -> every pair's transform [n_pairs, 3, 4]"""
        loc = np.ascontiguousarray(np.asarray(T_local, np.float64).reshape(-1, 12))
        out = np.zeros((int(n_pairs), 3, 4))
        rc = self._L.ppcr_gather_transforms(self._h, loc.ctypes.data if loc.size else None, int(n_pairs), out.ctypes.data)
        if rc != 0:
            raise PpcrError(rc, self._L.ppcr_comm_last_error().decode())
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.ppcr_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def voxel_filter(cloud, leaf, device_id=0):
    """ppcr_voxel_filter: pcl::VoxelGrid centroid down-sampling on the device -> float32 [k, 3]"""
    a = _cloud(cloud)
    out = np.zeros((max(a.shape[0], 1), 3), dtype=np.float32)
    k = C.c_int64(0)
    L = load()
    rc = L.ppcr_voxel_filter(int(device_id), a.ctypes.data if a.size else None, a.shape[0], a.shape[1] * 4, float(leaf),
                             out.ctypes.data, 12, C.byref(k))
    if rc != 0:
        raise PpcrError(rc, L.ppcr_last_error(None).decode())
    return out[:k.value].copy()


def nearest_sq_distances(queries, targets, device_id=0):
    """ppcr_nearest_sq_distances: exact 1-NN squared distances on the device -> float32 [nq]"""
    q, t = _cloud(queries), _cloud(targets)
    out = np.zeros(max(q.shape[0], 1), dtype=np.float32)
    L = load()
    rc = L.ppcr_nearest_sq_distances(int(device_id), q.ctypes.data if q.size else None, q.shape[0], q.shape[1] * 4,
                                     t.ctypes.data if t.size else None, t.shape[0], t.shape[1] * 4, out.ctypes.data)
    if rc != 0:
        raise PpcrError(rc, L.ppcr_last_error(None).decode())
    return out[:q.shape[0]].copy()
