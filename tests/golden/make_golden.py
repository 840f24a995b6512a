#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (dev tool, run in the build container).

The reference itself cannot be built or imported here (C++ only; PCL/FLANN/Ceres/Eigen are
absent), so these vectors come from an INDEPENDENT numpy/scipy restatement of the same
published semantics — scipy.spatial.cKDTree for the neighbour sets, numpy.linalg.svd for the
weighted Kabsch solve, scipy.special.logsumexp for the soft assignment — sharing no code with
oracle/ppcr_oracle.c or the HIP kernels.  They pin the oracle (tests/test_oracle_vs_golden.py)
and, through it and directly, the GPU path.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree
from scipy.special import logsumexp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from probabilistic_point_clouds_registration_amd import synth  # noqa: E402  (input generator only)


# --------------------------------------------------------------------------- numpy restatement
def nn_csr(src, tgt, radius, max_nn):
    """Radius search with the FLANN/PCL semantics (float d2 accumulated x,y,z; strict d2 < float(r*r);
    the max_nn closest, ties by target index; rows by ascending column)."""
    src = np.asarray(src, np.float32)[:, :3]
    tgt = np.asarray(tgt, np.float32)[:, :3]
    nt = tgt.shape[0]
    r2 = np.float32(float(radius) * float(radius))
    unbounded = max_nn <= 0 or max_nn >= nt
    tree = cKDTree(tgt.astype(np.float64))
    balls = tree.query_ball_point(src.astype(np.float64), r=float(radius) * (1 + 1e-5) + 1e-5)
    row_ptr = [0]
    cols, d2s = [], []
    for i, cand in enumerate(balls):
        cand = np.asarray(sorted(cand), dtype=np.int64)
        if cand.size:
            d = src[i][None, :] - tgt[cand]                 # float32
            dd = d[:, 0] * d[:, 0]
            dd = dd + d[:, 1] * d[:, 1]
            dd = dd + d[:, 2] * d[:, 2]
            keep = dd < r2
            cand, dd = cand[keep], dd[keep]
            if not unbounded and cand.size > max_nn:
                order = np.lexsort((cand, dd))[:max_nn]
                order = np.sort(order)                      # back to ascending column
                cand, dd = cand[order], dd[order]
            cols.append(cand.astype(np.int32))
            d2s.append(dd.astype(np.float32))
        row_ptr.append(row_ptr[-1] + (cand.size if cand.size else 0))
    col = np.concatenate(cols) if cols else np.zeros(0, np.int32)
    d2 = np.concatenate(d2s) if d2s else np.zeros(0, np.float32)
    return np.asarray(row_ptr, np.int32), col.astype(np.int32), d2.astype(np.float32)


def weights_rows(row_ptr, s, v, dim):
    w = np.zeros_like(s)
    for i in range(len(row_ptr) - 1):
        a, b = row_ptr[i], row_ptr[i + 1]
        if a == b:
            continue
        si = s[a:b]
        if np.isinf(v):
            lp = -si / 2
            w[a:b] = np.exp(lp - logsumexp(lp))
        else:
            lp = -(v + dim) / 2 * np.log1p(si / v)
            w[a:b] = np.exp(lp - logsumexp(lp)) * (v + dim) / (v + si)
    return w


def irls_step(src, tgt, row_ptr, col, R, t, v, dim):
    """weights at (R,t) -> (R_new, t_new, cost_at_old, cost_at_new_with_old_weights, w, s)"""
    x = np.repeat(src[:, :3].astype(np.float64), np.diff(row_ptr), axis=0)
    y = tgt[col, :3].astype(np.float64)
    res = y - (x @ R.T + t)
    s = (res ** 2).sum(1)
    w = weights_rows(row_ptr, s, v, dim)
    W = w.sum()
    if not W > 0:
        return np.eye(3), np.zeros(3), 0.0, 0.0, w, s
    mx = (w[:, None] * x).sum(0) / W
    my = (w[:, None] * y).sum(0) / W
    H = np.einsum("n,na,nb->ab", w, x - mx, y - my)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    Rn = Vt.T @ D @ U.T
    tn = my - Rn @ mx
    c_old = 0.5 * (w * s).sum()
    c_new = 0.5 * (w * ((y - (x @ Rn.T + tn)) ** 2).sum(1)).sum()
    return Rn, tn, c_old, c_new, w, s


def cost_floor(src, tgt, row_ptr, col, w):
    """rounding floor of a moment-based cost: 1e-14 * (Sxx + Syy) / 2 about the target bounding-box centre"""
    c = 0.5 * (tgt[:, :3].min(0).astype(np.float64) + tgt[:, :3].max(0).astype(np.float64))
    x = np.repeat(src[:, :3].astype(np.float64), np.diff(row_ptr), axis=0) - c
    y = tgt[col, :3].astype(np.float64) - c
    return 1e-14 * 0.5 * ((w * (x ** 2).sum(1)).sum() + (w * (y ** 2).sum(1)).sum())


def transform_inplace(cloud, R, t):
    p = cloud[:, :3].astype(np.float64)
    out = np.empty_like(p)
    for a in range(3):
        out[:, a] = ((R[a, 0] * p[:, 0] + R[a, 1] * p[:, 1]) + R[a, 2] * p[:, 2]) + t[a]
    cloud[:, :3] = out.astype(np.float32)


def align(src, tgt, radius, max_nn, v, n_iter, inner_steps, f_tol=1e-5, dim=3):
    src = src.copy()
    Tcum = np.eye(4)
    hist, costs, steps_l = [], [], []
    for _ in range(n_iter):
        row_ptr, col, _ = nn_csr(src, tgt, radius, max_nn)
        R, t = np.eye(3), np.zeros(3)
        steps = 0
        c0 = None
        while True:
            Rn, tn, c_old, c_new, w_, _ = irls_step(src, tgt, row_ptr, col, R, t, v, dim)
            if c0 is None:
                c0 = c_old
            steps += 1
            R, t = Rn, tn
            if steps >= inner_steps or (c_old - c_new) <= max(f_tol * c_old, cost_floor(src, tgt, row_ptr, col, w_)):
                break
        Tk = np.eye(4)
        Tk[:3, :3], Tk[:3, 3] = R, t
        Tcum = Tk @ Tcum
        hist.append(Tcum[:3, :4].copy())
        costs.append((c0, c_new))
        steps_l.append(steps)
        transform_inplace(src, R, t)
    return np.asarray(hist), np.asarray(costs), np.asarray(steps_l, np.int32), src


def voxel_grid(cloud, leaf):
    """pcl::VoxelGrid<PointXYZ> centroid down-sampling, numpy restatement (float32 throughout; the points of a voxel
    are added in ascending original index)."""
    a = np.asarray(cloud, np.float32)[:, :3]
    fin = np.isfinite(a).all(1)
    idx_pts = np.nonzero(fin)[0]
    a = a[fin]
    inv = np.float32(1.0) / np.float32(leaf)
    lo, hi = a.min(0), a.max(0)
    minb = np.floor(lo * inv).astype(np.int32)
    divb = np.floor(hi * inv).astype(np.int32) - minb + 1
    ijk = (np.floor(a * inv) - minb.astype(np.float32)).astype(np.int32)
    idx = ijk[:, 0] + ijk[:, 1] * divb[0] + ijk[:, 2] * divb[0] * divb[1]
    order = np.lexsort((idx_pts, idx))
    _, start, cnt = np.unique(idx[order], return_index=True, return_counts=True)
    out = np.zeros((len(start), 3), np.float32)
    for k, (s0, c) in enumerate(zip(start, cnt)):
        acc = np.zeros(3, np.float32)
        for j in order[s0:s0 + c]:
            acc = acc + a[j]
        out[k] = acc / np.float32(c)
    return out


# --------------------------------------------------------------------------- fixtures
def main():
    # 0. voxel-grid down-sampling (the step before the path)
    rng = np.random.default_rng(77)
    cloud = (rng.random((4000, 3)) * np.array([12, 9, 5]) - np.array([6, 2, 1])).astype(np.float32)
    cloud[::97, 2] = np.nan
    np.savez_compressed(os.path.join(HERE, "voxel_4k.npz"), cloud=cloud, leaf_1=voxel_grid(cloud, 1.0),
                        leaf_037=voxel_grid(cloud, 0.37))

    # 1. random 2k clouds at the benchmark density, m = 10 and m = 5, plus unbounded
    src, tgt, Rgt, tgt_t = synth.make_pair(2000, cfg=1, stride=3)
    out = dict(src=src, tgt=tgt, R_gt=Rgt, t_gt=tgt_t)
    for m in (10, 5, 0):
        rp, col, d2 = nn_csr(src, tgt, 1.0, m)
        out[f"row_ptr_m{m}"], out[f"col_m{m}"], out[f"d2_m{m}"] = rp, col, d2
    # weights + moments at a non-trivial theta for the m=10 association
    q = np.array([0.9999, 0.003, -0.002, 0.004])
    q = q / np.linalg.norm(q)
    R = synth.rodrigues(q[1:], 2 * np.arctan2(np.linalg.norm(q[1:]), q[0]))
    t = np.array([0.02, -0.01, 0.03])
    out["theta_q"], out["theta_t"] = q, t
    for name, v in (("t5", 5.0), ("gauss", np.inf)):
        rp, col = out["row_ptr_m10"], out["col_m10"]
        _, _, _, _, w, s = irls_step(src, tgt, rp, col, R, t, v, 3)
        out[f"w_{name}"], out[f"s_{name}"] = w, s
    np.savez_compressed(os.path.join(HERE, "nn_weights_2k.npz"), **out)

    # 2. regular grid with exact distance ties (the shape the reference's dead kd-tree test
    #    would have exercised: test/PointCloudRegistrationTest.cc:118-193, radius 3, m 5)
    g = synth.grid_test_cloud()
    Rz = synth.rodrigues([0, 0, 1], 0.10)
    gt = (g.astype(np.float64) @ Rz.T).astype(np.float32)
    out = dict(src=g, tgt=gt)
    for (r, m) in ((3.0, 5), (0.75, 4), (1.0, 0)):
        rp, col, d2 = nn_csr(g, gt, r, m)
        key = f"r{r}_m{m}"
        out[f"row_ptr_{key}"], out[f"col_{key}"], out[f"d2_{key}"] = rp, col, d2
    # ties with identical clouds: d2 == 0 for self, symmetric neighbours tie exactly
    rp, col, d2 = nn_csr(g, g, 0.75, 3)
    out["row_ptr_self"], out["col_self"], out["d2_self"] = rp, col, d2
    np.savez_compressed(os.path.join(HERE, "nn_grid_ties.npz"), **out)

    # 3. full-loop traces (per-iteration cumulative transforms) on the 2k pair
    src, tgt, Rgt, tgt_t = synth.make_pair(2000, cfg=1, stride=3)
    out = dict(src=src, tgt=tgt)
    for name, v, inner in (("t5_inner1", 5.0, 1), ("gauss_inner1", np.inf, 1), ("t5_conv", 5.0, 50)):
        hist, costs, steps, moved = align(src, tgt, 1.0, 10, v, 6, inner)
        out[f"hist_{name}"], out[f"costs_{name}"], out[f"steps_{name}"] = hist, costs, steps
        out[f"moved_{name}"] = moved
    np.savez_compressed(os.path.join(HERE, "align_trace_2k.npz"), **out)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
