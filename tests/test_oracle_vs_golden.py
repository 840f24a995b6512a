"""Oracle (oracle/ppcr_oracle.c) vs the committed golden fixtures (tests/golden/*.npz, produced by the
independent numpy/scipy restatement in tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import binding as po
from probabilistic_point_clouds_registration_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("m", [10, 5, 0])
def test_radius_search_random(method, m):
    g = load("nn_weights_2k.npz")
    rp, col, d2 = po.radius_search(g["src"], g["tgt"], 1.0, m, method=method)
    np.testing.assert_array_equal(rp, g[f"row_ptr_m{m}"])       # bit exact: index work
    np.testing.assert_array_equal(col, g[f"col_m{m}"])
    np.testing.assert_array_equal(d2, g[f"d2_m{m}"])            # float32 d2, same op order -> identical


@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("key,r,m", [("r3.0_m5", 3.0, 5), ("r0.75_m4", 0.75, 4), ("r1.0_m0", 1.0, 0)])
def test_radius_search_grid_ties(method, key, r, m):
    g = load("nn_grid_ties.npz")
    rp, col, d2 = po.radius_search(g["src"], g["tgt"], r, m, method=method)
    np.testing.assert_array_equal(rp, g[f"row_ptr_{key}"])
    np.testing.assert_array_equal(col, g[f"col_{key}"])
    np.testing.assert_array_equal(d2, g[f"d2_{key}"])


@pytest.mark.parametrize("method", [0, 1])
def test_radius_search_self_ties(method):
    g = load("nn_grid_ties.npz")
    rp, col, d2 = po.radius_search(g["src"], g["src"], 0.75, 3, method=method)
    np.testing.assert_array_equal(rp, g["row_ptr_self"])
    np.testing.assert_array_equal(col, g["col_self"])
    np.testing.assert_array_equal(d2, g["d2_self"])
    # every point finds itself at distance exactly zero
    n = g["src"].shape[0]
    for i in range(0, n, 97):
        assert i in col[rp[i]:rp[i + 1]]


def test_radius_search_edge_cases():
    tgt = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [5, 5, 5]], np.float32)
    src = np.array([[0, 0, 0], [10, 10, 10], [0.5, 0, 0]], np.float32)
    for method in (0, 1):
        # d2 == r2 is excluded (strict <): (1,0,0) is exactly at r = 1 from the origin query
        rp, col, d2 = po.radius_search(src, tgt, 1.0, 0, method=method)
        assert rp.tolist() == [0, 1, 1, 3] and col.tolist() == [0, 0, 1]
        # max_nn >= N_t  => unbounded (PCL: max_nn clipped to N then "return all")
        rp2, col2, _ = po.radius_search(src, tgt, 2.5, 4, method=method)
        rp3, col3, _ = po.radius_search(src, tgt, 2.5, 0, method=method)
        assert rp2.tolist() == rp3.tolist() and col2.tolist() == col3.tolist()
        # max_nn = 1 keeps the closest; tie (0.5 from both 0 and 1) resolved to the lower index
        rp4, col4, _ = po.radius_search(src, tgt, 2.5, 1, method=method)
        assert col4.tolist() == [0, 0]
        # empty source / empty target
        rp5, col5, _ = po.radius_search(np.zeros((0, 3), np.float32), tgt, 1.0, 3, method=method)
        assert rp5.tolist() == [0] and col5.size == 0
        rp6, col6, _ = po.radius_search(src, np.zeros((0, 3), np.float32), 1.0, 3, method=method)
        assert rp6.tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("name,v", [("t5", 5.0), ("gauss", float("inf"))])
def test_weights_and_errors_at_theta(name, v):
    g = load("nn_weights_2k.npz")
    rp, col = g["row_ptr_m10"], g["col_m10"]
    s = po.squared_errors(g["src"], g["tgt"], rp, col, g["theta_q"], g["theta_t"])
    np.testing.assert_allclose(s, g[f"s_{name}"], rtol=1e-11, atol=1e-15)
    w = po.update_weights(rp, s, v, 3)
    np.testing.assert_allclose(w, g[f"w_{name}"], rtol=1e-10, atol=1e-15)
    # moments agree with a direct numpy sum of the golden weights
    c = np.array([0.1, -0.2, 0.3])
    sums = po.accumulate(g["src"], g["tgt"], rp, col, g["theta_q"], g["theta_t"], v, 3, c)
    x = np.repeat(g["src"].astype(np.float64), np.diff(rp), axis=0) - c
    y = g["tgt"][col].astype(np.float64) - c
    wg = g[f"w_{name}"]
    exp = np.concatenate([[wg.sum()], (wg[:, None] * x).sum(0), (wg[:, None] * y).sum(0),
                          np.einsum("n,na,nb->ab", wg, x, y).reshape(9), [(wg * g[f"s_{name}"]).sum()],
                          [(wg * (x ** 2).sum(1)).sum()], [(wg * (y ** 2).sum(1)).sum()]])
    np.testing.assert_allclose(sums, exp, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("name,v,inner", [("t5_inner1", 5.0, 1), ("gauss_inner1", float("inf"), 1),
                                          ("t5_conv", 5.0, 50)])
def test_align_trace(name, v, inner):
    """Outer loop: per-iteration cumulative transforms within 1e-5 rad / 1e-5 m of the golden trace
    (north-star tolerance); in practice they agree to ~1e-12."""
    g = load("align_trace_2k.npz")
    res = po.align(g["src"], g["tgt"], 1.0, 10, v, 6, cost_drop_thresh=0.0, inner_max_steps=inner,
                   f_tol=1e-5, return_source=True)
    hist = g[f"hist_{name}"]
    assert res["n_iter"] == hist.shape[0] == 6                     # thresh 0 -> exactly n_iter
    np.testing.assert_array_equal(res["inner_steps"], g[f"steps_{name}"])
    for k in range(6):
        assert synth.rotation_angle(res["history"][k][:, :3], hist[k][:, :3]) < 1e-9
        assert np.linalg.norm(res["history"][k][:, 3] - hist[k][:, 3]) < 1e-9
    np.testing.assert_allclose(res["costs"], g[f"costs_{name}"], rtol=1e-7)
    # the moved source is re-rounded to float32 each iteration (SURVEY a-10): allow 1 ulp-level slack
    np.testing.assert_allclose(res["source"], g[f"moved_{name}"], rtol=0, atol=2e-6)
    # and the loop actually registers: final transform close to the generator's ground truth
    Rgt, tgt_t = synth.ground_truth(0)
    # (soft assignment converges slowly; only check that it moved most of the way from identity)
    assert synth.rotation_angle(res["history"][-1][:, :3], Rgt) < 0.5 * synth.GT_ANGLE
    assert np.linalg.norm(res["history"][-1][:, 3] - tgt_t) < 0.5 * np.linalg.norm(tgt_t)


def test_has_converged_rule():
    """src/prob_point_cloud_registration.cc:138-158: with the default thresholds the earliest stop is
    after 6 iterations when the cost never drops by >= 1 % (cost_drop_ starts at 0)."""
    g = load("align_trace_2k.npz")
    # a huge threshold makes every iteration 'unuseful'
    res = po.align(g["src"], g["tgt"], 1.0, 10, 5.0, 1000, cost_drop_thresh=2.0, n_cost_drop_it=5)
    assert res["n_iter"] == 6
    res = po.align(g["src"], g["tgt"], 1.0, 10, 5.0, 1000, cost_drop_thresh=2.0, n_cost_drop_it=2)
    assert res["n_iter"] == 3
    # n_iter caps
    res = po.align(g["src"], g["tgt"], 1.0, 10, 5.0, 4, cost_drop_thresh=2.0, n_cost_drop_it=5)
    assert res["n_iter"] == 4
    # degenerate association (no neighbour anywhere): cost_drop = 0/0 = NaN -> counter reset -> runs to n_iter
    far = g["src"] + np.float32(1000.0)
    res = po.align(far, g["tgt"], 1.0, 10, 5.0, 9, cost_drop_thresh=0.01, n_cost_drop_it=5)
    assert res["n_iter"] == 9
    assert np.allclose(res["history"][-1], np.eye(4)[:3])


def test_transform_cloud_rounding():
    # f64 math, f32 store, in place
    pts = np.array([[1.1, 2.2, 3.3], [-4.4, 5.5, -6.6]], np.float32)
    R = synth.rodrigues([0.3, -0.2, 0.9], 0.7)
    t = np.array([0.123456789, -9.87654321, 1e-3])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    exp = (pts.astype(np.float64) @ R.T + t).astype(np.float32)
    got = pts.copy()
    po.transform_cloud(got, T)
    assert np.max(np.abs(got.view(np.int32) - exp.view(np.int32))) <= 1   # same up to summation order
    # stride-4 clouds keep their padding lane untouched
    p4 = np.concatenate([pts, np.full((2, 1), 7.0, np.float32)], axis=1)
    po.transform_cloud(p4, T)
    np.testing.assert_array_equal(p4[:, :3], got)
    assert (p4[:, 3] == 7.0).all()


def test_voxel_filter_matches_numpy_restatement():
    """po_voxel_filter against the independent numpy restatement of pcl::VoxelGrid (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "voxel_4k.npz"))
    for key, leaf in (("leaf_1", 1.0), ("leaf_037", 0.37)):
        out = po.voxel_filter(g["cloud"], leaf)
        np.testing.assert_array_equal(out, g[key])
    assert po.voxel_filter(np.zeros((0, 3), np.float32), 1.0).shape == (0, 3)
