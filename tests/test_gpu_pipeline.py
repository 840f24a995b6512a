"""GPU tests of the device-paced align loop (ppcr_align / ppcr_align_report): the inner IRLS loop to function_tolerance,
the companion move and the per-iteration reports all run on the device, one iteration ahead of the host — and every
number must be the one the one-call-at-a-time loop (options run_ahead = 0 / ppcr_iterate + the report calls) and the
oracle produce.  Reference: src/prob_point_cloud_registration.cc:63-136 (loop), :96-100 + ..._iteration.hpp:52-57
(inner solve to function_tolerance), :110-129 (second cloud, reports).

Nothing here reads /root/reference (it does not exist on the GPU box)."""
import os

import numpy as np
import pytest

from oracle import binding as po
from probabilistic_point_clouds_registration_amd import _lib, synth

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-5    # rad   (BASELINE.json north_star)
TRANS_TOL = 1e-5  # m
REF_F_TOL = 10e-6  # function_tolerance of the reference (cc:97)


def _run(src, tgt, opts, m=10, dof=5.0, radius=1.0, companion=None, **align_kw):
    with _lib.Context(0) as c:
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_params(radius, m, dof, 3)
        c.set_target(tgt)
        c.set_source(src)
        if companion is not None:
            c.set_companion(companion)
        res = c.align(**align_kw)
        moved = c.get_source()
        comp = c.get_companion() if companion is not None else None
    return res, moved, comp


@pytest.mark.parametrize("dof", [5.0, float("inf"), 3.0, 10.0, 3.5])
def test_device_paced_inner_loop_is_the_host_paced_loop(dof):
    """inner_steps > 1: the device decides about the inner loop (LoopCtl in the fold-and-solve lane) and walks the later
    IRLS steps itself (inner_steps_kernel).  With the same kernels on both sides (fuse_k23 = 0) histories, costs, step
    counts and the moved source are IDENTICAL to the host-paced loop — also when the device's step budget runs out and
    the host takes an iteration over (inner_dev_steps 0 / 1), and when the inner loop is capped at 2 or 3 steps."""
    src, tgt, _, _ = synth.make_pair(20000, cfg=2, stride=3)
    full = np.concatenate([src, (src[::3] + np.float32(0.02))]).astype(np.float32)
    for inner, f_tol in ((100, REF_F_TOL), (100, 1e-9), (2, 1e-12), (3, 1e-12)):
        kw = dict(n_iter=6, cost_drop_thresh=0.0, inner_steps=inner, f_tol=f_tol)
        base, base_src, base_comp = _run(src, tgt, dict(run_ahead=0, fuse_k23=0), dof=dof, companion=full, **kw)
        assert base["n_iter"] == 6
        if f_tol < 1e-8:
            assert base["inner_steps"].max() >= min(inner, 3), "the case must need several inner steps"
        for dev_steps in (3, 1, 0, 8):
            res, moved, comp = _run(src, tgt, dict(fuse_k23=0, inner_dev_steps=dev_steps), dof=dof, companion=full, **kw)
            tag = f"dof={dof} inner={inner} f_tol={f_tol} dev_steps={dev_steps}"
            np.testing.assert_array_equal(res["inner_steps"], base["inner_steps"], err_msg=tag)
            np.testing.assert_array_equal(res["history"], base["history"], err_msg=tag)
            np.testing.assert_array_equal(res["costs"], base["costs"], err_msg=tag)
            np.testing.assert_array_equal(moved, base_src, err_msg=tag)
            np.testing.assert_array_equal(comp, base_comp, err_msg=tag)
        # the default (K23 of the first step folded into K1: another summation order) agrees to rounding
        res, moved, _ = _run(src, tgt, dict(), dof=dof, **kw)
        np.testing.assert_array_equal(res["inner_steps"], base["inner_steps"])
        np.testing.assert_allclose(res["history"], base["history"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(res["costs"], base["costs"], rtol=1e-9)
    # and the oracle, on the reference's own schedule
    ora = po.align(src, tgt, 1.0, 10, dof, 6, cost_drop_thresh=0.0, inner_max_steps=100, f_tol=REF_F_TOL)
    res, _, _ = _run(src, tgt, dict(), dof=dof, n_iter=6, cost_drop_thresh=0.0, inner_steps=100, f_tol=REF_F_TOL)
    np.testing.assert_array_equal(res["inner_steps"], ora["inner_steps"])
    for k in range(6):
        assert synth.rotation_angle(res["history"][k][:, :3], ora["history"][k][:, :3]) < 1e-8
        assert np.linalg.norm(res["history"][k][:, 3] - ora["history"][k][:, 3]) < 1e-8
    np.testing.assert_allclose(res["costs"], ora["costs"], rtol=1e-8)


def test_device_paced_loop_keeps_the_stopping_rule_exact():
    """The run-ahead stays exact with a device-paced inner loop: early stops, patience, chained calls."""
    src, tgt, _, _ = synth.make_pair(15000, cfg=2, stride=3)
    for case in (dict(n_iter=40, cost_drop_thresh=0.05, n_cost_drop_it=1), dict(n_iter=40, cost_drop_thresh=0.05, n_cost_drop_it=0),
                 dict(n_iter=9, cost_drop_thresh=0.3, n_cost_drop_it=2.5), dict(n_iter=1, cost_drop_thresh=0.0, n_cost_drop_it=5)):
        out = []
        for opts in (dict(run_ahead=0, fuse_k23=0), dict(fuse_k23=0), dict(fuse_k23=0, inner_dev_steps=1)):
            with _lib.Context(0) as c:
                for k, v in opts.items():
                    c.set_option(k, v)
                c.set_params(1.0, 10, 5.0, 3)
                c.set_target(tgt)
                c.set_source(src)
                r1 = c.align(inner_steps=100, f_tol=1e-8, **case)
                r2 = c.align(3, cost_drop_thresh=0.0, inner_steps=100, f_tol=1e-8)   # continues from the moved source
                out.append((r1, r2, c.get_source()))
        for (a1, a2, asrc) in out[1:]:
            assert a1["n_iter"] == out[0][0]["n_iter"], case
            for key in ("history", "costs", "inner_steps"):
                np.testing.assert_array_equal(a1[key], out[0][0][key], err_msg=str(case))
                np.testing.assert_array_equal(a2[key], out[0][1][key], err_msg=str(case))
            np.testing.assert_array_equal(asrc, out[0][2])


@pytest.mark.parametrize("with_companion", [True, False])
def test_align_report_delivers_the_references_per_iteration_reports(with_companion):
    """ppcr_align_report: cost, increment, mean distance to the ground truth (cc:115) and mean distance moved (cc:121)
    per iteration, computed on the device behind the solve and delivered through the callback — equal to what
    ppcr_iterate + ppcr_mse_ground_truth + ppcr_mse_previous report one call at a time, and to the oracle."""
    src, tgt, _, _ = synth.make_pair(9000, cfg=1, stride=3)
    full = np.concatenate([src, (src + np.float32(0.013))[::2]]).astype(np.float32)
    tracked0 = full if with_companion else src
    gt = (tracked0 + np.float32(0.05)).astype(np.float32)
    n_it, inner = 5, 100

    def handle(opts):
        c = _lib.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        c.set_params(1.0, 10, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        if with_companion:
            c.set_companion(full)
        c.set_ground_truth(gt)
        return c

    # one call at a time
    with handle(dict(fuse_k23=0)) as c:
        c.mse_previous()
        want = []
        for it in range(n_it):
            T, cost, st = c.iterate(inner_steps=inner, f_tol=REF_F_TOL)
            want.append(dict(T=T, cost=cost, steps=st, truth=c.mse_ground_truth(), moved=c.mse_previous()))
        final_tracked = c.get_companion() if with_companion else c.get_source()
    for opts in (dict(fuse_k23=0), dict(fuse_k23=0, inner_dev_steps=0), dict(fuse_k23=0, run_ahead=0)):
        seen = []
        with handle(opts) as c:
            rep = c.align_report(n_it, cost_drop_thresh=0.0, inner_steps=inner, f_tol=REF_F_TOL, report_truth=True,
                                 report_moved=True, on_iteration=lambda row: seen.append(row["iteration"]))
            got_tracked = c.get_companion() if with_companion else c.get_source()
            after = c.mse_ground_truth()
        assert rep["n_iter"] == n_it and seen == list(range(n_it))
        assert rep["rule"].iteration == n_it
        Tc = np.eye(4)
        # With a companion both loops leave the source's moves to the next association, so every kernel sees the same
        # data in the same order: identical bits.  Without one, the report calls of the one-at-a-time loop read the
        # SOURCE, which applies its pending move at once and restarts the temporal cut-off: the next association stores
        # each row's neighbours in another order and the moments differ in the last bits.
        exact = with_companion
        for it, (row, w) in enumerate(zip(rep["iterations"], want)):
            tag = f"{opts} iteration {it}"
            assert row["inner_steps"] == w["steps"], tag
            if exact:
                np.testing.assert_array_equal(row["T_step"], w["T"], err_msg=tag)
                np.testing.assert_array_equal(np.array(row["cost"]), w["cost"], err_msg=tag)
                assert row["mse_truth"] == w["truth"], tag
                assert row["moved"] == w["moved"], tag
            else:
                np.testing.assert_allclose(row["T_step"], w["T"], rtol=0, atol=1e-12, err_msg=tag)
                np.testing.assert_allclose(np.array(row["cost"]), w["cost"], rtol=1e-10, err_msg=tag)
                assert abs(row["mse_truth"] - w["truth"]) < 1e-9 and abs(row["moved"] - w["moved"]) < 1e-9, tag
            Tc = np.vstack([row["T_step"], [0, 0, 0, 1]]) @ Tc
            np.testing.assert_allclose(row["T_cum"], Tc[:3], rtol=0, atol=1e-15)
        if exact:
            np.testing.assert_array_equal(got_tracked, final_tracked)
            assert after == want[-1]["truth"]
        else:
            np.testing.assert_allclose(got_tracked, final_tracked, rtol=0, atol=1e-5)
            assert abs(after - want[-1]["truth"]) < 1e-6
    # against the oracle's host arithmetic
    cur = tracked0.copy()
    for row in rep["iterations"]:
        prev = cur.copy()
        po.transform_cloud(cur, np.vstack([row["T_step"], [0, 0, 0, 1]]))
        assert abs(row["mse_truth"] - po.calculate_mse(cur, gt)) < 1e-12
        assert abs(row["moved"] - po.calculate_mse(cur, prev)) < 1e-12
    # unrequested reports are NaN; a report without its cloud is refused
    with handle(dict()) as c:
        rep = c.align_report(2, cost_drop_thresh=0.0, inner_steps=1, report_moved=True)
        assert all(np.isnan(r["mse_truth"]) and np.isfinite(r["moved"]) for r in rep["iterations"])
    with _lib.Context(0) as c:
        c.set_params(1.0, 10, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        with pytest.raises(_lib.PpcrError, match="ground truth"):
            c.align_report(2, report_truth=True)


def test_align_report_continues_a_stop_rule():
    """rule_io: the loop continues from the caller's hasConverged() state and leaves the state it stopped in, so a front
    end that owns the rule (the C++ class) can mix align() with its own checks."""
    src, tgt, _, _ = synth.make_pair(8000, cfg=1, stride=3)
    with _lib.Context(0) as c:
        c.set_params(1.0, 5, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        whole = c.align(30, cost_drop_thresh=0.05, n_cost_drop_it=2, inner_steps=1)
    with _lib.Context(0) as c:
        c.set_params(1.0, 5, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        rule = _lib.StopRule(0, 0, 0.0)
        a = c.align_report(3, cost_drop_thresh=0.05, n_cost_drop_it=2, inner_steps=1, rule=rule)   # capped at 3
        assert a["n_iter"] == 3 and rule.iteration == 3
        b = c.align_report(30, cost_drop_thresh=0.05, n_cost_drop_it=2, inner_steps=1, rule=rule)  # carries on
        assert rule.iteration == whole["n_iter"]
        assert b["n_iter"] == whole["n_iter"]          # n_done counts the rule's iterations
        assert rule.check(30, 0.05, 2) != 0            # asking again gives the same verdict
        steps = [r["T_step"] for r in a["iterations"] + b["iterations"]]
        Tc = np.eye(4)
        for k, T in enumerate(steps):
            Tc = np.vstack([T, [0, 0, 0, 1]]) @ Tc
            np.testing.assert_allclose(Tc[:3], whole["history"][k], rtol=0, atol=1e-12)


@pytest.mark.parametrize("m", [11, 12])
def test_align_with_list_widths_that_have_no_steady_state_variant(m):
    """max_neighbours 11 and 12 are served by the 16-wide K1, which has no steady-state variant to fold K23 or the
    fold-and-solve step into: the pipelined loop must fall back to separate launches (round-2 advisor finding)."""
    src, tgt, _, _ = synth.make_pair(12000, cfg=2, stride=3)
    for inner in (1, 100):
        with _lib.Context(0) as c:
            c.set_params(1.0, m, 5.0, 3)
            c.set_target(tgt)
            c.set_source(src)
            res = c.align(6, cost_drop_thresh=0.0, inner_steps=inner, f_tol=REF_F_TOL)
        ora = po.align(src, tgt, 1.0, m, 5.0, 6, cost_drop_thresh=0.0, inner_max_steps=inner, f_tol=REF_F_TOL)
        assert res["n_iter"] == 6
        np.testing.assert_array_equal(res["inner_steps"], ora["inner_steps"])
        for k in range(6):
            assert synth.rotation_angle(res["history"][k][:, :3], ora["history"][k][:, :3]) < 1e-8
            assert np.linalg.norm(res["history"][k][:, 3] - ora["history"][k][:, 3]) < 1e-8


def test_align_many_paces_inner_loops_on_the_device():
    """ppcr_align_many drives device-paced inner loops of several handles from one thread; every pair equals its solo run."""
    pairs = [synth.make_pair(4000 + 700 * p, cfg=5, pair=p, stride=3)[:2] for p in range(5)]
    solo = []
    for s, t in pairs:
        with _lib.Context(0) as c:
            c.set_params(1.0, 10, 5.0, 3)
            c.set_target(t)
            c.set_source(s)
            solo.append(c.align(5, cost_drop_thresh=0.0, inner_steps=100, f_tol=REF_F_TOL)["history"][-1])
    for lanes, dev_steps in ((1, 3), (3, 3), (5, 1)):
        ctxs = []
        try:
            for s, t in pairs:
                c = _lib.Context(0)
                c.set_option("inner_dev_steps", dev_steps)
                c.set_params(1.0, 10, 5.0, 3)
                c.set_target(t)
                c.set_source(s)
                ctxs.append(c)
            T, done = _lib.align_many(ctxs, 5, lanes=lanes, cost_drop_thresh=0.0, inner_steps=100, f_tol=REF_F_TOL)
            assert list(done) == [5] * len(pairs)
            for p in range(len(pairs)):
                # lanes > 1 keeps the fold as its own launch (shares_device): another summation order than the solo run
                np.testing.assert_allclose(T[p], solo[p], rtol=0, atol=1e-10)
        finally:
            for c in ctxs:
                c.close()


def test_sequence_numbers_far_into_a_handles_life():
    """A pooled handle's sequence numbers grow by one per fold for days.  (i) Past 2^29 the product seq * kMaxDevSteps that
    orders the device-paced IRLS steps no longer fits 32 bits (it is a 64-bit word now: with the 32-bit one a later step
    could start before the previous one had solved); (ii) past 2^30 the next source upload restarts everything stamped
    with sequence numbers from zero.  Both must leave a registration with an inner loop exactly what a fresh handle gives."""
    src, tgt, _, _ = synth.make_pair(30_000, cfg=2)
    kw = dict(cost_drop_thresh=0.0, inner_steps=30, f_tol=1e-7)
    with _lib.Context(0) as c:
        c.set_params(1.0, 10, 5.0, 3)
        c.set_target(tgt)
        c.set_source(src)
        fresh = c.align(6, **kw)
        assert max(fresh["inner_steps"]) >= 3          # (the device walks several steps per iteration)
        for start in ((1 << 29) + 11, (1 << 30) + 7, 0x7FFFFF00):
            c.set_option("debug_mbox_seq", start)
            c.set_source(src)                          # (past 2^30: restarts the sequence numbers)
            again = c.align(6, **kw)
            np.testing.assert_array_equal(again["inner_steps"], fresh["inner_steps"])
            np.testing.assert_array_equal(again["history"], fresh["history"])
            np.testing.assert_array_equal(again["costs"], fresh["costs"])
