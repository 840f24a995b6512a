"""bench.py's rank logic with TWO REAL RANKS on a box without GPUs (round-2 review, item 8): the launch line the driver
uses (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N) with `--backend gloo --fake-register`,
which swaps the GPU registration for a deterministic CPU stand-in and leaves everything else as it is — sharding
(pair p on rank p % world), warm-up of the collective, timed windows between barriers, the gather of the transforms,
all_reduce(MAX) of the window times, the joint teardown (barrier + destroy on every rank BEFORE rank 0's long
single-rank tail), the verification of every gathered transform, one JSON line from rank 0, exit status 0 from both."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc, bench_args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH] + bench_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)


def _line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]        # rank 0 only
    return json.loads(lines[0])


def test_config5_two_ranks_shard_gather_verify_and_leave_together():
    out = _line(_torchrun(2, ["--gpus", "2", "--config", "5", "--points", "300", "--steps", "3", "--warmup", "1", "--windows", "2",
                              "--backend", "gloo", "--fake-register"]))
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["pairs"] == 64 and out["config"]["pairs_per_gpu"] == 32
    assert out["gathered_transforms"] == 64                      # every pair of both ranks arrived on rank 0
    assert out["batch_verification"]["pairs_checked"] == 64      # ... and equals its single-rank re-run (after the teardown)
    assert out["batch_verification"]["max_abs_diff_vs_single_rank_run"] == 0.0
    assert out["windows"]["count"] == 2 and out["value"] > 0
    assert "FAKE" in out["data"]


def test_weak_scaling_two_ranks_one_pair_each():
    out = _line(_torchrun(2, ["--gpus", "2", "--points", "500", "--steps", "4", "--warmup", "2", "--windows", "3", "--backend", "gloo",
                              "--fake-register"]))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["pairs"] == 2
    assert out["gathered_transforms"] == 2
    assert abs(out["ms_per_step"] * out["steps"] - out["windows"]["window_ms"]) < 1e-9


def test_world_size_must_match_gpus_also_under_torchrun():
    r = _torchrun(2, ["--gpus", "4", "--steps", "1", "--backend", "gloo", "--fake-register"])
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_gloo_without_the_test_switch_is_refused():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--backend", "gloo", "--steps", "1"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0 and "fake-register" in (r.stdout + r.stderr)
