"""bench.py's launch contract (round-1 verdict, item 1a): `--gpus N` never reports a job smaller than the one asked
for — it starts N ranks itself or exits non-zero — and the JSON line carries the agreed blocks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_more_gpus_than_devices_is_refused_not_downgraded():
    """Asking for more GPUs than the box has must fail loudly (here: 0 or 1 device), never run one rank and print a line."""
    sys.path.insert(0, ROOT)
    import bench
    have = bench.visible_gpu_count()   # counted from the KFD topology: the launcher never loads the HIP runtime
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "refusing" in r.stderr


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4", "--steps", "1"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_line_has_the_agreed_blocks():
    """A small run of the real thing (20k points, RCCL initialised with one rank): one JSON line with the median-of-
    windows value, roofline + traffic source, both CPU baselines, cold-start times, the converged-inner block and parity."""
    r = _run(["--n", "20000", "--steps", "4", "--warmup", "2", "--windows", "3", "--force-dist", "--cpu-iters", "3"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["windows"]["count"] == 3 and len(out["windows"]["it_per_s"]) == 3
    assert out["windows"]["min_it_per_s"] <= out["value"] <= out["windows"]["max_it_per_s"]
    assert abs(out["ms_per_step"] * out["steps"] - out["windows"]["window_ms"]) < 1e-9
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1 and "traffic_source" in out["roofline"]
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert out["cpu_baseline_refshape"]["cores"] == 1 and out["cpu_baseline_refshape"]["value"] > 0
    assert len(out["cold_ms_per_iteration"]) == 5
    assert out["converged_inner"]["mean_inner_steps"] >= 1 and out["converged_inner"]["it_per_s"] > 0
    assert out["parity"]["rot_err_rad"] < 1e-5 and out["parity"]["trans_err_m"] < 1e-5
    assert out["gathered_transforms"] == 1
    # round 3: the timed instantiation is the roofline's kernel (stand-alone figure beside it), set-up cost per pair,
    # window spread, and the C++ classes timed in a child process on both inner schedules
    assert "folded in" in out["roofline"]["kernel"] and out["roofline"]["standalone"]["avg_kernel_ms"] > 0
    assert out["setup_ms"]["total"] > 0 and out["setup_ms"]["grid_and_source_sort_kernels"] > 0
    assert out["windows"]["spread"] >= 0 and "settle" in out["windows"]
    for key in ("inner_steps_1", "default_inner_to_f_tol"):
        assert out["cpp_api"][key]["steady_it_per_s"] > 0, out["cpp_api"]


def test_launcher_counts_devices_without_the_hip_runtime():
    """spawn_ranks' device count must not load libamdhip64 (the parent of the ranks stays GPU-clean)."""
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.visible_gpu_count(); "
            "import os; maps = open('/proc/self/maps').read(); "
            "assert 'libamdhip64' not in maps and 'libhsa-runtime' not in maps, 'HIP/HSA runtime loaded'; print(n)") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert int(r.stdout.strip()) >= 0


@pytest.mark.gpu
def test_config5_batch_line_small():
    """--config 5 shape at a reduced size: 64 pairs on the one GPU of the test box, every gathered transform verified
    against its single-rank run inside bench.py itself."""
    r = _run(["--config", "5", "--n", "6000", "--steps", "3", "--warmup", "1", "--windows", "2", "--no-cpu-baseline",
              "--no-extras"])
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["config"]["pairs"] == 64 and out["config"]["pairs_per_gpu"] == 64 and out["scaling"] == "strong"
    assert out["gathered_transforms"] == 64
    assert out["batch_verification"]["pairs_checked"] == 64
    assert out["batch_verification"]["max_abs_diff_vs_single_rank_run"] < 1e-9
