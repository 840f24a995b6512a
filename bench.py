#!/usr/bin/env python3
"""Benchmark of the registration hot path on MI355X (contract: see the task statement / DESIGN.md §5).

A "step" is one outer registration iteration over one source/target pair:
    radius-NN association (K1) -> t/Gaussian weights + weighted moments (K23) -> 3x3 SVD -> in-place move of
    the source (K4, folded into the next K1),
i.e. BASELINE.json's metric "registration iterations/sec (1M<->1M pts, r=1.0, m=10)" with one inner IRLS step per
association, early termination disabled (cost_drop_thresh = 0).  Clouds are resident in HBM
before a timed window starts.

Timed region: `--windows` (default 5) windows; every window re-uploads the source (fresh start), runs `--warmup`
untimed iterations, and then times EXACTLY `--steps` iterations between barrier + synchronize on both sides (max over
ranks).  `value` is the MEDIAN window; min / max are reported next to it.

--gpus N without a torchrun environment: this process starts N ranks itself (torch.distributed.run as a child
process, before anything here touches a GPU) and exits with the child's status.  N ranks always means N GPUs: the
line never reports an n_gpus other than the one asked for.
  config 3 (default) / 4 / 2: every rank registers its own pair (weak scaling);
  config 5 (BASELINE configs[4]): 64 pairs of 250k, 64 / N per rank (strong scaling), the gathered transforms
  checked on rank 0 against a single-rank run of every pair.
The only collective of the job is the final all_gather of the transforms over RCCL, inside the timed window.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
REF_F_TOL = 10e-6      # function_tolerance of the reference (src/prob_point_cloud_registration.cc:97)
SIMD_CLOCK_GHZ = 2.4   # MI355X peak engine clock (MI355X_MICROARCH.md); prices roofline.valu_issue_frac


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps iterations (value = median)")
    ap.add_argument("--config", type=int, default=3,
                    help="BASELINE.json config number (3 = 1M headline, 4 = Gaussian, 2 = 100k, 5 = 64 x 250k batch; "
                         "6 / 7 = \"4b\": the 1M pair with -d 3 / -d 10, the other t models the fused kernel serves; "
                         "8 = the command line's own defaults: 200k points, radius 3, 20 neighbours, inner loop to f_tol; "
                         "9 / 10 = the same on non-uniform clouds: a LiDAR-like scene / a slab with a density gradient and dense blobs)")
    ap.add_argument("--n", "--points", dest="n", type=int, default=None,
                    help="override the cloud size (debugging; spell it --points behind torch.distributed.run, whose own "
                         "parser trips over --n)")
    ap.add_argument("--inner-steps", type=int, default=None,
                    help="IRLS steps per association (default 1: the metric's definition; config 8 = the command line's own "
                         "defaults: 100, to function_tolerance)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=0,
                    help="outer iterations of the CPU baseline sample (0 = auto: about 6 s of wall time, 3..30)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the cold-start / converged-inner / reference-shaped CPU legs (headline + roofline only)")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event per-kernel pass")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank, to exercise the collective path")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="ppcr_set_option knob applied to every handle (experiments; the default run sets none)")
    ap.add_argument("--lanes", type=int, default=8,
                    help="pairs in flight per GPU when a rank holds several pairs (ppcr_align_many: one host thread polls them all)")
    ap.add_argument("--pairs-per-gpu", type=int, default=0,
                    help="independent pairs per rank (0 = the config's own: 1, or 64 / N for config 5)")
    ap.add_argument("--no-verify", action="store_true", help="config 5: skip the single-rank re-run of every pair")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend: nccl (= RCCL, the real thing) or gloo (CPU: only with --fake-register)")
    ap.add_argument("--fake-register", action="store_true",
                    help="TEST SWITCH: replace the GPU registration by a deterministic CPU stand-in so that the rank logic "
                         "(shard, timed windows, gather, all_reduce(MAX), verification, teardown) can be run with real "
                         "ranks on a box without GPUs; the line it prints is labelled and measures nothing")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="untimed iterations run for at least this long before the first window (clocks settle)")
    ap.add_argument("--no-cpp-api", action="store_true", help="skip the cpp_api block (the C++ classes timed in a child process)")
    return ap.parse_args()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count():
    """GPUs this process tree can use, counted WITHOUT the HIP runtime (the parent of the ranks stays GPU-clean): the
    KFD topology lists one node per agent, GPU nodes are those with SIMDs; ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES narrow the set the usual way (a comma-separated list; empty = none)."""
    import glob
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            kv = dict(line.split()[:2] for line in open(props) if len(line.split()) >= 2)
        except OSError:
            continue
        if int(kv.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in os.environ:
            ids = [x for x in os.environ[var].split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


def spawn_ranks(a):
    """--gpus N > 1 outside torchrun: start the N ranks as a CHILD process (subprocess, never an exec) and hand its exit
    status on.  This parent does not load the HIP runtime at all: devices are counted from the KFD topology."""
    have = visible_gpu_count()
    if have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} requested but {have} device(s) visible; refusing to run a smaller job "
              f"under the requested label", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + [
               "--points" if arg == "--n" else arg for arg in sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def k1_instantiation(cfg, inner_steps, two_pass, fused=True, verlet=True, n_rows=None):
    """Template string of the nn_fast_kernel instantiation a config's steady-state iterations launch (mirrors
    launch_tile<M> in csrc/ppcr_nn_tile.hip and k23_form in csrc/ppcr_hip_iteration.inc): list width M = the narrowest compiled
    width holding max_neighbours; widths <= 10 have the steady-state variant (16-slot lists, 1728-candidate halo), wider
    ones keep 32 slots and the 2240-candidate halo; FTM = the K23 form folded in (8: t with v + dim = 8, 0: Gaussian,
    -3: another integer v + dim, -2: none — two-pass searches, wide lists, non-integer v + dim, fuse_k23 = 0)."""
    m = cfg["max_neighbours"]
    width = next(w for w in (4, 5, 8, 10, 16, 20, 32) if m <= w)
    steady = width <= 10
    ftm = -2
    if fused and steady and not two_pass:
        vpd = cfg["dof"] + 3
        if np.isinf(cfg["dof"]):
            ftm = 0
        elif vpd == 8:
            ftm = 8
        elif vpd == int(vpd) and 1 <= vpd <= 64:
            ftm = -3
    c, cap = (16, 1728) if steady else ((32, 2240) if width <= 24 else (48, 2048))
    multi = "scene" in cfg   # (clouds with a dense tail: the multi-level instantiation)
    if steady and not two_pass and not multi and verlet:
        # the Verlet variant (csrc/ppcr_device.hip.h: VerletLists): 24-slot scan lists, 1920-candidate halo, four workgroups
        # per CU; rows answered from their lists where the lists still hold.  Last template argument: 1, or 2 where the grid is
        # resident all at once (<= 1024 workgroups: a few failing rows are rebuilt inside the workgroup)
        rows = n_rows if n_rows is not None else cfg["n"]
        variant = 2 if (rows + 255) // 256 + 128 <= 1024 else 1
        return f"nn_fast_kernel<{width}, 24, 1920, false, {ftm}, false, {variant}>"
    if width in (16, 20) and not multi and verlet:
        # the mid widths' Verlet variant (32-slot lists, 36-slot scan lists, 1600-candidate halo), one- or two-pass
        return f"nn_fast_kernel<{width}, 36, 1600, false, -2, false, 1>"
    return f"nn_fast_kernel<{width}, {c}, {cap}, false, {ftm}, {'true' if multi else 'false'}, 0>"


def effective_cores():
    """Host cores this process can really use: the affinity mask capped by the container's CFS quota
    (the GPU boxes expose all hardware threads but cap CPU time, /sys/fs/cgroup/cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(period) + 0.5))
    except (OSError, ValueError):
        pass
    return (min(n, quota) if quota else n), n, quota


def cpu_baseline(src, tgt, cfg, iters, inner_steps, thresh=0.0):
    """The oracle (kind 'port': OpenMP-generous variant — every loop parallel over the host cores) timed on this
    box on a bounded sample: `iters` outer iterations of the SAME workload."""
    from oracle import binding as po  # checker-side import, only on this leg
    cores, visible, quota = effective_cores()
    threads = max(1, min(po.num_threads(), cores))
    tw = time.perf_counter()
    po.align(src, tgt, cfg["radius"], cfg["max_neighbours"], cfg["dof"], 1, cost_drop_thresh=thresh, inner_max_steps=inner_steps,
             threads=threads)  # warm-up (page-in, thread pool)
    tw = time.perf_counter() - tw
    if iters <= 0:
        iters = int(min(30, max(3, round(6.0 / max(tw, 1e-3)))))
    t0 = time.perf_counter()
    res = po.align(src, tgt, cfg["radius"], cfg["max_neighbours"], cfg["dof"], iters, cost_drop_thresh=thresh,
                   inner_max_steps=inner_steps, threads=threads)
    dt = time.perf_counter() - t0
    iters = len(res["history"])   # iterations really performed (hasConverged may stop a converged run early)
    return dict(value=iters / dt, unit="iterations/s", cores=threads, kind="port",
                sample=f"{iters} outer iterations of the same {src.shape[0]}<->{tgt.shape[0]} workload "
                       f"(oracle/ppcr_oracle.c, OpenMP x{threads}, grid NN; host shows {visible} hardware threads, "
                       f"container CPU quota {quota if quota else 'none'})",
                seconds=dt), res


def cpu_baseline_refshape(src, tgt, cfg, iters=3):
    """The oracle run the way the REFERENCE runs (SURVEY §8(d) ii): one thread (its search, residual and weight loops
    are serial, cc:72-81, weight_updater_callback.hpp:42-51), the spatial index rebuilt every outer iteration
    (cc:66-67) and the inner loop iterated to function_tolerance (cc:96-97).  A bounded sample; still kinder than the
    reference, which also builds a Ceres problem of nnz residual blocks per iteration."""
    from oracle import binding as po
    t0 = time.perf_counter()
    res = po.align(src, tgt, cfg["radius"], cfg["max_neighbours"], cfg["dof"], iters, cost_drop_thresh=0.0,
                   inner_max_steps=100, f_tol=REF_F_TOL, threads=1)
    dt = time.perf_counter() - t0
    n = len(res["history"])
    return dict(value=n / dt, unit="iterations/s", cores=1, kind="port",
                sample=f"{n} outer iterations of the same {src.shape[0]}<->{tgt.shape[0]} workload, one thread, grid "
                       f"rebuilt every iteration, inner IRLS to f_tol={REF_F_TOL:g} "
                       f"(mean {float(np.mean(res['inner_steps'])):.1f} inner steps)",
                seconds=dt, mean_inner_steps=float(np.mean(res["inner_steps"])))


def trace_phases(stderr_text, skip=0):
    """Median over the PPCR_TRACE lines (one per align(); the first `skip` are warm-up objects) of the first time each mark
    was passed, microseconds from the call's entry: validated (loop state ready), grid (the early grid build waited for),
    sorted (source sort enqueued), reserved (the association's buffers there), k1 (first association launched), enqueued
    (first train complete), consumed (first iteration's result on the host), ran / finished (loop over, stream idle)."""
    rows = []
    for ln in stderr_text.splitlines():
        if not ln.startswith("PPCR_TRACE "):
            continue
        first = {}
        for tok in ln.split()[1:]:
            k, _, v = tok.partition("=")
            if k not in first:
                first[k] = float(v)
        rows.append(first)
    rows = rows[skip:]
    if not rows:
        return None
    keys = [k for k in ("validated", "grid", "sorted", "reserved", "k1", "enqueued", "consumed", "ran", "finished") if all(k in r for r in rows)]
    return {k: round(float(np.median([r[k] for r in rows])), 1) for k in keys}


def cpp_api_block(src, tgt, cfg, a):
    """ppcr_cpp_api_test --bench: the built C++ classes over libppcr_hip.so, in their own process."""
    import tempfile
    exe = os.path.join(ROOT, "probabilistic_point_clouds_registration_amd", "ppcr_cpp_api_test")
    if not os.path.exists(exe):
        return {"error": "ppcr_cpp_api_test is not built (python -m probabilistic_point_clouds_registration_amd.build)"}
    block = {"program": "ppcr_cpp_api_test --bench (ProbPointCloudRegistration::align(), cost_drop_thresh = 0)",
             "method": "steady it/s = S / (align time of warm + S iterations - align time of warm iterations), S = 3 x steps, "
                       "a fresh object per measurement (all constructed before the first is timed), median of 7"}
    with tempfile.TemporaryDirectory(prefix="ppcr_bench_") as d:
        sp, tp = os.path.join(d, "src.f32"), os.path.join(d, "tgt.f32")
        np.ascontiguousarray(src[:, :3], dtype=np.float32).tofile(sp)
        np.ascontiguousarray(tgt[:, :3], dtype=np.float32).tofile(tp)
        dof = "inf" if np.isinf(cfg["dof"]) else repr(float(cfg["dof"]))
        for key, inner in (("inner_steps_1", 1), ("default_inner_to_f_tol", 100)):
            cmd = [exe, "--bench", sp, tp, repr(float(cfg["radius"])), str(cfg["max_neighbours"]), dof, str(a.warmup),
                   str(a.steps), str(inner), "7"]
            try:
                # PPCR_TRACE: the library marks the host wall clock through every align() (stderr, one line per call);
                # the marks of the fresh objects say WHERE a box spends align_fixed_overhead_ms
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, PPCR_TRACE="1"))
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                block[key] = json.loads(line[-1]) if (r.returncode == 0 and line) else {
                    "error": f"exit {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
                if "error" not in block[key]:
                    block[key]["fresh_align_phases_us"] = trace_phases(r.stderr, skip=3)
            except (OSError, subprocess.TimeoutExpired, ValueError) as e:
                block[key] = {"error": str(e)}
    return block


class FakeContext:
    """Stand-in of _lib.Context for --fake-register (the rank-logic test): same calls, no GPU, results that depend only
    on the clouds and the iteration count, a little sleep per iteration so that windows have a duration."""

    def __init__(self, device_id=0):
        self.src = self.tgt = None
        self.iters = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    def set_option(self, key, value):
        pass

    def set_params(self, *args):
        pass

    def set_target(self, tgt):
        self.tgt = np.asarray(tgt, np.float64)[:, :3].mean(axis=0)

    def set_source(self, src):
        self.src = np.asarray(src, np.float64)[:, :3].mean(axis=0)
        self.iters = 0

    def synchronize(self):
        pass

    def association_size(self):
        return 0, 0

    def _T(self):
        T = np.eye(4)[:3].copy()
        T[:, 3] = (self.tgt - self.src) * (1.0 - 0.5 ** self.iters)
        return T

    def align(self, n_iter, cost_drop_thresh=0.0, inner_steps=1, f_tol=1e-5, want_history=True, **kw):
        hist = []
        for _ in range(int(n_iter)):
            time.sleep(2e-4)
            self.iters += 1
            hist.append(self._T())
        return dict(n_iter=int(n_iter), history=np.array(hist).reshape(-1, 3, 4), inner_steps=np.ones(int(n_iter), np.int32),
                    costs=np.zeros((int(n_iter), 2)))


def fake_align_many(ctxs, k, lanes=1, **kw):
    out = [c.align(k)["history"][-1] if k > 0 else np.eye(4)[:3] for c in ctxs]
    return np.array(out), [k] * len(ctxs)


def run_rank(a):
    from probabilistic_point_clouds_registration_amd import batch, synth
    if a.fake_register:
        import types
        _lib = types.SimpleNamespace(Context=FakeContext, align_many=fake_align_many)
        a.no_profile = a.no_extras = a.no_cpu_baseline = a.no_cpp_api = True
        a.settle_ms = 0.0
    else:
        from probabilistic_point_clouds_registration_amd import _lib
        if a.backend != "nccl":
            raise SystemExit("bench.py: --backend gloo is for --fake-register runs only (the product's collective is RCCL)")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to mislabel the run")

    import torch  # device selection, synchronize, torch.distributed (RCCL)
    on_gpu = not a.fake_register
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        if torch.cuda.device_count() < (local_rank + 1):
            raise SystemExit(f"bench.py: rank {rank} has no device {local_rank}")
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if on_gpu:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=a.backend, rank=rank, world_size=world)
        assert dist.get_world_size() == a.gpus

    cfg = dict(synth.CONFIGS[a.config])
    if a.inner_steps is None:
        a.inner_steps = int(cfg.get("inner_steps", 1))
    n = a.n or cfg["n"]
    batch_cfg = "pairs" in cfg              # config 5: a fixed batch of independent pairs sharded over the ranks
    if a.pairs_per_gpu > 0:
        n_pairs = world * a.pairs_per_gpu
    elif batch_cfg:
        if cfg["pairs"] % world:
            raise SystemExit(f"config {a.config}: {cfg['pairs']} pairs do not divide over {world} ranks")
        n_pairs = cfg["pairs"]
    else:
        n_pairs = world
    pairs_per_gpu = n_pairs // world
    my_pairs = batch.shard_pairs(n_pairs, world, rank)   # pair p lives on rank p % world
    ctxs, clouds = [], []
    for p in my_pairs:
        src, tgt, _, _ = synth.make_config(a.config, pair=p, n=n)
        c = _lib.Context(local_rank)
        for kv in a.opt:
            k, v = kv.split("=")
            c.set_option(k, int(v))
        c.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
        c.set_target(tgt)
        c.set_source(src)
        ctxs.append(c)
        clouds.append((src, tgt))
    ctx = ctxs[0]
    src, tgt = clouds[0]   # this rank's first pair (rank 0: cpu baselines / parity)
    concurrent = len(ctxs) > 1 and a.lanes > 1
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")

    def barrier():
        for c in ctxs:
            c.synchronize()
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # with the inner loop run to function_tolerance a pair converges inside the window and its cost drop becomes rounding
    # noise of either sign: those workloads run their k iterations regardless of it
    timed_thresh = -1.0 if a.inner_steps > 1 else 0.0

    def run_iterations(k, inner, f_tol=1e-5, thresh=None):
        """k outer iterations on every pair of this rank -> {pair: final cumulative 3x4 of these k iterations}
        (thresh = 0 runs exactly k iterations while the cost still falls; a converged pair, whose cost drop is rounding
        noise of either sign, needs a negative threshold to keep going)"""
        out = {}
        thresh = timed_thresh if thresh is None else thresh
        if concurrent:
            # several resident pairs per GPU: a.lanes of them in flight, each on its own handle/stream
            T_fin, done = _lib.align_many(ctxs, k, lanes=a.lanes, cost_drop_thresh=thresh, inner_steps=inner, f_tol=f_tol)
            assert all(int(d) == k for d in done), f"early stop: {list(done)}"
            for j, p in enumerate(my_pairs):
                out[p] = T_fin[j]
        else:
            for p, c in zip(my_pairs, ctxs):
                res = c.align(k, cost_drop_thresh=thresh, inner_steps=inner, f_tol=f_tol)
                assert res["n_iter"] == k, f"early stop: {res['n_iter']}"
                out[p] = res["history"][-1] if k > 0 else np.eye(4)[:3]
        return out

    def fresh_start(inner, f_tol=1e-5):
        for c, (s, _) in zip(ctxs, clouds):
            c.set_source(s)
        if a.warmup > 0:
            run_iterations(a.warmup, inner, f_tol)

    if dist is not None:
        # the collective is warmed up once (communicator set-up and its buffers are not part of a step)
        batch.gather_transforms({p: np.eye(4)[:3] for p in my_pairs}, n_pairs, dist=dist, device=dev)

    # ---- settle: the first windows of a process used to come out ~8 % slow (clocks still ramping); run untimed
    # iterations of the same workload until --settle-ms have passed
    settle_iters = 0
    if a.settle_ms > 0:
        fresh_start(a.inner_steps)
        ts = time.perf_counter()
        while (time.perf_counter() - ts) * 1e3 < a.settle_ms:
            run_iterations(max(a.steps, 50), a.inner_steps, thresh=-1.0)
            settle_iters += max(a.steps, 50)
        barrier()

    # ---- timed windows --------------------------------------------------------------------------------------------
    window_s, gathered, local = [], None, None
    for w in range(max(1, a.windows)):
        fresh_start(a.inner_steps)
        barrier()
        t0 = time.perf_counter()
        local = run_iterations(a.steps, a.inner_steps)
        if dist is not None:
            # RCCL: the only collective of the job — final gather of the transforms
            gathered = batch.gather_transforms(local, n_pairs, dist=dist, device=dev)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        window_s.append(dt)
    if dist is None:
        gathered = batch.gather_transforms(local, n_pairs)
    dt = float(np.median(window_s))

    # ---- per-kernel durations with HIP events on the handle's own stream (separate passes so that the event records
    # cannot perturb the headline number; same workload, same warm-up, same K).  Pass A is the schedule of the timed
    # windows (K23 folded into K1: the kernel the roofline is quoted for); pass B runs K23 as its own kernel, so that
    # K1's duration is K1's alone (the stand-alone figure beside it).
    prof, prof_alone = {}, {}
    nnz = ctx.association_size()[1]
    if not a.no_profile:
        def profiled_pass(fuse):
            ctx.set_option("fuse_k23", fuse)
            ctx.set_source(src)
            if a.warmup > 0:
                ctx.align(a.warmup, cost_drop_thresh=timed_thresh, inner_steps=a.inner_steps, want_history=False)
            ctx.profile_enable(True)
            tp0 = time.perf_counter()
            ctx.align(a.steps, cost_drop_thresh=timed_thresh, inner_steps=a.inner_steps, want_history=False)
            ctx.synchronize()
            tp = time.perf_counter() - tp0
            out_ = ctx.profile_get()
            ctx.profile_enable(False)
            out_["_profiled_pass_ms_per_step"] = 1e3 * tp / a.steps
            return out_
        prof = profiled_pass(1)
        prof_alone = profiled_pass(0)
        ctx.set_option("fuse_k23", 1)

    # ---- set-up cost of one pair ("grid build of the static target reported separately"): wall time of the
    # uploads (H2D + repack, synchronous) and of the first association on a fresh handle, with the HIP-event durations
    # of the grid-build and source-sort kernels inside it
    setup = None
    if rank == 0 and not a.no_extras:
        with _lib.Context(local_rank) as sc:
            sc.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
            sc.set_target(tgt)            # first touch of a new handle: allocations
            sc.set_source(src)
            sc.associate()
            sc.synchronize()
            reps = []
            for _ in range(5):            # wall times, no event records in the way
                t0 = time.perf_counter()
                sc.set_target(tgt)        # (+ the first half of the grid build, enqueued on the device's second stream)
                t1 = time.perf_counter()
                sc.set_source(src)
                t2 = time.perf_counter()
                sc.associate()            # rest of K0 (bbox, keys, sort, gather, cell_start) + source sort + first K1
                sc.synchronize()
                t3 = time.perf_counter()
                reps.append(dict(h2d_target=1e3 * (t1 - t0), h2d_source=1e3 * (t2 - t1), first_associate_call=1e3 * (t3 - t2)))
            sc.profile_enable(True)       # kernel durations (HIP events) from one more repetition
            sc.set_target(tgt)
            sc.set_source(src)
            sc.associate()
            sc.synchronize()
            ks = sc.profile_get()
            sc.profile_enable(False)
            ms = lambda *names: sum(ks[n]["total_ms"] for n in names if n in ks)
            k0 = ms("bbox_kernel", "cell_key_kernel", "radix_sort", "gather_points_kernel", "cell_start_kernel")
            k1_first = ms("nn_fast_kernel", "nn_tile_cleanup_kernel", "nn_wide_kernel")
            med = {k: float(np.median([r[k] for r in reps])) for k in reps[0]}
            setup = dict(med, grid_and_source_sort_kernels=k0, first_association_kernel=k1_first,
                         total=med["h2d_target"] + med["h2d_source"] + med["first_associate_call"],
                         note="ms per pair, median of 5, host buffers in, warm handle: uploads are synchronous PCIe copies + "
                              "repack (+ bounding box); the target's grid build is enqueued on a second stream at the end of "
                              "set_target and runs under the source's copy; first_associate_call = what is left of it + "
                              "spatial sort of the source + the first association, wall time; the *_kernels figures are HIP-event "
                              "durations from one more, profiled repetition")

    # Every collective of the job is behind us.  All ranks leave the process group TOGETHER, here (a last barrier, then
    # destroy): rank 0's remaining work (extras, CPU baselines, the single-rank re-run of every pair) is its own and can
    # take minutes — no rank tears its communicator down while a peer may still be inside a collective, and no peer
    # waits in a collective for a rank that has gone.
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        dist_was, dist = True, None
    else:
        dist_was = False
    if rank != 0:
        return 0

    ns, nt = src.shape[0], tgt.shape[0]
    model = "Gaussian" if np.isinf(cfg["dof"]) else "t dof=%g" % cfg["dof"]
    out = {
        # BASELINE.json's metric; the label follows the cloud size actually run (configs other than the headline)
        "metric": "registration iterations/sec (%s<->%s pts, r=%.1f, m=%d)" % (
            (("1M", "1M") if ns == nt == 1000000 else (ns, nt)) + (cfg["radius"], cfg["max_neighbours"])),
        "value": n_pairs * a.steps / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / (a.steps * pairs_per_gpu),
        "higher_is_better": True,
        "scaling": "strong" if (batch_cfg and a.pairs_per_gpu == 0) else "weak",
        "vs_baseline": None,
        "dtype": "f32 (distances) + f64 (weights, moments, solve)",
        "data": "synthetic" if on_gpu else "FAKE REGISTRATION (--fake-register: rank-logic test, measures nothing)",
        "config": {"workload": (f"BASELINE configs[{a.config - 1}]: " if a.config <= 5 else
                                "the reference CLI's default parameters (radius 3, 20 neighbours): " if a.config == 8 else
                                f"the reference CLI's default parameters on a NON-UNIFORM cloud ({cfg['scene']} scene, synth.make_scene): " if "scene" in cfg else
                                f"BASELINE configs[3] variant 4b (t model, dof {cfg['dof']:g}): ")
                               + (f"batch of {n_pairs} independent pairs of " if n_pairs > 1 else "")
                               + f"{ns}<->{nt} synthetic clouds, radius={cfg['radius']}, "
                               f"max_neighbours={cfg['max_neighbours']}, {model}, "
                               f"{a.inner_steps} inner IRLS step(s)/iteration, cost_drop_thresh=0",
                   "pairs": n_pairs, "pairs_per_gpu": pairs_per_gpu, "lanes_per_gpu": (a.lanes if concurrent else 1),
                   "parallelism": f"{n_pairs} independent pair(s), {pairs_per_gpu} per GPU on {world} GPU(s); "
                                  "no data-path collective; final RCCL all_gather of the transforms"},
        "windows": {"count": len(window_s), "statistic": "median",
                    "it_per_s": [n_pairs * a.steps / w for w in window_s],
                    "min_it_per_s": n_pairs * a.steps / max(window_s), "max_it_per_s": n_pairs * a.steps / min(window_s),
                    "window_ms": 1e3 * dt,
                    "spread": (max(window_s) - min(window_s)) / dt,
                    "settle": f"{settle_iters} untimed iterations (>= {a.settle_ms:g} ms) before the first window"},
        "nnz": int(nnz),
    }
    # roofline of the dominant kernel (K1, nn_fast_kernel): algorithmic bytes B_nn = 16*Ns + 12*Nt + 4*nnz
    #  / average launch duration measured with HIP events above
    b_nn = 16.0 * ns + 12.0 * nt + 4.0 * nnz
    b_iter = 72.0 * ns + 12.0 * nt + 52.0 * nnz + (a.inner_steps - 1) * (32.0 * ns + 48.0 * nnz)
    if "nn_fast_kernel" in prof:
        k = prof["nn_fast_kernel"]
        avg_ms = k["total_ms"] / max(1, k["launches"])
        ach = b_nn / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch and instruction counts come from PMC counters, which need their own rocprofv3 --pmc
        # passes (tools/profile_round.sh); the committed summary of the latest passes is quoted — only for the very
        # instantiation these windows ran (entries are keyed by the kernel's template string)
        two_pass = "nn_wide_kernel" in prof
        fused_on = not any(kv.split("=")[0] == "fuse_k23" and int(kv.split("=")[1]) == 0 for kv in a.opt)
        verlet_on = not any(kv.split("=")[0] == "verlet" and int(kv.split("=")[1]) != 1 for kv in a.opt)
        kname = k1_instantiation(cfg, a.inner_steps, two_pass, fused_on, verlet_on, n)
        kname_alone = k1_instantiation(cfg, a.inner_steps, two_pass, False, verlet_on, n)
        traffic, traffic_source, traffic_alone, pmc = None, None, None, {}
        tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(tpath) and ns == nt == 1000000 and not two_pass:
            tj = json.load(open(tpath))
            ent = tj.get("entries", {})
            pmc = ent.get(kname) or {}
            traffic = pmc.get("traffic_bytes_per_launch")
            traffic_alone = (ent.get(kname_alone) or {}).get("traffic_bytes_per_launch")
            if traffic is not None:
                traffic_source = "profiles/k1_traffic.json (%s)" % tj.get("source", "separate rocprofv3 --pmc passes")
        out["roofline"] = {"bound": "hbm",
                           "kernel": kname + (": first pass of a two-pass radius search (rows that come back short go to "
                                              "nn_wide_kernel; K23 is its own kernel)" if two_pass else
                                              ": K1 with the previous iteration's source move in its prologue" +
                                              (" and K23 (weights + 19 moments) folded in" if ", -2, " not in kname else "") +
                                              ("; Verlet variant: a workgroup whose rows' lists still provably hold every possible "
                                               "neighbour re-measures the lists (16 gathers per row), the others search the grid and "
                                               "rebuild" if ", 24, 1920, " in kname else "")),
                           "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "traffic_source": traffic_source,
                           "avg_kernel_ms": avg_ms, "algorithmic_bytes_per_launch": b_nn,
                           "candidate_tests_per_s": None if two_pass else 27 * 3.8147 * ns / (avg_ms * 1e-3)}
        if traffic is not None:
            # what the kernel really pulled through the memory side, as a fraction of peak (the honest bandwidth figure)
            out["roofline"]["measured_traffic_frac"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            # ... and per regime, where the committed passes tell the Verlet variant's launches apart (tools/pmc_summary.py --split):
            # answering launches re-measure lists, list-building ones search with a wider acceptance and write every row's list
            regimes = {tag: (ent.get(kname + " " + tag) or {}).get("traffic_bytes_per_launch")
                       for tag in ("[answering launches]", "[list-building launches]")}
            if any(v is not None for v in regimes.values()):
                out["roofline"]["traffic_by_regime"] = {k.strip("[]"): v for k, v in regimes.items()}
                out["roofline"]["traffic_over_algorithmic_by_regime"] = {k.strip("[]"): (v / b_nn if v else None) for k, v in regimes.items()}
        if pmc.get("valu_per_wave"):
            # VALU issue: a wave64 VALU instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs at SIMD_CLOCK_GHZ
            simd_cycles = 1024 * avg_ms * 1e-3 * SIMD_CLOCK_GHZ * 1e9
            out["roofline"]["valu_issue_frac"] = pmc["valu_insts"] * 4.0 / simd_cycles
            out["roofline"]["valu_per_wave"] = pmc["valu_per_wave"]
            if pmc.get("lds_active_cycles"):
                out["roofline"]["lds_bank_conflict_share"] = pmc.get("lds_bank_conflict_cycles", 0.0) / pmc["lds_active_cycles"]
            out["roofline"]["counters_note"] = ("instruction counts per launch from the committed --pmc passes of this instantiation; "
                                               f"issue fraction priced at {SIMD_CLOCK_GHZ} GHz over this run's HIP-event duration")
        if "nn_fast_kernel" in prof_alone and not two_pass:
            ka = prof_alone["nn_fast_kernel"]
            alone_ms = ka["total_ms"] / max(1, ka["launches"])
            out["roofline"]["standalone"] = {"kernel": kname_alone + " (K1 alone, K23 as its own kernel)",
                                             "avg_kernel_ms": alone_ms, "achieved": b_nn / (alone_ms * 1e-3) / 1e9,
                                             "frac": b_nn / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_alone}
    else:
        out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                           "traffic": None, "traffic_source": None}
    out["iteration_roofline"] = {"algorithmic_bytes_per_iteration": b_iter,
                                 "notional_materialised_GBs_per_gpu": b_iter * a.steps * pairs_per_gpu / dt / 1e9,
                                 "notional_materialised_frac": b_iter * a.steps * pairs_per_gpu / dt / 1e9 / HBM_PEAK_GBS,
                                 "measured_traffic_frac": ((out["roofline"].get("traffic") or 0.0) * a.steps * pairs_per_gpu / dt / 1e9
                                                           / HBM_PEAK_GBS) if out["roofline"].get("traffic") else None,
                                 "note": "notional_*: B_iter is SURVEY §8(d)'s definition (weights and correspondences "
                                         "MATERIALISED, 600 MB at 1M) over the iteration time — NOT achieved bandwidth: the "
                                         "fused kernels never move those bytes; measured_traffic_frac: the PMC-counted bytes "
                                         "of K1 (the iteration's only large kernel) per iteration over the iteration time"}
    out["kernels_ms_per_launch"] = {k: v["total_ms"] / max(1, v["launches"]) for k, v in prof.items()
                                    if isinstance(v, dict)}
    out["kernels_ms_per_launch_k23_unfused"] = {k: v["total_ms"] / max(1, v["launches"]) for k, v in prof_alone.items()
                                                if isinstance(v, dict)}
    if "_profiled_pass_ms_per_step" in prof:
        out["profiled_pass_ms_per_step"] = prof["_profiled_pass_ms_per_step"]
    if setup is not None:
        out["setup_ms"] = setup

    if not a.no_extras:
        # cold start: the first associations after an upload have no temporal cut-off yet and the source still moves
        ctx.set_source(src)
        ctx.synchronize()
        cold = []
        for _ in range(5):
            tc = time.perf_counter()
            ctx.iterate(inner_steps=a.inner_steps)
            ctx.synchronize()
            cold.append(1e3 * (time.perf_counter() - tc))
        out["cold_ms_per_iteration"] = cold   # it0 also sorts the source into the grid's order (once per upload)
        # the reference's own schedule (inner IRLS to its function_tolerance, cc:96-100): what ppcr_align(inner_steps = 100)
        # — and with it the C++ class and the CLI by default — runs; the inner loop is paced by the device
        rates, inner_mean = [], 0.0
        for _ in range(max(1, a.windows)):
            ctx.set_source(src)
            if a.warmup > 0:
                ctx.align(a.warmup, cost_drop_thresh=0.0, inner_steps=100, f_tol=REF_F_TOL, want_history=False)
            ctx.synchronize()
            tc = time.perf_counter()
            res = ctx.align(a.steps, cost_drop_thresh=0.0, inner_steps=100, f_tol=REF_F_TOL)
            ctx.synchronize()
            rates.append(a.steps / (time.perf_counter() - tc))
            inner_mean = float(np.mean(res["inner_steps"]))
        out["converged_inner"] = {"it_per_s": float(np.median(rates)), "ms_per_iteration": 1e3 / float(np.median(rates)),
                                  "min_it_per_s": min(rates), "max_it_per_s": max(rates),
                                  "mean_inner_steps": inner_mean,
                                  "schedule": f"<=100 IRLS steps per association, f_tol={REF_F_TOL:g} "
                                              "(the C++ layer's and the reference's default); device-paced inner loop"}
        # Where the timed windows sit on the registration's way to its fixed point, and what the same kernels do once it is
        # there.  The windows time iterations warmup+1 .. warmup+steps after a fresh upload: the source still moves by
        # ~1e-2 .. 1e-3 radii per iteration there, so part of the workgroups search the grid again every iteration (their
        # rows' Verlet lists no longer provably hold every possible neighbour); `searched_share` is that part, counted by
        # the kernel itself.  `converged`: 3 x steps further iterations of the SAME registration after 60 more (moves
        # < 1e-4 radii): every row is answered from its list.
        try:
            # (cost_drop_thresh = -1: a converged registration's cost drop is 0 up to rounding, sometimes below: the rule
            #  must not end these calls early)
            ctx.set_source(src)
            if a.warmup > 0:
                ctx.align(a.warmup, cost_drop_thresh=-1.0, inner_steps=a.inner_steps, want_history=False)
            v0 = ctx.debug_verlet()
            ctx.align(a.steps, cost_drop_thresh=-1.0, inner_steps=a.inner_steps, want_history=False)
            v1 = ctx.debug_verlet()
            ctx.align(60, cost_drop_thresh=-1.0, inner_steps=a.inner_steps, want_history=False)
            ctx.synchronize()
            v2 = ctx.debug_verlet()
            crates = []
            for _ in range(3):
                tc0 = time.perf_counter()
                cres = ctx.align(3 * a.steps, cost_drop_thresh=-1.0, inner_steps=a.inner_steps, want_history=False)
                ctx.synchronize()
                assert int(cres["n_iter"]) == 3 * a.steps, f"early stop: {cres['n_iter']}"
                crates.append(3 * a.steps / (time.perf_counter() - tc0))
            v3 = ctx.debug_verlet()
            real = max(1, v1["workgroups"] - 128)
            out["steady_state"] = {
                "verlet_lists": bool(v1["rows"] > 0 and v1["trusted"]),
                "window": {"searched_share": (v1["rebuilt"] - v0["rebuilt"]) / (a.steps * real) if v1["rows"] else None,
                           "mean_list_length": v1["mean_list"] if v1["rows"] else None},
                "converged": {"it_per_s": float(np.median(crates)), "min_it_per_s": min(crates), "max_it_per_s": max(crates),
                              "searched_share": (v3["rebuilt"] - v2["rebuilt"]) / (9 * a.steps * real) if v3["rows"] else None,
                              "after_iterations": a.warmup + a.steps + 60},
                "note": "value (the headline) is the median window; converged.it_per_s is the same registration, same "
                        "kernels and schedule, once the source has stopped moving — never part of value"}
        except Exception as e:   # (a diagnostic block must not cost the line)
            out["steady_state"] = {"error": str(e)}
        # A SECOND trajectory through the same windows (round-5 review, task 6): the same target, permutation and noise draws
        # with three times the ground-truth motion and three times the noise — when the lists engage is a forecast from the
        # moves the registration makes (verlet_lists_pay_off), not a constant fitted to the headline pair's convergence ratio.
        try:
            src2 = synth.make_pair(n, cfg=synth.CONFIGS[a.config].get("clouds", a.config), stride=src.shape[1], motion_scale=3.0, noise_scale=3.0)[0]
            rates2, shares2 = [], []
            for _ in range(3):
                ctx.set_source(src2)
                if a.warmup > 0:
                    ctx.align(a.warmup, cost_drop_thresh=-1.0, inner_steps=a.inner_steps, want_history=False)
                ctx.synchronize()
                w0 = ctx.debug_verlet()
                t2 = time.perf_counter()
                r2 = ctx.align(a.steps, cost_drop_thresh=-1.0, inner_steps=a.inner_steps, want_history=False)
                ctx.synchronize()
                rates2.append(a.steps / (time.perf_counter() - t2))
                assert int(r2["n_iter"]) == a.steps
                w1 = ctx.debug_verlet()
                shares2.append((w1["rebuilt"] - w0["rebuilt"]) / (a.steps * max(1, w1["workgroups"] - 128)) if w1["rows"] else None)
            if isinstance(out.get("steady_state"), dict) and "error" not in out["steady_state"]:
                out["steady_state"]["second_trajectory"] = {
                    "it_per_s": float(np.median(rates2)), "min_it_per_s": min(rates2), "max_it_per_s": max(rates2),
                    "searched_share": shares2[-1],
                    "ratio_to_value": (float(np.median(rates2)) / out["value"]) if out.get("value") else None,
                    "what": "the same windows on the same pair with 3 x the ground-truth motion and 3 x the noise"}
        except Exception as e:
            if isinstance(out.get("steady_state"), dict):
                out["steady_state"]["second_trajectory"] = {"error": str(e)}
        # one whole registration the way the command line runs it by default, HOST BUFFERS IN: uploads, grid build,
        # source sort, first association and the loop until hasConverged() stops it (-c 0.01 -n 5, inner loop to
        # function_tolerance) — what a caller of align() waits for; the handle is warm (its buffers exist)
        ttc = []
        with _lib.Context(local_rank) as tc:
            tc.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
            for rep in range(4):
                t0 = time.perf_counter()
                tc.set_target(tgt)
                tc.set_source(src)
                res = tc.align(1000, cost_drop_thresh=0.01, n_cost_drop_it=5, inner_steps=100, f_tol=REF_F_TOL, want_history=False)
                tc.synchronize()
                if rep > 0:   # (the first repetition grows the handle's buffers)
                    ttc.append((1e3 * (time.perf_counter() - t0), int(res["n_iter"])))
        out["time_to_converge_ms"] = {"value": float(np.median([t for t, _ in ttc])), "iterations": ttc[0][1],
                                      "all": [t for t, _ in ttc],
                                      "schedule": "host buffers in; the CLI's defaults for the stopping rule (-c 0.01 -n 5, "
                                                  "<= 1000 iterations) and the inner loop (<= 100 IRLS steps, f_tol 1e-5); warm handle"}
        # the drop-in surface itself: ProbPointCloudRegistration::align() (one call into ppcr_align_report) timed in a
        # child process running the C++ API test program on the same clouds: -c 0 -i (warmup + steps), once with one
        # inner step per association, once with the class's default (inner loop to function_tolerance)
        if not a.no_cpp_api and not batch_cfg:
            out["cpp_api"] = cpp_api_block(src, tgt, cfg, a)

    if world == 1 and a.config == 3 and not a.no_extras and not a.fake_register and a.n is None:
        # The reference's OWN defaults (prob_point_cloud_registration_ex.cc:43-56: radius 3, 20 neighbours, inner loop to
        # function_tolerance) beside the headline, so that every default run times them: bench.py --config 8 / 9 / 10 in
        # short — a uniform 200k cloud (two-pass search) and the two pinned non-uniform scenes (multi-level search); three
        # windows of `steps` iterations each after a settle phase, median, same timing rules as `value`.
        shapes = {}
        for cid in (8, 9, 10):
            try:
                ccfg = synth.CONFIGS[cid]
                s_c, t_c, _, _ = synth.make_config(cid, pair=0)
                with _lib.Context(local_rank) as cc:
                    for kv in a.opt:
                        k, v = kv.split("=")
                        cc.set_option(k, int(v))
                    cc.set_params(ccfg["radius"], ccfg["max_neighbours"], ccfg["dof"], 3)
                    cc.set_target(t_c)
                    cc.set_source(s_c)
                    inner = int(ccfg.get("inner_steps", 1))
                    ts = time.perf_counter()
                    while (time.perf_counter() - ts) * 1e3 < 120.0:     # settle
                        cc.align(50, cost_drop_thresh=-1.0, inner_steps=inner, want_history=False)
                    rates = []
                    for _ in range(3):
                        cc.set_source(s_c)
                        if a.warmup > 0:
                            cc.align(a.warmup, cost_drop_thresh=-1.0, inner_steps=inner, want_history=False)
                        cc.synchronize()
                        tw = time.perf_counter()
                        r_c = cc.align(a.steps, cost_drop_thresh=-1.0, inner_steps=inner, want_history=False)
                        cc.synchronize()
                        rates.append(a.steps / (time.perf_counter() - tw))
                        assert int(r_c["n_iter"]) == a.steps, f"early stop: {r_c['n_iter']}"
                    shapes[f"config_{cid}"] = {"it_per_s": float(np.median(rates)), "min_it_per_s": min(rates), "max_it_per_s": max(rates),
                                              "points": [int(s_c.shape[0]), int(t_c.shape[0])],
                                              "cloud": ccfg.get("scene", "uniform"), "radius": ccfg["radius"],
                                              "max_neighbours": ccfg["max_neighbours"], "inner_steps_max": inner}
            except Exception as e:   # (a side block must not cost the line)
                shapes[f"config_{cid}"] = {"error": str(e)}
        shapes["note"] = ("the command line's default parameters (radius 3, max_neighbours 20, inner IRLS loop to function_tolerance) "
                          "on 200k-point clouds: uniform / LiDAR-like scene / slab with dense blobs; median of 3 windows; never part of value")
        out["cli_default_shapes"] = shapes

    if batch_cfg or world > 1 or dist_was:
        ok = np.isfinite(gathered).all(axis=(1, 2))
        out["gathered_transforms"] = int(ok.sum())
        assert out["gathered_transforms"] == n_pairs, f"gathered {int(ok.sum())} of {n_pairs} transforms"
        if batch_cfg and world == 1 and not a.no_extras:
            # pairs per second END TO END: host buffers in (upload, grid build, source sort, K iterations), several
            # pairs in flight per GPU so that one pair's uploads overlap another pair's iterations (ppcr_batch_run)
            host_pairs = [synth.make_config(a.config, pair=p, n=n)[:2] for p in range(min(n_pairs, 64))]   # (64 x 250k: 0.5 GB of host memory)
            for _ in range(2):   # warm-up: ppcr_batch_run keeps its handles, four of them have grown their buffers after this
                _lib.batch_run(host_pairs[:8], cfg["radius"], cfg["max_neighbours"], cfg["dof"], n_iter=a.steps + a.warmup,
                               inner_steps=a.inner_steps, device_ids=(local_rank,), lanes_per_device=4)
            e2e = {}
            for lanes in (1, 2, 4, 6):
                best = 0.0
                for _ in range(2):
                    t0 = time.perf_counter()
                    _lib.batch_run(host_pairs, cfg["radius"], cfg["max_neighbours"], cfg["dof"], n_iter=a.steps + a.warmup,
                                   inner_steps=a.inner_steps, device_ids=(local_rank,), lanes_per_device=lanes)
                    best = max(best, len(host_pairs) / (time.perf_counter() - t0))
                e2e[f"lanes_{lanes}"] = best
            out["pairs_per_s_end_to_end"] = dict(e2e, pairs=len(host_pairs), iterations_per_pair=a.steps + a.warmup,
                                                 note="ppcr_batch_run on one GPU, host buffers in: upload + grid build + "
                                                      "source sort + iterations per pair, `lanes` pairs in flight (best of two "
                                                      "calls each); its pooled handles are warm (two untimed batches before)")
        if batch_cfg and not a.no_verify:
            # every gathered transform against a single-rank, single-stream run of the same pair and schedule
            worst = 0.0
            with _lib.Context(local_rank) as chk:
                chk.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
                for p in range(n_pairs):
                    s, t, _, _ = synth.make_config(a.config, pair=p, n=n)
                    chk.set_target(t)
                    chk.set_source(s)
                    # (the schedule of the timed windows: same stopping threshold, same function_tolerance)
                    if a.warmup > 0:
                        chk.align(a.warmup, cost_drop_thresh=timed_thresh, inner_steps=a.inner_steps, want_history=False)
                    one_res = chk.align(a.steps, cost_drop_thresh=timed_thresh, inner_steps=a.inner_steps)
                    assert one_res["n_iter"] == a.steps, f"pair {p}: early stop in the re-run: {one_res['n_iter']}"
                    one = one_res["history"][-1]
                    worst = max(worst, float(np.abs(one - gathered[p]).max()))
            out["batch_verification"] = {"pairs_checked": n_pairs, "max_abs_diff_vs_single_rank_run": worst}
            assert worst < 1e-9, f"a gathered transform differs from its single-rank run by {worst}"
            # ... and four sampled pairs against the ORACLE (the comparison above is GPU against GPU): same clouds, same
            # schedule, transform within the north-star's 1e-5 rad / 1e-5 m
            if rank == 0 and not a.no_cpu_baseline:
                from oracle import binding as po  # noqa: F811  (checker side)
                sample = sorted({0, n_pairs // 3, (2 * n_pairs) // 3, n_pairs - 1})
                rot_w = tr_w = 0.0
                for p in sample:
                    s, t, _, _ = synth.make_config(a.config, pair=p, n=n)
                    hist = po.align(s, t, cfg["radius"], cfg["max_neighbours"], cfg["dof"], a.warmup + a.steps,
                                    cost_drop_thresh=timed_thresh, inner_max_steps=a.inner_steps)["history"]
                    # (the gathered transform is the timed call's own: what came after the warm-up iterations)
                    to4 = lambda T: np.vstack([T, [0.0, 0.0, 0.0, 1.0]])
                    ora = (to4(hist[-1]) @ np.linalg.inv(to4(hist[a.warmup - 1]) if a.warmup > 0 else np.eye(4)))[:3]
                    rot_w = max(rot_w, float(synth.rotation_angle(gathered[p][:, :3], ora[:, :3])))
                    tr_w = max(tr_w, float(np.linalg.norm(gathered[p][:, 3] - ora[:, 3])))
                out["batch_verification"]["oracle_sample"] = {"pairs": sample, "rot_err_rad": rot_w, "trans_err_m": tr_w}
                assert rot_w < 1e-5 and tr_w < 1e-5, f"a gathered transform differs from the oracle's: {rot_w} rad, {tr_w} m"

    if world == 1 and not a.no_cpu_baseline:
        from oracle import binding as po  # noqa: F401  (checker side)
        cb, ora = cpu_baseline(src, tgt, cfg, a.cpu_iters, a.inner_steps, timed_thresh)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        if not a.no_extras:
            out["cpu_baseline_refshape"] = cpu_baseline_refshape(src, tgt, cfg)
            if "converged_inner" in out:
                out["converged_inner"]["speedup_vs_cpu_baseline_refshape"] = (
                    out["converged_inner"]["it_per_s"] / out["cpu_baseline_refshape"]["value"])
        # parity attached to the timing: GPU vs oracle after the same number of iterations from the same start
        with _lib.Context(local_rank) as chk:
            chk.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
            chk.set_target(tgt)
            chk.set_source(src)
            n_par = len(ora["history"])
            g = chk.align(n_par, cost_drop_thresh=timed_thresh, inner_steps=a.inner_steps)  # the oracle ran the same schedule
        assert g["n_iter"] == n_par, (g["n_iter"], n_par)
        out["parity"] = {"iterations": n_par,
                         "rot_err_rad": synth.rotation_angle(g["history"][-1][:, :3], ora["history"][-1][:, :3]),
                         "trans_err_m": float(np.linalg.norm(g["history"][-1][:, 3] - ora["history"][-1][:, 3])),
                         "vs": "oracle (CPU restatement); the reference itself cannot be built (PCL/Ceres absent)"}
    # the JSON line is the LAST line of stdout: everything that may still print (RCCL's version banner sits in the C
    # library's stdout buffer until exit) is shut down and flushed first
    for c in ctxs:
        c.close()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out))
    sys.stdout.flush()
    return 0


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return spawn_ranks(a)
    return run_rank(a)


if __name__ == "__main__":
    sys.exit(main())
