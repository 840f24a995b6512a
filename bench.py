#!/usr/bin/env python3
"""Benchmark of the registration hot path on MI355X (contract: see the task statement / DESIGN.md §7).

A "step" is one outer registration iteration over one source/target pair:
    radius-NN association (K1) -> t/Gaussian weights + weighted moments (K23) -> host 3x3 SVD
    -> in-place move of the source (K4),
i.e. BASELINE.json's metric "registration iterations/sec (1M<->1M pts, r=1.0, m=10)" with one inner
IRLS step per association, early termination disabled (cost_drop_thresh = 0).
Clouds are resident in HBM before the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank registers its own
independent 1M<->1M pair (weak scaling, no data-path collective); the only collective is the final
all_gather of the 3x4 transforms over RCCL, inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from probabilistic_point_clouds_registration_amd import _lib, batch, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (3 = 1M headline, 4 = Gaussian)")
    ap.add_argument("--n", type=int, default=None, help="override the cloud size (debugging)")
    ap.add_argument("--inner-steps", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=0,
                    help="outer iterations of the CPU baseline sample (0 = auto: about 6 s of wall time, 3..30)")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event per-kernel pass")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank, to exercise the collective path")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="ppcr_set_option knob applied to every handle (experiments; the default run sets none)")
    ap.add_argument("--lanes", type=int, default=4,
                    help="pairs in flight per GPU when --pairs-per-gpu > 1 (ppcr_align_many host worker threads)")
    ap.add_argument("--pairs-per-gpu", type=int, default=1,
                    help="independent pairs each rank registers back to back (BASELINE configs[4]: --config 5 --pairs-per-gpu 8)")
    return ap.parse_args()


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


def cpu_baseline(src, tgt, cfg, iters, inner_steps):
    """The oracle (kind 'port': OpenMP-generous variant — grid built per call, all loops parallel) timed on
    this box's host cores on a bounded sample: `iters` outer iterations of the SAME workload."""
    from oracle import binding as po  # checker-side import, only on this leg
    cores, visible, quota = effective_cores()
    threads = max(1, min(po.num_threads(), cores))
    tw = time.perf_counter()
    po.align(src, tgt, cfg["radius"], cfg["max_neighbours"], cfg["dof"], 1, inner_max_steps=inner_steps,
             threads=threads)  # warm-up (page-in, thread pool)
    tw = time.perf_counter() - tw
    if iters <= 0:
        iters = int(min(30, max(3, round(6.0 / max(tw, 1e-3)))))
    t0 = time.perf_counter()
    res = po.align(src, tgt, cfg["radius"], cfg["max_neighbours"], cfg["dof"], iters,
                   inner_max_steps=inner_steps, threads=threads)
    dt = time.perf_counter() - t0
    iters = len(res["history"])   # iterations really performed (hasConverged may stop a converged run early)
    return dict(value=iters / dt, unit="iterations/s", cores=threads, kind="port",
                sample=f"{iters} outer iterations of the same {src.shape[0]}<->{tgt.shape[0]} workload "
                       f"(oracle/ppcr_oracle.c, OpenMP x{threads}, grid NN; host shows {visible} hardware threads, "
                       f"container CPU quota {quota if quota else 'none'})",
                seconds=dt), res


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch  # device selection, synchronize, torch.distributed (RCCL)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    cfg = dict(synth.CONFIGS[a.config])
    n = a.n or cfg["n"]
    # weak scaling: rank r registers its own pair(s): pair p lives on rank p % world (batch.shard_pairs)
    n_pairs = world * a.pairs_per_gpu
    my_pairs = batch.shard_pairs(n_pairs, world, rank)
    ctxs = []
    for p in my_pairs:
        src, tgt, Rgt, tgt_t = synth.make_pair(n, cfg=a.config, pair=p)
        c = _lib.Context(local_rank)
        for kv in a.opt:
            k, v = kv.split("=")
            c.set_option(k, int(v))
        c.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
        c.set_target(tgt)
        c.set_source(src)
        ctxs.append(c)
    ctx = ctxs[0]
    src, tgt, Rgt, tgt_t = synth.make_pair(n, cfg=a.config, pair=my_pairs[0])   # rank 0's first pair (cpu baseline/parity)

    def barrier():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # warm-up (also builds the grid and sorts the source once)
    if a.warmup > 0:
        if len(ctxs) > 1 and a.lanes > 1:
            # same concurrency as the timed region: the runtime creates its extra hardware queues on first
            # concurrent use (a one-off ~50 ms stall measured when this ran sequentially)
            _lib.align_many(ctxs, a.warmup, lanes=a.lanes, cost_drop_thresh=0.0, inner_steps=a.inner_steps)
        else:
            for c in ctxs:
                c.align(a.warmup, cost_drop_thresh=0.0, inner_steps=a.inner_steps, want_history=False)
    if dist is not None:
        # the collective is warmed up too (communicator set-up and its buffers are not part of a step)
        batch.gather_transforms({p: np.eye(4)[:3] for p in my_pairs}, n_pairs, dist=dist,
                                device=torch.device("cuda", local_rank))
    barrier()
    gathered = None
    t0 = time.perf_counter()
    local = {}
    if len(ctxs) > 1 and a.lanes > 1:
        # several resident pairs per GPU: a.lanes of them in flight, each on its own handle/stream
        T_fin, done = _lib.align_many(ctxs, a.steps, lanes=a.lanes, cost_drop_thresh=0.0, inner_steps=a.inner_steps)
        assert all(int(d) == a.steps for d in done), f"early stop inside the timed region: {list(done)}"
        for k, p in enumerate(my_pairs):
            local[p] = T_fin[k]
    else:
        for p, c in zip(my_pairs, ctxs):
            res = c.align(a.steps, cost_drop_thresh=0.0, inner_steps=a.inner_steps)
            assert res["n_iter"] == a.steps, f"early stop inside the timed region: {res['n_iter']}"
            local[p] = res["history"][-1]
    if dist is not None:
        # RCCL: the only collective of the job — final gather of the transforms (batch.gather_transforms)
        gathered = batch.gather_transforms(local, n_pairs, dist=dist, device=torch.device("cuda", local_rank))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-kernel durations with HIP events on the handle's own stream (separate pass so that the event
    # records cannot perturb the headline number; same workload, same K)
    prof = {}
    nnz = ctx.association_size()[1]
    if not a.no_profile:
        ctx.profile_enable(True)
        tp0 = time.perf_counter()
        ctx.align(a.steps, cost_drop_thresh=0.0, inner_steps=a.inner_steps, want_history=False)
        ctx.synchronize()
        tp = time.perf_counter() - tp0
        prof = ctx.profile_get()
        ctx.profile_enable(False)
        prof["_profiled_pass_ms_per_step"] = 1e3 * tp / a.steps

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ns, nt = src.shape[0], tgt.shape[0]
    out = {
        # BASELINE.json's metric; the label follows the cloud size actually run (configs other than the headline)
        "metric": "registration iterations/sec (%s<->%s pts, r=%.1f, m=%d)" % (
            (("1M", "1M") if ns == nt == 1000000 else (ns, nt)) + (cfg["radius"], cfg["max_neighbours"])),
        "value": n_pairs * a.steps / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / (a.steps * a.pairs_per_gpu),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (distances) + f64 (weights, moments, solve)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{a.config - 1}]: {ns}<->{nt} synthetic clouds, radius={cfg['radius']}, "
                               f"max_neighbours={cfg['max_neighbours']}, "
                               f"{'Gaussian' if np.isinf(cfg['dof']) else 't dof=%g' % cfg['dof']}, "
                               f"{a.inner_steps} inner IRLS step(s)/iteration, cost_drop_thresh=0",
                   "pairs": n_pairs, "lanes_per_gpu": (a.lanes if a.pairs_per_gpu > 1 else 1),
                   "parallelism": f"{n_pairs} independent pair(s), {a.pairs_per_gpu} per GPU on {world} GPU(s); "
                                                     "no data-path collective; final RCCL all_gather of the transforms"},
        "nnz": int(nnz),
    }
    # roofline of the dominant kernel (K1, nn_topm_kernel): algorithmic bytes B_nn = 16*Ns + 12*Nt + 4*nnz
    #  / average launch duration measured with HIP events above
    b_nn = 16.0 * ns + 12.0 * nt + 4.0 * nnz
    b_iter = 72.0 * ns + 12.0 * nt + 52.0 * nnz + (a.inner_steps - 1) * (32.0 * ns + 48.0 * nnz)
    if "nn_topm_kernel" in prof:
        k = prof["nn_topm_kernel"]
        avg_ms = k["total_ms"] / max(1, k["launches"])
        ach = b_nn / (avg_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from PMC counters: a separate rocprofv3 --pmc run (profiles/k1_traffic.json)
        tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(tpath) and a.config in (3, 4) and a.n is None:
            traffic = json.load(open(tpath)).get("traffic_bytes_per_launch")
        out["roofline"] = {"bound": "hbm", "kernel": "nn_tile_kernel (K1)", "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "avg_kernel_ms": avg_ms, "algorithmic_bytes_per_launch": b_nn,
                           "candidate_tests_per_s": 27 * 3.8147 * ns / (avg_ms * 1e-3)}
    else:
        out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                           "traffic": None}
    out["iteration_roofline"] = {"algorithmic_bytes_per_iteration": b_iter,
                                 "achieved_GBs_per_gpu": b_iter * a.steps * a.pairs_per_gpu / dt / 1e9,
                                 "frac_of_hbm_peak": b_iter * a.steps * a.pairs_per_gpu / dt / 1e9 / HBM_PEAK_GBS}
    out["kernels_ms_per_launch"] = {k: v["total_ms"] / max(1, v["launches"]) for k, v in prof.items()
                                    if isinstance(v, dict)}
    if "_profiled_pass_ms_per_step" in prof:
        out["profiled_pass_ms_per_step"] = prof["_profiled_pass_ms_per_step"]

    if world == 1 and not a.no_cpu_baseline:
        cb, ora = cpu_baseline(src, tgt, cfg, a.cpu_iters, a.inner_steps)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        # parity attached to the timing: GPU vs oracle after the same number of iterations from the same start
        chk = _lib.Context(local_rank)
        chk.set_params(cfg["radius"], cfg["max_neighbours"], cfg["dof"], 3)
        chk.set_target(tgt)
        chk.set_source(src)
        n_par = int(ora["n_iter"]) if "n_iter" in ora else len(ora["history"])
        g = chk.align(n_par, cost_drop_thresh=0.0, inner_steps=a.inner_steps)
        chk.close()
        assert g["n_iter"] == len(ora["history"]), (g["n_iter"], len(ora["history"]))
        out["parity"] = {"iterations": n_par,
                         "rot_err_rad": synth.rotation_angle(g["history"][-1][:, :3], ora["history"][-1][:, :3]),
                         "trans_err_m": float(np.linalg.norm(g["history"][-1][:, 3] - ora["history"][-1][:, 3])),
                         "vs": "oracle (CPU restatement); the reference itself cannot be built (PCL/Ceres absent)"}
    elif world > 1:
        out["gathered_transforms"] = int(np.isfinite(gathered).all(axis=(1, 2)).sum()) if gathered is not None else 0
    print(json.dumps(out))
    sys.stdout.flush()
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
