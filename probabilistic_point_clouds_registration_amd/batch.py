"""Multi-pair batched registration: independent source/target pairs sharded over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" for the CPU tests).  Pairs
are independent, so there is NO data-path collective: pair p is registered entirely on rank p % world.
The only communication is the final gather of the 3x4 transforms (16-f64-per-pair class payload, a few KB:
latency-bound, link bandwidth irrelevant) — `gather_transforms` below, one all_gather.
"""
import numpy as np


def shard_pairs(n_pairs, world_size, rank):
    """Indices of the pairs this rank registers (round-robin: pair p -> rank p % world_size)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, n_pairs, world_size))


def register_pair_hip(src, tgt, params, device_id=0, n_iter=20, inner_steps=1, cost_drop_thresh=0.0, n_cost_drop_it=5):
    """Register one pair on one GPU through the C ABI; returns the final cumulative 3x4 transform."""
    from . import _lib
    with _lib.Context(device_id) as ctx:
        ctx.set_params(params["radius"], params["max_neighbours"], params["dof"], 3)
        ctx.set_target(tgt)
        ctx.set_source(src)
        res = ctx.align(n_iter, cost_drop_thresh=cost_drop_thresh, n_cost_drop_it=n_cost_drop_it, inner_steps=inner_steps)
    return res["history"][-1] if res["n_iter"] > 0 else np.eye(4)[:3]


def register_local_pairs_hip(make_pair, n_pairs, world_size, rank, device_id=0, lanes=2, n_iter=20, inner_steps=1,
                             cost_drop_thresh=0.0, n_cost_drop_it=5):
    """This rank's share through ppcr_batch_run: `lanes` pairs in flight on the rank's GPU, the next pairs' uploads,
    grid builds and source sorts prepared on a second thread meanwhile. All pairs must share one parameter set. -> {pair: 3x4}"""
    from . import _lib
    mine = shard_pairs(n_pairs, world_size, rank)
    if not mine:
        return {}
    clouds, params = [], None
    for p in mine:
        src, tgt, prm = make_pair(p)
        if params is not None and prm != params:
            raise ValueError("ppcr_batch_run takes one parameter set per batch")
        params = prm
        clouds.append((src, tgt))
    T, _ = _lib.batch_run(clouds, params["radius"], params["max_neighbours"], params["dof"], n_iter=n_iter,
                          cost_drop_thresh=cost_drop_thresh, n_cost_drop_it=n_cost_drop_it, inner_steps=inner_steps,
                          device_ids=(device_id,), lanes_per_device=lanes)
    return {p: T[k] for k, p in enumerate(mine)}


def register_local_pairs(make_pair, n_pairs, world_size, rank, register=register_pair_hip, **kw):
    """Run this rank's share. make_pair(p) -> (src, tgt, params). Returns {pair index: 3x4 transform}."""
    out = {}
    for p in shard_pairs(n_pairs, world_size, rank):
        src, tgt, params = make_pair(p)
        out[p] = np.asarray(register(src, tgt, params, **kw), dtype=np.float64).reshape(3, 4)
    return out


def gather_transforms(local, n_pairs, dist=None, device=None):
    """All ranks end up with the [n_pairs, 3, 4] array of transforms.

    local: {pair index: 3x4}. dist: an initialised torch.distributed module (None => single process).
    device: torch device for the collective buffers ("cuda:k" with nccl/RCCL, "cpu" with gloo).
    """
    result = np.full((n_pairs, 3, 4), np.nan)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        for p, T in local.items():
            result[p] = T
        return result
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (n_pairs + world - 1) // world
    buf = torch.full((per_rank, 13), float("nan"), dtype=torch.float64)
    for k, p in enumerate(shard_pairs(n_pairs, world, rank)):
        buf[k, 0] = float(p)
        buf[k, 1:] = torch.from_numpy(np.asarray(local[p], dtype=np.float64).reshape(12))
    if device is not None:
        buf = buf.to(device)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)          # the one collective of the batched path
    for g in gathered:
        g = g.cpu().numpy()
        for row in g:
            if not np.isnan(row[0]):
                result[int(row[0])] = row[1:].reshape(3, 4)
    return result
