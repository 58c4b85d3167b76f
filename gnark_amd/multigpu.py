"""Multi-GPU MSM: one process per GPU, partial results exchanged with `torch.distributed` (backend "nccl" = RCCL over
xGMI on the GPU box; "gloo" in CPU tests).  The reference has no multi-GPU path -- this is new:

  partition B (base-point range): rank g owns slice g of the bases and scalars, runs a complete Pippenger MSM on it and
      contributes ONE Jacobian point; all_gather of world x {96,144,192,288} bytes, then a local sum.
  partition A (scalar windows):   every rank holds all bases/scalars, rank g accumulates windows [lo_g, hi_g) only and
      contributes its window sums; all_gather, then Horner over all windows.

RCCL has no user-defined reduction, so "all-reduce of partial bucket sums" is all_gather + local group additions (the
payload is a few hundred bytes per MSM -- latency-bound, bandwidth irrelevant)."""
from __future__ import annotations

import numpy as np

from . import ecc
from .device import curve_id, jac_words


def shard_range(n: int, rank: int, world: int):
    """contiguous slice [lo, hi) of n items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_u64(arr: np.ndarray, dist, device=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy().view(np.uint64).reshape(arr.shape) for o in out]


def combine_partials(curve, group: int, partials, lib=None) -> np.ndarray:
    acc = np.ascontiguousarray(partials[0], dtype=np.uint64)
    for p in partials[1:]:
        acc = ecc.jac_add(curve, group, acc, p, lib=lib)
    return acc


def msm_base_sharded(ctx, curve, group: int, points_shard, scalars_shard, n_shard: int, dist, device=None) -> np.ndarray:
    """partition B: every rank passes ITS slice; every rank returns the full MSM (Jacobian).
    points_shard may be an ecc.PrecomputedBases built from the rank's slice (pinned key with window tables)."""
    if isinstance(points_shard, ecc.PrecomputedBases):
        part = points_shard.MultiExp(scalars_shard)
    else:
        part = ecc.MultiExp(ctx, curve, group, points_shard, scalars_shard, n=n_shard)
    if dist is None or dist.get_world_size() == 1:
        return part
    return combine_partials(curve, group, _all_gather_u64(part, dist, device), lib=ctx.lib)


def msm_window_sharded(ctx, curve, group: int, points, scalars, n: int, dist, device=None) -> np.ndarray:
    """partition A: every rank passes ALL points/scalars and accumulates only its share of the Pippenger windows."""
    cid = curve_id(curve)
    cbits, nwin = ecc.plan(curve, group, n, lib=ctx.lib)
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    lo, hi = shard_range(nwin, rank, world)
    mine = np.zeros((0, jac_words(cid, group)), dtype=np.uint64)
    if hi > lo:
        mine, _, _ = ecc.MultiExpWindows(ctx, curve, group, points, scalars, n, lo, hi)
    if world == 1:
        return ecc.combine_windows(curve, group, mine, cbits, lib=ctx.lib)
    # pad to the largest share so that all_gather sees equal shapes
    share = (nwin + world - 1) // world
    padded = np.zeros((share, jac_words(cid, group)), dtype=np.uint64)
    padded[: hi - lo] = mine
    gathered = _all_gather_u64(padded, dist, device)
    windows = np.concatenate([g[: shard_range(nwin, r, world)[1] - shard_range(nwin, r, world)[0]] for r, g in enumerate(gathered)])
    return ecc.combine_windows(curve, group, windows, cbits, lib=ctx.lib)


def groth16_prove_sharded(pk, solution, nb_public: int, r, s, dist, device=None):
    """One proof over a key sharded by base-point range (groth16.ProvingKey(..., shard=(rank, world))): every rank uploads
    the solution, computes H redundantly (20 ms at 2^24, cheaper than shipping 512 MiB of h over xGMI) and runs the five MSMs
    over its slices; one all_gather of 3 G1Jac + 1 G2Jac per rank, then every rank finishes identically."""
    from . import groth16
    part = groth16.ProvePartial(pk, solution, nb_public)
    if dist is not None and dist.get_world_size() > 1:
        part = groth16.SumPartials(pk.curve, _all_gather_u64(part, dist, device), lib=pk.ctx.lib)
    return groth16.Finish(pk, part, r, s)
