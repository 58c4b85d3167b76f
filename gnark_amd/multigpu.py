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
    if device is not None and dist.get_backend() != "gloo":
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy().view(np.uint64).reshape(arr.shape) for o in out]


def _via_cpu(t, dist) -> bool:
    """gloo moves CPU tensors only: on a GPU box without RCCL peers (two ranks sharing one GPU in a smoke run) device tensors
    are staged through the host; with the nccl (= RCCL) backend they travel over xGMI directly"""
    return t.is_cuda and dist.get_backend() == "gloo"


def _send(t, dst, dist):
    dist.send(t.cpu() if _via_cpu(t, dist) else t, dst=dst)


def _recv(t, src, dist):
    if _via_cpu(t, dist):
        tmp = t.cpu()
        dist.recv(tmp, src=src)
        t.copy_(tmp)
    else:
        dist.recv(t, src=src)


def _broadcast(t, src, dist):
    if _via_cpu(t, dist):
        tmp = t.cpu()
        dist.broadcast(tmp, src=src)
        t.copy_(tmp)
    else:
        dist.broadcast(t, src=src)


def _gather(t, parts, dst, dist):
    if _via_cpu(t, dist):
        tmp = [p.cpu() for p in parts] if parts is not None else None
        dist.gather(t.cpu(), tmp, dst=dst)
        if parts is not None:
            for p, q in zip(parts, tmp):
                p.copy_(q)
    else:
        dist.gather(t, parts, dst=dst)


def _scatter(out, pieces, src, dist):
    if _via_cpu(out, dist):
        tmp = out.cpu()
        dist.scatter(tmp, [p.cpu() for p in pieces] if pieces is not None else None, src=src)
        out.copy_(tmp)
    else:
        dist.scatter(out, pieces, src=src)


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
    """partition A: every rank passes ALL points/scalars and accumulates only its share of the Pippenger windows.
    With pinned bases (ecc.PrecomputedBases holding the whole vector) the windows run on the table path
    (ga_msm_table_run_windows) and the partial results simply add -- no Horner step."""
    if isinstance(points, ecc.PrecomputedBases):
        world = 1 if dist is None else dist.get_world_size()
        rank = 0 if dist is None else dist.get_rank()
        lo, hi = shard_range(points.info()["windows"], rank, world)
        part = points.MultiExpWindows(scalars, lo, hi)
        if world == 1:
            return part
        return combine_partials(curve, group, _all_gather_u64(part, dist, device), lib=ctx.lib)
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


def groth16_prove_sharded(pk, solution, nb_public: int, r, s, dist, device=None, replicate_h=False, replicate_uploads=False):
    """One proof over a key sharded by base-point range (groth16.ProvingKey(..., shard=(rank, world))), one process per GPU.

    Serial work is not replicated: every rank uploads only the wire range of W its bases cover; the three chains of computeH
    (FFT_coset(iFFT(.)) of the solver's A, B, C) run on ranks 0, 1, 2 -- three uploads over three PCIe links, on a helper thread
    BESIDE the rank's witness MSMs (second lane of the context) -- and travel to rank 0 over xGMI (send/recv, 32 B x n each);
    rank 0 finishes h and scatters the slices (32 B x n / world per peer); every rank runs the MSM over its slice of pk.G1.Z;
    one all_gather of 3 G1Jac + 1 G2Jac per rank; every rank finishes identically.
    replicate_h=True keeps the round-1 scheme (every rank recomputes h) for comparison; replicate_uploads=True makes the chain
    owners upload their whole vectors themselves even with 3+ ranks."""
    from . import groth16
    world = 1 if dist is None else dist.get_world_size()
    if world == 1 or replicate_h:
        part = groth16.ProvePartial(pk, solution, nb_public)
        if world > 1:
            part = groth16.SumPartials(pk.curve, _all_gather_u64(part, dist, device), lib=pk.ctx.lib)
        return groth16.Finish(pk, part, r, s)
    import threading
    import torch
    rank = dist.get_rank()
    lay = groth16.ShardLayout(pk)
    n = lay["n"]
    dev = device if device is not None else torch.device("cpu")
    sync = (lambda: torch.cuda.synchronize(dev)) if device is not None else (lambda: None)
    owner = [0, 1, 2 if world >= 3 else 0]
    vecs = [solution.A, solution.B, solution.C]
    windowed = lay["win_count"] > 1          # window-sharded key: every rank needs all of h
    share = (n - 1 + world - 1) // world + 1
    # device buffers as torch tensors, so that the collective library can move them; the prover library gets raw pointers
    bufs = {}
    for k in range(3):
        if rank == owner[k] or rank == 0:
            bufs[k] = torch.empty((n, 4), dtype=torch.int64, device=dev)
    if windowed and 0 not in bufs:
        bufs[0] = torch.empty((n, 4), dtype=torch.int64, device=dev)
    sync()
    state = {"pieces": None, "error": None}

    nc = int(np.asarray(solution.A).shape[0])                 # constraints: the length of the solver's A, B, C
    cshare = (nc + world - 1) // world                         # every rank uploads rows [rank*cshare, ...) of A, B and C
    sliced = world >= 3 and not replicate_uploads              # (with one or two ranks the chain owners upload whole vectors)

    def h_side():
        """the H side of the proof, beside the witness MSMs of the same rank (ga_g16_h_chain* / ga_g16_h_combine take the
        context's second lane when the device is busy, common.hip.h LaneLock).  With 3+ ranks no PCIe link carries a whole vector:
        every rank uploads 1/N of A, B and C over its own link and the pieces are gathered on the chain owners over xGMI
        (N x 55 GB/s of PCIe in parallel, 7 links x 150 GB/s into each owner) -- 3.6 ms of upload at N = 8 instead of 9.6 ms."""
        try:
            if device is not None:
                torch.cuda.set_device(dev)   # the current device is per thread
            if sliced:
                lo, hi = min(rank * cshare, nc), min((rank + 1) * cshare, nc)
                for k in range(3):
                    piece = torch.zeros((cshare, 4), dtype=torch.int64, device=dev)
                    if hi > lo:
                        src = np.ascontiguousarray(np.asarray(vecs[k])[lo:hi]).view(np.int64)
                        piece[: hi - lo].copy_(torch.from_numpy(src))
                    parts = [torch.empty_like(piece) for _ in range(world)] if rank == owner[k] else None
                    _gather(piece, parts, owner[k], dist)
                    if rank == owner[k]:
                        flat = torch.cat(parts)[:nc]
                        bufs[k][:nc].copy_(flat)
                        sync()
                        groth16.HChainDevice(pk, bufs[k].data_ptr(), nc)
            else:
                for k in range(3):
                    if rank == owner[k]:
                        groth16.HChain(pk, vecs[k], bufs[k].data_ptr())
            for k in range(3):                       # b and c travel to rank 0
                if owner[k] != 0:
                    if rank == owner[k]:
                        _send(bufs[k], 0, dist)
                    elif rank == 0:
                        _recv(bufs[k], owner[k], dist)
            if rank == 0:
                sync()
                groth16.HCombine(pk, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
                if not windowed:                     # one equally sized (padded) slice per rank
                    pieces = []
                    for q in range(world):
                        lo, hi = shard_range(n - 1, q, world)
                        t = torch.zeros((share, 4), dtype=torch.int64, device=dev)
                        t[: hi - lo] = bufs[0][lo:hi]
                        pieces.append(t)
                    state["pieces"] = pieces
                sync()
        except Exception as e:                       # re-raised on the main thread
            state["error"] = e

    helper = None
    if rank in owner or sliced:                # with sliced uploads every rank takes part in the gathers
        helper = threading.Thread(target=h_side)
        helper.start()
    part = groth16.WitnessPartial(pk, solution.W, nb_public)
    if helper is not None:
        helper.join()
        if state["error"] is not None:
            raise state["error"]
    if windowed:
        _broadcast(bufs[0], 0, dist)
        sync()
        z = groth16.ZPartial(pk, bufs[0].data_ptr())
        return _finish_gathered(pk, part, z, r, s, dist, device)
    mine = torch.empty((share, 4), dtype=torch.int64, device=dev)
    _scatter(mine, state["pieces"], 0, dist)
    sync()
    z = groth16.ZPartial(pk, mine.data_ptr())
    return _finish_gathered(pk, part, z, r, s, dist, device)


def _finish_gathered(pk, part, z, r, s, dist, device):
    """all_gather of every rank's A | B1 | K | B2 sums and Z sum, host additions, epilogue with (r, s) -- identical on every rank"""
    from . import groth16
    fp = part.shape[0] // 15
    both = np.concatenate([part, z])
    gathered = _all_gather_u64(both, dist, device)
    total = groth16.SumPartials(pk.curve, [g[: 15 * fp] for g in gathered], lib=pk.ctx.lib)
    zs = combine_partials(pk.curve, 0, [g[15 * fp:] for g in gathered], lib=pk.ctx.lib)
    total[6 * fp: 9 * fp] = ecc.jac_add(pk.curve, 0, np.ascontiguousarray(total[6 * fp: 9 * fp]), zs, lib=pk.ctx.lib)
    return groth16.Finish(pk, total, r, s)
