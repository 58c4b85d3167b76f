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


class ShardedProofError(RuntimeError):
    """a rank failed inside a sharded proof.  Raised on EVERY rank, after the last collective of the proof: the collectives of
    groth16_prove_sharded form one fixed schedule that every rank walks to its end whatever happened to its own library calls, so
    one rank's error (an OOM while its scratch grows, a bad input) can never leave the others waiting in a collective."""


def _fault(point: str, rank: int):
    """test hook: GA_MGPU_FAULT="<rank>:<point>" makes that rank fail at that point of a sharded proof (tests/test_multigpu_gloo.py
    and tests/test_bench_contract.py use it to show that a failing rank ends the proof on every rank instead of hanging it)"""
    import os
    spec = os.environ.get("GA_MGPU_FAULT", "")
    if spec and spec == "%d:%s" % (rank, point):
        raise RuntimeError("injected fault at '%s' on rank %d (GA_MGPU_FAULT)" % (point, rank))


class _Guard:
    """first local error of a rank; later local steps are skipped, collectives are not"""

    def __init__(self):
        self.error = None

    def run(self, fn, default=None):
        if self.error is not None:
            return default
        try:
            return fn()
        except Exception as e:   # noqa: BLE001 -- whatever it was, the schedule goes on and the error travels with the last all_gather
            self.error = e
            return default


def groth16_prove_sharded(pk, solution, nb_public: int, r, s, dist, device=None, replicate_h=False, replicate_uploads=False,
                          force_collectives=False):
    """One proof over a key sharded by base-point range (groth16.ProvingKey(..., shard=(rank, world))), one process per GPU.

    Serial work is not replicated: every rank uploads only the wire range of W its bases cover; the three chains of computeH
    (FFT_coset(iFFT(.)) of the solver's A, B, C) run on ranks 0, 1, 2 -- three uploads over three PCIe links, on a helper thread
    BESIDE the rank's witness MSMs (second lane of the context) -- and travel to rank 0 over xGMI (send/recv, 32 B x n each);
    rank 0 finishes h and scatters the slices (32 B x n / world per peer); every rank runs the MSM over its slice of pk.G1.Z;
    one all_gather of 3 G1Jac + 1 G2Jac per rank (+ one ok word); every rank finishes identically.
    replicate_h=True keeps the round-1 scheme (every rank recomputes h) for comparison; replicate_uploads=True makes the chain
    owners upload their whole vectors themselves even with 3+ ranks.
    force_collectives=True walks the full schedule (sliced uploads gathered on the owners, scatter / broadcast, all_gather) even
    with ONE rank: the RCCL self-test of a 1-GPU box (bench.py "nccl_selftest", tests/test_gpu_parity.py).

    Failure behaviour: local steps run under a guard, collectives always run, and the ok word of the final all_gather turns any
    rank's error into a ShardedProofError on every rank."""
    from . import groth16
    world = 1 if dist is None else dist.get_world_size()
    if (world == 1 and not force_collectives) or replicate_h:
        if world == 1:
            return groth16.Finish(pk, groth16.ProvePartial(pk, solution, nb_public), r, s)
        g = _Guard()
        fp15 = 15 * _fp_limbs(pk)
        part = g.run(lambda: groth16.ProvePartial(pk, solution, nb_public), np.zeros(fp15, dtype=np.uint64))
        total, _ = _gather_partials(pk, g, part, None, dist, device)
        return groth16.Finish(pk, total, r, s)
    import threading
    import torch
    rank = dist.get_rank()
    lay = groth16.ShardLayout(pk)
    n = lay["n"]
    dev = device if device is not None else torch.device("cpu")
    sync = (lambda: torch.cuda.synchronize(dev)) if device is not None else (lambda: None)
    owner = [0, 1 if world >= 2 else 0, 2 if world >= 3 else 0]
    vecs = [solution.A, solution.B, solution.C]
    windowed = lay["win_count"] > 1          # window-sharded key: every rank needs all of h
    share = (n - 1 + world - 1) // world + 1
    # device buffers as torch tensors, so that the collective library can move them; the prover library gets raw pointers
    bufs = {}
    for k in range(3):
        if rank == owner[k] or rank == 0:
            bufs[k] = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    if windowed and 0 not in bufs:
        bufs[0] = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    sync()
    g = _Guard()          # main thread
    gh = _Guard()         # helper thread (its collectives are issued between start() and join(): one thread at a time talks to the group)
    state = {"pieces": None}

    nc = int(np.asarray(solution.A).shape[0])                 # constraints: the length of the solver's A, B, C
    cshare = (nc + world - 1) // world                         # every rank uploads rows [rank*cshare, ...) of A, B and C
    sliced = (world >= 3 or force_collectives) and not replicate_uploads   # (with one or two ranks the chain owners upload whole vectors)

    def h_side():
        """the H side of the proof, beside the witness MSMs of the same rank (ga_g16_h_chain* / ga_g16_h_combine take the
        context's second lane when the device is busy, common.hip.h LaneLock).  With 3+ ranks no PCIe link carries a whole vector:
        every rank uploads 1/N of A, B and C over its own link and the pieces are gathered on the chain owners over xGMI
        (N x 55 GB/s of PCIe in parallel, 7 links x 150 GB/s into each owner) -- 3.6 ms of upload at N = 8 instead of 9.6 ms."""
        if device is not None:
            torch.cuda.set_device(dev)   # the current device is per thread
        gh.run(lambda: _fault("h_side", rank))
        if sliced:
            lo, hi = min(rank * cshare, nc), min((rank + 1) * cshare, nc)
            for k in range(3):
                piece = torch.zeros((cshare, 4), dtype=torch.int64, device=dev)

                def fill(k=k, piece=piece):
                    if hi > lo:
                        src = np.ascontiguousarray(np.asarray(vecs[k])[lo:hi]).view(np.int64)
                        piece[: hi - lo].copy_(torch.from_numpy(src))
                gh.run(fill)
                parts = [torch.empty_like(piece) for _ in range(world)] if rank == owner[k] else None
                _gather(piece, parts, owner[k], dist)
                if rank == owner[k]:
                    def chain(k=k, parts=parts):
                        bufs[k][:nc].copy_(torch.cat(parts)[:nc])
                        sync()
                        groth16.HChainDevice(pk, bufs[k].data_ptr(), nc)
                    gh.run(chain)
        else:
            for k in range(3):
                if rank == owner[k]:
                    gh.run(lambda k=k: groth16.HChain(pk, vecs[k], bufs[k].data_ptr()))
        for k in range(3):                       # b and c travel to rank 0
            if owner[k] != 0:
                if rank == owner[k]:
                    _send(bufs[k], 0, dist)
                elif rank == 0:
                    _recv(bufs[k], owner[k], dist)
        if rank == 0:
            def combine():
                sync()
                groth16.HCombine(pk, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
            gh.run(combine)
            if not windowed:                     # one equally sized (padded) slice per rank
                pieces = []
                for q in range(world):
                    lo, hi = shard_range(n - 1, q, world)
                    t = torch.zeros((share, 4), dtype=torch.int64, device=dev)
                    t[: hi - lo] = bufs[0][lo:hi]
                    pieces.append(t)
                state["pieces"] = pieces
            sync()

    helper = None
    if rank in owner or sliced:                # with sliced uploads every rank takes part in the gathers
        helper = threading.Thread(target=h_side)
        helper.start()

    def witness():
        _fault("witness", rank)
        return groth16.WitnessPartial(pk, solution.W, nb_public)
    part = g.run(witness, np.zeros(15 * _fp_limbs(pk), dtype=np.uint64))
    if helper is not None:
        helper.join()
        if gh.error is not None and g.error is None:
            g.error = gh.error
    if windowed:
        _broadcast(bufs[0], 0, dist)
        sync()
        h_ptr = bufs[0].data_ptr()
    else:
        mine = torch.zeros((share, 4), dtype=torch.int64, device=dev)
        if rank == 0 and state["pieces"] is None:    # the helper died outside its guard: the schedule still needs something to scatter
            g.error = g.error or RuntimeError("the H side of rank 0 did not finish")
            state["pieces"] = [torch.zeros((share, 4), dtype=torch.int64, device=dev) for _ in range(world)]
        _scatter(mine, state["pieces"], 0, dist)
        sync()
        h_ptr = mine.data_ptr()

    def zpart():
        _fault("z", rank)
        return groth16.ZPartial(pk, h_ptr)
    z = g.run(zpart, np.zeros(3 * _fp_limbs(pk), dtype=np.uint64))
    total, _ = _gather_partials(pk, g, part, z, dist, device)
    return groth16.Finish(pk, total, r, s)


def _fp_limbs(pk) -> int:
    from .device import FP_LIMBS
    return FP_LIMBS[pk.curve]


def _gather_partials(pk, guard, part, z, dist, device):
    """the one exchange of results: all_gather of every rank's A | B1 | K | B2 sums (+ its Z sum) and an ok word, host additions.
    Raises ShardedProofError on every rank if any rank's guard holds an error."""
    from . import groth16
    fp = _fp_limbs(pk)
    ok = np.array([0 if guard.error is not None else 1], dtype=np.uint64)
    pieces = [np.asarray(part, dtype=np.uint64)] + ([np.asarray(z, dtype=np.uint64)] if z is not None else []) + [ok]
    gathered = _all_gather_u64(np.concatenate(pieces), dist, device)
    bad = [q for q, gq in enumerate(gathered) if int(gq[-1]) != 1]
    if bad:
        mine = "" if guard.error is None else ": %r" % (guard.error,)
        raise ShardedProofError("sharded proof failed on rank(s) %s%s" % (bad, mine)) from guard.error
    total = groth16.SumPartials(pk.curve, [gq[: 15 * fp] for gq in gathered], lib=pk.ctx.lib)
    if z is not None:
        zs = combine_partials(pk.curve, 0, [gq[15 * fp: 18 * fp] for gq in gathered], lib=pk.ctx.lib)
        total[6 * fp: 9 * fp] = ecc.jac_add(pk.curve, 0, np.ascontiguousarray(total[6 * fp: 9 * fp]), zs, lib=pk.ctx.lib)
    return total, gathered


def agree(dist, ok: bool, err=None, device=None):
    """Every rank learns whether ALL ranks succeeded at a local step (all_reduce(MIN) of a flag), and if not, why
    (all_gather_object of the texts): the step after which a multi-rank program may enter collectives again.  Returns (ok, text)."""
    if dist is None or not dist.is_initialized():
        return bool(ok), (None if ok else str(err))
    import torch
    on_dev = device is not None and dist.get_backend() != "gloo"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t.item()) == 1:
        return True, None
    texts = [None] * dist.get_world_size()
    dist.all_gather_object(texts, None if ok else str(err)[:300])
    return False, "; ".join("rank %d: %s" % (q, t) for q, t in enumerate(texts) if t)


def collective_selftest(dist, device=None, words: int = 1 << 16) -> dict:
    """every collective this module uses (all_gather, gather, scatter, broadcast, all_reduce; send/recv with 2+ ranks), once, on
    DEVICE tensors with checkable contents.  With the nccl backend this is the RCCL path of the multi-GPU prover; it also runs at
    world = 1, where a 1-GPU box can execute it (bench.py "nccl_selftest")."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    res = {"backend": dist.get_backend(), "world": world, "device_tensors": device is not None and dist.get_backend() != "gloo"}
    pat = lambda q: (torch.arange(words, dtype=torch.int64, device=dev) * 7 + 1000003 * (q + 1))
    mine = pat(rank)
    got = _all_gather_u64(mine.cpu().numpy().view(np.uint64), dist, device)
    res["all_gather"] = all(np.array_equal(gq.view(np.int64), pat(q).cpu().numpy()) for q, gq in enumerate(got))
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    _gather(mine, parts, 0, dist)
    res["gather"] = True if rank != 0 else all(torch.equal(p, pat(q)) for q, p in enumerate(parts))
    out = torch.empty_like(mine)
    _scatter(out, [pat(q + 100) for q in range(world)] if rank == 0 else None, 0, dist)
    res["scatter"] = bool(torch.equal(out, pat(rank + 100)))
    b = pat(200) if rank == 0 else torch.zeros_like(mine)
    _broadcast(b, 0, dist)
    res["broadcast"] = bool(torch.equal(b, pat(200)))
    on_dev = res["device_tensors"]
    t = torch.tensor([rank + 1], dtype=torch.int64, device=dev if on_dev else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["all_reduce"] = int(t.item()) == world
    if world >= 2:   # (a rank cannot send to itself: the point-to-point pair needs two ranks)
        if rank == 1:
            _send(pat(300), 0, dist)
        elif rank == 0:
            r_ = torch.zeros_like(mine)
            _recv(r_, 1, dist)
            res["send_recv"] = bool(torch.equal(r_, pat(300)))
    else:
        res["send_recv"] = "skipped: needs two ranks"
    if device is not None:
        torch.cuda.synchronize(dev)
    res["ok"] = all(v is True for k, v in res.items() if k in ("all_gather", "gather", "scatter", "broadcast", "all_reduce")) and res.get("send_recv", True) in (True, "skipped: needs two ranks")
    return res
