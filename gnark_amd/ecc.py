"""Multi-scalar multiplication -- mirror of gnark-crypto's `G1Jac.MultiExp` / `G2Jac.MultiExp` as the reference
calls them (backend/groth16/bn254/prove.go:194,207,227,237,283), on gnark's own memory images."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .device import Context, DeviceBuffer, _ptr, affine_words, as_u64, curve_id, jac_words

G1, G2 = _lib.G1, _lib.G2


def _arg(x, flag):
    if isinstance(x, DeviceBuffer):
        return C.c_void_p(x.ptr), flag
    if isinstance(x, int):
        return C.c_void_p(x), flag
    return _ptr(x), 0


def MultiExp(ctx: Context, curve, group: int, points, scalars, n: int | None = None, montgomery: bool = True) -> np.ndarray:
    """sum_i scalars[i] * points[i]  ->  one Jacobian point {X,Y,Z} (uint64 limbs, Montgomery).

    points : (n, 2*fp) [G1] or (n, 4*fp) [G2] uint64 array (G1Affine/G2Affine images) or a DeviceBuffer
    scalars: (n, 4) uint64 array (fr.Element images) or a DeviceBuffer
    Errors mirror MultiExp's "len(points) != len(scalars)" check.
    """
    cid = curve_id(curve)
    if not isinstance(points, (DeviceBuffer, int)):
        points = as_u64(points, affine_words(cid, group))
        n_p = points.shape[0]
    else:
        n_p = n
    if not isinstance(scalars, (DeviceBuffer, int)):
        scalars = as_u64(scalars, 4)
        n_s = scalars.shape[0]
    else:
        n_s = n
    if n_p is None or n_s is None:
        raise ValueError("n is required when both operands are device buffers")
    if n_p != n_s:
        raise ValueError("len(points) != len(scalars)")
    bp, f1 = _arg(points, _lib.BASES_ON_DEVICE)
    sp, f2 = _arg(scalars, _lib.SCALARS_ON_DEVICE)
    flags = f1 | f2 | (_lib.SCALARS_MONTGOMERY if montgomery else 0)
    out = np.zeros(jac_words(cid, group), dtype=np.uint64)
    ctx.lib.check(ctx.lib.ga_msm(ctx.handle, cid, group, bp, sp, n_p, flags, _ptr(out)))
    return out


class PrecomputedBases:
    """Bases pinned on the device together with [2^(c*w)]P for every Pippenger window w (ga_msm_table_*): the GPU analogue of
    keeping `pk.G1.A` etc. resident ("PinToGPU", provingkey.go:37-42) with ICICLE's PrecomputeFactor."""

    def __init__(self, ctx: Context, curve, group: int, points, n: int | None = None, batched: bool = False):
        """batched: the table will mostly serve MultiExpBatch (GA_TABLE_BATCHED: a narrower window is planned)"""
        cid = curve_id(curve)
        if not isinstance(points, (DeviceBuffer, int)):
            points = as_u64(points, affine_words(cid, group))
            n = points.shape[0]
        if n is None:
            raise ValueError("n is required for device-resident points")
        bp, f1 = _arg(points, _lib.BASES_ON_DEVICE)
        h = C.c_void_p()
        ctx.lib.check(ctx.lib.ga_msm_table_create(ctx.handle, cid, group, bp, n, f1 | (_lib.TABLE_BATCHED if batched else 0), C.byref(h)))
        self.ctx, self.curve, self.group, self.n, self.handle = ctx, cid, group, n, h

    def info(self):
        c, nw, nb = C.c_int(), C.c_int(), C.c_uint64()
        self.ctx.lib.check(self.ctx.lib.ga_msm_table_info(self.handle, C.byref(c), C.byref(nw), C.byref(nb)))
        return {"window_bits": c.value, "windows": nw.value, "table_bytes": nb.value}

    def MultiExpWindows(self, scalars, win_lo: int, win_hi: int, montgomery: bool = True) -> np.ndarray:
        """windows [win_lo, win_hi) only (ga_msm_table_run_windows): partial results of disjoint ranges add up to MultiExp's"""
        if not isinstance(scalars, (DeviceBuffer, int)):
            scalars = as_u64(scalars, 4)
            if scalars.shape[0] != self.n:
                raise ValueError("len(points) != len(scalars)")
        sp, f2 = _arg(scalars, _lib.SCALARS_ON_DEVICE)
        out = np.zeros(jac_words(self.curve, self.group), dtype=np.uint64)
        flags = f2 | (_lib.SCALARS_MONTGOMERY if montgomery else 0)
        self.ctx.lib.check(self.ctx.lib.ga_msm_table_run_windows(self.handle, sp, flags, int(win_lo), int(win_hi), _ptr(out)))
        return out

    def MultiExp(self, scalars, montgomery: bool = True) -> np.ndarray:
        if not isinstance(scalars, (DeviceBuffer, int)):
            scalars = as_u64(scalars, 4)
            if scalars.shape[0] != self.n:
                raise ValueError("len(points) != len(scalars)")
        sp, f2 = _arg(scalars, _lib.SCALARS_ON_DEVICE)
        out = np.zeros(jac_words(self.curve, self.group), dtype=np.uint64)
        self.ctx.lib.check(self.ctx.lib.ga_msm_table_run(self.handle, sp, f2 | (_lib.SCALARS_MONTGOMERY if montgomery else 0), _ptr(out)))
        return out

    def MultiExpBatch(self, scalar_vectors, montgomery: bool = True) -> np.ndarray:
        """k scalar vectors over these bases in one pass (ga_msm_table_run_batch): (k, jac_words) -- row j equals MultiExp(vector j).
        All vectors host arrays or all DeviceBuffers."""
        k = len(scalar_vectors)
        on_dev = all(isinstance(v, (DeviceBuffer, int)) for v in scalar_vectors)
        if not on_dev:
            if any(isinstance(v, (DeviceBuffer, int)) for v in scalar_vectors):
                raise ValueError("a batch mixes host and device scalar vectors")
            scalar_vectors = [as_u64(v, 4) for v in scalar_vectors]
            if any(v.shape[0] != self.n for v in scalar_vectors):
                raise ValueError("len(points) != len(scalars)")
        ptrs, flags = [], (_lib.SCALARS_MONTGOMERY if montgomery else 0)
        for v in scalar_vectors:
            p, f = _arg(v, _lib.SCALARS_ON_DEVICE)
            ptrs.append(p)
            flags |= f
        arr = (C.c_void_p * k)(*[C.cast(p, C.c_void_p).value for p in ptrs])
        out = np.zeros((k, jac_words(self.curve, self.group)), dtype=np.uint64)
        self.ctx.lib.check(self.ctx.lib.ga_msm_table_run_batch(self.handle, arr, k, flags, _ptr(out)))
        return out

    def KzgOpen(self, poly, point):
        """kzg.Open(p, point, pk) over this (monomial G1) SRS: returns (ClaimedValue fr image, H as G1Jac)."""
        n = None
        if not isinstance(poly, (DeviceBuffer, int)):
            poly = as_u64(poly, 4)
            n = poly.shape[0]
        else:
            raise ValueError("pass the coefficient array (host); device polynomials go through ga_kzg_open directly")
        pp, f = _arg(poly, _lib.SCALARS_ON_DEVICE)
        z = as_u64(np.asarray(point).reshape(1, 4), 4)
        val = np.zeros(4, dtype=np.uint64)
        out = np.zeros(jac_words(self.curve, self.group), dtype=np.uint64)
        self.ctx.lib.check(self.ctx.lib.ga_kzg_open(self.handle, pp, n, f, _ptr(z), _ptr(val), _ptr(out)))
        return val, out

    def free(self):
        if self.handle:
            self.ctx.lib.ga_msm_table_destroy(self.handle)
            self.handle = None


def MultiExpWindows(ctx: Context, curve, group: int, points, scalars, n: int, win_lo: int, win_hi: int, montgomery=True):
    """Window-sharded MSM (multi-GPU partitioning A): Jacobian window sums for windows [win_lo, win_hi)."""
    cid = curve_id(curve)
    c, nw = plan(curve, group, n, lib=ctx.lib)
    hi = nw if win_hi < 0 else win_hi
    if not isinstance(points, (DeviceBuffer, int)):
        points = as_u64(points, affine_words(cid, group))
    if not isinstance(scalars, (DeviceBuffer, int)):
        scalars = as_u64(scalars, 4)
    bp, f1 = _arg(points, _lib.BASES_ON_DEVICE)
    sp, f2 = _arg(scalars, _lib.SCALARS_ON_DEVICE)
    flags = f1 | f2 | (_lib.SCALARS_MONTGOMERY if montgomery else 0)
    out = np.zeros((hi - win_lo, jac_words(cid, group)), dtype=np.uint64)
    cc, nn = C.c_int(), C.c_int()
    ctx.lib.check(ctx.lib.ga_msm_windows(ctx.handle, cid, group, bp, sp, n, flags, win_lo, win_hi, _ptr(out), C.byref(cc), C.byref(nn)))
    return out, cc.value, nn.value


def plan(curve, group: int, n: int, lib=None):
    lib = lib or _lib.load()
    c, nw = C.c_int(), C.c_int()
    lib.check(lib.ga_msm_plan(curve_id(curve), group, n, C.byref(c), C.byref(nw)))
    return c.value, nw.value


def combine_windows(curve, group: int, windows: np.ndarray, window_bits: int, lib=None) -> np.ndarray:
    lib = lib or _lib.load()
    cid = curve_id(curve)
    windows = as_u64(windows, jac_words(cid, group))
    out = np.zeros(jac_words(cid, group), dtype=np.uint64)
    lib.check(lib.ga_msm_combine_windows(cid, group, _ptr(windows), windows.shape[0], window_bits, _ptr(out)))
    return out


def jac_add(curve, group, a, b, lib=None):
    lib = lib or _lib.load()
    cid = curve_id(curve)
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros(jac_words(cid, group), dtype=np.uint64)
    lib.check(lib.ga_jac_add(cid, group, _ptr(a), _ptr(b), _ptr(out)))
    return out


def jac_to_affine(curve, group, a, lib=None):
    lib = lib or _lib.load()
    cid = curve_id(curve)
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros(affine_words(cid, group), dtype=np.uint64)
    lib.check(lib.ga_jac_to_affine(cid, group, _ptr(a), _ptr(out)))
    return out


def jac_scalar_mul(curve, group, a, k_canonical: int, lib=None):
    lib = lib or _lib.load()
    cid = curve_id(curve)
    a = np.ascontiguousarray(a, dtype=np.uint64)
    k = np.array([(k_canonical >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros(jac_words(cid, group), dtype=np.uint64)
    lib.check(lib.ga_jac_scalar_mul(cid, group, _ptr(a), _ptr(k), _ptr(out)))
    return out


def generator_mul(curve, group, k_canonical: int, lib=None):
    lib = lib or _lib.load()
    cid = curve_id(curve)
    k = np.array([(k_canonical >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros(jac_words(cid, group), dtype=np.uint64)
    lib.check(lib.ga_generator_mul(cid, group, _ptr(k), _ptr(out)))
    return out
