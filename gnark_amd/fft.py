"""Mirror of gnark-crypto's `fft.Domain` as the reference uses it (backend/groth16/bn254/setup.go:101,
prove.go:346-389): `NewDomain(n)`, `FFT(a, DIF|DIT, on_coset)`, `FFTInverse(...)`, plus `compute_h`."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .device import Context, DeviceBuffer, _ptr, as_u64, curve_id

DIF, DIT = _lib.DIF, _lib.DIT


class Domain:
    def __init__(self, ctx: Context, curve, cardinality: int):
        """fft.NewDomain(m): cardinality is rounded up to the next power of two, as gnark-crypto does."""
        n = 1
        while n < cardinality:
            n *= 2
        self.ctx, self.curve, self.Cardinality = ctx, curve_id(curve), n
        h = C.c_void_p()
        ctx.lib.check(ctx.lib.ga_domain_create(ctx.handle, self.curve, n, C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            self.ctx.lib.ga_domain_destroy(self.handle)
            self.handle = None

    def _run(self, a, direction, decimation, on_coset):
        if isinstance(a, DeviceBuffer):
            self.ctx.lib.check(self.ctx.lib.ga_fft(self.handle, C.c_void_p(a.ptr), direction, decimation, int(on_coset), 1))
            return a
        a = as_u64(a, 4)
        if a.shape[0] != self.Cardinality:
            raise ValueError(f"len(a)={a.shape[0]} != domain cardinality {self.Cardinality}")
        out = a.copy()
        self.ctx.lib.check(self.ctx.lib.ga_fft(self.handle, _ptr(out), direction, decimation, int(on_coset), 0))
        return out

    def FFT(self, a, decimation: int, on_coset: bool = False):
        return self._run(a, _lib.FFT_FORWARD, decimation, on_coset)

    def FFTInverse(self, a, decimation: int, on_coset: bool = False):
        return self._run(a, _lib.FFT_INVERSE, decimation, on_coset)

    def compute_h(self, a, b, c) -> np.ndarray:
        """computeH (prove.go:346-389): a, b, c are the solver's A, B, C (len = #constraints <= n)."""
        a, b, c = as_u64(a, 4), as_u64(b, 4), as_u64(c, 4)
        if not (a.shape == b.shape == c.shape):
            raise ValueError("a, b, c must have the same length")
        out = np.zeros((self.Cardinality, 4), dtype=np.uint64)
        self.ctx.lib.check(self.ctx.lib.ga_compute_h(self.handle, _ptr(a), _ptr(b), _ptr(c), a.shape[0], _ptr(out), 0))
        return out


def NewDomain(ctx: Context, curve, m: int) -> Domain:
    return Domain(ctx, curve, m)
