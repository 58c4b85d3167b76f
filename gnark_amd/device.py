"""Device context and buffers (the DeviceSlice / runtime part of the ICICLE wrapper the reference uses,
backend/accelerated/icicle/groth16/bn254/icicle.go:120,321,475,795)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

FP_LIMBS = {_lib.BN254: 4, _lib.BLS12_381: 6}
FR_LIMBS = 4
CURVE_IDS = {"bn254": _lib.BN254, "bls12-381": _lib.BLS12_381, "bls12_381": _lib.BLS12_381,
             _lib.BN254: _lib.BN254, _lib.BLS12_381: _lib.BLS12_381}


def curve_id(curve) -> int:
    try:
        return CURVE_IDS[curve]
    except KeyError:
        raise ValueError(f"unsupported curve {curve!r} (built: bn254, bls12-381)") from None


def affine_words(curve: int, group: int) -> int:
    return FP_LIMBS[curve] * (2 if group == _lib.G1 else 4)


def jac_words(curve: int, group: int) -> int:
    return FP_LIMBS[curve] * (3 if group == _lib.G1 else 6)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def as_u64(a, words=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if words is not None and (a.ndim != 2 or a.shape[1] != words):
        raise ValueError(f"expected an (n, {words}) uint64 array, got shape {a.shape}")
    return a


class Context:
    """One per (process, device).  `lib` lets tests bind a different build of the same ABI."""

    def __init__(self, device: int = 0, lib: _lib.Library | None = None):
        self.lib = lib or _lib.load()
        h = C.c_void_p()
        self.lib.check(self.lib.ga_ctx_create(device, C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            self.lib.ga_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self):
        name = C.create_string_buffer(256)
        tot, free = C.c_uint64(), C.c_uint64()
        self.lib.check(self.lib.ga_device_info(self.handle, name, 256, C.byref(tot), C.byref(free)))
        return {"name": name.value.decode(), "total_bytes": tot.value, "free_bytes": free.value}

    def malloc(self, nbytes: int) -> "DeviceBuffer":
        p = C.c_void_p()
        self.lib.check(self.lib.ga_malloc(self.handle, nbytes, C.byref(p)))
        return DeviceBuffer(self, p.value, nbytes)

    def to_device(self, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = self.malloc(a.nbytes)
        if a.nbytes:
            self.lib.check(self.lib.ga_copy_to_device(self.handle, buf.ptr, _ptr(a), a.nbytes))
        return buf

    def sync(self):
        self.lib.check(self.lib.ga_sync(self.handle))

    def lane_stats(self):
        """how this context's ga_g16_prove calls were scheduled (ga_g16_lane_stats)"""
        out = (C.c_uint64 * 6)()
        self.lib.check(self.lib.ga_g16_lane_stats(self.handle, out))
        keys = ("lanes01_proofs", "lanes23_proofs", "queued_proofs", "split_proofs", "lanes01_scratch_bytes", "lanes23_scratch_bytes")
        return dict(zip(keys, (int(v) for v in out)))

    # profiling (ICICLE_STEP_PROFILE analogue)
    def profile(self, on: bool):
        self.lib.check(self.lib.ga_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        self.lib.check(self.lib.ga_profile_reset(self.handle))

    def profile_read(self):
        buf = C.create_string_buffer(1 << 20)
        self.lib.check(self.lib.ga_profile_read(self.handle, buf, len(buf)))
        out = []
        for item in buf.value.decode().split(";"):
            if item:
                k, v = item.split("=")
                out.append((k, float(v)))
        return out

    def microbench(self):
        buf = C.create_string_buffer(4096)
        self.lib.check(self.lib.ga_microbench(self.handle, buf, len(buf)))
        return dict((k, float(v)) for k, v in (it.split("=") for it in buf.value.decode().split(";") if it))


class DeviceBuffer:
    def __init__(self, ctx: Context, ptr: int, nbytes: int):
        self.ctx, self.ptr, self.nbytes = ctx, ptr, nbytes

    def to_host(self, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            self.ctx.lib.check(self.ctx.lib.ga_copy_to_host(self.ctx.handle, _ptr(out), self.ptr, out.nbytes))
        return out

    def offset(self, nbytes: int) -> int:
        return self.ptr + nbytes

    def free(self):
        if self.ptr:
            self.ctx.lib.check(self.ctx.lib.ga_free(self.ctx.handle, self.ptr))
            self.ptr = 0
