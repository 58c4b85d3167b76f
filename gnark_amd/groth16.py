"""Mirror of `backend/accelerated/icicle/groth16` (groth16_icicle.go:79-194, provingkey.go:37-42): a ProvingKey
whose G1/G2 vectors live on the GPU, `Prove(pk, solution, ...) -> Proof`, and `Proof.WriteTo` bytes.

The solver is out of scope: `Prove` takes what `r1cs.Solve` returns -- W, A, B, C as fr.Element images."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .device import FP_LIMBS, Context, _ptr, as_u64, curve_id


@dataclass
class Solution:
    """constraint/bn254/system.go:162-165 R1CSSolution: W (nbWires), A, B, C (nbConstraints)."""
    W: np.ndarray
    A: np.ndarray
    B: np.ndarray
    C: np.ndarray


@dataclass
class Proof:
    """backend/groth16/bn254/prove.go:34-39: affine images, Montgomery limbs.  Commitments / CommitmentPok are filled by the
    caller from ProvingKey.Commit + FoldPok when the circuit uses api.Commit (BSB22)."""
    curve: int
    Ar: np.ndarray
    Bs: np.ndarray
    Krs: np.ndarray
    _lib: object = None
    Commitments: np.ndarray = None     # (n, 2*fp) G1Affine
    CommitmentPok: np.ndarray = None   # (2*fp,) G1Affine

    def raw(self) -> np.ndarray:
        return np.concatenate([self.Ar, self.Bs, self.Krs]).astype(np.uint64)

    def WriteTo(self) -> bytes:
        """Proof.WriteTo (marshal.go:33-58): compressed Ar | Bs | Krs | u32 n | commitments | CommitmentPok."""
        lib = self._lib or _lib.load()
        raw = self.raw()
        fp = FP_LIMBS[self.curve]
        n = C.c_size_t()
        if self.Commitments is None or len(self.Commitments) == 0:
            out = np.zeros(512, dtype=np.uint8)
            lib.check(lib.ga_g16_proof_marshal(self.curve, _ptr(raw), _ptr(out), out.nbytes, C.byref(n)))
        else:
            com = as_u64(np.asarray(self.Commitments).reshape(-1, 2 * fp), 2 * fp)
            pok = as_u64(np.asarray(self.CommitmentPok).reshape(1, 2 * fp), 2 * fp)
            out = np.zeros(512 + 64 * com.shape[0], dtype=np.uint8)
            lib.check(lib.ga_g16_proof_marshal_bsb22(self.curve, _ptr(raw), _ptr(com), com.shape[0], _ptr(pok), _ptr(out),
                                                     out.nbytes, C.byref(n)))
        return out[: n.value].tobytes()


    def WriteRawTo(self) -> bytes:
        """Proof.WriteRawTo (marshal.go:25-30): uncompressed points."""
        lib = self._lib or _lib.load()
        raw = self.raw()
        fp = FP_LIMBS[self.curve]
        ncom = 0 if self.Commitments is None else len(self.Commitments)
        com = as_u64(np.asarray(self.Commitments).reshape(-1, 2 * fp), 2 * fp) if ncom else None
        pok = as_u64(np.asarray(self.CommitmentPok).reshape(1, 2 * fp), 2 * fp) if self.CommitmentPok is not None else None
        out = np.zeros(1024 + 128 * ncom, dtype=np.uint8)
        n = C.c_size_t()
        lib.check(lib.ga_g16_proof_marshal_raw(self.curve, _ptr(raw), _ptr(com) if ncom else None, ncom,
                                               _ptr(pok) if pok is not None else None, _ptr(out), out.nbytes, C.byref(n)))
        return out[: n.value].tobytes()


KEY_FORMAT_COMPRESSED, KEY_FORMAT_RAW, KEY_FORMAT_DUMP = 0, 1, 2


def _key_struct(curve, *, domain_cardinality, alpha1, beta1, delta1, A, B, Z, K, beta2, delta2, B2, infinityA, infinityB,
                commitment_keys=(), **_ignored):
    """ga_g16_key over host arrays (returned together with the arrays that must stay alive)"""
    cid = curve_id(curve)
    fp = FP_LIMBS[cid]
    g1 = lambda v: as_u64(np.asarray(v).reshape(-1, 2 * fp), 2 * fp)
    g2 = lambda v: as_u64(np.asarray(v).reshape(-1, 4 * fp), 4 * fp)
    keep = dict(A=g1(A), B=g1(B), Z=g1(Z), K=g1(K), B2=g2(B2), alpha1=g1(alpha1), beta1=g1(beta1), delta1=g1(delta1), beta2=g2(beta2),
                delta2=g2(delta2), ia=np.ascontiguousarray(infinityA, dtype=np.uint8), ib=np.ascontiguousarray(infinityB, dtype=np.uint8))
    key = _lib.G16Key()
    key.curve, key.domain_cardinality = cid, int(domain_cardinality)
    key.g1_alpha, key.g1_beta, key.g1_delta = (keep[k].ctypes.data for k in ("alpha1", "beta1", "delta1"))
    key.g1_a, key.len_a = keep["A"].ctypes.data, keep["A"].shape[0]
    key.g1_b, key.len_b = keep["B"].ctypes.data, keep["B"].shape[0]
    key.g1_z, key.len_z = keep["Z"].ctypes.data, keep["Z"].shape[0]
    key.g1_k, key.len_k = keep["K"].ctypes.data, keep["K"].shape[0]
    key.g2_beta, key.g2_delta = keep["beta2"].ctypes.data, keep["delta2"].ctypes.data
    key.g2_b, key.len_b2 = keep["B2"].ctypes.data, keep["B2"].shape[0]
    key.infinity_a, key.infinity_b = keep["ia"].ctypes.data, keep["ib"].ctypes.data
    key.nb_wires = keep["ia"].shape[0]
    key.nb_infinity_a, key.nb_infinity_b = int(keep["ia"].sum()), int(keep["ib"].sum())
    cks = [(g1(b), g1(e)) for b, e in commitment_keys]
    if cks:
        nck = len(cks)
        keep["cks"] = cks
        keep["bas"] = (C.c_void_p * nck)(*[b.ctypes.data for b, _ in cks])
        keep["sig"] = (C.c_void_p * nck)(*[e.ctypes.data for _, e in cks])
        keep["lens"] = (C.c_uint64 * nck)(*[b.shape[0] for b, _ in cks])
        key.nb_commitments, key.ck_basis, key.ck_basis_exp_sigma, key.ck_len = nck, keep["bas"], keep["sig"], keep["lens"]
    return key, keep


def WriteKey(ctx: Context, curve, fileobj, fmt: int = KEY_FORMAT_COMPRESSED, **key_fields) -> int:
    """ProvingKey.WriteTo (compressed) / WriteRawTo / WriteDump (marshal.go:231-300,378-445) of a key given by its host arrays
    (same keyword fields as ProvingKey); fileobj: an open binary file.  Returns the number of bytes written."""
    key, keep = _key_struct(curve, **key_fields)
    fileobj.flush()
    n = C.c_uint64()
    ctx.lib.check(ctx.lib.ga_g16_key_write_fd(ctx.handle, C.byref(key), int(fmt), fileobj.fileno(), C.byref(n)))
    del keep
    return int(n.value)


def ParseProof(curve, data: bytes, lib=None, max_commitments: int = 64) -> "Proof":
    """Proof.ReadFrom (marshal.go:62-86) on WriteTo or WriteRawTo bytes"""
    lib = lib or _lib.load()
    cid = curve_id(curve)
    fp = FP_LIMBS[cid]
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.zeros(8 * fp, dtype=np.uint64)
    coms = np.zeros((max_commitments, 2 * fp), dtype=np.uint64)
    pok = np.zeros(2 * fp, dtype=np.uint64)
    n, used = C.c_uint32(), C.c_size_t()
    lib.check(lib.ga_g16_proof_unmarshal(cid, _ptr(buf), buf.shape[0], _ptr(out), _ptr(coms), max_commitments, C.byref(n), _ptr(pok), C.byref(used)))
    p = Proof(cid, out[: 2 * fp].copy(), out[2 * fp: 6 * fp].copy(), out[6 * fp:].copy(), lib)
    p.Commitments, p.CommitmentPok, p.bytes_read = coms[: n.value].copy(), pok, int(used.value)
    return p


class ProvingKey:
    """setup.go:25-48 fields, uploaded once ("PinToGPU"); free with FreeGPUResources() (icicle.go:1493)."""

    @classmethod
    def from_handle(cls, ctx: Context, curve, handle, *, nb_wires: int, domain_cardinality: int, shard=(0, 1)):
        """wrap a key that was built directly through ga_g16_builder_* (gnark_amd/synth.py pin_key_chunked)"""
        self = cls.__new__(cls)
        self.ctx, self.curve, self.shard = ctx, curve_id(curve), (int(shard[0]), int(shard[1]))
        self.handle, self.nb_wires, self.domain_cardinality, self.nb_commitments = handle, int(nb_wires), int(domain_cardinality), 0
        return self

    @classmethod
    def ReadFrom(cls, ctx: Context, curve, source, *, precompute: int = 0, shard=(0, 1), k_remove=()):
        """ProvingKey.ReadFrom / UnsafeReadFrom / ReadDump (marshal.go:305-373,449-539) straight into HBM: source is bytes or an
        open binary file; the format (compressed / raw points / dump) is recognised from the stream.  k_remove: see __init__."""
        cid = curve_id(curve)
        self = cls.__new__(cls)
        self.ctx, self.curve, self.shard = ctx, cid, (int(shard[0]), int(shard[1]))
        rem = np.ascontiguousarray(k_remove, dtype=np.uint64)
        h, used = C.c_void_p(), C.c_uint64()
        lib = ctx.lib
        if isinstance(source, (bytes, bytearray, memoryview)):
            buf = np.frombuffer(bytes(source), dtype=np.uint8)
            lib.check(lib.ga_g16_pk_read_mem(ctx.handle, cid, _ptr(buf), buf.shape[0], int(precompute), int(shard[0]), int(shard[1]),
                                             _ptr(rem) if rem.size else None, rem.size, C.byref(h), C.byref(used)))
        else:
            lib.check(lib.ga_g16_pk_read_fd(ctx.handle, cid, source.fileno(), int(precompute), int(shard[0]), int(shard[1]),
                                            _ptr(rem) if rem.size else None, rem.size, C.byref(h), C.byref(used)))
        self.handle, self.bytes_read = h, int(used.value)
        lay = ShardLayout(self)
        self.nb_wires, self.domain_cardinality, self.nb_commitments = lay["nb_wires"], lay["n"], None
        return self

    def __init__(self, ctx: Context, curve, *, domain_cardinality, alpha1, beta1, delta1, A, B, Z, K, beta2, delta2, B2,
                 infinityA, infinityB, precompute: int = 0, shard=(0, 1), commitment_keys=(), k_remove=(), staged_chunk: int = 0,
                 window_shard=(0, 1)):
        """commitment_keys: [(Basis, BasisExpSigma)] per pk.CommitmentKeys[i] (setup.go:276-287); k_remove: sorted wire ids of the
        private committed wires and the commitment wires, left out of the K MSM (prove.go:231-235).
        staged_chunk > 0 builds the key through ga_g16_builder_* in chunks of that many points (the cgo-safe call pattern of
        go/backend/accelerated/mi355x) instead of the struct-of-pointers ga_g16_pk_create."""
        cid = curve_id(curve)
        fp = FP_LIMBS[cid]
        self.ctx, self.curve = ctx, cid
        g1 = lambda v: as_u64(np.asarray(v).reshape(-1, 2 * fp), 2 * fp)
        g2 = lambda v: as_u64(np.asarray(v).reshape(-1, 4 * fp), 4 * fp)
        A, B, Z, K, B2 = g1(A), g1(B), g1(Z), g1(K), g2(B2)
        alpha1, beta1, delta1, beta2, delta2 = g1(alpha1), g1(beta1), g1(delta1), g2(beta2), g2(delta2)
        ia = np.ascontiguousarray(infinityA, dtype=np.uint8)
        ib = np.ascontiguousarray(infinityB, dtype=np.uint8)
        if ia.shape != ib.shape:
            raise ValueError("InfinityA and InfinityB must have nbWires entries each")
        self.shard = (int(shard[0]), int(shard[1]))
        self.nb_commitments = len(commitment_keys)
        self.nb_wires = int(ia.shape[0])
        self.domain_cardinality = int(domain_cardinality)
        if staged_chunk:
            lib = ctx.lib
            b = C.c_void_p()
            lib.check(lib.ga_g16_builder_create(ctx.handle, cid, int(domain_cardinality), ia.shape[0], int(shard[0]), int(shard[1]), C.byref(b)))
            try:
                for which, arr in enumerate((A, B, Z, K, B2)):
                    lib.check(lib.ga_g16_builder_reserve(b, which, arr.shape[0]))
                    for lo in range(0, arr.shape[0], int(staged_chunk)):
                        part = arr[lo:lo + int(staged_chunk)].copy()   # a fresh buffer per call, wiped afterwards: nothing may be retained
                        lib.check(lib.ga_g16_builder_append(b, which, _ptr(part), part.shape[0]))
                        part[:] = 0
                for which, pt in enumerate((alpha1, beta1, delta1, beta2, delta2)):
                    lib.check(lib.ga_g16_builder_set_point(b, which, _ptr(pt)))
                lib.check(lib.ga_g16_builder_set_infinity(b, 0, _ptr(ia), ia.shape[0]))
                lib.check(lib.ga_g16_builder_set_infinity(b, 1, _ptr(ib), ib.shape[0]))
                for bas, sig in commitment_keys:
                    bas, sig = g1(bas), g1(sig)
                    if bas.shape != sig.shape:
                        raise ValueError("Basis and BasisExpSigma must have the same length")
                    lib.check(lib.ga_g16_builder_add_commitment_key(b, _ptr(bas), _ptr(sig), bas.shape[0]))
                rem = np.ascontiguousarray(k_remove, dtype=np.uint64)
                if rem.size:
                    lib.check(lib.ga_g16_builder_set_k_remove(b, _ptr(rem), rem.size))
                if int(window_shard[1]) > 1:
                    lib.check(lib.ga_g16_builder_set_window_shard(b, int(window_shard[0]), int(window_shard[1])))
            except Exception:
                lib.ga_g16_builder_destroy(b)
                raise
            h = C.c_void_p()
            lib.check(lib.ga_g16_builder_finish(b, int(precompute), C.byref(h)))
            self.handle = h
            return
        key = _lib.G16Key()
        key.curve, key.domain_cardinality = cid, int(domain_cardinality)
        key.g1_alpha, key.g1_beta, key.g1_delta = alpha1.ctypes.data, beta1.ctypes.data, delta1.ctypes.data
        key.g1_a, key.len_a = A.ctypes.data, A.shape[0]
        key.g1_b, key.len_b = B.ctypes.data, B.shape[0]
        key.g1_z, key.len_z = Z.ctypes.data, Z.shape[0]
        key.g1_k, key.len_k = K.ctypes.data, K.shape[0]
        key.g2_beta, key.g2_delta = beta2.ctypes.data, delta2.ctypes.data
        key.g2_b, key.len_b2 = B2.ctypes.data, B2.shape[0]
        key.infinity_a, key.infinity_b = ia.ctypes.data, ib.ctypes.data
        key.nb_wires = ia.shape[0]
        key.nb_infinity_a, key.nb_infinity_b = int(ia.sum()), int(ib.sum())
        key.precompute = int(precompute)   # 0 auto, 1 always, -1 never (window-multiple tables, ga_g16_key.precompute)
        key.shard_index, key.shard_count = int(shard[0]), int(shard[1])   # multi-GPU: pin slice k of N of every base vector
        key.window_shard_index, key.window_shard_count = int(window_shard[0]), int(window_shard[1])
        self.shard = (int(shard[0]), int(shard[1]))
        cks = [(g1(b), g1(e)) for b, e in commitment_keys]
        if cks:
            nck = len(cks)
            bas = (C.c_void_p * nck)(*[b.ctypes.data for b, _ in cks])
            sig = (C.c_void_p * nck)(*[e.ctypes.data for _, e in cks])
            lens = (C.c_uint64 * nck)(*[b.shape[0] for b, _ in cks])
            for b, e in cks:
                if b.shape != e.shape:
                    raise ValueError("Basis and BasisExpSigma must have the same length")
            key.nb_commitments, key.ck_basis, key.ck_basis_exp_sigma, key.ck_len = nck, bas, sig, lens
        rem = np.ascontiguousarray(k_remove, dtype=np.uint64)
        if rem.size:
            key.k_remove, key.len_k_remove = rem.ctypes.data_as(C.POINTER(C.c_uint64)), rem.size
        self.nb_commitments = len(cks)
        h = C.c_void_p()
        ctx.lib.check(ctx.lib.ga_g16_pk_create(ctx.handle, C.byref(key), C.byref(h)))
        self.handle = h
        self.nb_wires = int(key.nb_wires)
        self.domain_cardinality = int(domain_cardinality)

    def Commit(self, index: int, values):
        """pk.CommitmentKeys[index].Commit(values) and .ProveKnowledge(values) (prove.go:84,114) on the pinned bases:
        returns (commitment, pok) as G1Affine images."""
        if self.handle is None:
            raise _lib.GnarkAmdError("proving key has been freed")
        fp = FP_LIMBS[self.curve]
        v = as_u64(np.asarray(values, dtype=np.uint64).reshape(-1, 4), 4)
        com, pok = np.zeros(2 * fp, dtype=np.uint64), np.zeros(2 * fp, dtype=np.uint64)
        lib = self.ctx.lib
        lib.check(lib.ga_g16_commit(self.handle, int(index), _ptr(v) if v.shape[0] else None, v.shape[0], _ptr(com), _ptr(pok)))
        return com, pok

    def FreeGPUResources(self):
        if self.handle:
            self.ctx.lib.ga_g16_pk_destroy(self.handle)
            self.handle = None


def Prove(pk: ProvingKey, solution: Solution, nb_public: int, r: np.ndarray, s: np.ndarray) -> Proof:
    """groth16.Prove from the solver's output (prove.go:130-315).  r, s: fr.Element images (Montgomery) of the
    prover's randomness -- gnark samples them with crypto/rand (prove.go:171-177); the caller supplies them here."""
    if pk.handle is None:
        raise _lib.GnarkAmdError("proving key has been freed")
    W, A, B, Cc = (as_u64(x, 4) for x in (solution.W, solution.A, solution.B, solution.C))
    if W.shape[0] != pk.nb_wires:
        raise ValueError(f"len(W)={W.shape[0]} != nbWires={pk.nb_wires}")
    if not (A.shape == B.shape == Cc.shape):
        raise ValueError("A, B, C must have the same length")
    r, s = as_u64(np.asarray(r).reshape(1, 4), 4), as_u64(np.asarray(s).reshape(1, 4), 4)
    fp = FP_LIMBS[pk.curve]
    out = np.zeros(8 * fp, dtype=np.uint64)
    lib = pk.ctx.lib
    lib.check(lib.ga_g16_prove(pk.handle, _ptr(W), _ptr(A), _ptr(B), _ptr(Cc), A.shape[0], nb_public, _ptr(r), _ptr(s), _ptr(out)))
    return Proof(pk.curve, out[: 2 * fp].copy(), out[2 * fp: 6 * fp].copy(), out[6 * fp:].copy(), lib)


def ProveOneShot(ctx: Context, curve, solution: Solution, nb_public: int, r, s, *, domain_cardinality, alpha1, beta1, delta1, A, B, Z, K,
                 beta2, delta2, B2, infinityA, infinityB, k_remove=()) -> Proof:
    """groth16.Prove with the key NOT kept on the device (the reference's and the Go package's default, PinToGPU = false,
    icicle.go:797-805): ga_g16_prove_oneshot uploads the key as plain vectors WHILE the proof runs and frees it afterwards.  Same
    proof bytes as ProvingKey(precompute=-1) + Prove + FreeGPUResources."""
    cid = curve_id(curve)
    fp = FP_LIMBS[cid]
    g1 = lambda v: as_u64(np.asarray(v).reshape(-1, 2 * fp), 2 * fp)
    g2 = lambda v: as_u64(np.asarray(v).reshape(-1, 4 * fp), 4 * fp)
    A, B, Z, K, B2 = g1(A), g1(B), g1(Z), g1(K), g2(B2)
    alpha1, beta1, delta1, beta2, delta2 = g1(alpha1), g1(beta1), g1(delta1), g2(beta2), g2(delta2)
    ia = np.ascontiguousarray(infinityA, dtype=np.uint8)
    ib = np.ascontiguousarray(infinityB, dtype=np.uint8)
    if ia.shape != ib.shape:
        raise ValueError("InfinityA and InfinityB must have nbWires entries each")
    key = _lib.G16Key()
    key.curve, key.domain_cardinality = cid, int(domain_cardinality)
    key.g1_alpha, key.g1_beta, key.g1_delta = alpha1.ctypes.data, beta1.ctypes.data, delta1.ctypes.data
    key.g1_a, key.len_a = A.ctypes.data, A.shape[0]
    key.g1_b, key.len_b = B.ctypes.data, B.shape[0]
    key.g1_z, key.len_z = Z.ctypes.data, Z.shape[0]
    key.g1_k, key.len_k = K.ctypes.data, K.shape[0]
    key.g2_beta, key.g2_delta = beta2.ctypes.data, delta2.ctypes.data
    key.g2_b, key.len_b2 = B2.ctypes.data, B2.shape[0]
    key.infinity_a, key.infinity_b = ia.ctypes.data, ib.ctypes.data
    key.nb_wires = ia.shape[0]
    key.nb_infinity_a, key.nb_infinity_b = int(np.count_nonzero(ia)), int(np.count_nonzero(ib))
    key.precompute = -1
    key.shard_index, key.shard_count = 0, 1
    rem = np.ascontiguousarray(k_remove, dtype=np.uint64)
    if rem.size:
        key.k_remove, key.len_k_remove = rem.ctypes.data_as(C.POINTER(C.c_uint64)), rem.size
    W, Av, Bv, Cc = (as_u64(x, 4) for x in (solution.W, solution.A, solution.B, solution.C))
    if W.shape[0] != ia.shape[0]:
        raise ValueError(f"len(W)={W.shape[0]} != nbWires={ia.shape[0]}")
    if not (Av.shape == Bv.shape == Cc.shape):
        raise ValueError("A, B, C must have the same length")
    r, s = as_u64(np.asarray(r).reshape(1, 4), 4), as_u64(np.asarray(s).reshape(1, 4), 4)
    out = np.zeros(8 * fp, dtype=np.uint64)
    lib = ctx.lib
    lib.check(lib.ga_g16_prove_oneshot(ctx.handle, C.byref(key), _ptr(W), _ptr(Av), _ptr(Bv), _ptr(Cc), Av.shape[0], nb_public, _ptr(r), _ptr(s),
                                       _ptr(out)))
    return Proof(cid, out[: 2 * fp].copy(), out[2 * fp: 6 * fp].copy(), out[6 * fp:].copy(), lib)


def ProvePartial(pk: ProvingKey, solution: Solution, nb_public: int) -> np.ndarray:
    """Device part of a proof on this key's shard (ga_g16_prove_partial): Jacobian sums A | B1 | K+Z | B2 before
    randomisation, as one uint64 vector (3 G1Jac + 1 G2Jac) ready for an all_gather."""
    if pk.handle is None:
        raise _lib.GnarkAmdError("proving key has been freed")
    W, A, B, Cc = (as_u64(x, 4) for x in (solution.W, solution.A, solution.B, solution.C))
    if W.shape[0] != pk.nb_wires:
        raise ValueError(f"len(W)={W.shape[0]} != nbWires={pk.nb_wires}")
    fp = FP_LIMBS[pk.curve]
    out = np.zeros(3 * 3 * fp + 6 * fp, dtype=np.uint64)
    lib = pk.ctx.lib
    lib.check(lib.ga_g16_prove_partial(pk.handle, _ptr(W), _ptr(A), _ptr(B), _ptr(Cc), A.shape[0], nb_public, _ptr(out)))
    return out


def ShardLayout(pk: ProvingKey) -> dict:
    """where this key's shard sits: slice [off_z, off_z+len_z) of h / pk.G1.Z, wire range [w_lo, w_hi) of W (ga_g16_shard_layout)"""
    out = (C.c_uint64 * 8)()
    pk.ctx.lib.check(pk.ctx.lib.ga_g16_shard_layout(pk.handle, out))
    d = dict(zip(("off_z", "len_z", "w_lo", "w_hi", "n", "nb_wires", "win_index", "win_count"), (int(v) for v in out)))
    t = (C.c_uint64 * 2)()
    pk.ctx.lib.check(pk.ctx.lib.ga_g16_table_layout(pk.handle, t))
    d["tables"] = {k: bool(t[0] >> b & 1) for k, b in (("A", 0), ("B", 1), ("Z", 2), ("K", 3), ("B2", 4))}          # vectors with a window table
    d["wire_indexed"] = {k: bool(t[1] >> b & 1) for k, b in (("A", 0), ("B", 1), ("K", 3), ("B2", 4))}            # ... sharing the one witness sort
    return d


def WitnessPartial(pk: ProvingKey, W, nb_public: int) -> np.ndarray:
    """the four witness MSMs over this key's shard: A | B1 | K (G1Jac) | B2 (G2Jac) as one uint64 vector (ga_g16_witness_partial)"""
    W = as_u64(W, 4)
    if W.shape[0] != pk.nb_wires:
        raise ValueError(f"len(W)={W.shape[0]} != nbWires={pk.nb_wires}")
    fp = FP_LIMBS[pk.curve]
    out = np.zeros(15 * fp, dtype=np.uint64)
    pk.ctx.lib.check(pk.ctx.lib.ga_g16_witness_partial(pk.handle, _ptr(W), nb_public, _ptr(out)))
    return out


def HChain(pk: ProvingKey, v, out_dev_ptr: int):
    """out_dev <- n * FFT_coset(iFFT(v)) for one of the solver's A, B, C (ga_g16_h_chain); out_dev: n fr elements on pk's device.
    The chain is unscaled (the 1/n lives in HCombine's point-wise step): valid only as an input of HCombine."""
    v = as_u64(v, 4)
    pk.ctx.lib.check(pk.ctx.lib.ga_g16_h_chain(pk.handle, _ptr(v), v.shape[0], C.c_void_p(out_dev_ptr)))


def HChainDevice(pk: ProvingKey, buf_dev_ptr: int, n_constraints: int):
    """buf_dev (n fr elements of room, the first n_constraints filled) <- n * FFT_coset(iFFT(.)) in place (ga_g16_h_chain_dev);
    unscaled like HChain: valid only as an input of HCombine"""
    pk.ctx.lib.check(pk.ctx.lib.ga_g16_h_chain_dev(pk.handle, C.c_void_p(buf_dev_ptr), int(n_constraints)))


def HCombine(pk: ProvingKey, a_dev_ptr: int, b_dev_ptr: int, c_dev_ptr: int):
    """a_dev <- h = iFFT_coset((a*b - c)/(g^n - 1)), bit-reversed (ga_g16_h_combine)"""
    pk.ctx.lib.check(pk.ctx.lib.ga_g16_h_combine(pk.handle, C.c_void_p(a_dev_ptr), C.c_void_p(b_dev_ptr), C.c_void_p(c_dev_ptr)))


def ZPartial(pk: ProvingKey, h_slice_dev_ptr: int) -> np.ndarray:
    """MSM of this shard's slice of pk.G1.Z with the matching slice of h (device pointer to element off_z): G1Jac"""
    out = np.zeros(3 * FP_LIMBS[pk.curve], dtype=np.uint64)
    pk.ctx.lib.check(pk.ctx.lib.ga_g16_z_partial(pk.handle, C.c_void_p(h_slice_dev_ptr), _ptr(out)))
    return out


def ProveMulti(pks, solution: Solution, nb_public: int, r, s) -> Proof:
    """One proof over several devices from one process (ga_g16_prove_multi): pks[i] = shard i of len(pks), each in its own
    Context.  What the Go shim calls under mi355x.WithDevices."""
    W, A, B, Cc = (as_u64(x, 4) for x in (solution.W, solution.A, solution.B, solution.C))
    r, s = as_u64(np.asarray(r).reshape(1, 4), 4), as_u64(np.asarray(s).reshape(1, 4), 4)
    pk0 = pks[0]
    fp = FP_LIMBS[pk0.curve]
    out = np.zeros(8 * fp, dtype=np.uint64)
    hs = (C.c_void_p * len(pks))(*[p.handle for p in pks])
    lib = pk0.ctx.lib
    lib.check(lib.ga_g16_prove_multi(hs, len(pks), _ptr(W), _ptr(A), _ptr(B), _ptr(Cc), A.shape[0], nb_public, _ptr(r), _ptr(s), _ptr(out)))
    return Proof(pk0.curve, out[: 2 * fp].copy(), out[2 * fp: 6 * fp].copy(), out[6 * fp:].copy(), lib)


def SumPartials(curve, parts, lib=None) -> np.ndarray:
    """component-wise group addition of several ProvePartial outputs (host arithmetic, ga_jac_add)"""
    from . import ecc
    cid = curve_id(curve)
    fp = FP_LIMBS[cid]
    cuts = [(0, 3 * fp, 0), (3 * fp, 6 * fp, 0), (6 * fp, 9 * fp, 0), (9 * fp, 15 * fp, 1)]
    acc = np.ascontiguousarray(parts[0], dtype=np.uint64).copy()
    for p in parts[1:]:
        for lo, hi, grp in cuts:
            acc[lo:hi] = ecc.jac_add(cid, grp, acc[lo:hi], np.ascontiguousarray(p[lo:hi]), lib=lib)
    return acc


def Finish(pk: ProvingKey, partials_sum: np.ndarray, r: np.ndarray, s: np.ndarray) -> Proof:
    """host epilogue with the prover's randomness on the summed partials (ga_g16_finish)"""
    fp = FP_LIMBS[pk.curve]
    ps = np.ascontiguousarray(partials_sum, dtype=np.uint64)
    r, s = as_u64(np.asarray(r).reshape(1, 4), 4), as_u64(np.asarray(s).reshape(1, 4), 4)
    out = np.zeros(8 * fp, dtype=np.uint64)
    lib = pk.ctx.lib
    lib.check(lib.ga_g16_finish(pk.handle, _ptr(ps), _ptr(r), _ptr(s), _ptr(out)))
    return Proof(pk.curve, out[: 2 * fp].copy(), out[2 * fp: 6 * fp].copy(), out[6 * fp:].copy(), lib)


# ---- BSB22 host helpers (what the Go shim gets from gnark-crypto; here through the library's host code) -------------------
COMMITMENT_DST = b"bsb22-commitment"   # constraint/commitment.go:7
FOLD_DST = b"G16-BSB22"                # prove.go:123: the fold challenge hashes the commitment WIRE VALUES (32-byte BE each), not the points


def HashToField(curve, msg: bytes, dst: bytes, count: int = 1, lib=None) -> np.ndarray:
    """fr.Hash(msg, dst, count) -> (count, 4) fr.Element images (Montgomery)."""
    lib = lib or _lib.load()
    m = np.frombuffer(bytes(msg), dtype=np.uint8) if len(msg) else np.zeros(1, dtype=np.uint8)
    d = np.frombuffer(bytes(dst), dtype=np.uint8)
    out = np.zeros((count, 4), dtype=np.uint64)
    lib.check(lib.ga_hash_to_field(curve_id(curve), _ptr(m), len(msg), _ptr(d), len(dst), count, _ptr(out)))
    return out


def MarshalG1(curve, affine, lib=None) -> bytes:
    """G1Affine.Marshal(): uncompressed big-endian x | y (what SerializeCommitment hashes, constraint/commitment.go:76-89)."""
    lib = lib or _lib.load()
    cid = curve_id(curve)
    a = as_u64(np.asarray(affine).reshape(1, 2 * FP_LIMBS[cid]), 2 * FP_LIMBS[cid])
    out = np.zeros(96, dtype=np.uint8)
    n = C.c_size_t()
    lib.check(lib.ga_g1_marshal_uncompressed(cid, _ptr(a), _ptr(out), out.nbytes, C.byref(n)))
    return out[: n.value].tobytes()


def FoldPok(curve, poks, challenge, lib=None) -> np.ndarray:
    """proof.CommitmentPok.Fold(poks, challenge) (prove.go:127): sum_i challenge^i poks[i]."""
    lib = lib or _lib.load()
    cid = curve_id(curve)
    fp = FP_LIMBS[cid]
    p = as_u64(np.asarray(poks).reshape(-1, 2 * fp), 2 * fp)
    ch = as_u64(np.asarray(challenge).reshape(1, 4), 4)
    out = np.zeros(2 * fp, dtype=np.uint64)
    lib.check(lib.ga_g16_fold_pok(cid, _ptr(p), p.shape[0], _ptr(ch), _ptr(out)))
    return out
