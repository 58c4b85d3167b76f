"""ctypes binding of libgnark_amd.so (include/gnark_amd.h).

The shared library is the product; this module only declares its prototypes.  There is no CPU
fallback: if the hipcc-built library is missing, `load()` raises with build instructions.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "libgnark_amd.so")

GA_OK = 0
BN254, BLS12_381 = 0, 1
G1, G2 = 0, 1
BASES_ON_DEVICE, SCALARS_ON_DEVICE, SCALARS_MONTGOMERY, TABLE_BATCHED = 0x1, 0x2, 0x4, 0x10
FFT_FORWARD, FFT_INVERSE = 0, 1
DIF, DIT = 0, 1


class GnarkAmdError(RuntimeError):
    pass


class G16Key(C.Structure):
    """struct ga_g16_key"""
    _fields_ = [
        ("curve", C.c_int),
        ("domain_cardinality", C.c_uint64),
        ("g1_alpha", C.c_void_p), ("g1_beta", C.c_void_p), ("g1_delta", C.c_void_p),
        ("g1_a", C.c_void_p), ("len_a", C.c_uint64),
        ("g1_b", C.c_void_p), ("len_b", C.c_uint64),
        ("g1_z", C.c_void_p), ("len_z", C.c_uint64),
        ("g1_k", C.c_void_p), ("len_k", C.c_uint64),
        ("g2_beta", C.c_void_p), ("g2_delta", C.c_void_p),
        ("g2_b", C.c_void_p), ("len_b2", C.c_uint64),
        ("infinity_a", C.c_void_p), ("infinity_b", C.c_void_p),
        ("nb_wires", C.c_uint64), ("nb_infinity_a", C.c_uint64), ("nb_infinity_b", C.c_uint64),
        ("precompute", C.c_int32),
        ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
        ("nb_commitments", C.c_uint32),
        ("ck_basis", C.POINTER(C.c_void_p)), ("ck_basis_exp_sigma", C.POINTER(C.c_void_p)), ("ck_len", C.POINTER(C.c_uint64)),
        ("k_remove", C.POINTER(C.c_uint64)), ("len_k_remove", C.c_uint64),
        ("window_shard_index", C.c_uint32), ("window_shard_count", C.c_uint32),
    ]


class PlonkQuotientIn(C.Structure):
    """struct ga_plonk_quotient_in"""
    _fields_ = [("nb_bsb", C.c_uint32)] + [(k, C.c_void_p) for k in ("l", "r", "o", "z", "ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3")] + [
        ("qcp", C.POINTER(C.c_void_p)), ("pi2", C.POINTER(C.c_void_p)),
        ("lagrange_mask", C.c_uint64),
        ("bl", C.c_void_p), ("br", C.c_void_p), ("bo", C.c_void_p), ("bz", C.c_void_p),
        ("alpha", C.c_void_p), ("beta", C.c_void_p), ("gamma", C.c_void_p),
        ("flags", C.c_uint32),
    ]


_P = C.c_void_p
_PROTOS = {
    "ga_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ga_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "ga_ctx_destroy": (None, [_P]),
    "ga_last_error": (C.c_char_p, []),
    "ga_version": (C.c_char_p, []),
    "ga_device_info": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ga_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "ga_free": (C.c_int, [_P, _P]),
    "ga_copy_to_device": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "ga_copy_to_host": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "ga_sync": (C.c_int, [_P]),
    "ga_msm": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_size_t, C.c_uint, _P]),
    "ga_msm_windows": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_size_t, C.c_uint, C.c_int, C.c_int, _P,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ga_msm_plan": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ga_msm_combine_windows": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "ga_msm_table_create": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t, C.c_uint, C.POINTER(_P)]),
    "ga_msm_table_destroy": (None, [_P]),
    "ga_msm_table_run": (C.c_int, [_P, _P, C.c_uint, _P]),
    "ga_msm_table_run_windows": (C.c_int, [_P, _P, C.c_uint, C.c_int, C.c_int, _P]),
    "ga_msm_table_run_batch": (C.c_int, [_P, _P, C.c_uint32, C.c_uint, _P]),
    "ga_msm_table_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "ga_jac_add": (C.c_int, [C.c_int, C.c_int, _P, _P, _P]),
    "ga_jac_to_affine": (C.c_int, [C.c_int, C.c_int, _P, _P]),
    "ga_jac_scalar_mul": (C.c_int, [C.c_int, C.c_int, _P, _P, _P]),
    "ga_domain_create": (C.c_int, [_P, C.c_int, C.c_uint64, C.POINTER(_P)]),
    "ga_domain_destroy": (None, [_P]),
    "ga_fft": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ga_compute_h": (C.c_int, [_P, _P, _P, _P, C.c_uint64, _P, C.c_int]),
    "ga_plonk_quotient": (C.c_int, [_P, _P, C.POINTER(PlonkQuotientIn), _P]),
    "ga_plonk_pk_create": (C.c_int, [_P, _P, C.POINTER(PlonkQuotientIn), C.POINTER(_P)]),
    "ga_plonk_pk_destroy": (None, [_P]),
    "ga_plonk_quotient_pinned": (C.c_int, [_P, C.POINTER(PlonkQuotientIn), _P]),
    "ga_plonk_build_z": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "ga_fr_linear_combination": (C.c_int, [_P, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), _P, _P, C.c_int]),
    "ga_fr_poly_evaluate": (C.c_int, [_P, C.c_int, _P, C.c_uint64, _P, _P, C.c_int]),
    "ga_kzg_open": (C.c_int, [_P, _P, C.c_size_t, C.c_uint, _P, _P, _P]),
    "ga_fr_vec_mul": (C.c_int, [_P, C.c_int, _P, _P, C.c_size_t, _P, C.c_int]),
    "ga_fr_batch_invert": (C.c_int, [_P, C.c_int, _P, C.c_uint64, C.c_int]),
    "ga_g16_pk_create": (C.c_int, [_P, C.POINTER(G16Key), C.POINTER(_P)]),
    "ga_g16_pk_destroy": (None, [_P]),
    "ga_g16_builder_create": (C.c_int, [_P, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "ga_g16_builder_reserve": (C.c_int, [_P, C.c_int, C.c_uint64]),
    "ga_g16_builder_append": (C.c_int, [_P, C.c_int, _P, C.c_uint64]),
    "ga_g16_builder_set_point": (C.c_int, [_P, C.c_int, _P]),
    "ga_g16_builder_set_infinity": (C.c_int, [_P, C.c_int, _P, C.c_uint64]),
    "ga_g16_builder_add_commitment_key": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "ga_g16_builder_set_k_remove": (C.c_int, [_P, _P, C.c_uint64]),
    "ga_g16_builder_set_window_shard": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "ga_g16_builder_finish": (C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    "ga_g16_builder_destroy": (None, [_P]),
    "ga_g16_prove": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P, _P]),
    "ga_g16_prove_oneshot": (C.c_int, [_P, C.POINTER(G16Key), _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P, _P]),
    "ga_g16_prove_partial": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P]),
    "ga_g16_finish": (C.c_int, [_P, _P, _P, _P, _P]),
    "ga_g16_shard_layout": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ga_g16_lane_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ga_g16_table_layout": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ga_g16_witness_partial": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "ga_g16_h_chain": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "ga_g16_h_chain_dev": (C.c_int, [_P, _P, C.c_uint64]),
    "ga_g16_h_combine": (C.c_int, [_P, _P, _P, _P]),
    "ga_g16_z_partial": (C.c_int, [_P, _P, _P]),
    "ga_g16_prove_multi": (C.c_int, [C.POINTER(_P), C.c_uint32, _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P, _P]),
    "ga_g16_pk_read_mem": (C.c_int, [_P, C.c_int, _P, C.c_size_t, C.c_int32, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "ga_g16_pk_read_fd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int32, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "ga_g16_key_write_fd": (C.c_int, [_P, C.POINTER(G16Key), C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "ga_g16_proof_unmarshal": (C.c_int, [C.c_int, _P, C.c_size_t, _P, _P, C.c_uint32, C.POINTER(C.c_uint32), _P, C.POINTER(C.c_size_t)]),
    "ga_point_unmarshal": (C.c_int, [C.c_int, C.c_int, _P, C.c_size_t, _P, C.POINTER(C.c_size_t)]),
    "ga_g16_proof_marshal": (C.c_int, [C.c_int, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ga_g16_commit": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, _P, _P]),
    "ga_g16_fold_pok": (C.c_int, [C.c_int, _P, C.c_uint64, _P, _P]),
    "ga_hash_to_field": (C.c_int, [C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.c_uint32, _P]),
    "ga_expand_message_xmd": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, _P]),
    "ga_g16_proof_marshal_bsb22": (C.c_int, [C.c_int, _P, _P, C.c_uint32, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ga_g16_proof_marshal_raw": (C.c_int, [C.c_int, _P, _P, C.c_uint32, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ga_g1_marshal_uncompressed": (C.c_int, [C.c_int, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ga_profile_enable": (C.c_int, [_P, C.c_int]),
    "ga_profile_reset": (C.c_int, [_P]),
    "ga_profile_read": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "ga_gen_bases": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64, C.c_size_t, _P, _P]),
    "ga_gen_bases_at": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_size_t, _P, _P]),
    "ga_gen_scalars": (C.c_int, [_P, C.c_int, C.c_uint64, C.c_size_t, _P]),
    "ga_fr_dot": (C.c_int, [_P, C.c_int, _P, _P, C.c_size_t, _P]),
    "ga_generator_mul": (C.c_int, [C.c_int, C.c_int, _P, _P]),
    "ga_microbench": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "ga_clock_probe": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_double)]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


class Library:
    """A loaded libgnark_amd with typed prototypes; `check(rc)` raises GnarkAmdError with ga_last_error()."""

    def __init__(self, path: str = DEFAULT_PATH):
        if not os.path.exists(path):
            raise GnarkAmdError(
                f"{path} not found: build the HIP library first "
                "(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gnark_amd/csrc -j8`). "
                "gnark_amd has no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self.dll, name)   # AttributeError here == a symbol of include/gnark_amd.h is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, rc: int):
        if rc != GA_OK:
            raise GnarkAmdError(f"libgnark_amd error {rc}: {self.dll.ga_last_error().decode()}")


_default = None


def load() -> Library:
    """The in-tree HIP build.  GA_LIB_PATH points at another hipcc build of the same sources (A/B experiments with compile-time
    knobs, tools/build_variant.sh); there is no CPU fallback either way."""
    global _default
    if _default is None:
        _default = Library(os.environ.get("GA_LIB_PATH", DEFAULT_PATH))
    return _default
