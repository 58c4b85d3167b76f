"""Golden-fixture directories in the layout a gnark checkout writes with
go/backend/accelerated/mi355x/internal/fixtures/gen_fixtures_test.go (TEST INFRASTRUCTURE).

A case directory holds what `groth16.Prove` saw and what it produced, byte for byte:

    meta.json      {"curve": "bn254" | "bls12-381", "nb_public": cs.GetNbPublicVariables() (the constant-one wire included),
                    "commitments": [{"private_committed": [...], "public_and_commitment_committed": [...], "commitment_index": k}, ...],
                    "producer": "..."}
    pk.bin         ProvingKey.WriteRawTo            (backend/groth16/bn254/marshal.go:231-300)
    solution.bin   R1CSSolution.WriteTo: W | A | B | C, each an fr.Vector (constraint/bn254/system.go:167-185; the vector framing is
                   gnark-crypto's: u32 big-endian length, then 32-byte big-endian canonical elements -- restated, not in the tree)
    r.bin, s.bin   the prover's randomness, 32-byte big-endian canonical (needs the groth16_rs.patch hook: prove.go:171-177 samples them)
    proof.bin      Proof.WriteTo  (compressed)      (marshal.go:33-58)
    proof.raw      Proof.WriteRawTo

`run_case` loads such a directory through the C ABI -- ga_g16_pk_read_mem on pk.bin, ga_g16_prove on the solution, for BSB22 circuits
ga_g16_commit + ga_g16_fold_pok -- and compares the proof bytes with the directory's.  Until a box with a Go toolchain has produced
real directories under tests/golden/gnark/, `write_case` produces the same layout from the Python oracle, so that the consumer is
exercised on every run; a directory made by gnark itself is then the first DIRECT byte-parity check against the reference's prover."""
import json
import os

import numpy as np

import pyref
from gnark_amd import groth16
from helpers import arr_to_fr, fr_to_arr

CURVES = {"bn254": pyref.BN254, "bls12-381": pyref.BLS12_381}


def _fr_vector(c, vals) -> bytes:
    return len(vals).to_bytes(4, "big") + b"".join(int(v % c.r).to_bytes(32, "big") for v in vals)


def _read_fr_vector(c, data: bytes, at: int):
    n = int.from_bytes(data[at:at + 4], "big")
    at += 4
    if at + 32 * n > len(data):
        raise ValueError("solution.bin: a vector of %d elements does not fit the remaining %d bytes" % (n, len(data) - at))
    vals = [int.from_bytes(data[at + 32 * i: at + 32 * i + 32], "big") for i in range(n)]
    if any(v >= c.r for v in vals):
        raise ValueError("solution.bin: element not below the field modulus")
    return vals, at + 32 * n


def parse_solution(c, data: bytes):
    """R1CSSolution.ReadFrom: W, A, B, C as lists of canonical integers"""
    out, at = [], 0
    for _ in range(4):
        v, at = _read_fr_vector(c, data, at)
        out.append(v)
    if at != len(data):
        raise ValueError("solution.bin: %d trailing bytes" % (len(data) - at))
    return out


def write_case(path, c, cs, pk, w, r, s, producer="oracle/pyref.py (layout self-test, NOT gnark)"):
    """the oracle's Setup/Prove written in the fixture layout (what gen_fixtures_test.go writes from gnark)"""
    os.makedirs(path, exist_ok=True)
    A, B, C = pyref.r1cs_solve(c, cs, w)
    if cs.commitments:
        ar, bs, krs, coms, pok = pyref.groth16_prove_bsb22(pk, cs, w, r, s)
    else:
        ar, bs, krs = pyref.groth16_prove(pk, cs, w, r, s)
        coms, pok = (), None
    meta = {"curve": c.name, "nb_public": cs.nb_public, "producer": producer,
            "commitments": [{"private_committed": list(cm.private_committed), "public_and_commitment_committed": list(cm.public_and_commitment_committed),
                             "commitment_index": cm.commitment_index} for cm in cs.commitments]}
    files = {"meta.json": json.dumps(meta, indent=1).encode(), "pk.bin": pyref.pk_write(pk, raw=True),
             "solution.bin": _fr_vector(c, w) + _fr_vector(c, A) + _fr_vector(c, B) + _fr_vector(c, C),
             "r.bin": int(r).to_bytes(32, "big"), "s.bin": int(s).to_bytes(32, "big"),
             "proof.bin": pyref.proof_bytes(c, ar, bs, krs, coms, pok), "proof.raw": pyref.proof_bytes_raw(c, ar, bs, krs, coms, pok)}
    for name, data in files.items():
        with open(os.path.join(path, name), "wb") as f:
            f.write(data)
    return path


def case_dirs(root):
    """every directory under root that holds a complete case"""
    if not os.path.isdir(root):
        return []
    need = ("meta.json", "pk.bin", "solution.bin", "r.bin", "s.bin", "proof.bin")
    return sorted(os.path.join(root, d) for d in os.listdir(root) if all(os.path.exists(os.path.join(root, d, f)) for f in need))


def run_case(ctx, path, precompute=0):
    """prove the case on ctx's device through the C ABI; returns (proof bytes, proof raw bytes or None, expected, expected raw or None)"""
    meta = json.load(open(os.path.join(path, "meta.json")))
    c = CURVES[meta["curve"]]
    rd = lambda name: open(os.path.join(path, name), "rb").read()
    w, A, B, C = parse_solution(c, rd("solution.bin"))
    r, s = int.from_bytes(rd("r.bin"), "big"), int.from_bytes(rd("s.bin"), "big")
    cms = meta.get("commitments") or []
    removed = sorted({j for cm in cms for j in cm["private_committed"]} | {cm["commitment_index"] for cm in cms})
    lib = ctx.lib
    dpk = groth16.ProvingKey.ReadFrom(ctx, c.name, rd("pk.bin"), precompute=precompute, k_remove=removed)
    try:
        if dpk.nb_wires != len(w):
            raise ValueError("pk.bin has %d wires, solution.bin %d" % (dpk.nb_wires, len(w)))
        sol = groth16.Solution(W=fr_to_arr(c, w), A=fr_to_arr(c, A), B=fr_to_arr(c, B), C=fr_to_arr(c, C))
        proof = groth16.Prove(dpk, sol, int(meta["nb_public"]), fr_to_arr(c, [r]), fr_to_arr(c, [s]))
        if cms:   # prove.go:60-127: commitments + folded proof of knowledge from the solved wire values
            coms, poks = [], []
            for i, cm in enumerate(cms):
                com, pok = dpk.Commit(i, fr_to_arr(c, [w[j] for j in cm["private_committed"]]))
                coms.append(com)
                poks.append(pok)
            ser = b"".join(int(w[cm["commitment_index"]]).to_bytes(32, "big") for cm in cms)
            challenge = groth16.HashToField(c.name, ser, groth16.FOLD_DST, 1, lib=lib)
            proof.Commitments = np.stack(coms)
            proof.CommitmentPok = groth16.FoldPok(c.name, np.stack(poks), challenge, lib=lib)
    finally:
        dpk.FreeGPUResources()
    raw_path = os.path.join(path, "proof.raw")
    return proof.WriteTo(), proof.WriteRawTo(), rd("proof.bin"), (rd("proof.raw") if os.path.exists(raw_path) else None)


def oracle_cases(tmp_root, curves=("bn254", "bls12-381")):
    """examples/cubic and the two-commitment circuit, both curves, written by the oracle in the fixture layout"""
    out = []
    for name in curves:
        c = CURVES[name]
        rng = pyref.Xoshiro(0x601D + len(name))
        cs, w = pyref.cubic_r1cs(), pyref.cubic_witness(3)
        pk, _, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(5)])
        out.append(write_case(os.path.join(tmp_root, "cubic_" + name), c, cs, pk, w, rng.field(c.r), rng.field(c.r)))
        cs2 = pyref.commit_r1cs()
        pk2, _, _ = pyref.groth16_setup(c, cs2, [rng.field(c.r) for _ in range(5 + len(cs2.commitments) + 1)])
        w2 = pyref.commit_solve(c, cs2, 3, 11, lambda i, ww: pyref.commitment_hint(pk2, cs2, i, ww)[1])
        out.append(write_case(os.path.join(tmp_root, "two_commitments_" + name), c, cs2, pk2, w2, rng.field(c.r), rng.field(c.r)))
    return out
