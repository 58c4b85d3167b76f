#!/usr/bin/env python3
"""Generate tests/golden/ from the fixtures the reference ships (run in the build container, where /root/reference
exists; the GPU box only sees the committed outputs).

  kzg4096_bls12381.npz   std/evmprecompiles/kzg_trusted_setup.json (EIP-4844 ceremony): g1_monomial, g1_lagrange
                         (4096 each) and g2_monomial[0..3], decompressed with oracle/pyref.py to gnark memory images
                         (Montgomery limbs) + the compressed bytes of the first 8 entries (decompression KATs)
  vk_*.bin               backend/solidity/testdata/blank_groth16_{bn254,bls12381}_nocommit.vk (serialized VKs, raw)
  bellman_bls12381.json  the first (vk, proof, inputs) tuple of backend/groth16/bellman_test.go:26-40 (base64 as in the file) and, under
                         "tuples", all twelve of :26-84 with their `ok` flags (6 the verifier must accept, 6 it must not)
  fft_constants.json     the domain constants the reference ships as literals: VK_DOMAIN_SIZE / VK_INV_DOMAIN_SIZE / VK_OMEGA /
                         VK_COSET_SHIFT of backend/solidity/testdata/blank_plonk_{bn254,bls12381}_nocommit.sol
  vk_blank_plonk_*.bin, vk_blank_groth16_*_commit.bin   the serialized keys of backend/solidity/testdata (decoded in the tests per
                         backend/plonk/bn254/marshal.go:177-203 and backend/groth16/bn254/marshal.go:151-230)
  expand_msg_xmd.json    the 16 expand_message_xmd (SHA-256) vectors of std/hash/expand/expand_test.go:44-140 (32, 48 and 128 output bytes)
"""
import base64
import json
import os
import re
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref  # noqa: E402
from helpers import g1_to_arr, g2_to_arr  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    c = pyref.BLS12_381
    ts = json.load(open(os.path.join(REF, "std/evmprecompiles/kzg_trusted_setup.json")))
    hexb = lambda s: bytes.fromhex(s[2:])
    mono = [pyref.g1_decompress(c, hexb(s)) for s in ts["g1_monomial"]]
    lag = [pyref.g1_decompress(c, hexb(s)) for s in ts["g1_lagrange"]]
    g2m = [pyref.g2_decompress(c, hexb(s)) for s in ts["g2_monomial"][:4]]
    np.savez_compressed(
        os.path.join(OUT, "kzg4096_bls12381.npz"),
        g1_monomial=g1_to_arr(c, mono), g1_lagrange=g1_to_arr(c, lag), g2_monomial=g2_to_arr(c, g2m),
        g1_monomial_compressed=np.frombuffer(b"".join(hexb(s) for s in ts["g1_monomial"][:8]), dtype=np.uint8),
        g1_lagrange_compressed=np.frombuffer(b"".join(hexb(s) for s in ts["g1_lagrange"][:8]), dtype=np.uint8),
        g2_monomial_compressed=np.frombuffer(b"".join(hexb(s) for s in ts["g2_monomial"][:4]), dtype=np.uint8))
    for name in ("blank_groth16_bn254_nocommit.vk", "blank_groth16_bls12381_nocommit.vk"):
        shutil.copyfile(os.path.join(REF, "backend/solidity/testdata", name), os.path.join(OUT, "vk_" + name.replace(".vk", ".bin")))
    src = open(os.path.join(REF, "backend/groth16/bellman_test.go")).read()
    strs = re.findall(r'"([A-Za-z0-9+/=]{40,})"', src)
    body = src[src.index("for _, test := range"):src.index("// decode verifying key")]
    tuples = re.findall(r'\{\s*"([A-Za-z0-9+/=]+)",\s*"([A-Za-z0-9+/=]+)",\s*"([A-Za-z0-9+/=]*)",\s*(true|false),\s*\}', body)
    assert len(tuples) == 12 and sum(t[3] == "true" for t in tuples) == 6
    json.dump({"vk": strs[0], "proof": strs[1], "inputs": strs[2] if len(strs) > 2 else "",
               "tuples": [{"vk": v, "proof": pr, "inputs": i, "ok": ok == "true"} for v, pr, i, ok in tuples]},
              open(os.path.join(OUT, "bellman_bls12381.json"), "w"), indent=0)
    # FFT-convention literals: the solidity templates rendered for the blank PLONK keys
    consts = {}
    for curve, fn in (("bn254", "blank_plonk_bn254_nocommit.sol"), ("bls12-381", "blank_plonk_bls12381_nocommit.sol")):
        sol = open(os.path.join(REF, "backend/solidity/testdata", fn)).read()
        get = lambda name: int(re.search(r"uint256 private constant %s = (\d+);" % name, sol).group(1))
        consts[curve] = {k: str(get(k)) for k in ("VK_DOMAIN_SIZE", "VK_INV_DOMAIN_SIZE", "VK_OMEGA", "VK_COSET_SHIFT", "R_MOD")
                         if re.search(r"uint256 private constant %s = " % k, sol)}
    json.dump(consts, open(os.path.join(OUT, "fft_constants.json"), "w"), indent=1)
    for name in ("blank_plonk_bn254_nocommit.vk", "blank_plonk_bls12381_nocommit.vk", "blank_plonk_bn254_commit.vk", "blank_plonk_bls12381_commit.vk",
                 "blank_groth16_bn254_commit.vk", "blank_groth16_bls12381_commit.vk"):
        if os.path.exists(os.path.join(REF, "backend/solidity/testdata", name)):
            shutil.copyfile(os.path.join(REF, "backend/solidity/testdata", name), os.path.join(OUT, "vk_" + name.replace(".vk", ".bin")))
    # expand_message_xmd known answers (std/hash/expand/expand_test.go:44-140, "adapted from gnark-crypto/field/hash"): they pin
    # the hash-to-field used by the BSB22 commitment hint and the PoK fold challenge
    src = open(os.path.join(REF, "std/hash/expand/expand_test.go")).read()
    dst = re.search(r'dst := "([^"]+)"', src).group(1)
    vecs = re.findall(r'\{\s*"([^"]*)",\s*(0x[0-9a-fA-F]+),\s*"([0-9a-f]+)",\s*\}', src)
    assert len(vecs) == 16
    json.dump({"dst": dst, "vectors": [{"msg": m, "len_in_bytes": int(n, 16), "uniform_bytes_hex": h} for m, n, h in vecs]},
              open(os.path.join(OUT, "expand_msg_xmd.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
