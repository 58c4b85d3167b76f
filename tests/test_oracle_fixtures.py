"""Pin the oracle to the fixtures the reference ships.  Inputs: tests/golden/, generated from
/root/reference by tools/make_golden.py (committed, so these run on any box)."""
import base64
import json
import os

import numpy as np
import pytest

import oracle
import pyref
from helpers import BLS12_381, BN254, arr_to_g1_affine, arr_to_g2_affine, fr_to_arr, g1_to_arr, jac_to_affine_py

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def kzg():
    return np.load(os.path.join(GOLD, "kzg4096_bls12381.npz"))


def test_g2_2_65_literals():
    """std/algebra/emulated/sw_bn254/g2.go:87-96 and sw_bls12381/g2.go:100-110: [2^65]G2."""
    P = pyref.g2_group(BN254).mul(BN254.g2, 1 << 65)
    assert P == ((6099622139700402640581725571890015148411145321742729577177999911575645303725,
                  9870328428465937988383794519490899227160817120884239055108452134207619193487),
                 (16268382111792290652321980382595025991160708296314050973435867558225525677485,
                  15377126855853471483498618408547895055706247905282062963450025729940352455943))
    Q = pyref.g2_group(BLS12_381).mul(BLS12_381.g2, 1 << 65)
    assert Q == ((1307001654908388153254394944417118155033503188409787277795273489312551176370209873126740711463572657296916966732684,
                  1066804690119577865989830850277879393407029322116864061755683314318400220056817483617033672656485029228353937929571),
                 (1233864651366532660795929818904272589705597977637697925481983092108793193162343169655985724823869788077854535468808,
                  2703972434797875065063829955607449483769333186572810763171217085444622779819503421195150761462859837038921185079043))
    # the C oracle's G2 arithmetic reproduces both literals
    for c, want in ((BN254, P), (BLS12_381, Q)):
        assert jac_to_affine_py(c, 1, oracle.generator_mul(c.cid, 1, 1 << 65)) == want


def test_kzg_setup_decompression_kats(kzg):
    c = BLS12_381
    comp = kzg["g1_monomial_compressed"].tobytes()
    for i in range(8):
        P = pyref.g1_decompress(c, comp[48 * i:48 * (i + 1)])
        assert P == arr_to_g1_affine(c, kzg["g1_monomial"][i])
        assert pyref.g1_compress(c, P) == comp[48 * i:48 * (i + 1)]
    assert arr_to_g1_affine(c, kzg["g1_monomial"][0]) == c.g1            # [tau^0]G = the generator
    comp2 = kzg["g2_monomial_compressed"].tobytes()
    for i in range(4):
        Q = pyref.g2_decompress(c, comp2[96 * i:96 * (i + 1)])
        assert Q == arr_to_g2_affine(c, kzg["g2_monomial"][i])
        assert pyref.g2_compress(c, Q) == comp2[96 * i:96 * (i + 1)]
    assert arr_to_g2_affine(c, kzg["g2_monomial"][0]) == c.g2


def test_kzg_lagrange_relations_pin_msm_and_root_of_unity(kzg):
    """sum_i L_i = G and sum_i w^i L_i = [tau]G with w = 7^((r-1)/4096): a 4096-point BLS12-381 MSM of a real ceremony
    that only holds for gnark's root of unity and natural ordering."""
    c = BLS12_381
    n = 4096
    L = kzg["g1_lagrange"]
    ones = fr_to_arr(c, [1] * n)
    assert jac_to_affine_py(c, 0, oracle.msm(c.cid, 0, L, ones, nthreads=8)) == c.g1
    w = c.fr_root_of_unity(n)
    pw = [pow(w, i, c.r) for i in range(n)]
    got = jac_to_affine_py(c, 0, oracle.msm(c.cid, 0, L, fr_to_arr(c, pw), nthreads=8))
    assert got == arr_to_g1_affine(c, kzg["g1_monomial"][1])


def test_kzg_joint_msm_ntt_relation(kzg):
    """MSM(p, g1_monomial) == MSM(NTT(p), g1_lagrange) for a random polynomial p (T4): pins the oracle's FFT direction,
    ordering and MSM together against the ceremony file."""
    c = BLS12_381
    n = 4096
    rng = pyref.Xoshiro(4844)
    p = fr_to_arr(c, [rng.field(c.r) for _ in range(n)])
    evals_bitrev = oracle.fft(c.cid, p, 0, pyref.DIF, False)          # natural -> bit-reversed evaluations
    idx = np.array([pyref.bitrev(i, 12) for i in range(n)])
    evals = evals_bitrev[idx]                                          # natural-order evaluations p(w^i)
    lhs = jac_to_affine_py(c, 0, oracle.msm(c.cid, 0, kzg["g1_monomial"], p, nthreads=8))
    rhs = jac_to_affine_py(c, 0, oracle.msm(c.cid, 0, kzg["g1_lagrange"], evals, nthreads=8))
    assert lhs == rhs and lhs is not None


@pytest.mark.parametrize("name,c", [("bn254", BN254), ("bls12381", BLS12_381)])
def test_serialized_verifying_keys_decode(name, c):
    """backend/solidity/testdata/*.vk: layout of marshal.go:99-125 -- every point decodes on-curve, G2 points are in
    the r-torsion, and re-compression reproduces the file bytes (pins the compressed encodings used for proofs)."""
    raw = open(os.path.join(GOLD, f"vk_blank_groth16_{name}_nocommit.bin"), "rb").read()
    b = c.fp_bytes
    off = 0
    G2 = pyref.g2_group(c)

    def g1():
        nonlocal off
        P = pyref.g1_decompress(c, raw[off:off + b])
        assert pyref.g1_compress(c, P) == raw[off:off + b]
        off += b
        return P

    def g2():
        nonlocal off
        Q = pyref.g2_decompress(c, raw[off:off + 2 * b])
        assert pyref.g2_compress(c, Q) == raw[off:off + 2 * b]
        assert G2.mul(Q, c.r) is None
        off += 2 * b
        return Q
    g1(); g1(); g2(); g2(); g1(); g2()           # alpha1 beta1 beta2 gamma2 delta1 delta2
    nk = int.from_bytes(raw[off:off + 4], "big")
    off += 4
    assert nk == 4
    for _ in range(nk):
        g1()
    assert len(raw) - off == 8 and raw[off:] == bytes(8)


def test_bellman_tuple_points_decode():
    """backend/groth16/bellman_test.go:26-40: the (vk, proof) blobs are ZCash-format BLS12-381 points; the proof is
    A (G1) | B (G2) | C (G1) -- the wire order of marshal.go:41-49."""
    c = BLS12_381
    t = json.load(open(os.path.join(GOLD, "bellman_bls12381.json")))
    proof = base64.b64decode(t["proof"])
    assert len(proof) == 192
    A = pyref.g1_decompress(c, proof[:48])
    B = pyref.g2_decompress(c, proof[48:144])
    Cp = pyref.g1_decompress(c, proof[144:])
    assert pyref.g1_group(c).mul(A, c.r) is None and pyref.g2_group(c).mul(B, c.r) is None and pyref.g1_group(c).mul(Cp, c.r) is None
    assert pyref.g1_compress(c, A) + pyref.g2_compress(c, B) + pyref.g1_compress(c, Cp) == proof
    vk = base64.b64decode(t["vk"])
    pyref.g1_decompress(c, vk[:48])
    pyref.g1_decompress(c, vk[48:96])
    pyref.g2_decompress(c, vk[96:192])


def test_expand_message_xmd_reference_vectors():
    """std/hash/expand/expand_test.go:52-140 (16 vectors: 32, 48 and 128 output bytes) pin the oracle's hash-to-field, i.e. the
    BSB22 commitment hint (prove.go:88-98) and the PoK fold challenge (prove.go:123)."""
    import json
    g = json.load(open(os.path.join(GOLD, "expand_msg_xmd.json")))
    assert len(g["vectors"]) == 16
    for v in g["vectors"]:
        assert pyref.expand_message_xmd(v["msg"].encode(), g["dst"].encode(), v["len_in_bytes"]).hex() == v["uniform_bytes_hex"]


def test_groth16_bsb22_equation_in_the_exponent():
    """Groth16 with commitments (setup.go:133-178,260-287, prove.go:60-127,231-235, verify.go:75-125) restated in the oracle:
    the verifier's equation holds in the exponent, with the commitments standing in for the private committed wires, and the
    folded proof of knowledge is [sigma_i]-related to the commitments."""
    for c in (pyref.BN254, pyref.BLS12_381):
        rng = pyref.Xoshiro(99)
        cs = pyref.commit_r1cs()
        toxic = [rng.field(c.r) for _ in range(8)]
        pk, vk, dl = pyref.groth16_setup(c, cs, toxic)
        assert len(pk.K) == 2 and [len(b) for b, _ in pk.commitment_keys] == [2, 1] and len(vk.K) == 4
        w = pyref.commit_solve(c, cs, 5, 7, lambda i, ww: pyref.commitment_hint(pk, cs, i, ww)[1])
        r, s = rng.field(c.r), rng.field(c.r)
        ar, bs, krs, coms, pok = pyref.groth16_prove_bsb22(pk, cs, w, r, s)
        mod = c.r
        alpha, beta, gamma, delta, tau = toxic[:5]
        A, B, C = pyref.r1cs_solve(c, cs, w)
        a_dl = (alpha + sum(x * k for x, k in zip([w[i] for i in range(len(w)) if not pk.infinityA[i]], dl["A"])) + r * delta) % mod
        b_dl = (beta + sum(x * k for x, k in zip([w[i] for i in range(len(w)) if not pk.infinityB[i]], dl["B"])) + s * delta) % mod
        removed = {j for cm in cs.commitments for j in cm.private_committed} | {cm.commitment_index for cm in cs.commitments}
        kw = [w[i] for i in range(cs.nb_public, len(w)) if i not in removed]
        h = pyref.compute_h(c, A, B, C, pk.n)
        krs_dl = (sum(x * k for x, k in zip(kw, dl["K"])) + sum(x * k for x, k in zip(h, dl["Z"])) + s * a_dl + r * b_dl - r * s * delta) % mod
        G1 = pyref.g1_group(c)
        assert G1.mul(c.g1, krs_dl) == krs
        # kSum of verify.go:112-121 in the exponent: public wires, commitment wires (hash values) and the commitments themselves
        ginv = pow(gamma, -1, mod)
        com_dl = [sum(w[j] * k for j, k in zip(cm.private_committed, dl["CK"][i])) % mod for i, cm in enumerate(cs.commitments)]
        assert [G1.mul(c.g1, d) for d in com_dl] == coms
        vk_wires = [i for i in range(len(w)) if i < cs.nb_public or i in {cm.commitment_index for cm in cs.commitments}]
        ksum = (sum(w[i] * dl["kk"][i] * ginv for i in vk_wires) + sum(com_dl)) % mod
        assert a_dl * b_dl % mod == (alpha * beta + ksum * gamma + krs_dl * delta) % mod
        ser = b"".join(int(w[cm.commitment_index]).to_bytes(32, "big") for cm in cs.commitments)
        ch = pyref.fr_hash(c, ser, pyref.FOLD_DST, 1)[0]
        assert pok == G1.mul(c.g1, sum(sg * pow(ch, i, mod) * com_dl[i] for i, sg in enumerate(dl["sigmas"])) % mod)
        # proof wire format with commitments (marshal.go:33-58): 3 points, u32 count, commitments, pok
        nb = c.fp_bytes
        assert len(pyref.proof_bytes(c, ar, bs, krs, coms, pok)) == 4 * nb + 4 + 2 * nb + nb
