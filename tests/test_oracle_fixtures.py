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


# ---- pairing + Groth16 Verify, pinned by the reference's own (vk, proof, inputs, ok) tuples ---------------------------------------
def test_pairing_is_bilinear_and_nondegenerate():
    """the oracle's ate pairing (pyref.miller_loop / final_exponentiation) on both curves: e(aP, bQ) = e(P, Q)^(ab) != 1, order r"""
    for c in (BN254, BLS12_381):
        G1, G2, F12 = pyref.g1_group(c), pyref.g2_group(c), pyref.Fp12Ops(c)
        e = pyref.pairing(c, c.g1, c.g2)
        assert e != F12.one and F12.pow(e, c.r) == F12.one
        a, b = 0x1234567, 0x89ABCDEF01
        assert pyref.pairing(c, G1.mul(c.g1, a), G2.mul(c.g2, b)) == F12.pow(e, a * b % c.r)
        assert pyref.pairing_check(c, [(G1.mul(c.g1, a), c.g2), (G1.neg(c.g1), G2.mul(c.g2, a))])
        assert not pyref.pairing_check(c, [(G1.mul(c.g1, a), c.g2), (G1.neg(c.g1), G2.mul(c.g2, a + 1))])
        assert F12.mul(e, F12.inv(e)) == F12.one


def test_bellman_tuples_pin_the_verifier():
    """backend/groth16/bellman_test.go:26-84: twelve (vk, proof, inputs) tuples produced OUTSIDE gnark (bellman, BLS12-381); the
    reference's Verify accepts the six marked ok.  The oracle's restatement of verify.go:38-145 accepts exactly those six and
    rejects the other six -- the pin that lets the GPU tests check proof bytes with a verifier instead of known toxic waste."""
    c = BLS12_381
    t = json.load(open(os.path.join(GOLD, "bellman_bls12381.json")))["tuples"]
    assert len(t) == 12 and sum(x["ok"] for x in t) == 6
    for x in t:
        vkb, prb, inb = base64.b64decode(x["vk"]), base64.b64decode(x["proof"]), base64.b64decode(x["inputs"])
        vk, pacc, _, _, used = pyref.vk_read(c, vkb)
        assert pacc == [] and set(vkb[used:]) <= {0}            # (the test vectors are zero-padded for "commitment stuff", :92-93)
        ar, bs, krs = pyref.g1_decompress(c, prb[:48]), pyref.g2_decompress(c, prb[48:144]), pyref.g1_decompress(c, prb[144:192])
        pw = [int.from_bytes(inb[i:i + 32], "big") for i in range(0, len(inb), 32)]
        assert len(pw) == len(vk.K) - 1
        assert pyref.groth16_verify(c, vk, (ar, bs, krs), pw) is x["ok"], x["proof"][:16]


@pytest.mark.parametrize("c", [BN254, BLS12_381], ids=lambda c: c.name)
def test_oracle_verifier_accepts_oracle_proofs_and_rejects_tampering(c):
    """Setup -> Prove -> Verify inside the oracle (examples/cubic and the two-commitment circuit): the verifier pinned above
    accepts what the oracle's prover (prove.go:52-315 restated) produces, through the proof BYTES, and rejects a wrong public input,
    a tampered commitment and a proof point moved off its value."""
    rng = pyref.Xoshiro(0xBE11)
    G1 = pyref.g1_group(c)
    # cubic
    cs, w = pyref.cubic_r1cs(), pyref.cubic_witness(3)
    pk, vk, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(5)])
    r, s = rng.field(c.r), rng.field(c.r)
    blob = pyref.proof_bytes(c, *pyref.groth16_prove(pk, cs, w, r, s))
    ar, bs, krs, coms, pok = pyref.proof_read(c, blob)[:5]
    assert pyref.groth16_verify(c, vk, (ar, bs, krs, coms, pok), w[1:cs.nb_public])
    assert not pyref.groth16_verify(c, vk, (ar, bs, krs, coms, pok), [(w[1] + 1) % c.r])
    assert not pyref.groth16_verify(c, vk, (G1.add(ar, c.g1), bs, krs, coms, pok), w[1:cs.nb_public])
    # two commitments (BSB22)
    cs = pyref.commit_r1cs()
    pk, vk, _ = pyref.groth16_setup(c, cs, [rng.field(c.r) for _ in range(8)])
    w = pyref.commit_solve(c, cs, 5, 7, lambda i, ww: pyref.commitment_hint(pk, cs, i, ww)[1])
    blob = pyref.proof_bytes(c, *pyref.groth16_prove_bsb22(pk, cs, w, r, s))
    proof = pyref.proof_read(c, blob)[:5]
    pacc = pyref.vk_public_and_commitment_committed(cs)
    assert pacc == [[1], [2]]
    assert pyref.groth16_verify(c, vk, proof, w[1:cs.nb_public], pacc)
    assert not pyref.groth16_verify(c, vk, proof, [(w[1] + 1) % c.r], pacc)
    bad = (proof[0], proof[1], proof[2], [G1.add(proof[3][0], c.g1), proof[3][1]], proof[4])
    assert not pyref.groth16_verify(c, vk, bad, w[1:cs.nb_public], pacc)


# ---- the FFT convention and the remaining encodings, from literals the reference ships ------------------------------------------
def _fr_be(b):
    return int.from_bytes(b, "big")


@pytest.mark.parametrize("c", [BN254, BLS12_381], ids=lambda c: c.name)
def test_domain_constants_match_the_reference_literals(c, emu_ctx):
    """backend/solidity/testdata/blank_plonk_{bn254,bls12381}_nocommit.sol:42-65 (VK_DOMAIN_SIZE, VK_INV_DOMAIN_SIZE, VK_OMEGA,
    VK_COSET_SHIFT) and the head of the serialized blank_plonk_*.vk (backend/plonk/bn254/marshal.go:177-203: Size, SizeInv,
    Generator, NbPublicVariables, CosetShift): gnark's root of unity / coset shift as literals, against the oracle's domain
    AND the library's (a size-8 transform of the unit vector e_1 spells out the powers of the domain generator)."""
    from gnark_amd import fft
    k = json.load(open(os.path.join(GOLD, "fft_constants.json")))[c.name]
    n, ninv, omega, shift = (int(k[x]) for x in ("VK_DOMAIN_SIZE", "VK_INV_DOMAIN_SIZE", "VK_OMEGA", "VK_COSET_SHIFT"))
    assert int(k["R_MOD"]) == c.r and n == 8
    assert c.fr_root_of_unity(n) == omega and pow(n, -1, c.r) == ninv and c.fr_gen == shift
    for name in ("nocommit", "commit"):
        raw = open(os.path.join(GOLD, "vk_blank_plonk_%s_%s.bin" % (c.name.replace("-", ""), name)), "rb").read()
        if _fr_be(raw[:8]) == 0:                                      # keyVersionMarker, then the version (marshal.go:233-245, setup.go:32)
            assert _fr_be(raw[8:16]) == 1
            raw = raw[16:]
        size = _fr_be(raw[:8])                                        # (else: the legacy layout starts with Size, marshal.go:262-285)
        assert size == (n if name == "nocommit" else 2 * n)           # (the circuit with a commitment has a 16-point domain)
        if name == "nocommit":
            assert _fr_be(raw[8:40]) == ninv and _fr_be(raw[40:72]) == omega
        assert _fr_be(raw[8:40]) == pow(size, -1, c.r) and _fr_be(raw[40:72]) == c.fr_root_of_unity(size) and _fr_be(raw[80:112]) == shift
        nb = c.fp_bytes
        off = 112
        G1 = pyref.g1_group(c)
        for _ in range(8):                                            # S1 S2 S3 Ql Qr Qm Qo Qk: compressed G1 points
            P = pyref.g1_decompress(c, raw[off:off + nb])
            assert G1.on_curve(P) and pyref.g1_compress(c, P) == raw[off:off + nb]
            off += nb
        nq = int.from_bytes(raw[off:off + 4], "big")                  # Qcp
        off += 4 + nq * nb
        assert nq == (1 if name == "commit" else 0)
        assert pyref.g1_decompress(c, raw[off:off + nb]) == c.g1      # Kzg.G1 = [1]G1
        off += nb
        assert pyref.g2_decompress(c, raw[off:off + 2 * nb]) == c.g2  # Kzg.G2[0] = [1]G2
        tau2 = pyref.g2_decompress(c, raw[off + 2 * nb:off + 4 * nb])  # Kzg.G2[1] = [tau]G2
        assert pyref.g2_group(c).mul(tau2, c.r) is None
    # the library's domain of size 8: FFT(e_1)[k] = omega^k (DIT: bit-reversed in, natural out; e_1 sits at bitrev(1) = 4)
    d = fft.Domain(emu_ctx, c.name, 8)
    try:
        e = [0] * 8
        e[4] = 1
        got = [pyref.from_mont_limbs(row, c.r) for row in d.FFT(fr_to_arr(c, e), fft.DIT)]
        assert got == [pow(omega, i, c.r) for i in range(8)]
        e = [0] * 8
        e[1] = 1
        got = [pyref.from_mont_limbs(row, c.r) for row in d.FFT(fr_to_arr(c, e), fft.DIF, True)]      # coset: (shift*omega^k), bit-reversed out
        assert [got[pyref.bitrev(i, 3)] for i in range(8)] == [shift * pow(omega, i, c.r) % c.r for i in range(8)]
        got = [pyref.from_mont_limbs(row, c.r) for row in d.FFTInverse(fr_to_arr(c, [1] * 8), fft.DIF)]
        assert got[0] == 1 and not any(got[1:])                       # 1/n inside the inverse: iFFT(1,...,1) = e_0
    finally:
        d.close()


@pytest.mark.parametrize("c", [BN254, BLS12_381], ids=lambda c: c.name)
def test_groth16_commit_vk_fixture_decodes(c):
    """backend/solidity/testdata/blank_groth16_*_commit.vk per backend/groth16/bn254/marshal.go:151-230: the key of a circuit with
    one commitment -- K, PublicAndCommitmentCommitted, one pedersen.VerifyingKey (G, GSigmaNeg in G2).  Every byte is consumed and
    every G2 point is r-torsion."""
    raw = open(os.path.join(GOLD, "vk_blank_groth16_%s_commit.bin" % c.name.replace("-", "")), "rb").read()
    vk, pacc, beta1, delta1, used = pyref.vk_read(c, raw)
    assert used == len(raw) and len(pacc) == 1 and len(vk.commitment_g2_sigma_neg) == 1
    G2 = pyref.g2_group(c)
    for Q in (vk.beta2, vk.gamma2, vk.delta2, vk.commitment_g2, vk.commitment_g2_sigma_neg[0]):
        assert G2.on_curve(Q) and G2.mul(Q, c.r) is None
    assert len(vk.K) >= 2 and all(pyref.g1_group(c).on_curve(P) for P in vk.K + [vk.alpha1, beta1, delta1])
