"""The bucket loops run in an unreduced representation (field29.hip.h); tools/lazy_bounds.py is the interval analysis that
justifies the constants.  It must hold for every (curve, group) the kernels are instantiated for, with headroom."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lazy_bounds  # noqa: E402


@pytest.mark.parametrize("curve", ["bn254", "bls12-381"])
@pytest.mark.parametrize("fp2", [False, True], ids=["G1", "G2"])
def test_bounds_hold_with_headroom(curve, fp2):
    out = lazy_bounds.check(curve, fp2)
    assert max(out["X"], out["Y"], out["P"], out["R"]) < out["limit"] - 2.5   # >= 2.5 bits below R'


def test_constants_match_the_kernels():
    src = open(os.path.join(ROOT, "gnark_amd", "csrc", "msm.hip.h")).read()
    g1 = src[src.index("__device__ __forceinline__ void madd29(const LdsAcc29<Fe<P>>"):src.index("__device__ __forceinline__ void madd29(const LdsAcc29<Fe2<P>>")]
    g2 = src[src.index("__device__ __forceinline__ void madd29(const LdsAcc29<Fe2<P>>"):src.index("// acc = 2*(qx, qy) for an affine q")]
    dbl = src[src.index("__device__ __forceinline__ void mdbl29("):src.index("// acc += q with the exceptional cases")]
    cpl = src[src.index("__device__ __forceinline__ bool madd29_complete("):src.index("// one task = the sorted pairs")]
    subs = lambda s: [int(x) for x in re.findall(r"f29_sub(?:_wide|_raw)?<(\d+)", s)]   # K of every subtraction flavour
    k = lazy_bounds.G1
    assert subs(g1) == [k["Kx"], k["Ky"], k["K3"], k["Kq"]]
    assert "f29_mul_sub<8>(" in g1            # Y3: negation constant checked in lazy_bounds.check (Kms)
    k = lazy_bounds.G2
    assert subs(g2) == [k["Kx"], k["Ky"], k["K3"], k["Kq"]]
    assert "f29_mul_sub<P::FP2Z_K>(" in g2
    assert g2.count("f29_partial_reduce(") == len(k["partial_reduce"])
    # the doubling and the zero tests of the complete loop use the constants lazy_bounds.check_mdbl / check() assume
    assert [int(x) for x in re.findall(r"f29_sub<(\d+)>", dbl)] == [4, 8] and "KMS = Lazy<F>::FP2 ? P::FP2Z_K : 8" in dbl
    assert "KS = Lazy<F>::FP2 ? 4 : 8" in cpl and lazy_bounds.G1["Kx"] == lazy_bounds.G1["Ky"] == 8 and lazy_bounds.G2["Kx"] == lazy_bounds.G2["Ky"] == 4
    f29 = open(os.path.join(ROOT, "gnark_amd", "csrc", "field29.hip.h")).read()
    prod = f29[f29.index("GA_HD_BIG F29x2<P> f29_mul("):f29.index("GA_HD_BIG F29x2<P> f29_sqr(")]
    assert subs(prod) == [] and "kp_limb<P, K>(i) - a.c1.l[i]" in prod and "constexpr int K = P::FP2Z_K;" in prod   # schoolbook on columns only
    sq = f29[f29.index("GA_HD_BIG F29x2<P> f29_sqr("):f29.index("GA_HD_BIG F29<P> f29_sqr(")]
    assert subs(sq) == [k["KQ"]]


@pytest.mark.parametrize("curve", ["bn254", "bls12-381"])
@pytest.mark.parametrize("fp2", [False, True], ids=["G1", "G2"])
def test_general_addition_bounds(curve, fp2):
    """msm.hip.h::add29 (lazy window reduction): fixed point of the bounds when both operands are earlier sums"""
    out = lazy_bounds.check_add(curve, fp2)
    assert max(out["X"], out["Y"], out["P"], out["R"]) < out["limit"] - 2.5


def test_general_addition_constants_match_the_kernel():
    src = open(os.path.join(ROOT, "gnark_amd", "csrc", "msm.hip.h")).read()
    body = src[src.index("__device__ __forceinline__ void add29(Lazy4<F>& a"):src.index("msm_reduce_groups29_kernel(")]
    subs = [int(x) for x in re.findall(r"f29_sub<(\d+)>", body)]
    for k in (lazy_bounds.ADD_G1, lazy_bounds.ADD_G2):
        assert subs == [k["KP"], k["KR"], k["K3"], k["Kq"]]
    assert "Lazy<F>::FP2 ? P::FP2Z_K : 8" in body and lazy_bounds.ADD_G1["Kms"] == 8 and lazy_bounds.ADD_G2["Kms"] == 16
    assert body.count("f29_partial_reduce(") == len(lazy_bounds.ADD_G2["partial_reduce"]) and not lazy_bounds.ADD_G1["partial_reduce"]


@pytest.mark.parametrize("curve", sorted(lazy_bounds.CURVES))
@pytest.mark.parametrize("fp2", [False, True], ids=["G1", "G2"])
def test_affine_doubling_into_the_accumulator(curve, fp2):
    """msm.hip.h::mdbl29 (the doubling case of the complete bucket loop): its own subtraction constants and Fp2 operand bounds,
    and the mixed additions that follow still satisfy theirs when the accumulator starts from a doubling's output"""
    out = lazy_bounds.check_mdbl(curve, fp2)
    limit = lazy_bounds.CURVES[curve][1] * lazy_bounds.CURVES[curve][2]
    assert all(v < limit - 2 for v in out.values())


@pytest.mark.parametrize("curve", sorted(lazy_bounds.CURVES))
@pytest.mark.parametrize("fp2", [False, True], ids=["G1", "G2"])
def test_repeated_doubling_bounds(curve, fp2):
    """msm.hip.h::dbl29 (the doubling chain of the window-table build): fixed point of the bounds, with headroom"""
    out = lazy_bounds.check_dbl(curve, fp2)
    assert all(v < out["limit"] - 2 for k, v in out.items() if k != "limit")


def test_doubling_constants_match_the_kernel():
    src = open(os.path.join(ROOT, "gnark_amd", "csrc", "msm.hip.h")).read()
    d = src[src.index("__device__ __forceinline__ void dbl29("):src.index("// A lane carries TableBatch<F>::K points")]
    assert [int(x) for x in re.findall(r"f29_sub<(\d+)>", d)] == [4, 8] and "KMS = Lazy<F>::FP2 ? P::FP2Z_K : 8" in d


def test_plonk_constraint_kernel_bounds():
    """plonk.hip.h::plonk_constraints29_kernel: every intermediate of the PLONK constraint expression on unreduced limbs stays below
    2^261 and every shifted factor below 2^256 for BN254's Fr with the maximum of 16 BSB22 gates; the subtraction constants are the
    kernel's; the 255-bit Fr of BLS12-381 does NOT satisfy the bounds (which is why that curve keeps the packed kernel)"""
    out = lazy_bounds.check_plonk_constraints()
    assert out["res"] < 40 and out["limit"] > 160                       # 27 r of the 169 r that fit
    with pytest.raises(AssertionError):
        lazy_bounds.check_plonk_constraints(r=lazy_bounds.BLS12_381_R)
    src = open(os.path.join(ROOT, "gnark_amd", "csrc", "plonk.hip.h")).read()
    body = src[src.index("plonk_constraints29_kernel(PlonkPtrs P"):src.index("// out[bitrev(i)] = in[i]")]
    assert [int(x) for x in re.findall(r"f29_sub<(\d+)>", body)] == [lazy_bounds.PLONK_SUB["ord"], lazy_bounds.PLONK_SUB["zm1"]]
    assert "Radix<FrP>::NL * Radix<FrP>::L - FrP::BITS >= 7" in src       # the dispatch that keeps the 255-bit field on the packed kernel
