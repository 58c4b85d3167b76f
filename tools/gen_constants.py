#!/usr/bin/env python3
"""Generate the field / curve constant headers from the moduli.

  gnark_amd/csrc/constants.h   product header, 32-bit limbs (device arithmetic uses v_mad_u64_u32)
  oracle/oracle_constants.h    oracle header, 64-bit limbs (plain C, unsigned __int128)

Moduli sources in the reference: backend/groth16/bn254/solidity.go:64-65,
std/math/emulated/emparams/emparams.go:142-157,225-241.  Everything else (R, R^2, -p^-1,
roots of unity) is derived here from the moduli.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyref import BN254, BLS12_381  # noqa: E402


def limbs(x, n, w):
    return [(x >> (w * i)) & ((1 << w) - 1) for i in range(n)]


def arr(x, n, w):
    sfx = "u" if w == 32 else "ull"
    return "{" + ", ".join(f"0x{v:0{w // 4}x}{sfx}" for v in limbs(x, n, w)) + "}"


FP2_LAZY_K = 16   # operands of the lazy Fp2 product are below 16p (asserted by tools/lazy_bounds.py)


def fp2_lazy_offset(mod, n64, K=FP2_LAZY_K):
    """Column constants Z[0..2NL-2] for field29.hip.h's Fp2 product: a multiple of p in a redundant base-2^L representation whose
    every column dominates the corresponding column of a1*b1 (operands < K*p, normalized limbs), so that
    a0*b0 - a1*b1 + Z can be formed column by column in unsigned 64-bit arithmetic before ONE Montgomery reduction."""
    n32 = 2 * n64
    L = 29 if n32 == 8 else 28
    NL = (32 * n32 + L - 1) // L + (1 if (32 * n32) % L == 0 else 0)
    top = (K * mod >> (L * (NL - 1))) + 1
    A = [2 ** L - 1] * (NL - 1) + [top]
    ncol = 2 * NL - 1
    colb = [sum(A[i] * A[j - i] for i in range(NL) if 0 <= j - i < NL) for j in range(ncol)]
    t = [0] * ncol            # t[j]: units of 2^L column j borrows from column j+1
    prev = 0
    for j in range(ncol - 1):
        t[j] = (colb[j] + prev + (1 << L) - 1) >> L
        prev = t[j]
    need_top = colb[ncol - 1] + prev
    zint = -(-(need_top << (L * (ncol - 1))) // mod) * mod
    z = [(zint >> (L * j)) & ((1 << L) - 1) for j in range(ncol - 1)] + [zint >> (L * (ncol - 1))]
    Z = [z[j] + (t[j] << L) - (t[j - 1] if j else 0) for j in range(ncol)]
    assert sum(Z[j] << (L * j) for j in range(ncol)) == zint and zint % mod == 0
    assert all(Z[j] >= colb[j] for j in range(ncol)) and max(Z) + max(colb) + NL * (1 << (2 * L)) < 1 << 63
    R = 1 << (L * NL)
    return Z, zint, -(-zint // R)     # the offset adds less than this to a reduced value


def field_block(name, mod, n64, w, extra=None):
    n = n64 * (64 // w)
    R = 1 << (64 * n64)
    ty = "uint32_t" if w == 32 else "uint64_t"
    inv = (-pow(mod, -1, 1 << w)) % (1 << w)
    out = [f"struct {name} {{"]
    out.append(f"    static constexpr int N = {n};          // {w}-bit limbs")
    out.append(f"    static constexpr int N64 = {n64};")
    out.append(f"    static constexpr int BITS = {mod.bit_length()};")
    out.append(f"    static constexpr {ty} INV = 0x{inv:x}{'u' if w == 32 else 'ull'};   // -mod^-1 mod 2^{w}")
    out.append(f"    static constexpr {ty} MOD[{n}] = {arr(mod, n, w)};")
    out.append(f"    static constexpr {ty} ONE[{n}] = {arr(R % mod, n, w)};   // R mod p")
    out.append(f"    static constexpr {ty} R2[{n}] = {arr(R * R % mod, n, w)};    // R^2 mod p")
    out.append(f"    static constexpr {ty} PM2[{n}] = {arr(mod - 2, n, w)};   // p-2 (Fermat inverse exponent)")
    out.append(f"    static constexpr uint32_t MU12 = {(1 << (mod.bit_length() + 8)) // mod}u;   // floor(2^(BITS+8) / p): Barrett quotient estimate (field29.hip.h)")
    if w == 32 and name.endswith("_Fp"):
        Z, zint, add = fp2_lazy_offset(mod, n64)
        out.append(f"    // lazy Fp2 product (field29.hip.h): redundant-digit columns of a multiple of p dominating a1*b1 for operands < {FP2_LAZY_K}p;")
        out.append(f"    // it adds < {add / mod:.3f} p to the reduced real part")
        out.append(f"    static constexpr int FP2Z_K = {FP2_LAZY_K};")
        out.append(f"    static constexpr uint64_t FP2Z[{len(Z)}] = {{" + ", ".join(f"0x{v:x}ull" for v in Z) + "};")
    for k, v in (extra or {}).items():
        if isinstance(v, int) and k.isupper() and k.startswith("I_"):
            out.append(f"    static constexpr int {k[2:]} = {v};")
        else:
            out.append(f"    static constexpr {ty} {k}[{n}] = {arr(v * R % mod, n, w)};   // Montgomery form")
    out.append("};")
    return "\n".join(out)


def c_field_block(name, mod, n64, extra=None):
    """plain-C flavour for the oracle (no structs with static members)."""
    R = 1 << (64 * n64)
    inv = (-pow(mod, -1, 1 << 64)) % (1 << 64)
    out = [f"#define {name}_N {n64}", f"#define {name}_BITS {mod.bit_length()}",
           f"static const uint64_t {name}_INV = 0x{inv:x}ull;",
           f"static const uint64_t {name}_MOD[{n64}] = {arr(mod, n64, 64)};",
           f"static const uint64_t {name}_ONE[{n64}] = {arr(R % mod, n64, 64)};",
           f"static const uint64_t {name}_R2[{n64}] = {arr(R * R % mod, n64, 64)};"]
    for k, v in (extra or {}).items():
        if k.startswith("I_"):
            out.append(f"#define {name}_{k[2:]} {v}")
        else:
            out.append(f"static const uint64_t {name}_{k}[{n64}] = {arr(v * R % mod, n64, 64)};")
    return "\n".join(out)


def main():
    hdr = ["// GENERATED by tools/gen_constants.py -- do not edit.", "#pragma once", "#include <stdint.h>", ""]
    prod = list(hdr) + ["namespace ga {", ""]
    orc = list(hdr)
    for c, tag in ((BN254, "BN254"), (BLS12_381, "BLS12_381")):
        wmax = c.fr_root_of_unity(1 << c.fr_adicity)
        fr_extra = {"ROOT": wmax, "ROOT_INV": pow(wmax, -1, c.r), "GEN": c.fr_gen,
                    "GEN_INV": pow(c.fr_gen, -1, c.r), "I_ADICITY": c.fr_adicity}
        fp_extra = {"G1X": c.g1[0], "G1Y": c.g1[1], "G2X0": c.g2[0][0], "G2X1": c.g2[0][1],
                    "G2Y0": c.g2[1][0], "G2Y1": c.g2[1][1], "B1": c.b, "B2_0": c.b2[0], "B2_1": c.b2[1]}
        prod.append(field_block(f"{tag}_Fp", c.p, c.fp_limbs, 32, fp_extra))
        prod.append("")
        prod.append(field_block(f"{tag}_Fr", c.r, c.fr_limbs, 32, fr_extra))
        prod.append("")
        orc.append(c_field_block(f"{tag}_FP", c.p, c.fp_limbs, fp_extra))
        orc.append("")
        orc.append(c_field_block(f"{tag}_FR", c.r, c.fr_limbs, fr_extra))
        orc.append("")
    prod.append("}  // namespace ga")
    with open(os.path.join(ROOT, "gnark_amd", "csrc", "constants.h"), "w") as f:
        f.write("\n".join(prod) + "\n")
    with open(os.path.join(ROOT, "oracle", "oracle_constants.h"), "w") as f:
        f.write("\n".join(orc) + "\n")
    print("wrote constants.h, oracle_constants.h")


if __name__ == "__main__":
    main()
