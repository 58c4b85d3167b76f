// Unpacked field representation for hot loops: NL limbs of L bits (9 x 29 for the 254/255-bit fields, 14 x 28 for the
// 381-bit one), one limb per 32-bit register, values kept in the Montgomery domain of R' = 2^(NL*L):
//
//     hat(x) = x * 2^(NL*L) mod p        (memory format is x * 2^(32N); hat = memory value * 2^S, S = NL*L - 32N)
//
// Why: on gfx950 every carry costs an issue slot as expensive as the multiply (profiles/r01_e_microbench.json).  In this
// representation a Montgomery product is 2*NL^2 v_mad_u64_u32 + one shift/mask per column -- no operand unpacking, no
// repacking, no final conditional subtraction -- and additions/subtractions are limb-wise with one carry sweep and NO
// modular correction: values are only kept below 2^(NL*L - 2), far above p (7 spare bits for BN254, 11 for BLS12-381),
// by adding fixed multiples of p on subtraction.  Exact reduction happens once, when a result leaves the hot loop.
#pragma once
#include "ec.hip.h"

namespace ga {

template <class P>
struct F29 {
    static constexpr int NL = Radix<P>::NL;
    uint32_t l[NL];
};

// k*p as limbs prepared for borrow-free subtraction:  a - b + k*p  computed limb-wise never goes negative as long as
// b's limbs are normalized (< 2^L) and b < k*p: every limb but the top lends 2^L to its lower neighbour.
template <class P, int K>
GA_HD uint32_t kp_limb(int i) {
    typedef Radix<P> R;
    // k*p in plain limbs (host/compile-time folded): multiply the modulus limbs by K with carry
    uint64_t carry = 0;
    uint32_t v = 0;
    for (int j = 0; j <= i; j++) {
        uint64_t t = (uint64_t)mod_limb<P>(j) * (uint32_t)K + carry;
        v = (uint32_t)(t & R::MASK);
        carry = t >> R::L;
        if (j == R::NL - 1) v = (uint32_t)t;   // top limb keeps the overflow
    }
    // lend: limb i gains 2^L (if not the top) and loses 1 (if not the bottom)
    uint32_t r = v;
    if (i < R::NL - 1) r += (1u << R::L);
    if (i > 0) r -= 1;
    return r;
}

// limb i of K*p in plain normalized limbs (the top limb keeps the overflow)
template <class P, int K>
GA_HD uint32_t kp_plain_limb(int i) {
    typedef Radix<P> R;
    uint64_t carry = 0;
    uint32_t v = 0;
    for (int j = 0; j <= i; j++) {
        uint64_t t = (uint64_t)mod_limb<P>(j) * (uint32_t)K + carry;
        v = (uint32_t)(t & R::MASK);
        carry = t >> R::L;
        if (j == R::NL - 1) v = (uint32_t)t;
    }
    return v;
}

template <class P>
GA_HD F29<P> f29_zero() {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < F29<P>::NL; i++) r.l[i] = 0;
    return r;
}

template <class P>
GA_HD bool f29_is_zero_limbs(const F29<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < F29<P>::NL; i++) o |= a.l[i];
    return o == 0;
}

// carry sweep: limbs back below 2^L (the top limb keeps the excess)
template <class P>
GA_HD void f29_normalize(F29<P>& a) {
    typedef Radix<P> R;
#pragma unroll
    for (int i = 0; i < R::NL - 1; i++) {
        a.l[i + 1] += a.l[i] >> R::L;
        a.l[i] &= R::MASK;
    }
}

template <class P>
GA_HD F29<P> f29_add(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < F29<P>::NL; i++) r.l[i] = a.l[i] + b.l[i];
    f29_normalize(r);
    return r;
}

// a - b + K*p   (requires b < K*p, b normalized)
template <int K, class P>
GA_HD F29<P> f29_sub(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < F29<P>::NL; i++) r.l[i] = a.l[i] + kp_limb<P, K>(i) - b.l[i];
    f29_normalize(r);
    return r;
}

// limb-wise sum without the carry sweep (limbs grow by one bit; the consumer must tolerate it)
template <class P>
GA_HD F29<P> f29_add_raw(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < F29<P>::NL; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
// a - b + K*p limb-wise without the carry sweep (b normalized, b < K*p)
template <int K, class P>
GA_HD F29<P> f29_sub_raw(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < F29<P>::NL; i++) r.l[i] = a.l[i] + kp_limb<P, K>(i) - b.l[i];
    return r;
}
// a - b + K*p where b is an UN-normalized sum with limbs below W*2^L (b < K*p as a value): every limb but the top lends
// W*2^L to its lower neighbour, so no limb goes negative; one carry sweep at the end.  Lets sums like PPP + 2Q skip their own
// sweeps.
template <int K, int W, class P>
GA_HD F29<P> f29_sub_wide(const F29<P>& a, const F29<P>& b) {
    typedef Radix<P> R;
    F29<P> r;
#pragma unroll
    for (int i = 0; i < R::NL; i++) {
        uint32_t k = kp_limb<P, K>(i);   // = limb + 2^L (not top) - 1 (not bottom): widen the loan from 1 to W units
        if (i < R::NL - 1) k += (uint32_t)(W - 1) << R::L;
        if (i > 0) k -= (uint32_t)(W - 1);
        r.l[i] = a.l[i] + k - b.l[i];
    }
    f29_normalize(r);
    return r;
}

// a*b / 2^(NL*L) (+ a multiple of p): normalized limbs in, normalized limbs out; result < a*b/2^(NL*L) + p.
// Column accumulators: the compiler keeps independent per-column MAD chains and merges the carries.  (Product scanning with one
// running accumulator was measured in round 1: no faster for the 9-limb fields, 13x slower for the 14-limb field -- dropped.)
template <class P>
GA_HD_BIG F29<P> f29_mul(const F29<P>& a, const F29<P>& b) {
    typedef Radix<P> R;
    constexpr int NL = R::NL, L = R::L;
    const uint32_t inv = P::INV & R::MASK;
    F29<P> r;
    uint64_t col[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)a.l[i] * b.l[j];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t m = ((uint32_t)col[i] * inv) & R::MASK;
#pragma unroll
        for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)m * mod_limb<P>(j);
        col[i + 1] += col[i] >> L;
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
        if (k + 1 < NL) {
            r.l[k] = (uint32_t)col[NL + k] & R::MASK;
            col[NL + k + 1] += col[NL + k] >> L;
        } else {
            r.l[k] = (uint32_t)col[NL + k];
        }
    }
    return r;
}

// Montgomery reduction of 2*NL product columns (column sums below 2^63): the second half of f29_mul on its own
template <class P>
GA_HD F29<P> f29_reduce_cols(uint64_t (&col)[2 * Radix<P>::NL]) {
    typedef Radix<P> R;
    constexpr int NL = R::NL, L = R::L;
    const uint32_t inv = P::INV & R::MASK;
    F29<P> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t m = ((uint32_t)col[i] * inv) & R::MASK;
#pragma unroll
        for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)m * mod_limb<P>(j);
        col[i + 1] += col[i] >> L;
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
        if (k + 1 < NL) {
            r.l[k] = (uint32_t)col[NL + k] & R::MASK;
            col[NL + k + 1] += col[NL + k] >> L;
        } else {
            r.l[k] = (uint32_t)col[NL + k];
        }
    }
    return r;
}

// a*b - c*d (+ a multiple of p) with ONE reduction: the columns of a*b and of (K*p - c)*d are accumulated together.
// Requires c < K*p; result < (a*b + K*p*d) / 2^(NL*L) + p.  Saves a Montgomery reduction and a limb-wise subtraction per
// difference of products (Y3 of the mixed addition).
template <int K, class P>
GA_HD_BIG F29<P> f29_mul_sub(const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d) {
    typedef Radix<P> R;
    constexpr int NL = R::NL;
    F29<P> nc;
#pragma unroll
    for (int i = 0; i < NL; i++) nc.l[i] = kp_limb<P, K>(i) - c.l[i];
    f29_normalize(nc);
    uint64_t col[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) {
            col[i + j] += (uint64_t)a.l[i] * b.l[j];
            col[i + j] += (uint64_t)nc.l[i] * d.l[j];
        }
    return f29_reduce_cols<P>(col);
}

// memory image (x * 2^(32N), canonical) -> hat(x) as an ordinary canonical 32N-bit integer (table storage format)
template <class P>
GA_HD Fe<P> f29_hat_packed(const Fe<P>& x) {
    Fe<P> t = x;
#pragma unroll
    for (int k = 0; k < Radix<P>::S; k++) t = dbl(t);   // * 2^S mod p
    return t;
}
// packed integer -> limbs (no arithmetic: bit-field extraction only)
template <class P>
GA_HD F29<P> f29_unpack(const Fe<P>& t) {
    typedef Radix<P> R;
    F29<P> r;
#pragma unroll
    for (int i = 0; i < R::NL; i++) r.l[i] = take_bits<P::N, R::L>(t.l, i * R::L);
    return r;
}
// memory image -> hat(x), canonical limbs
template <class P>
GA_HD F29<P> f29_from_mem(const Fe<P>& x) { return f29_unpack(f29_hat_packed(x)); }

// hat(x) with any value < 2^(32N) -> canonical memory image x * 2^(32N) mod p
template <class P>
GA_HD_BIG Fe<P> f29_to_mem(const F29<P>& a) {
    typedef Radix<P> R;
    // mont'(a, 2^(32N) mod p) = a * 2^(32N) / 2^(NL*L) = a * 2^-S : hat -> memory scaling; P::ONE is 2^(32N) mod p
    F29<P> one;
#pragma unroll
    for (int i = 0; i < R::NL; i++) one.l[i] = take_bits<P::N, R::L>(P::ONE, i * R::L);
    F29<P> v = f29_mul(a, one);   // < a*p/2^(NL*L) + p < 2p
    uint32_t t[P::N];
#pragma unroll
    for (int w = 0; w < P::N; w++) {
        const int k0 = (32 * w) / R::L, off = (32 * w) % R::L;
        uint64_t u = (uint64_t)v.l[k0] >> off;
        if (k0 + 1 < R::NL) u |= (uint64_t)v.l[k0 + 1] << (R::L - off);
        if (2 * R::L - off < 32 && k0 + 2 < R::NL) u |= (uint64_t)v.l[k0 + 2] << (2 * R::L - off);
        t[w] = (uint32_t)u;
    }
    reduce_once<P>(t);
    reduce_once<P>(t);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = t[i];
    return r;
}

// one step towards [0, p): v - floor(v / 2^BITS) * p   (>= 0; result < 2^BITS + q*(2^BITS - p), i.e. < 4p for v < 2^(BITS+3))
template <class P>
GA_HD F29<P> f29_partial_reduce(const F29<P>& a) {
    typedef Radix<P> R;
    constexpr int TOP = P::BITS / R::L, OFF = P::BITS % R::L;
    static_assert(TOP == R::NL - 1, "the modulus' top bit lives in the top limb");
    const uint32_t q = a.l[TOP] >> OFF;
    F29<P> r;
    uint64_t c = 0;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < R::NL; i++) {
        uint64_t u = (uint64_t)q * mod_limb<P>(i) + c;   // limb i of q*p
        c = u >> R::L;
        int32_t d = (int32_t)a.l[i] - (int32_t)((uint32_t)u & R::MASK) + borrow;
        if (i < R::NL - 1) {
            r.l[i] = (uint32_t)d & R::MASK;
            borrow = d >> R::L;   // arithmetic shift: 0 or -1
        } else {
            r.l[i] = (uint32_t)d;
        }
    }
    return r;
}

// Barrett step: v - q*p with q = ((v >> (BITS-4)) * MU12) >> 12, q <= floor(v/p) <= q+2  =>  result in [0, 3p)
// (valid for any normalized v < 2^(NL*L)); used where sums keep doubling (NTT butterflies)
template <class P>
GA_HD F29<P> f29_reduce_3p(const F29<P>& a) {
    typedef Radix<P> R;
    constexpr int SH = P::BITS - 4, TOP = R::NL - 1;
    static_assert(SH >= R::L * TOP, "quotient estimate comes from the top limb");
    const uint32_t q = ((a.l[TOP] >> (SH - R::L * TOP)) * P::MU12) >> 12;
    F29<P> r;
    uint64_t c = 0;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < R::NL; i++) {
        uint64_t u = (uint64_t)q * mod_limb<P>(i) + c;
        c = u >> R::L;
        int32_t d = (int32_t)a.l[i] - (int32_t)((uint32_t)u & R::MASK) + borrow;
        if (i < R::NL - 1) {
            r.l[i] = (uint32_t)d & R::MASK;
            borrow = d >> R::L;
        } else {
            r.l[i] = (uint32_t)d;
        }
    }
    return r;
}

// exact test v == 0 (mod p) of a lazy value (normalized limbs, any v < 2^(NL*L)): one Barrett step to [0, 3p), then the three
// candidates 0, p, 2p.  Used only where the addition law needs it (the complete variant of the bucket loop, msm.hip.h).
template <class P>
GA_HD bool f29_is_zero_mod_p(const F29<P>& a) {
    typedef Radix<P> R;
    const F29<P> r = f29_reduce_3p(a);
    uint32_t z0 = 0, z1 = 0, z2 = 0;
#pragma unroll
    for (int i = 0; i < R::NL; i++) {
        z0 |= r.l[i];
        z1 |= r.l[i] ^ kp_plain_limb<P, 1>(i);
        z2 |= r.l[i] ^ kp_plain_limb<P, 2>(i);
    }
    return (z0 == 0) | (z1 == 0) | (z2 == 0);
}

// limbs (value < 3p) -> canonical packed words
template <class P>
GA_HD Fe<P> f29_pack_canonical(const F29<P>& v) {
    typedef Radix<P> R;
    uint32_t t[P::N];
#pragma unroll
    for (int w = 0; w < P::N; w++) {
        const int k0 = (32 * w) / R::L, off = (32 * w) % R::L;
        uint64_t u = (uint64_t)v.l[k0] >> off;
        if (k0 + 1 < R::NL) u |= (uint64_t)v.l[k0 + 1] << (R::L - off);
        if (2 * R::L - off < 32 && k0 + 2 < R::NL) u |= (uint64_t)v.l[k0 + 2] << (2 * R::L - off);
        t[w] = (uint32_t)u;
    }
    reduce_once<P>(t);
    reduce_once<P>(t);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = t[i];
    return r;
}

// ---- Fp2 = Fp[u]/(u^2+1) in the lazy representation ------------------------------------------------------------------
template <class P>
struct F29x2 {
    F29<P> c0, c1;
};

template <class P> GA_HD F29x2<P> f29_add(const F29x2<P>& a, const F29x2<P>& b) { return {f29_add(a.c0, b.c0), f29_add(a.c1, b.c1)}; }
template <int K, class P> GA_HD F29x2<P> f29_sub(const F29x2<P>& a, const F29x2<P>& b) {
    return {f29_sub<K>(a.c0, b.c0), f29_sub<K>(a.c1, b.c1)};
}
template <class P> GA_HD F29x2<P> f29_add_raw(const F29x2<P>& a, const F29x2<P>& b) { return {f29_add_raw(a.c0, b.c0), f29_add_raw(a.c1, b.c1)}; }
template <int K, int W, class P> GA_HD F29x2<P> f29_sub_wide(const F29x2<P>& a, const F29x2<P>& b) {
    return {f29_sub_wide<K, W>(a.c0, b.c0), f29_sub_wide<K, W>(a.c1, b.c1)};
}
template <class P> GA_HD F29x2<P> f29_partial_reduce(const F29x2<P>& a) { return {f29_partial_reduce(a.c0), f29_partial_reduce(a.c1)}; }
template <class P> GA_HD bool f29_is_zero_limbs(const F29x2<P>& a) { return f29_is_zero_limbs(a.c0) & f29_is_zero_limbs(a.c1); }
template <class P> GA_HD bool f29_is_zero_mod_p(const F29x2<P>& a) { return f29_is_zero_mod_p(a.c0) & f29_is_zero_mod_p(a.c1); }

// Fp2 product, schoolbook on unreduced columns: real part a0*b0 + (K*p - a1)*b1, imaginary part a0*b1 + a1*b0 -- every term is
// non-negative, so there is no 64-bit subtraction (a v_sub_co/v_subb pair costs more than a multiply on gfx950) and only two
// column sets are live; 4*NL^2 limb products + 2 reductions.  K = P::FP2Z_K (operands below K*p; bounds in tools/lazy_bounds.py).
// Measured against it in round 1 and dropped: Karatsuba on the unreduced columns (fewer multiplies, 51 64-bit subtractions:
// slower) and Karatsuba on reduced operands (three reductions and five limb-wise sweeps).
template <class P>
GA_HD_BIG F29x2<P> f29_mul(const F29x2<P>& a, const F29x2<P>& b) {
    typedef Radix<P> R;
    constexpr int NL = R::NL;
        constexpr int K = P::FP2Z_K;
        F29<P> n1;
#pragma unroll
        for (int i = 0; i < NL; i++) n1.l[i] = kp_limb<P, K>(i) - a.c1.l[i];
        f29_normalize(n1);
        F29x2<P> r;
        {   // one column set live at a time
            uint64_t c0[2 * NL];
#pragma unroll
            for (int k = 0; k < 2 * NL; k++) c0[k] = 0;
#pragma unroll
            for (int i = 0; i < NL; i++)
#pragma unroll
                for (int j = 0; j < NL; j++) {
                    c0[i + j] += (uint64_t)a.c0.l[i] * b.c0.l[j];
                    c0[i + j] += (uint64_t)n1.l[i] * b.c1.l[j];
                }
            r.c0 = f29_reduce_cols<P>(c0);
        }
        {
            uint64_t c1[2 * NL];
#pragma unroll
            for (int k = 0; k < 2 * NL; k++) c1[k] = 0;
#pragma unroll
            for (int i = 0; i < NL; i++)
#pragma unroll
                for (int j = 0; j < NL; j++) {
                    c1[i + j] += (uint64_t)a.c0.l[i] * b.c1.l[j];
                    c1[i + j] += (uint64_t)a.c1.l[i] * b.c0.l[j];
                }
            r.c1 = f29_reduce_cols<P>(c1);
        }
        return r;
}
// complex squaring: (a0+a1)(a0-a1) + 2 a0 a1 u
template <class P>
GA_HD_BIG F29x2<P> f29_sqr(const F29x2<P>& a) {
    F29<P> t = f29_mul(a.c0, a.c1);
    F29<P> r0 = f29_mul(f29_add(a.c0, a.c1), f29_sub<8>(a.c0, a.c1));
    return {r0, f29_add(t, t)};
}
// a*b - c*d in Fp2 with two reductions (instead of four): real = a0*b0 + (Kp-a1)*b1 + (Kp-c0)*d0 + c1*d1,
// imaginary = a0*b1 + a1*b0 + (Kp-c0)*d1 + (Kp-c1)*d0.  Four products per column: 4*NL*2^(2L) + NL*2^(2L) < 2^64.
template <int K, class P>
GA_HD_BIG F29x2<P> f29_mul_sub(const F29x2<P>& a, const F29x2<P>& b, const F29x2<P>& c, const F29x2<P>& d) {
    typedef Radix<P> R;
    constexpr int NL = R::NL;
    static_assert(5.0 * NL * (double)(1ull << (2 * R::L)) < 18446744073709551616.0, "column sums must fit 64 bits");
    F29<P> na1, nc0, nc1;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        na1.l[i] = kp_limb<P, K>(i) - a.c1.l[i];
        nc0.l[i] = kp_limb<P, K>(i) - c.c0.l[i];
        nc1.l[i] = kp_limb<P, K>(i) - c.c1.l[i];
    }
    f29_normalize(na1);
    f29_normalize(nc0);
    f29_normalize(nc1);
    F29x2<P> r;
    {
        uint64_t col[2 * NL];
#pragma unroll
        for (int k = 0; k < 2 * NL; k++) col[k] = 0;
#pragma unroll
        for (int i = 0; i < NL; i++)
#pragma unroll
            for (int j = 0; j < NL; j++) {
                col[i + j] += (uint64_t)a.c0.l[i] * b.c0.l[j];
                col[i + j] += (uint64_t)na1.l[i] * b.c1.l[j];
                col[i + j] += (uint64_t)nc0.l[i] * d.c0.l[j];
                col[i + j] += (uint64_t)c.c1.l[i] * d.c1.l[j];
            }
        r.c0 = f29_reduce_cols<P>(col);
    }
    {
        uint64_t col[2 * NL];
#pragma unroll
        for (int k = 0; k < 2 * NL; k++) col[k] = 0;
#pragma unroll
        for (int i = 0; i < NL; i++)
#pragma unroll
            for (int j = 0; j < NL; j++) {
                col[i + j] += (uint64_t)a.c0.l[i] * b.c1.l[j];
                col[i + j] += (uint64_t)a.c1.l[i] * b.c0.l[j];
                col[i + j] += (uint64_t)nc0.l[i] * d.c1.l[j];
                col[i + j] += (uint64_t)nc1.l[i] * d.c0.l[j];
            }
        r.c1 = f29_reduce_cols<P>(col);
    }
    return r;
}

template <class P>
GA_HD_BIG F29<P> f29_sqr(const F29<P>& a) {
    // a_i*a_j (i<j) computed once against the doubled limb: NL(NL+1)/2 products instead of NL^2 (45 vs 81 for 9 limbs).
    // Doubled limbs stay below 2^(L+1), a column holds at most NL/2 doubled products + one square + the NL reduction
    // products: < 2^(2L+1) * NL < 2^63 for both limb layouts.
    typedef Radix<P> R;
    constexpr int NL = R::NL, L = R::L;
    const uint32_t inv = P::INV & R::MASK;
    F29<P> r;
    uint32_t d[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) d[i] = a.l[i] << 1;
    uint64_t col[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        col[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
        for (int j = i + 1; j < NL; j++) col[i + j] += (uint64_t)a.l[i] * d[j];
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t m = ((uint32_t)col[i] * inv) & R::MASK;
#pragma unroll
        for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)m * mod_limb<P>(j);
        col[i + 1] += col[i] >> L;
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
        if (k + 1 < NL) {
            r.l[k] = (uint32_t)col[NL + k] & R::MASK;
            col[NL + k + 1] += col[NL + k] >> L;
        } else {
            r.l[k] = (uint32_t)col[NL + k];
        }
    }
    return r;
}

// ---- uniform view used by the table kernels: Lazy<Fe<P>> / Lazy<Fe2<P>> ------------------------------------------------
// ---- inversion in the lazy representation (Fermat; products only, so every intermediate stays below 2p) ---------------------
// a = hat(x) with any value below 2^(NL*L - 2)  ->  hat(1/x)   (0 -> 0)
template <class P>
GA_HD_BIG F29<P> f29_inv(const F29<P>& a) {
    // hat(1) = 2^(NL*L) mod p = ONE * 2^S: S doublings of the canonical packed value
    Fe<P> h;
#pragma unroll
    for (int i = 0; i < P::N; i++) h.l[i] = P::ONE[i];
    F29<P> r = f29_unpack(f29_hat_packed(h));
    for (int i = P::N - 1; i >= 0; i--) {
        const uint32_t e = P::PM2[i];
        for (int b = 31; b >= 0; b--) {
            r = f29_sqr(r);
            if ((e >> b) & 1) r = f29_mul(r, a);
        }
    }
    return r;
}
template <class P>
GA_HD_BIG F29x2<P> f29_inv(const F29x2<P>& a) {
    // 1/(a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2)
    const F29<P> n = f29_inv(f29_add(f29_sqr(a.c0), f29_sqr(a.c1)));
    F29<P> zero = f29_zero<P>();
    return {f29_mul(a.c0, n), f29_sub<2>(zero, f29_mul(a.c1, n))};   // the product is below 2p
}
// lazy value -> canonical packed words of the SAME (hat) domain: the storage format of the window tables
template <class P>
GA_HD Fe<P> f29_pack_hat(const F29<P>& v) { return f29_pack_canonical(f29_reduce_3p(v)); }
template <class P>
GA_HD Fe2<P> f29_pack_hat(const F29x2<P>& v) { return {f29_pack_hat(v.c0), f29_pack_hat(v.c1)}; }

template <class F> struct Lazy;
template <class P> struct Lazy<Fe<P>> {
    typedef P Params;
    typedef F29<P> T;
    static constexpr int NW = Radix<P>::NL;          // 32-bit words per coordinate
    static constexpr bool FP2 = false;
    GA_HD static T from_mem(const Fe<P>& x) { return f29_from_mem(x); }
    GA_HD static Fe<P> hat_packed(const Fe<P>& x) { return f29_hat_packed(x); }
    GA_HD static T unpack(const Fe<P>& x) { return f29_unpack(x); }
    GA_HD static Fe<P> to_mem(const T& x) { return f29_to_mem(x); }
    GA_HD static uint32_t word(const T& x, int i) { return x.l[i]; }
    GA_HD static void set_word(T& x, int i, uint32_t v) { x.l[i] = v; }
};
template <class P> struct Lazy<Fe2<P>> {
    typedef P Params;
    typedef F29x2<P> T;
    static constexpr int NW = 2 * Radix<P>::NL;
    static constexpr bool FP2 = true;
    GA_HD static T from_mem(const Fe2<P>& x) { return {f29_from_mem(x.c0), f29_from_mem(x.c1)}; }
    GA_HD static Fe2<P> hat_packed(const Fe2<P>& x) { return {f29_hat_packed(x.c0), f29_hat_packed(x.c1)}; }
    GA_HD static T unpack(const Fe2<P>& x) { return {f29_unpack(x.c0), f29_unpack(x.c1)}; }
    GA_HD static Fe2<P> to_mem(const T& x) { return {f29_to_mem(x.c0), f29_to_mem(x.c1)}; }
    GA_HD static uint32_t word(const T& x, int i) { return i < Radix<P>::NL ? x.c0.l[i] : x.c1.l[i - Radix<P>::NL]; }
    GA_HD static void set_word(T& x, int i, uint32_t v) {
        if (i < Radix<P>::NL) x.c0.l[i] = v;
        else x.c1.l[i - Radix<P>::NL] = v;
    }
};

}  // namespace ga
