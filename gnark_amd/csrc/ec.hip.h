// Quadratic extension Fp2 = Fp[u]/(u^2+1) and short-Weierstrass (a = 0) group law for G1 (over Fp) and
// G2 (over Fp2) of BN254 and BLS12-381.
//
// Memory images match gnark-crypto: G1Affine{X,Y}, G2Affine{X,Y E2{A0,A1}}, infinity = all-zero coordinates,
// G1Jac{X,Y,Z}.  Buckets use extended-Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2): a mixed addition costs 8M+2S and the formulas below are made complete by branching on
// P == 0 (doubling / inverse), which DummySetup-style keys with identical bases need (setup.go:526-540).
#pragma once
#include "field.hip.h"

namespace ga {

// ---- Fp2 ---------------------------------------------------------------------------------------
template <class P>
struct Fe2 {
    Fe<P> c0, c1;   // c0 + c1*u   (gnark E2{A0, A1})
};

template <class P> GA_HD Fe2<P> add(const Fe2<P>& a, const Fe2<P>& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
template <class P> GA_HD Fe2<P> sub(const Fe2<P>& a, const Fe2<P>& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
template <class P> GA_HD Fe2<P> dbl(const Fe2<P>& a) { return {dbl(a.c0), dbl(a.c1)}; }
template <class P> GA_HD Fe2<P> neg(const Fe2<P>& a) { return {neg(a.c0), neg(a.c1)}; }
template <class P> GA_HD bool is_zero(const Fe2<P>& a) { return is_zero(a.c0) & is_zero(a.c1); }
template <class P> GA_HD bool eq(const Fe2<P>& a, const Fe2<P>& b) { return eq(a.c0, b.c0) & eq(a.c1, b.c1); }

// Karatsuba product / complex squaring with every base-field product inlined (hot loops) ...
template <class P>
GA_HD_BIG Fe2<P> mul_body(const Fe2<P>& a, const Fe2<P>& b) {
    Fe<P> v0 = mul_hot(a.c0, b.c0);
    Fe<P> v1 = mul_hot(a.c1, b.c1);
    Fe<P> s = mul_hot(add(a.c0, a.c1), add(b.c0, b.c1));
    return {sub(v0, v1), sub(sub(s, v0), v1)};
}
template <class P>
GA_HD_BIG Fe2<P> sqr_body(const Fe2<P>& a) {
    Fe<P> t = mul_hot(a.c0, a.c1);
    Fe<P> r0 = mul_hot(add(a.c0, a.c1), sub(a.c0, a.c1));
    return {r0, dbl(t)};
}
template <class P>
GA_HD_BIG Fe<P> sqr_body(const Fe<P>& a) { return mul_hot(a, a); }

// ... and as shared out-of-line functions (cold kernels, host code)
template <class P>
GA_HD_CALL Fe2<P> mul(const Fe2<P>& a, const Fe2<P>& b) { return mul_body(a, b); }
template <class P>
GA_HD_CALL Fe2<P> sqr(const Fe2<P>& a) { return sqr_body(a); }

// INL = true: fully inlined product (accumulator stays in registers, no call ABI traffic); false: shared functions
template <class P>
GA_HD Fe<P> mul_inl(const Fe<P>& a, const Fe<P>& b) { return mul_hot(a, b); }
template <class P>
GA_HD Fe2<P> mul_inl(const Fe2<P>& a, const Fe2<P>& b) { return mul_body(a, b); }
template <bool INL, class F>
GA_HD F fmul(const F& a, const F& b) {
    if constexpr (INL) return mul_inl(a, b);
    else return mul(a, b);
}
template <bool INL, class F>
GA_HD F fsqr(const F& a) {
    if constexpr (INL) return sqr_body(a);
    else return sqr(a);
}

template <class P>
GA_HD_CALL Fe2<P> inv(const Fe2<P>& a) {
    Fe<P> d = inv(add(sqr(a.c0), sqr(a.c1)));
    return {mul(a.c0, d), neg(mul(a.c1, d))};
}

// zero / one for either field type
template <class F> struct FieldTraits;
template <class P> struct FieldTraits<Fe<P>> {
    GA_HD static Fe<P> zero() { return fe_zero<P>(); }
    GA_HD static Fe<P> one() { return fe_one<P>(); }
};
template <class P> struct FieldTraits<Fe2<P>> {
    GA_HD static Fe2<P> zero() { return {fe_zero<P>(), fe_zero<P>()}; }
    GA_HD static Fe2<P> one() { return {fe_one<P>(), fe_zero<P>()}; }
};

// ---- points ------------------------------------------------------------------------------------
template <class F> struct Affine { F x, y; };           // (0,0) = infinity
template <class F> struct Jac { F x, y, z; };            // gnark G1Jac / G2Jac image
template <class F> struct XYZZ { F x, y, zz, zzz; };     // zz == 0  <=> infinity

template <class F> GA_HD bool is_inf(const Affine<F>& p) { return is_zero(p.x) & is_zero(p.y); }
template <class F> GA_HD bool is_inf(const XYZZ<F>& p) { return is_zero(p.zz); }

template <class F>
GA_HD XYZZ<F> xyzz_inf() {
    F z = FieldTraits<F>::zero();
    F o = FieldTraits<F>::one();
    return {o, o, z, z};
}

template <class F>
GA_HD XYZZ<F> to_xyzz(const Affine<F>& p) {
    if (is_inf(p)) return xyzz_inf<F>();
    F o = FieldTraits<F>::one();
    return {p.x, p.y, o, o};
}

template <class F>
GA_HD Affine<F> neg(const Affine<F>& p) { return {p.x, neg(p.y)}; }
template <class F>
GA_HD XYZZ<F> neg(const XYZZ<F>& p) { return {p.x, neg(p.y), p.zz, p.zzz}; }

// 2*(affine) -> XYZZ   (mdbl-2008-s-1, a = 0)
template <bool INL, class F>
GA_HD XYZZ<F> dbl_affine_t(const Affine<F>& p) {
    if (is_inf(p) || is_zero(p.y)) return xyzz_inf<F>();
    F U = dbl(p.y);
    F V = fsqr<INL>(U);
    F W = fmul<INL>(U, V);
    F S = fmul<INL>(p.x, V);
    F xx = fsqr<INL>(p.x);
    F M = add(dbl(xx), xx);
    F X3 = sub(fsqr<INL>(M), dbl(S));
    F Y3 = sub(fmul<INL>(M, sub(S, X3)), fmul<INL>(W, p.y));
    return {X3, Y3, V, W};
}
template <class F>
GA_HD_CALL XYZZ<F> dbl_affine(const Affine<F>& p) { return dbl_affine_t<false>(p); }

// 2*P   (dbl-2008-s-1, a = 0)
template <class F>
GA_HD_CALL XYZZ<F> dbl(const XYZZ<F>& p) {
    if (is_inf(p) || is_zero(p.y)) return xyzz_inf<F>();
    F U = dbl(p.y);
    F V = sqr(U);
    F W = mul(U, V);
    F S = mul(p.x, V);
    F xx = sqr(p.x);
    F M = add(dbl(xx), xx);
    F X3 = sub(sqr(M), dbl(S));
    F Y3 = sub(mul(M, sub(S, X3)), mul(W, p.y));
    return {X3, Y3, mul(V, p.zz), mul(W, p.zzz)};
}

// acc + affine   (madd-2008-s), complete.  INL = true is the bucket-accumulation hot loop: everything inlined so
// that the accumulator never has its address taken (a call returning a point through memory forces the whole
// accumulator into scratch -- measured as 13.7 GB of WRITE_SIZE per 2^22 MSM before this split).
template <bool INL, class F>
GA_HD XYZZ<F> madd_t(const XYZZ<F>& a, const Affine<F>& q) {
    if (is_inf(q)) return a;
    if (is_inf(a)) return to_xyzz(q);
    F U2 = fmul<INL>(q.x, a.zz);
    F S2 = fmul<INL>(q.y, a.zzz);
    F Pp = sub(U2, a.x);
    F R = sub(S2, a.y);
    if (is_zero(Pp)) {
        if (is_zero(R)) return dbl_affine_t<INL>(q);
        return xyzz_inf<F>();
    }
    F PP = fsqr<INL>(Pp);
    F PPP = fmul<INL>(Pp, PP);
    F Q = fmul<INL>(a.x, PP);
    F X3 = sub(sub(fsqr<INL>(R), PPP), dbl(Q));
    F Y3 = sub(fmul<INL>(R, sub(Q, X3)), fmul<INL>(a.y, PPP));
    return {X3, Y3, fmul<INL>(a.zz, PP), fmul<INL>(a.zzz, PPP)};
}
template <class F>
GA_HD_BIG XYZZ<F> madd(const XYZZ<F>& a, const Affine<F>& q) { return madd_t<false>(a, q); }

// a + b   (add-2008-s), complete
template <class F>
GA_HD_CALL XYZZ<F> add(const XYZZ<F>& a, const XYZZ<F>& b) {
    if (is_inf(b)) return a;
    if (is_inf(a)) return b;
    F U1 = mul(a.x, b.zz);
    F U2 = mul(b.x, a.zz);
    F S1 = mul(a.y, b.zzz);
    F S2 = mul(b.y, a.zzz);
    F Pp = sub(U2, U1);
    F R = sub(S2, S1);
    if (is_zero(Pp)) {
        if (is_zero(R)) return dbl(a);
        return xyzz_inf<F>();
    }
    F PP = sqr(Pp);
    F PPP = mul(Pp, PP);
    F Q = mul(U1, PP);
    F X3 = sub(sub(sqr(R), PPP), dbl(Q));
    F Y3 = sub(mul(R, sub(Q, X3)), mul(S1, PPP));
    return {X3, Y3, mul(mul(a.zz, b.zz), PP), mul(mul(a.zzz, b.zzz), PPP)};
}

// XYZZ -> Jacobian without inversion: Z := ZZZ, X := X*ZZ^2, Y := Y*ZZZ^2   (Z^2 = ZZ^3, Z^3 = ZZZ^3)
template <class F>
GA_HD Jac<F> to_jac(const XYZZ<F>& p) {
    if (is_inf(p)) {
        F o = FieldTraits<F>::one();
        return {o, o, FieldTraits<F>::zero()};   // gnark's canonical Jacobian infinity (1,1,0)
    }
    return {mul(p.x, sqr(p.zz)), mul(p.y, sqr(p.zzz)), p.zzz};
}

template <class F>
GA_HD XYZZ<F> from_jac(const Jac<F>& p) {
    if (is_zero(p.z)) return xyzz_inf<F>();
    F zz = sqr(p.z);
    return {p.x, p.y, zz, mul(zz, p.z)};
}

// XYZZ -> affine (two inversions folded into one)
template <class F>
GA_HD_CALL Affine<F> to_affine(const XYZZ<F>& p) {
    if (is_inf(p)) return {FieldTraits<F>::zero(), FieldTraits<F>::zero()};
    F i = inv(mul(p.zz, p.zzz));          // 1/(zz*zzz)
    F izz = mul(i, p.zzz);                // 1/zz
    F izzz = mul(i, p.zz);                // 1/zzz
    return {mul(p.x, izz), mul(p.y, izzz)};
}

// [k]P for a little-endian scalar of nwords 32-bit words (canonical integer, NOT Montgomery)
template <class F>
GA_HD_CALL XYZZ<F> scalar_mul(const XYZZ<F>& p, const uint32_t* k, int nwords) {
    XYZZ<F> r = xyzz_inf<F>();
    for (int i = nwords - 1; i >= 0; i--) {
        for (int b = 31; b >= 0; b--) {
            r = dbl(r);
            if ((k[i] >> b) & 1) r = add(r, p);
        }
    }
    return r;
}

// [k1]P1 + [k2]P2 (Straus: one doubling chain for both scalars, 4-bit windows over two tables of 15 multiples): 256 doublings +
// <= 128 + 28 additions instead of the 2 x (256 + ~128) of two separate double-and-add loops.  Host epilogue of a proof.
template <class F>
inline XYZZ<F> scalar_mul2(const XYZZ<F>& p1, const uint32_t* k1, const XYZZ<F>& p2, const uint32_t* k2, int nwords) {
    XYZZ<F> t1[15], t2[15];
    t1[0] = p1;
    t2[0] = p2;
    for (int d = 1; d < 15; d++) {
        t1[d] = add(t1[d - 1], p1);
        t2[d] = add(t2[d - 1], p2);
    }
    XYZZ<F> r = xyzz_inf<F>();
    for (int w = nwords * 8 - 1; w >= 0; w--) {
        for (int q = 0; q < 4; q++) r = dbl(r);
        const uint32_t d1 = (k1[w / 8] >> (4 * (w % 8))) & 15u, d2 = (k2[w / 8] >> (4 * (w % 8))) & 15u;
        if (d1) r = add(r, t1[d1 - 1]);
        if (d2) r = add(r, t2[d2 - 1]);
    }
    return r;
}

template <class F>
GA_HD_CALL XYZZ<F> scalar_mul_u32(const XYZZ<F>& p, uint32_t k) {
    XYZZ<F> r = xyzz_inf<F>();
    for (int b = 31; b >= 0; b--) {
        r = dbl(r);
        if ((k >> b) & 1) r = add(r, p);
    }
    return r;
}

}  // namespace ga
