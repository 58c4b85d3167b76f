// Prime-field arithmetic for the MI355X prover kernels.
//
// Memory image == gnark-crypto's fp.Element / fr.Element: little-endian 64-bit limbs holding a*R mod p,
// R = 2^(64*N64)  (SURVEY Appendix A).  On the device the same bytes are viewed as 2*N64 32-bit limbs
// because CDNA4's widest integer multiplier is v_mad_u64_u32 (32x32+64 -> 64); R is unchanged, so values
// round-trip bit-for-bit with the Go side.
//
// Replaces (as an independent design): gnark-crypto field arithmetic reached from
// backend/groth16/bn254/prove.go:194-283,362-386 and the ICICLE field kernels reached from
// backend/accelerated/icicle/groth16/bn254/icicle.go:1425-1480.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "constants.h"

#define GA_HD __host__ __device__ __forceinline__
// Large bodies: inlined into kernels on the device, real functions in host code (keeps host objects small).
// GA_HD_CALL bodies are real functions on the device as well: out-of-line Fp2 / point arithmetic keeps the hot
// loops inside the instruction cache and the build time sane (a G2 point addition is ~40 Montgomery products).
#if defined(__HIP_DEVICE_COMPILE__)
#define GA_HD_BIG __host__ __device__ __forceinline__
#else
#define GA_HD_BIG __host__ __device__ inline __attribute__((noinline))
#endif
#define GA_HD_CALL __host__ __device__ inline __attribute__((noinline))
// keep a loaded value (and therefore its load) alive up to this point without using it
#if defined(__HIP_DEVICE_COMPILE__)
#define GA_KEEP_LIVE(x) asm volatile("" ::"v"(x))
#else
#define GA_KEEP_LIVE(x) asm volatile("" ::"r"(x))
#endif

namespace ga {

template <class P>
struct Fe {
    static constexpr int N = P::N;
    uint32_t l[N];
};

// ---- raw multi-word helpers -------------------------------------------------------------------

template <int N>
GA_HD uint32_t add_words(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (uint64_t)a[i] + b[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}

template <int N>
GA_HD uint32_t sub_words(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (int64_t)a[i] - (int64_t)b[i];
        r[i] = (uint32_t)c;
        c >>= 32;   // arithmetic shift: 0 or -1
    }
    return (uint32_t)(c & 1);   // borrow
}

template <class P>
GA_HD bool geq_mod(const uint32_t* a) {
#pragma unroll
    for (int i = P::N - 1; i >= 0; i--) {
        if (a[i] > P::MOD[i]) return true;
        if (a[i] < P::MOD[i]) return false;
    }
    return true;
}

// r = a - p if a >= p  (branch-free: compute a-p, keep it when no borrow)
template <class P>
GA_HD void reduce_once(uint32_t* a) {
    uint32_t t[P::N];
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        c += (int64_t)a[i] - (int64_t)P::MOD[i];
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    bool keep = (c == 0);
#pragma unroll
    for (int i = 0; i < P::N; i++) a[i] = keep ? t[i] : a[i];
}

// ---- field API --------------------------------------------------------------------------------

template <class P>
GA_HD Fe<P> fe_zero() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = 0;
    return r;
}

template <class P>
GA_HD Fe<P> fe_one() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = P::ONE[i];
    return r;
}

template <class P>
GA_HD Fe<P> fe_const(const uint32_t (&c)[P::N]) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = c[i];
    return r;
}

template <class P>
GA_HD bool is_zero(const Fe<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) o |= a.l[i];
    return o == 0;
}

template <class P>
GA_HD bool eq(const Fe<P>& a, const Fe<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

template <class P>
GA_HD Fe<P> add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    add_words<P::N>(r.l, a.l, b.l);   // both moduli leave >= 1 spare bit: no carry out of the top word
    reduce_once<P>(r.l);
    return r;
}

template <class P>
GA_HD Fe<P> dbl(const Fe<P>& a) { return add(a, a); }

template <class P>
GA_HD Fe<P> sub(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint32_t borrow = sub_words<P::N>(r.l, a.l, b.l);
    uint32_t m = 0u - borrow;   // all-ones when we must add p back
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        c += (uint64_t)r.l[i] + (P::MOD[i] & m);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class P>
GA_HD Fe<P> neg(const Fe<P>& a) {
    Fe<P> r;
    if (is_zero(a)) return a;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        c += (int64_t)P::MOD[i] - (int64_t)a.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

// Montgomery product a*b*R^-1 mod p, R = 2^(32N) (gnark's R), computed carry-free in a smaller radix.
//
// gfx950 has no multiply with carry-in: v_mad_u64_u32 is 32x32+64 -> 64 at ~half rate, and a 64-bit add or an add-with-
// carry costs the same issue slot as the multiply (profiles/r01_e_microbench.json: all ~30-35 Gop/s/lane-group), so a
// 32-bit-limb CIOS spends more time on carry plumbing (v_lshl_add_u64, v_mov to build {x,0} pairs) than on products.
// Instead the operands are unpacked to L-bit limbs (L = 29 for the 254/255-bit fields, 28 for the 381-bit one) so that
// a whole column  sum_i a_i*b_j + sum_i m_i*p_j  (2*NL products < 2^(2L)) fits a 64-bit accumulator: every product is ONE
// v_mad_u64_u32 accumulating in place, and carries are resolved once per column with a shift.
//   a' = a << S  (S = NL*L - 32N), so that  a'*b / 2^(NL*L) = a*b / 2^(32N)  -- the memory format keeps gnark's R.
//   result < 1.25 p  ->  one conditional subtraction.
template <class P>
struct Radix {
    static constexpr int L = (P::N == 8) ? 29 : 28;
    static constexpr int NL = (32 * P::N + L - 1) / L + ((32 * P::N) % L == 0 ? 1 : 0);
    static constexpr int S = NL * L - 32 * P::N;
    static constexpr uint32_t MASK = (1u << L) - 1;
    static_assert(S > 0 && S < L, "pre-shift must fit one limb");
    static_assert(2 * L + 6 <= 64, "column sums must fit 64 bits");
};

// bits [lo, lo+L) of the little-endian word array w[0..N), zero outside
template <int N, int L>
GA_HD uint32_t take_bits(const uint32_t* w, int lo) {
    if (lo + L <= 0 || lo >= 32 * N) return 0;
    uint64_t v;
    if (lo < 0) {
        v = (uint64_t)w[0] << (-lo);
    } else {
        const int i = lo >> 5, off = lo & 31;
        v = (uint64_t)w[i] >> off;
        if (i + 1 < N) v |= (uint64_t)w[i + 1] << (32 - off);
    }
    return (uint32_t)v & ((1u << L) - 1);
}

// p as L-bit limbs and -p^-1 mod 2^L, folded to immediates by the compiler
template <class P>
GA_HD uint32_t mod_limb(int j) { return take_bits<P::N, Radix<P>::L>(P::MOD, j * Radix<P>::L); }

// The same product for HOST code (the proof epilogue, the Horner step of a raw MSM: 15 windows x 17 doublings took 0.2 ms of a 2.7 ms
// 2^20 MSM through the limb-unpacking form above, which is shaped for the GPU's multiplier): plain CIOS on 64-bit limbs with the
// compiler's 128-bit products.  Memory format and result are identical (canonical, below p).  The functional emulation keeps the
// device form on the host, since testing that form is what it is for.
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(GA_HIP_EMULATION)
#define GA_HOST_MUL64 1
template <class P>
inline uint64_t mont_inv64() {   // -p^-1 mod 2^64 from P::INV = -p^-1 mod 2^32: one Newton step
    const uint64_t p0 = (uint64_t)P::MOD[0] | ((uint64_t)P::MOD[1] << 32);
    uint64_t x = (uint64_t)(0u - P::INV);   // p^-1 mod 2^32
    x *= 2 - p0 * x;                       // p^-1 mod 2^64
    return 0 - x;
}
template <class P>
inline Fe<P> mul_host64(const Fe<P>& a, const Fe<P>& b) {
    static_assert(P::N % 2 == 0, "64-bit limbs");
    constexpr int M = P::N / 2;
    typedef unsigned __int128 u128;
    uint64_t A[M], B[M], Q[M], t[M + 2];
    memcpy(A, a.l, 8 * M);
    memcpy(B, b.l, 8 * M);
    for (int i = 0; i < M; i++) Q[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    const uint64_t inv = mont_inv64<P>();
    for (int i = 0; i < M; i++) {
        u128 c = 0;
        for (int j = 0; j < M; j++) {
            c += (u128)A[j] * B[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[M];
        t[M] = (uint64_t)c;
        t[M + 1] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * inv;
        c = ((u128)m * Q[0] + t[0]) >> 64;
        for (int j = 1; j < M; j++) {
            c += (u128)m * Q[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[M];
        t[M - 1] = (uint64_t)c;
        t[M] = t[M + 1] + (uint64_t)(c >> 64);
    }
    // the moduli leave spare bits in the top word (254 of 256, 381 of 384): t < 2p < 2^(32N), t[M] == 0
    uint32_t w[P::N];
    memcpy(w, t, 8 * M);
    reduce_once<P>(w);
    Fe<P> r;
    for (int i = 0; i < P::N; i++) r.l[i] = w[i];
    return r;
}
#endif

template <class P>
GA_HD_BIG Fe<P> mul_body(const Fe<P>& a, const Fe<P>& b) {
#ifdef GA_HOST_MUL64
    return mul_host64(a, b);
#else
    typedef Radix<P> R;
    constexpr int N = P::N, L = R::L, NL = R::NL;
    uint32_t al[NL], bl[NL], pl[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        al[i] = take_bits<N, L>(a.l, i * L - R::S);   // a << S
        bl[i] = take_bits<N, L>(b.l, i * L);
        pl[i] = mod_limb<P>(i);
    }
    // column accumulators of the schoolbook product
    uint64_t col[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)al[i] * bl[j];
    // Montgomery reduction, one L-bit digit per row; -p^-1 mod 2^L == low L bits of (-p^-1 mod 2^32)
    const uint32_t inv = P::INV & R::MASK;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t m = ((uint32_t)col[i] * inv) & R::MASK;
#pragma unroll
        for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)m * pl[j];
        col[i + 1] += col[i] >> L;   // low L bits of col[i] are now zero
    }
    // carry-normalise the high half and repack to 32-bit words
    uint32_t rl[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        rl[k] = (uint32_t)col[NL + k] & R::MASK;
        if (k + 1 < NL) col[NL + k + 1] += col[NL + k] >> L;
        else rl[k] = (uint32_t)col[NL + k];   // top limb keeps everything (value < 2p < 2^(32N))
    }
    uint32_t t[N];
#pragma unroll
    for (int w = 0; w < N; w++) {
        // word w = bits [32w, 32w+32) of sum rl[k] << (k*L)
        const int k0 = (32 * w) / L, off = (32 * w) % L;
        uint64_t v = (uint64_t)rl[k0] >> off;
        if (k0 + 1 < NL) v |= (uint64_t)rl[k0 + 1] << (L - off);
        if (2 * L - off < 32 && k0 + 2 < NL) v |= (uint64_t)rl[k0 + 2] << (2 * L - off);
        t[w] = (uint32_t)v;
    }
    reduce_once<P>(t);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    return r;
#endif
}

template <class P>
GA_HD_CALL Fe<P> mul_call(const Fe<P>& a, const Fe<P>& b) { return mul_body(a, b); }

// by-value variant: operands and result travel in VGPRs (AMDGPU calling convention), no memory traffic: the cold PLONK passes
// call it (one shared copy per field).  The hot loops inline their products (out-of-line products for the 12-limb field there
// measured -20..25 %, BLS12-381 G1 accumulate 42 -> 51 ms, round 1).
template <class P>
GA_HD_CALL Fe<P> mul_val(Fe<P> a, Fe<P> b) { return mul_body(a, b); }
template <class P>
GA_HD Fe<P> mul_hot(const Fe<P>& a, const Fe<P>& b) { return mul_body(a, b); }

// 8-limb fields (BN254 Fp/Fr, BLS12-381 Fr): inlined.  12-limb BLS12-381 Fp: one shared out-of-line copy.
template <class P>
GA_HD Fe<P> mul(const Fe<P>& a, const Fe<P>& b) {
    if constexpr (P::N > 8) return mul_call(a, b);
    else return mul_body(a, b);
}

template <class P>
GA_HD Fe<P> sqr(const Fe<P>& a) { return mul(a, a); }

// a * R^-1 : leave Montgomery form (fr.Element.BigInt / FromMontgomery)
template <class P>
GA_HD Fe<P> from_mont(const Fe<P>& a) {
    Fe<P> one = fe_zero<P>();
    one.l[0] = 1;
    return mul(a, one);
}

template <class P>
GA_HD Fe<P> to_mont(const Fe<P>& a) { return mul(a, fe_const<P>(P::R2)); }

// a^e, e given as N 32-bit words (little-endian)
template <class P>
GA_HD_CALL Fe<P> pow_words(const Fe<P>& a, const uint32_t* e, int nwords) {
    Fe<P> r = fe_one<P>();
    for (int i = nwords - 1; i >= 0; i--) {
        for (int b = 31; b >= 0; b--) {
            r = sqr(r);
            if ((e[i] >> b) & 1) r = mul(r, a);
        }
    }
    return r;
}

template <class P>
GA_HD_CALL Fe<P> pow_u64(const Fe<P>& a, uint64_t e) {
    Fe<P> r = fe_one<P>();
    Fe<P> x = a;
    while (e) {
        if (e & 1) r = mul(r, x);
        x = sqr(x);
        e >>= 1;
    }
    return r;
}

// Fermat inverse (0 -> 0)
template <class P>
GA_HD Fe<P> inv(const Fe<P>& a) {
    uint32_t e[P::N];
#pragma unroll
    for (int i = 0; i < P::N; i++) e[i] = P::PM2[i];
    return pow_words(a, e, P::N);
}

// a * k for a small unsigned k (Montgomery-form-preserving), via double-and-add
template <class P>
GA_HD Fe<P> mul_small(const Fe<P>& a, uint32_t k) {
    Fe<P> r = fe_zero<P>();
    Fe<P> x = a;
    while (k) {
        if (k & 1) r = add(r, x);
        x = dbl(x);
        k >>= 1;
    }
    return r;
}

// ---- 16-byte vector access (coalesced 16 B/lane loads) ---------------------------------------

struct alignas(16) u32x4 {
    uint32_t x, y, z, w;
};

// One 16-byte vector load that the compiler may neither split nor narrow.  Left alone, LLVM fuses a point load with the 29-bit
// limb extraction that follows and emits dozens of 2-byte global_load_ushort at odd offsets (52 per G2 point) instead of eight
// global_load_dwordx4.  Round 1 stopped that with a `volatile` access -- which also made every 16-byte piece a system-scope
// (sc0 sc1) load followed by its own s_waitcnt vmcnt(0): four serialized memory round trips per G1 table entry, eight per G2
// entry, up to 24 per bucket in the window reduction.  An empty asm on the loaded VALUE hides it from the combiner just as well
// and leaves the load an ordinary one: the pieces of a point are issued back to back and waited for once.
typedef uint32_t ga_v4u __attribute__((vector_size(16)));
// n consecutive 16-byte pieces: all loads first, then the values are made opaque (an asm right after each load would make the
// compiler wait for that load before issuing the next one)
template <int NV>
GA_HD void load16n(const void* p, ga_v4u (&v)[NV]) {
    const ga_v4u* q = reinterpret_cast<const ga_v4u*>(p);
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = q[i];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < NV; i++) asm volatile("" : "+v"(v[i]));
#endif
}
GA_HD u32x4 load16(const void* p) {
    ga_v4u v[1];
    load16n<1>(p, v);
    u32x4 r;
    r.x = v[0][0];
    r.y = v[0][1];
    r.z = v[0][2];
    r.w = v[0][3];
    return r;
}

template <class P>
GA_HD Fe<P> load_fe(const void* p) {
    Fe<P> r;
    ga_v4u v[P::N / 4];
    load16n<P::N / 4>(p, v);
#pragma unroll
    for (int i = 0; i < P::N / 4; i++) {
        r.l[4 * i + 0] = v[i][0];
        r.l[4 * i + 1] = v[i][1];
        r.l[4 * i + 2] = v[i][2];
        r.l[4 * i + 3] = v[i][3];
    }
    return r;
}

// plain (non-volatile) variant: the compiler may narrow / reschedule it; measured faster for L2-resident twiddles
template <class P>
GA_HD Fe<P> load_fe_plain(const void* p) {
    Fe<P> r;
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; i++) {
        u32x4 v = q[i];
        r.l[4 * i + 0] = v.x;
        r.l[4 * i + 1] = v.y;
        r.l[4 * i + 2] = v.z;
        r.l[4 * i + 3] = v.w;
    }
    return r;
}

template <class P>
GA_HD void store_fe(void* p, const Fe<P>& a) {
    u32x4* q = reinterpret_cast<u32x4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; i++) {
        u32x4 v;
        v.x = a.l[4 * i + 0];
        v.y = a.l[4 * i + 1];
        v.z = a.l[4 * i + 2];
        v.w = a.l[4 * i + 3];
        q[i] = v;
    }
}

// whole-struct 16-byte vector copies (points)
template <class T>
GA_HD T load_pod(const void* p) {
    static_assert(sizeof(T) % 16 == 0, "16-byte multiples only");
    T r;
    ga_v4u v[sizeof(T) / 16];
    load16n<(int)(sizeof(T) / 16)>(p, v);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 16; i++) memcpy(reinterpret_cast<char*>(&r) + 16 * i, &v[i], 16);
    return r;
}

template <class T>
GA_HD void store_pod(void* p, const T& a) {
    static_assert(sizeof(T) % 16 == 0, "16-byte multiples only");
    u32x4* d = reinterpret_cast<u32x4*>(p);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 16; i++) {
        u32x4 v;
        memcpy(&v, reinterpret_cast<const char*>(&a) + 16 * i, 16);
        d[i] = v;
    }
}

}  // namespace ga
