// PLONK quotient on the device: computeNumerator + divideByZH of backend/plonk/bn254/prove.go:841-1123,1287-1350
// (SURVEY 8f row 4), and the grand-product polynomial of iop.BuildRatioCopyConstraint (call site prove.go:645-655).
//
// The reference keeps 13+2k polynomials of size n on the CPU and, for each of the rho = |domain1|/n cosets
// coset_i = g*w1^i, converts every polynomial back to canonical form, rescales it and runs a forward FFT (two FFTs per
// polynomial and coset, prove.go:1033-1058: "we do **a lot** of FFT here").  Here the canonical coefficients stay resident
// in HBM in bit-reversed order (one inverse DIF per polynomial, once), each coset costs ONE out-of-place coset DIT per
// polynomial with the coset scaling fused into its first pass, the pointwise constraint (gate + alpha*ordering +
// alpha^2*(Z-1)*L1, blinding included) is one kernel that also applies divideByZH's 1/(X^n-1) factor and writes the result
// at its bit-reversed slot, and the final inverse coset transform of size rho*n produces h in canonical order.
// 1/(x-1) for L1 comes from a batched Montgomery-trick inversion kernel (the reference's batchInvert, prove.go:1134-1147).
// Field elements are mathematically unique, so the coefficients equal the reference's bit for bit.
#pragma once
#include "ntt.hip.h"

namespace ga {

// PLONK_MAX_BSB, PLONK_NB_FIXED and PlonkQuotientArgs live in common.hip.h (shared with the ABI translation unit)
enum { PX_L = 0, PX_R, PX_O, PX_Z, PX_QL, PX_QR, PX_QM, PX_QO, PX_QK, PX_S1, PX_S2, PX_S3 };

struct PlonkPtrs {
    const uint32_t* p[PLONK_NB_FIXED + 2 * PLONK_MAX_BSB];   // [12 + 2*i] = Qcp_i, [13 + 2*i] = committed polynomial i
    int nb_bsb;
};

struct PlonkConsts {
    uint32_t alpha[8], beta[8], gamma[8], cs[8], css[8];
    uint32_t bl[2][8], br[2][8], bo[2][8], bz[3][8];   // blinding coefficients already multiplied by (coset^n - 1)
    uint32_t lone[8];      // (coset^n - 1) / n
    uint32_t omega[8];     // generator of the small domain
    uint32_t zh_inv[8];    // 1 / (coset^n - 1): divideByZH's factor for this coset (prove.go:1327-1350)
    // 32 * c mod r of the constants that the lazy constraint kernel multiplies by (plonk_constraints29_kernel): a product of limb
    // vectors divides by 2^261 where the memory format wants 2^256, so one factor carries the 2^5
    uint32_t sh[11][8];
};
enum { PSH_OMEGA = 0, PSH_BL1, PSH_BR1, PSH_BO1, PSH_BZ2, PSH_BETA, PSH_CS, PSH_CSS, PSH_LONE, PSH_ALPHA, PSH_ZHINV };

template <class FrP>
__device__ __forceinline__ Fe<FrP> plonk_c(const uint32_t* w) {
    Fe<FrP> r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.l[k] = w[k];
    return r;
}

// x_j = coset * w^j from the two power tables (lo: coset*w^k, k < 2^lo_bits; hi: w^(k << lo_bits)), plain Montgomery form
template <class FrP>
__device__ __forceinline__ Fe<FrP> plonk_point(const uint32_t* lo, const uint32_t* hi, int lo_bits, uint64_t j) {
    Fe<FrP> a = load_fe<FrP>(lo + (j & ((1ull << lo_bits) - 1)) * 8);
    uint64_t h = j >> lo_bits;
    return h ? mul(a, load_fe<FrP>(hi + h * 8)) : a;
}

template <class FrP>
__global__ void plonk_x_minus_one_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                         int lo_bits, uint64_t n) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    store_fe(out + j * 8, sub(plonk_point<FrP>(lo, hi, lo_bits, j), fe_one<FrP>()));
}

// In-place batched inversion (Montgomery's trick; zeros stay zero like fr.BatchInvert).  Thread t owns elements
// t, t+T, t+2T, ... (T = total threads): every access is coalesced across the wave; tmp holds the running products.
template <class FrP>
__global__ void fr_batch_inverse_kernel(uint32_t* __restrict__ v, uint32_t* __restrict__ tmp, uint64_t n) {
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Fe<FrP> acc = fe_one<FrP>();
    for (uint64_t i = t; i < n; i += T) {
        store_fe(tmp + i * 8, acc);   // product of the (non-zero) elements before i
        Fe<FrP> x = load_fe<FrP>(v + i * 8);
        if (!is_zero(x)) acc = mul(acc, x);
    }
    acc = inv(acc);
    uint64_t last = t + ((n - 1 - t) / T) * T;
    for (uint64_t i = last;; i -= T) {
        Fe<FrP> x = load_fe<FrP>(v + i * 8);
        if (!is_zero(x)) {
            store_fe(v + i * 8, mul(acc, load_fe<FrP>(tmp + i * 8)));
            acc = mul(acc, x);
        }
        if (i == t) break;
    }
}

// allConstraints of prove.go:950-981 at point j of the current coset, times 1/(X^n-1) on this coset (divideByZH,
// prove.go:1311-1316), stored where the big inverse DIT expects evaluation rho*j + i: block*n + bitrev_n(j) (prove.go:1073).
template <class FrP>
__global__ void __launch_bounds__(128)
plonk_constraints_kernel(PlonkPtrs P, PlonkConsts K, const uint32_t* __restrict__ x_lo, const uint32_t* __restrict__ x_hi, int lo_bits,
                         const uint32_t* __restrict__ inv_xm1, uint32_t* __restrict__ out_block, uint64_t n, int logn) {
    typedef Fe<FrP> F;
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    // ~45 products per point against 14 x 32 B of loads: one shared out-of-line multiply keeps the kernel small
    auto mul = [](const F& u, const F& v) { return mul_val(u, v); };
    auto at = [&](int id, uint64_t idx) { return load_fe<FrP>(P.p[id] + idx * 8); };
    const F x = plonk_point<FrP>(x_lo, x_hi, lo_bits, j);
    const F xw = mul(x, plonk_c<FrP>(K.omega));   // next point of the coset: ZS is Z shifted by one (prove.go:602,972)
    const F beta = plonk_c<FrP>(K.beta), gamma = plonk_c<FrP>(K.gamma), alpha = plonk_c<FrP>(K.alpha);
    // blinded wires (prove.go:957-973): p + b(x)*(x^n - 1), the factor (coset^n - 1) is folded into the coefficients
    F l = add(at(PX_L, j), add(plonk_c<FrP>(K.bl[0]), mul(plonk_c<FrP>(K.bl[1]), x)));
    F r = add(at(PX_R, j), add(plonk_c<FrP>(K.br[0]), mul(plonk_c<FrP>(K.br[1]), x)));
    F o = add(at(PX_O, j), add(plonk_c<FrP>(K.bo[0]), mul(plonk_c<FrP>(K.bo[1]), x)));
    auto bz = [&](const F& pt) {
        return add(plonk_c<FrP>(K.bz[0]), mul(pt, add(plonk_c<FrP>(K.bz[1]), mul(pt, plonk_c<FrP>(K.bz[2])))));
    };
    F z = add(at(PX_Z, j), bz(x));
    F zs = add(at(PX_Z, j + 1 == n ? 0 : j + 1), bz(xw));
    // gate (prove.go:868-885)
    F gate = add(mul(at(PX_QL, j), l), mul(at(PX_QR, j), r));
    gate = add(gate, mul(mul(at(PX_QM, j), l), r));
    gate = add(gate, mul(at(PX_QO, j), o));
    gate = add(gate, at(PX_QK, j));
    for (int t = 0; t < P.nb_bsb; t++) gate = add(gate, mul(at(PLONK_NB_FIXED + 2 * t, j), at(PLONK_NB_FIXED + 2 * t + 1, j)));
    // ordering (prove.go:898-923)
    F id = mul(x, beta);
    F a = add(add(gamma, l), id);
    F b = add(add(mul(id, plonk_c<FrP>(K.cs)), r), gamma);
    F c = add(add(mul(id, plonk_c<FrP>(K.css)), o), gamma);
    F rr = mul(mul(mul(a, b), c), z);
    a = add(add(mul(at(PX_S1, j), beta), l), gamma);
    b = add(add(mul(at(PX_S2, j), beta), r), gamma);
    c = add(add(mul(at(PX_S3, j), beta), o), gamma);
    F ll = mul(mul(mul(a, b), c), zs);
    F ord = sub(ll, rr);
    // local (prove.go:926-934, 380-385)
    F lone = mul(plonk_c<FrP>(K.lone), load_fe<FrP>(inv_xm1 + j * 8));
    F loc = mul(sub(z, fe_one<FrP>()), lone);
    F res = add(mul(add(mul(loc, alpha), ord), alpha), gate);
    res = mul(res, plonk_c<FrP>(K.zh_inv));
    store_fe(out_block + bitrev64(j, logn) * 8, res);
}

// The same expression in the lazy representation (field29.hip.h) for scalar fields with >= 7 spare bits in nine 29-bit limbs
// (BN254's Fr; BASELINE config 5): values are memory images x * 2^256 held as unreduced limbs -- additions are limb-wise with one carry
// sweep and no modular correction, a product is f29_mul (162 multiply-adds + a shift / mask per column, ~250 instructions inline
// against ~360 + a call for the packed product).  f29_mul divides by 2^261, the memory format by 2^256: a product takes one factor
// times 32 -- a constant as its precomputed image 32 c mod r (PlonkConsts::sh: canonical, so the other factor may be anything below
// 2^261), a variable by a 5-bit limb shift (it must then be below 2^256 = 5.29 r).  Every intermediate's bound: tools/lazy_bounds.py
// check_plonk_constraints (largest value 27 r of the 169 r that fit); one exact reduction at the store.
template <class FrP>
__device__ __forceinline__ F29<FrP> plonk_shl5(const F29<FrP>& a) {   // 32 a for a normalized a < 2^256
    constexpr int L = Radix<FrP>::L, NL = Radix<FrP>::NL;
    constexpr uint32_t MASK = (1u << L) - 1;
    F29<FrP> r;
    r.l[0] = (a.l[0] << 5) & MASK;
#pragma unroll
    for (int i = 1; i < NL - 1; i++) r.l[i] = ((a.l[i] << 5) & MASK) | (a.l[i - 1] >> (L - 5));
    r.l[NL - 1] = (a.l[NL - 1] << 5) | (a.l[NL - 2] >> (L - 5));
    return r;
}
template <class FrP>
__global__ void __launch_bounds__(128)
plonk_constraints29_kernel(PlonkPtrs P, PlonkConsts K, const uint32_t* __restrict__ x_lo, const uint32_t* __restrict__ x_hi, int lo_bits,
                           const uint32_t* __restrict__ inv_xm1, uint32_t* __restrict__ out_block, uint64_t n, int logn) {
    static_assert(Radix<FrP>::NL * Radix<FrP>::L - 32 * FrP::N == 5 && Radix<FrP>::NL * Radix<FrP>::L - FrP::BITS >= 7,
                  "the shift-by-5 products and the bounds of tools/lazy_bounds.py need nine 29-bit limbs over a <= 254-bit modulus");
    typedef F29<FrP> E;
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    auto at = [&](int id, uint64_t idx) { return f29_unpack(load_fe<FrP>(P.p[id] + idx * 8)); };   // canonical
    auto cst = [&](const uint32_t* w) { return f29_unpack(plonk_c<FrP>(w)); };                       // canonical
    auto mulc = [&](int sh_id, const E& v) { return f29_mul(cst(K.sh[sh_id]), v); };                 // constant x anything below 2^261
    auto mulv = [&](const E& small, const E& v) { return f29_mul(plonk_shl5<FrP>(small), v); };      // `small` below 2^256
    E x = f29_unpack(load_fe<FrP>(x_lo + (j & ((1ull << lo_bits) - 1)) * 8));
    if (const uint64_t h = j >> lo_bits) x = mulv(x, f29_unpack(load_fe<FrP>(x_hi + h * 8)));
    const E xw = mulc(PSH_OMEGA, x);   // next point of the coset: ZS is Z shifted by one (prove.go:602,972)
    const E gamma = cst(K.gamma);
    // blinded wires (prove.go:957-973)
    const E l = f29_add(at(PX_L, j), f29_add(cst(K.bl[0]), mulc(PSH_BL1, x)));
    const E r = f29_add(at(PX_R, j), f29_add(cst(K.br[0]), mulc(PSH_BR1, x)));
    const E o = f29_add(at(PX_O, j), f29_add(cst(K.bo[0]), mulc(PSH_BO1, x)));
    auto bz = [&](const E& pt) { return f29_add(cst(K.bz[0]), mulv(pt, f29_add(cst(K.bz[1]), mulc(PSH_BZ2, pt)))); };
    const E z = f29_add(at(PX_Z, j), bz(x));
    const E zs = f29_add(at(PX_Z, j + 1 == n ? 0 : j + 1), bz(xw));
    // gate (prove.go:868-885): the circuit-constant factor is canonical, so it is the one that takes the shift
    E gate = f29_add(mulv(at(PX_QL, j), l), mulv(at(PX_QR, j), r));
    gate = f29_add(gate, mulv(mulv(at(PX_QM, j), l), r));
    gate = f29_add(gate, mulv(at(PX_QO, j), o));
    gate = f29_add(gate, at(PX_QK, j));
    for (int t = 0; t < P.nb_bsb; t++) gate = f29_add(gate, mulv(at(PLONK_NB_FIXED + 2 * t, j), at(PLONK_NB_FIXED + 2 * t + 1, j)));
    // ordering (prove.go:898-923): each three-term factor stays below 2^256, the running product is always the un-shifted operand
    const E id = mulc(PSH_BETA, x);
    E a = f29_add(f29_add(gamma, l), id);
    E b = f29_add(f29_add(mulc(PSH_CS, id), r), gamma);
    E c = f29_add(f29_add(mulc(PSH_CSS, id), o), gamma);
    const E rr = mulv(z, mulv(c, mulv(a, b)));
    a = f29_add(f29_add(mulc(PSH_BETA, at(PX_S1, j)), l), gamma);
    b = f29_add(f29_add(mulc(PSH_BETA, at(PX_S2, j)), r), gamma);
    c = f29_add(f29_add(mulc(PSH_BETA, at(PX_S3, j)), o), gamma);
    const E ll = mulv(zs, mulv(c, mulv(a, b)));
    const E ord = f29_sub<8>(ll, rr);
    // local (prove.go:926-934, 380-385)
    const E lone = mulc(PSH_LONE, f29_unpack(load_fe<FrP>(inv_xm1 + j * 8)));
    const E loc = mulv(lone, f29_sub<2>(z, f29_unpack(fe_one<FrP>())));
    E res = f29_add(mulc(PSH_ALPHA, f29_add(mulc(PSH_ALPHA, loc), ord)), gate);
    res = mulc(PSH_ZHINV, res);
    store_fe(out_block + bitrev64(j, logn) * 8, f29_pack_canonical(f29_reduce_3p(res)));
}

// out[bitrev(i)] = in[i]
template <class FrP>
__global__ void fr_bitrev_copy_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, uint64_t n, int logn) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_fe(out + bitrev64(i, logn) * 8, load_fe<FrP>(in + i * 8));
}

// ---- grand product (iop.BuildRatioCopyConstraint) --------------------------------------------------------------------
// ratio[i] = prod_k (e_k[i] + beta*id_k(i) + gamma) and its denominator prod_k (e_k[i] + beta*id(perm[k*n+i]) + gamma)
template <class FrP>
__global__ void plonk_ratio_terms_kernel(const uint32_t* __restrict__ L, const uint32_t* __restrict__ R, const uint32_t* __restrict__ O,
                                         const int64_t* __restrict__ perm, const uint32_t* __restrict__ w_lo,
                                         const uint32_t* __restrict__ w_hi, int lo_bits, PlonkConsts K, uint32_t* __restrict__ num,
                                         uint32_t* __restrict__ den, uint64_t n) {
    typedef Fe<FrP> F;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F beta = plonk_c<FrP>(K.beta), gamma = plonk_c<FrP>(K.gamma);
    const F shift[3] = {fe_one<FrP>(), plonk_c<FrP>(K.cs), plonk_c<FrP>(K.css)};
    const uint32_t* e[3] = {L, R, O};
    auto idv = [&](uint64_t pos) {   // evaluation of the identity permutation at flat position pos: u^(pos / n) * w^(pos % n)
        uint64_t k = pos / n, j = pos % n;
        F w = plonk_point<FrP>(w_lo, w_hi, lo_bits, j);
        return k == 0 ? w : mul(w, shift[k]);
    };
    F a = fe_one<FrP>(), b = fe_one<FrP>();
    for (int k = 0; k < 3; k++) {
        F v = add(load_fe<FrP>(e[k] + i * 8), gamma);
        a = mul(a, add(v, mul(beta, idv((uint64_t)k * n + i))));
        b = mul(b, add(v, mul(beta, idv((uint64_t)perm[(uint64_t)k * n + i]))));
    }
    store_fe(num + i * 8, a);
    store_fe(den + i * 8, b);
}

// exclusive prefix product over n elements, three phases: per-chunk products, scan of the chunk products (one block),
// per-chunk rescan with the carried-in prefix.  z[0] = 1, z[i] = prod_{t<i} r[t].
constexpr int PROD_CHUNK = 256;
template <class FrP>
__global__ void fr_chunk_product_kernel(const uint32_t* __restrict__ num, const uint32_t* __restrict__ den_inv,
                                        uint32_t* __restrict__ chunk_prod, uint64_t n) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = c * PROD_CHUNK;
    if (lo >= n) return;
    uint64_t hi = lo + PROD_CHUNK < n ? lo + PROD_CHUNK : n;
    Fe<FrP> acc = fe_one<FrP>();
    for (uint64_t i = lo; i < hi; i++) acc = mul(acc, mul(load_fe<FrP>(num + i * 8), load_fe<FrP>(den_inv + i * 8)));
    store_fe(chunk_prod + c * 8, acc);
}
template <class FrP>
__global__ void __launch_bounds__(256) fr_chunk_scan_kernel(uint32_t* __restrict__ chunk_prod, uint64_t nchunks) {
    // one block: each thread multiplies a contiguous segment of chunk products, the 256 segment products are scanned in LDS
    // (Hillis-Steele, 8 steps), then every thread rewrites its segment as exclusive prefixes
    __shared__ uint32_t sh[256 * 8];
    const uint32_t tid = threadIdx.x;
    const uint64_t seg = (nchunks + 255) / 256;
    const uint64_t lo = (uint64_t)tid * seg < nchunks ? (uint64_t)tid * seg : nchunks;
    const uint64_t hi = lo + seg < nchunks ? lo + seg : nchunks;
    Fe<FrP> acc = fe_one<FrP>();
    for (uint64_t c = lo; c < hi; c++) acc = mul_val(acc, load_fe<FrP>(chunk_prod + c * 8));
    store_fe(sh + tid * 8, acc);
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        Fe<FrP> v = load_fe<FrP>(sh + tid * 8);
        if (tid >= off) v = mul_val(v, load_fe<FrP>(sh + (tid - off) * 8));
        __syncthreads();
        store_fe(sh + tid * 8, v);
        __syncthreads();
    }
    acc = tid ? load_fe<FrP>(sh + (tid - 1) * 8) : fe_one<FrP>();
    for (uint64_t c = lo; c < hi; c++) {
        Fe<FrP> p = load_fe<FrP>(chunk_prod + c * 8);
        store_fe(chunk_prod + c * 8, acc);
        acc = mul_val(acc, p);
    }
}
template <class FrP>
__global__ void fr_chunk_apply_kernel(const uint32_t* __restrict__ num, const uint32_t* __restrict__ den_inv,
                                      const uint32_t* __restrict__ chunk_prefix, uint32_t* __restrict__ z, uint64_t n) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = c * PROD_CHUNK;
    if (lo >= n) return;
    uint64_t hi = lo + PROD_CHUNK < n ? lo + PROD_CHUNK : n;
    Fe<FrP> acc = load_fe<FrP>(chunk_prefix + c * 8);
    for (uint64_t i = lo; i < hi; i++) {
        store_fe(z + i * 8, acc);
        acc = mul(acc, mul(load_fe<FrP>(num + i * 8), load_fe<FrP>(den_inv + i * 8)));
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------

// two-level power table of `base` with first factor c0 (plain Montgomery form): lo[k] = c0*base^k, hi[k] = base^(k << LO)
template <class FrP>
int plonk_pow_tables(Ctx* ctx, const char* key, const Fe<FrP>& base, const Fe<FrP>& c0, uint64_t n, bool hat, uint32_t** d_lo,
                     uint32_t** d_hi) {
    typedef Fe<FrP> F;
    const uint64_t nlo = 1ull << NTT_POW_LO_BITS;
    const uint64_t nhi = (n >> NTT_POW_LO_BITS) ? (n >> NTT_POW_LO_BITS) : 1;
    std::vector<uint32_t> buf((nlo + nhi) * 8);
    F acc = c0;
    for (uint64_t k = 0; k < nlo; k++) {
        F st = hat ? f29_hat_packed(acc) : acc;
        memcpy(&buf[k * 8], st.l, 32);
        acc = mul(acc, base);
    }
    F step = base;
    for (int k = 0; k < NTT_POW_LO_BITS; k++) step = sqr(step);
    acc = fe_one<FrP>();
    for (uint64_t k = 0; k < nhi; k++) {
        F st = hat ? f29_hat_packed(acc) : acc;
        memcpy(&buf[(nlo + k) * 8], st.l, 32);
        acc = mul(acc, step);
    }
    void* d;
    GA_CHECK(ctx->scratch_get(key, buf.size() * 4, &d));
    // the staging vector dies at return: synchronous copy (pageable memory)
    GA_HIP_CHECK(hipMemcpyAsync(d, buf.data(), buf.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    GA_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *d_lo = (uint32_t*)d;
    *d_hi = (uint32_t*)d + nlo * 8;
    return GA_OK;
}

template <class FrP>
Fe<FrP> plonk_host_pow(Fe<FrP> a, uint64_t e) {
    Fe<FrP> r = fe_one<FrP>();
    while (e) {
        if (e & 1) r = mul(r, a);
        a = sqr(a);
        e >>= 1;
    }
    return r;
}

template <class FrP>
Fe<FrP> plonk_root_of_unity(int logn) {
    Fe<FrP> w = fe_const<FrP>(FrP::ROOT);
    for (int k = 0; k < FrP::ADICITY - logn; k++) w = sqr(w);
    return w;
}

// Circuit-constant part of the quotient, pinned in HBM (the "huge memory footprint" precomputation prove.go:1030-1034 decides
// against on a CPU): the evaluations of Ql, Qr, Qm, Qo, S1, S2, S3 and every Qcp on each of the rho cosets, and 1/(x-1) on
// each coset -- (7 + k + 1) * rho * n * 32 B, 4.3 GB at n = 2^22.  Per proof only L, R, O, Z, Qk and the BSB22 commitment
// polynomials are transformed: 6 + 4*6 + 1 transforms instead of 12 + 4*12 + 1.
struct PlonkFixed {
    Ctx* ctx = nullptr;
    Domain *d0 = nullptr, *d1 = nullptr;
    uint32_t nb_bsb = 0;
    int nslots = 0;
    int slot[PLONK_NB_FIXED + 2 * PLONK_MAX_BSB];   // -1: per-proof polynomial; else index into evals
    uint32_t* evals = nullptr;     // [rho][nslots][n] fr
    uint32_t* inv_xm1 = nullptr;   // [rho][n] fr
};
inline bool plonk_is_fixed(int p) {
    if (p >= PLONK_NB_FIXED) return ((p - PLONK_NB_FIXED) & 1) == 0;               // Qcp_i fixed, Pi2_i per proof
    return p == PX_QL || p == PX_QR || p == PX_QM || p == PX_QO || p == PX_S1 || p == PX_S2 || p == PX_S3;
}

// mode 0: everything per call; mode 1: build `fx` from the fixed polynomials of A (no quotient); mode 2: quotient with `fx`
template <class FrP>
int plonk_quotient(Domain* d0, Domain* d1, const PlonkQuotientArgs& A, void* h_out, int mode = 0, PlonkFixed* fx = nullptr) {
    typedef Fe<FrP> F;
    Ctx* ctx = d0->ctx;
    const uint64_t n = d0->n, N = d1->n;
    const int logn = d0->logn;
    if (n < 2 || N < n || N % n != 0 || A.nb_bsb > (uint32_t)PLONK_MAX_BSB || d1->ctx != ctx) {
        set_error("plonk quotient: need n >= 2, |domain1| a multiple of |domain0|, at most %d BSB22 gates", PLONK_MAX_BSB);
        return GA_ERR_INVALID;
    }
    if (mode == 2 && (fx->nb_bsb != A.nb_bsb || fx->d0 != d0 || fx->d1 != d1)) {
        set_error("plonk quotient: the pinned key was built for %u BSB22 gates / other domains", fx->nb_bsb);
        return GA_ERR_INVALID;
    }
    const uint64_t rho = N / n;
    const int logrho = ilog2_u64(rho);
    const int np = PLONK_NB_FIXED + 2 * (int)A.nb_bsb;
    hipStream_t st = ctx->stream;
    uint32_t *canon, *work, *invb, *tmpb, *cres;
    GA_CHECK(ctx->scratch_get("plonk_canon", (size_t)np * n * 32, (void**)&canon));
    GA_CHECK(ctx->scratch_get("plonk_work", (size_t)np * n * 32, (void**)&work));
    GA_CHECK(ctx->scratch_get("plonk_inv", n * 32, (void**)&invb));
    GA_CHECK(ctx->scratch_get("plonk_tmp", n * 32, (void**)&tmpb));
    GA_CHECK(ctx->scratch_get("plonk_cres", N * 32, (void**)&cres));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    auto skip = [&](int p) { return mode == 1 ? !plonk_is_fixed(p) : (mode == 2 && plonk_is_fixed(p)); };
    // ---- canonical coefficients in bit-reversed order, once ---------------------------------------------------------
    for (int p = 0; p < np; p++) {
        if (skip(p)) continue;
        uint32_t* dst = canon + (size_t)p * n * 8;
        uint32_t* stage = work + (size_t)p * n * 8;
        const bool lag = (A.lagrange_mask >> p) & 1;
        {
            StageTimer tm(ctx, "plonk_h2d");
            GA_HIP_CHECK(hipMemcpyAsync(lag ? dst : stage, A.polys[p], n * 32, A.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        }
        if (lag) {
            // Lagrange regular -> canonical bit-reversed: inverse DIF with the 1/n (p.ToCanonical, prove.go:1036)
            GA_CHECK(ntt_run<FrP>(d0, dst, /*inverse=*/true, /*dit=*/false, scale_none(), scale_const(d0->ninv)));
        } else {
            StageTimer tm(ctx, "plonk_bitrev");
            hipLaunchKernelGGL((fr_bitrev_copy_kernel<FrP>), dim3(blocks), dim3(256), 0, st, dst, stage, n, logn);
            GA_KERNEL_CHECK();
        }
    }
    // ---- constants ------------------------------------------------------------------------------------------------------
    auto ld = [](const void* p, int k = 0) {
        F f;
        memcpy(f.l, (const char*)p + 32 * k, 32);
        return f;
    };
    const F g = fe_const<FrP>(FrP::GEN);
    const F w0 = plonk_root_of_unity<FrP>(logn), w1 = plonk_root_of_unity<FrP>(d1->logn);
    F ninv = fe_one<FrP>();
    {
        F half = inv(add(fe_one<FrP>(), fe_one<FrP>()));
        for (int k = 0; k < logn; k++) ninv = mul(ninv, half);
    }
    PlonkConsts K;
    auto put = [](uint32_t* dst, const F& v) { memcpy(dst, v.l, 32); };
    memset(&K, 0, sizeof(K));
    if (mode != 1) {
        put(K.alpha, ld(A.alpha));
        put(K.beta, ld(A.beta));
        put(K.gamma, ld(A.gamma));
    }
    put(K.cs, g);              // prove.go:891-893
    put(K.css, sqr(g));
    put(K.omega, w0);
    PlonkPtrs P;
    memset(&P, 0, sizeof(P));
    P.nb_bsb = (int)A.nb_bsb;
    F coset = fe_one<FrP>();
    for (uint64_t i = 0; i < rho; i++) {
        for (int p = 0; p < np; p++)
            P.p[p] = (mode == 2 && plonk_is_fixed(p)) ? fx->evals + ((size_t)(i * fx->nslots + fx->slot[p])) * n * 8 : work + (size_t)p * n * 8;
        coset = mul(coset, i == 0 ? g : w1);                       // shifters, prove.go:936-941,998
        const F cexp = sub(plonk_host_pow<FrP>(coset, n), fe_one<FrP>());   // (coset^n - 1), prove.go:999-1000
        if (mode != 1)
        for (int k = 0; k < 2; k++) {
            put(K.bl[k], mul(ld(A.bl, k), cexp));
            put(K.br[k], mul(ld(A.br, k), cexp));
            put(K.bo[k], mul(ld(A.bo, k), cexp));
        }
        if (mode != 1)
            for (int k = 0; k < 3; k++) put(K.bz[k], mul(ld(A.bz, k), cexp));
        put(K.lone, mul(cexp, ninv));
        put(K.zh_inv, inv(cexp));
        {   // 32 c mod r of the constants the lazy constraint kernel multiplies by (five modular doublings each, host)
            auto sh5 = [&](const uint32_t* c) {
                F v = ld(c);
                for (int k = 0; k < 5; k++) v = add(v, v);
                return v;
            };
            const uint32_t* src[11] = {K.omega, K.bl[1], K.br[1], K.bo[1], K.bz[2], K.beta, K.cs, K.css, K.lone, K.alpha, K.zh_inv};
            for (int q = 0; q < 11; q++) put(K.sh[q], sh5(src[q]));   // (index = PSH_*)
        }
        // evaluations on coset*H: forward DIT with the coset powers fused into the first pass (prove.go:1033-1058)
        // (round 3: the coset lives in the twiddle table -- ntt_coset_table, built once per domain and coset and kept -- instead of
        // a scaling of the input by coset^i: two products per element and transform less)
        uint32_t *x_lo, *x_hi, *s_lo = nullptr, *s_hi = nullptr;
        const uint32_t* coset_tw = nullptr;
        const bool fold = ctx->tun.ntt_coset_fold != 0;
        if (fold) GA_CHECK(ntt_coset_table<FrP>(d0, coset, &coset_tw));
        else GA_CHECK(plonk_pow_tables<FrP>(ctx, "plonk_scale_tab", coset, fe_one<FrP>(), n, d0->lazy, &s_lo, &s_hi));
        GA_CHECK(plonk_pow_tables<FrP>(ctx, "plonk_x_tab", w0, coset, n, false, &x_lo, &x_hi));
        for (int p = 0; p < np; p++) {
            if (skip(p)) continue;
            uint32_t* dst = mode == 1 ? fx->evals + ((size_t)(i * fx->nslots + fx->slot[p])) * n * 8 : work + (size_t)p * n * 8;
            if (fold) GA_CHECK(ntt_run<FrP>(d0, dst, /*inverse=*/false, /*dit=*/true, scale_none(), scale_none(), canon + (size_t)p * n * 8, coset_tw));
            else GA_CHECK(ntt_run<FrP>(d0, dst, /*inverse=*/false, /*dit=*/true, scale_pow(s_lo, s_hi, /*bitrev=*/true), scale_none(), canon + (size_t)p * n * 8));
        }
        uint32_t* inv_i = mode == 0 ? invb : fx->inv_xm1 + (size_t)i * n * 8;
        if (mode != 2) {
            StageTimer tm(ctx, "plonk_batch_inverse");
            hipLaunchKernelGGL((plonk_x_minus_one_kernel<FrP>), dim3(blocks), dim3(256), 0, st, inv_i, x_lo, x_hi, NTT_POW_LO_BITS, n);
            const uint64_t threads = n < 65536 ? (n + 63) / 64 : n / 64;   // >= 64 elements per thread amortise the inversion
            const unsigned ib = (unsigned)((threads + 63) / 64);
            hipLaunchKernelGGL((fr_batch_inverse_kernel<FrP>), dim3(ib), dim3(64), 0, st, inv_i, tmpb, n);
            GA_KERNEL_CHECK();
        }
        if (mode != 1) {
            StageTimer tm(ctx, "plonk_constraints");
            uint64_t block = 0;   // bitrev_N(rho*j + i) = bitrev_rho(i)*n + bitrev_n(j)
            for (int b = 0; b < logrho; b++) block |= ((i >> b) & 1) << (logrho - 1 - b);
            if constexpr (Radix<FrP>::NL * Radix<FrP>::L - FrP::BITS >= 7)   // (BN254: lazy representation; BLS12-381's 255-bit Fr leaves 6 spare bits: packed)
                hipLaunchKernelGGL((plonk_constraints29_kernel<FrP>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, P, K, x_lo, x_hi,
                                   NTT_POW_LO_BITS, inv_i, cres + block * n * 8, n, logn);
            else
                hipLaunchKernelGGL((plonk_constraints_kernel<FrP>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, P, K, x_lo, x_hi,
                                   NTT_POW_LO_BITS, inv_i, cres + block * n * 8, n, logn);
            GA_KERNEL_CHECK();
        }
    }
    if (mode == 1) {
        GA_HIP_CHECK(hipStreamSynchronize(st));
        return GA_OK;
    }
    // ---- a.ToCanonical(bigDomain).ToRegular() from LagrangeCoset/BitReverse (prove.go:1319): inverse DIT on the coset ----
    GA_CHECK(ntt_fft<FrP>(d1, cres, GA_FFT_INVERSE, GA_DIT, 1));
    {
        StageTimer tm(ctx, "plonk_d2h");
        GA_HIP_CHECK(hipMemcpyAsync(h_out, cres, N * 32, A.on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    }
    GA_HIP_CHECK(hipStreamSynchronize(st));
    return GA_OK;
}

template <class FrP>
int plonk_fixed_create(Domain* d0, Domain* d1, const PlonkQuotientArgs& A, PlonkFixed** out) {
    const uint64_t n = d0->n, rho = d1->n / (d0->n ? d0->n : 1);
    if (n < 2 || d1->n % n != 0 || A.nb_bsb > (uint32_t)PLONK_MAX_BSB || d1->ctx != d0->ctx) {
        set_error("plonk key: need n >= 2, |domain1| a multiple of |domain0|, at most %d BSB22 gates", PLONK_MAX_BSB);
        return GA_ERR_INVALID;
    }
    PlonkFixed* fx = new PlonkFixed();
    fx->ctx = d0->ctx;
    fx->d0 = d0;
    fx->d1 = d1;
    fx->nb_bsb = A.nb_bsb;
    const int np = PLONK_NB_FIXED + 2 * (int)A.nb_bsb;
    for (int p = 0; p < PLONK_NB_FIXED + 2 * PLONK_MAX_BSB; p++) fx->slot[p] = (p < np && plonk_is_fixed(p)) ? fx->nslots++ : -1;
    const size_t ev = (size_t)rho * fx->nslots * n * 32, iv = (size_t)rho * n * 32;
    if (device_malloc((void**)&fx->evals, ev) != hipSuccess || device_malloc((void**)&fx->inv_xm1, iv) != hipSuccess) {
        set_error("plonk key: hipMalloc of %zu bytes failed", ev + iv);
        hipFree(fx->evals);
        delete fx;
        return GA_ERR_NOMEM;
    }
    int rc = plonk_quotient<FrP>(d0, d1, A, nullptr, 1, fx);
    if (rc != GA_OK) {
        hipFree(fx->evals);
        hipFree(fx->inv_xm1);
        delete fx;
        return rc;
    }
    *out = fx;
    return GA_OK;
}
inline void plonk_fixed_destroy(PlonkFixed* fx) {
    if (!fx) return;
    hipFree(fx->evals);
    hipFree(fx->inv_xm1);
    delete fx;
}

// ---- kzg.Open: p(z) and the coefficients of (p(X) - p(z)) / (X - z) -----------------------------------------------------
// gnark-crypto's dividePolyByXminusA is a sequential Horner recurrence q_{k-1} = p_k + z*q_k (call sites
// backend/plonk/bn254/prove.go:681,788,827).  Unrolled, q_k = z^-(k+1) * sum_{j>k} p_j z^j: an element-wise scaling by z^j, a
// suffix SUM (field additions only, three-phase scan) and an element-wise scaling by z^-(k+1); p(z) is the total sum.
template <class FrP>
__global__ void kzg_scale_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ lo,
                                 const uint32_t* __restrict__ hi, int lo_bits, uint64_t n, uint64_t shift) {
    // out[i] = in[i + shift] * pow(i + shift)  (pow from the two-level table), i < n
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_fe(out + i * 8, mul(load_fe<FrP>(in + (i + shift) * 8), plonk_point<FrP>(lo, hi, lo_bits, i + shift)));
}
constexpr int SUM_CHUNK = 256;
template <class FrP>
__global__ void fr_chunk_sum_kernel(const uint32_t* __restrict__ t, uint32_t* __restrict__ chunk_sum, uint64_t n) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = c * SUM_CHUNK;
    if (lo >= n) return;
    uint64_t hi = lo + SUM_CHUNK < n ? lo + SUM_CHUNK : n;
    Fe<FrP> acc = fe_zero<FrP>();
    for (uint64_t i = lo; i < hi; i++) acc = add(acc, load_fe<FrP>(t + i * 8));
    store_fe(chunk_sum + c * 8, acc);
}
template <class FrP>
__global__ void __launch_bounds__(256) fr_chunk_suffix_scan_kernel(uint32_t* __restrict__ chunk_sum, uint64_t nchunks) {
    // chunk_sum[c] <- sum of the chunks AFTER c (exclusive suffix sum); one block, segment per thread + LDS scan
    __shared__ uint32_t sh[256 * 8];
    const uint32_t tid = threadIdx.x;
    const uint64_t seg = (nchunks + 255) / 256;
    const uint64_t lo = (uint64_t)tid * seg < nchunks ? (uint64_t)tid * seg : nchunks;
    const uint64_t hi = lo + seg < nchunks ? lo + seg : nchunks;
    Fe<FrP> acc = fe_zero<FrP>();
    for (uint64_t c = lo; c < hi; c++) acc = add(acc, load_fe<FrP>(chunk_sum + c * 8));
    store_fe(sh + tid * 8, acc);
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {   // inclusive suffix scan over the 256 segment sums
        Fe<FrP> v = load_fe<FrP>(sh + tid * 8);
        if (tid + off < 256) v = add(v, load_fe<FrP>(sh + (tid + off) * 8));
        __syncthreads();
        store_fe(sh + tid * 8, v);
        __syncthreads();
    }
    acc = tid + 1 < 256 ? load_fe<FrP>(sh + (tid + 1) * 8) : fe_zero<FrP>();   // everything after this thread's segment
    for (uint64_t c = hi; c-- > lo;) {
        Fe<FrP> v = load_fe<FrP>(chunk_sum + c * 8);
        store_fe(chunk_sum + c * 8, acc);
        acc = add(acc, v);
    }
}
template <class FrP>
__global__ void fr_chunk_suffix_apply_kernel(uint32_t* __restrict__ t, const uint32_t* __restrict__ chunk_after, uint64_t n) {
    // t[i] <- sum_{j >= i} t[j]  (inclusive suffix sum)
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t lo = c * SUM_CHUNK;
    if (lo >= n) return;
    uint64_t hi = lo + SUM_CHUNK < n ? lo + SUM_CHUNK : n;
    Fe<FrP> acc = load_fe<FrP>(chunk_after + c * 8);
    for (uint64_t i = hi; i-- > lo;) {
        acc = add(acc, load_fe<FrP>(t + i * 8));
        store_fe(t + i * 8, acc);
    }
}

// d_quot (device, n elements: n-1 quotient coefficients then a zero) and *value (host, Montgomery) from d_poly (device, n coeffs)
template <class FrP>
int kzg_divide_by_linear(Ctx* ctx, const uint32_t* d_poly, uint64_t n, const void* z_mont, uint32_t* d_quot, void* value_out) {
    typedef Fe<FrP> F;
    hipStream_t st = ctx->stream;
    F z;
    memcpy(z.l, z_mont, 32);
    if (n == 0) {
        memset(value_out, 0, 32);
        return GA_OK;
    }
    uint32_t *t, *cs;
    const uint64_t nchunks = (n + SUM_CHUNK - 1) / SUM_CHUNK;
    GA_CHECK(ctx->scratch_get("kzg_t", n * 32, (void**)&t));
    GA_CHECK(ctx->scratch_get("kzg_chunks", nchunks * 32, (void**)&cs));
    const unsigned blocks = (unsigned)((n + 255) / 256), cb = (unsigned)((nchunks + 63) / 64);
    uint32_t *lo, *hi;
    StageTimer tm(ctx, "kzg_divide");
    GA_CHECK(plonk_pow_tables<FrP>(ctx, "plonk_x_tab", z, fe_one<FrP>(), n, false, &lo, &hi));
    hipLaunchKernelGGL((kzg_scale_kernel<FrP>), dim3(blocks), dim3(256), 0, st, d_poly, t, lo, hi, NTT_POW_LO_BITS, n, (uint64_t)0);
    hipLaunchKernelGGL((fr_chunk_sum_kernel<FrP>), dim3(cb), dim3(64), 0, st, t, cs, n);
    hipLaunchKernelGGL((fr_chunk_suffix_scan_kernel<FrP>), dim3(1), dim3(256), 0, st, cs, nchunks);
    hipLaunchKernelGGL((fr_chunk_suffix_apply_kernel<FrP>), dim3(cb), dim3(64), 0, st, t, cs, n);
    GA_KERNEL_CHECK();
    GA_HIP_CHECK(hipMemcpyAsync(value_out, t, 32, hipMemcpyDeviceToHost, st));   // S_0 = p(z)
    GA_HIP_CHECK(hipMemsetAsync(d_quot + (n - 1) * 8, 0, 32, st));
    if (n > 1) {
        if (is_zero(z)) {
            // q_k = p_{k+1}
            GA_HIP_CHECK(hipMemcpyAsync(d_quot, d_poly + 8, (n - 1) * 32, hipMemcpyDeviceToDevice, st));
        } else {
            // q_k = S_{k+1} * z^-(k+1): table of zinv^j, element i of the output reads index i+1
            GA_CHECK(plonk_pow_tables<FrP>(ctx, "plonk_scale_tab", inv(z), fe_one<FrP>(), n, false, &lo, &hi));
            hipLaunchKernelGGL((kzg_scale_kernel<FrP>), dim3((unsigned)((n - 1 + 255) / 256)), dim3(256), 0, st, t, d_quot, lo, hi,
                               NTT_POW_LO_BITS, n - 1, (uint64_t)1);
            GA_KERNEL_CHECK();
        }
    }
    GA_HIP_CHECK(hipStreamSynchronize(st));
    return GA_OK;
}

// out[i] = sum_k s_k * v_k[i]: the folding of kzg.BatchOpenSinglePoint and the linear combinations of the linearised polynomial
// (backend/plonk/bn254/prove.go:1352-1460) as one pass over the vectors
constexpr int LINCOMB_MAX = 16;
struct LinCombArgs {
    const uint32_t* v[LINCOMB_MAX];
    uint32_t s[LINCOMB_MAX][8];
    int k;
};
template <class FrP>
__global__ void fr_lincomb_kernel(LinCombArgs A, uint32_t* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<FrP> acc = fe_zero<FrP>();
    for (int k = 0; k < A.k; k++) acc = add(acc, mul_val(load_fe<FrP>(A.v[k] + i * 8), plonk_c<FrP>(A.s[k])));
    store_fe(out + i * 8, acc);
}
template <class FrP>
int fr_lincomb(Ctx* ctx, uint64_t n, int k, const void* const* vecs, const void* scalars, void* out, bool on_device) {
    if (k < 1 || k > LINCOMB_MAX) {
        set_error("linear combination of %d vectors: 1..%d supported per call", k, LINCOMB_MAX);
        return GA_ERR_INVALID;
    }
    if (n == 0) return GA_OK;
    hipStream_t st = ctx->stream;
    LinCombArgs A;
    memset(&A, 0, sizeof(A));
    A.k = k;
    uint32_t *stage = nullptr, *dout = (uint32_t*)out;
    if (!on_device) {
        GA_CHECK(ctx->scratch_get("lincomb_in", (size_t)(k + 1) * n * 32, (void**)&stage));
        dout = stage + (size_t)k * n * 8;
    }
    for (int j = 0; j < k; j++) {
        memcpy(A.s[j], (const char*)scalars + 32 * j, 32);
        if (on_device) {
            A.v[j] = (const uint32_t*)vecs[j];
        } else {
            GA_HIP_CHECK(hipMemcpyAsync(stage + (size_t)j * n * 8, vecs[j], n * 32, hipMemcpyHostToDevice, st));
            A.v[j] = stage + (size_t)j * n * 8;
        }
    }
    {
        StageTimer tm(ctx, "fr_lincomb");
        hipLaunchKernelGGL((fr_lincomb_kernel<FrP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, dout, n);
        GA_KERNEL_CHECK();
    }
    if (!on_device) GA_HIP_CHECK(hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, st));
    GA_HIP_CHECK(hipStreamSynchronize(st));
    return GA_OK;
}

// fr.BatchInvert on a vector (host or device memory), in place
template <class FrP>
int fr_batch_inverse(Ctx* ctx, void* v, uint64_t n, bool on_device) {
    if (n == 0) return GA_OK;
    uint32_t *buf, *tmp;
    GA_CHECK(ctx->scratch_get("plonk_inv", n * 32, (void**)&buf));
    GA_CHECK(ctx->scratch_get("plonk_tmp", n * 32, (void**)&tmp));
    hipStream_t st = ctx->stream;
    uint32_t* d = on_device ? (uint32_t*)v : buf;
    if (!on_device) GA_HIP_CHECK(hipMemcpyAsync(buf, v, n * 32, hipMemcpyHostToDevice, st));
    {
        StageTimer tm(ctx, "plonk_batch_inverse");
        const uint64_t threads = n < 65536 ? (n + 63) / 64 : n / 64;
        hipLaunchKernelGGL((fr_batch_inverse_kernel<FrP>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, d, tmp, n);
        GA_KERNEL_CHECK();
    }
    if (!on_device) GA_HIP_CHECK(hipMemcpyAsync(v, buf, n * 32, hipMemcpyDeviceToHost, st));
    GA_HIP_CHECK(hipStreamSynchronize(st));
    return GA_OK;
}

// entries of a device-resident permutation outside [0, 3n) would index the identity tables out of bounds
static __global__ void plonk_perm_check_kernel(const int64_t* __restrict__ perm, uint64_t n3, uint32_t* __restrict__ bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3 && (perm[i] < 0 || (uint64_t)perm[i] >= n3)) atomicAdd(bad, 1u);
}

// iop.BuildRatioCopyConstraint: Z in Lagrange form (regular layout) from L, R, O (Lagrange regular) and the permutation.
template <class FrP>
int plonk_build_z(Domain* d0, const void* L, const void* R, const void* O, const int64_t* perm, const void* beta, const void* gamma,
                  bool on_device, void* z_out) {
    typedef Fe<FrP> F;
    Ctx* ctx = d0->ctx;
    const uint64_t n = d0->n;
    hipStream_t st = ctx->stream;
    uint32_t *lro, *num, *den, *tmp, *z, *cp;
    int64_t* dperm;
    const uint64_t nchunks = (n + PROD_CHUNK - 1) / PROD_CHUNK;
    GA_CHECK(ctx->scratch_get("plonk_z_lro", 3 * n * 32, (void**)&lro));
    GA_CHECK(ctx->scratch_get("plonk_z_num", n * 32, (void**)&num));
    GA_CHECK(ctx->scratch_get("plonk_z_den", n * 32, (void**)&den));
    GA_CHECK(ctx->scratch_get("plonk_tmp", n * 32, (void**)&tmp));
    GA_CHECK(ctx->scratch_get("plonk_z_out", n * 32, (void**)&z));
    GA_CHECK(ctx->scratch_get("plonk_z_chunks", nchunks * 32, (void**)&cp));
    GA_CHECK(ctx->scratch_get("plonk_z_perm", 3 * n * 8, (void**)&dperm));
    const hipMemcpyKind kin = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const void* src[3] = {L, R, O};
    for (int k = 0; k < 3; k++) GA_HIP_CHECK(hipMemcpyAsync(lro + (size_t)k * n * 8, src[k], n * 32, kin, st));
    GA_HIP_CHECK(hipMemcpyAsync(dperm, perm, 3 * n * 8, kin, st));
    if (on_device) {   // a host permutation was range-checked by the entry point
        uint32_t* d_bad;
        GA_CHECK(ctx->scratch_get("plonk_perm_bad", 256, (void**)&d_bad));
        GA_HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, st));
        hipLaunchKernelGGL(plonk_perm_check_kernel, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, st, (const int64_t*)dperm, 3 * n, d_bad);
        GA_KERNEL_CHECK();
        uint32_t bad = 0;
        GA_HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
        GA_HIP_CHECK(hipStreamSynchronize(st));
        if (bad) {
            set_error("ga_plonk_build_z: %u entries of the device-resident permutation are outside [0, 3n)", bad);
            return GA_ERR_INVALID;
        }
    }
    PlonkConsts K;
    memset(&K, 0, sizeof(K));
    memcpy(K.beta, beta, 32);
    memcpy(K.gamma, gamma, 32);
    const F g = fe_const<FrP>(FrP::GEN);
    memcpy(K.cs, g.l, 32);
    F gg = sqr(g);
    memcpy(K.css, gg.l, 32);
    uint32_t *w_lo, *w_hi;
    GA_CHECK(plonk_pow_tables<FrP>(ctx, "plonk_x_tab", plonk_root_of_unity<FrP>(d0->logn), fe_one<FrP>(), n, false, &w_lo, &w_hi));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    {
        StageTimer tm(ctx, "plonk_z_terms");
        hipLaunchKernelGGL((plonk_ratio_terms_kernel<FrP>), dim3(blocks), dim3(256), 0, st, lro, lro + n * 8, lro + 2 * n * 8, dperm, w_lo,
                           w_hi, NTT_POW_LO_BITS, K, num, den, n);
        GA_KERNEL_CHECK();
    }
    {
        StageTimer tm(ctx, "plonk_batch_inverse");
        const uint64_t threads = n < 65536 ? (n + 63) / 64 : n / 64;
        hipLaunchKernelGGL((fr_batch_inverse_kernel<FrP>), dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, den, tmp, n);
        GA_KERNEL_CHECK();
    }
    {
        StageTimer tm(ctx, "plonk_z_prefix_product");
        const unsigned cb = (unsigned)((nchunks + 63) / 64);
        hipLaunchKernelGGL((fr_chunk_product_kernel<FrP>), dim3(cb), dim3(64), 0, st, num, den, cp, n);
        hipLaunchKernelGGL((fr_chunk_scan_kernel<FrP>), dim3(1), dim3(256), 0, st, cp, nchunks);
        hipLaunchKernelGGL((fr_chunk_apply_kernel<FrP>), dim3(cb), dim3(64), 0, st, num, den, cp, z, n);
        GA_KERNEL_CHECK();
    }
    GA_HIP_CHECK(hipMemcpyAsync(z_out, z, n * 32, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    GA_HIP_CHECK(hipStreamSynchronize(st));
    return GA_OK;
}

}  // namespace ga
