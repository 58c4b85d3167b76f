// Proving-key and proof wire formats, parsed ON THE DEVICE.
//
//   ProvingKey.WriteTo / WriteRawTo / ReadFrom     backend/groth16/bn254/marshal.go:231-373   (compressed / uncompressed points)
//   ProvingKey.WriteDump / ReadDump                 backend/groth16/bn254/marshal.go:378-539   (raw memory images of the slices)
//   Proof.ReadFrom                                  backend/groth16/bn254/marshal.go:62-86
//
// A 2^24-constraint key is 6-9 GiB and 84 M points.  gnark decodes it on the CPU (one square root per compressed point: minutes
// on all cores); here the file streams through two pinned staging buffers straight into HBM and the big-endian decoding, the
// Montgomery conversion, the on-curve check and the square roots of compressed points run in a kernel, one thread per point
// (~4e10 field products for a compressed 2^24 key: well under a second).  The dump format needs no arithmetic at all: the slices
// are gnark's memory images and go from the file to their place in HBM with no host copy in between.
//
// Byte layouts owned by gnark-crypto v0.21.0 (go.mod:10; NOT in /root/reference) and restated from its published code:
//   * curve.Encoder: uint64 / uint32 big-endian; a []G1Affine / []G2Affine as u32 BE length + points; a []bool through
//     binary.Write, i.e. one byte per entry and NO length prefix (which is why ReadFrom sizes InfinityA from nbWires first);
//     points as big-endian coordinates (G2: A1 | A0) with the flag bits of SURVEY Appendix A in the first byte.
//   * fft.Domain.WriteTo: Cardinality u64 BE, then CardinalityInv, Generator, GeneratorInv, FrMultiplicativeGen,
//     FrMultiplicativeGenInv as 32-byte big-endian canonical fr elements, then (versions with fft.WithoutPrecompute) one byte
//     withPrecompute.  The reader accepts both: it decodes [alpha]1 under either hypothesis and keeps the one that is a curve point.
//   * utils/unsafe: WriteMarker = the 8 bytes of uint64(0xdeadbeef) in native byte order; WriteSlice = u64 LE length + the raw
//     memory of the slice.
//   * pedersen.ProvingKey.WriteTo: Basis then BasisExpSigma, each a []G1Affine.
// The reference repository pins the point encodings (serialized verifying keys, bellman tuples: tests/test_oracle_fixtures.py);
// the key-level framing has no fixture in the reference tree, so the test-side big-integer restatement writes the same layouts
// from marshal.go and the tests compare bytes both ways (library-written == checker-written, and each reads the other's).
#pragma once
#include <sys/stat.h>
#include <errno.h>
#include <unistd.h>

#include "hostops.hip.h"

namespace ga {

// ---- field decoding helpers (host and device) -------------------------------------------------------------------------------
template <class P>
GA_HD Fe<P> fe_from_be_bytes(const uint8_t* b, uint8_t first_mask, bool* canonical) {
    Fe<P> x;
#pragma unroll
    for (int w = 0; w < P::N; w++) {
        const uint8_t* q = b + 4 * (P::N - 1 - w);
        uint32_t b0 = q[0];
        if (w == P::N - 1) b0 &= first_mask;
        x.l[w] = (b0 << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    *canonical = !geq_mod<P>(x.l);
    return to_mont(x);
}
template <class P>
GA_HD void fe_to_be_bytes(const Fe<P>& x_mont, uint8_t* out) {
    Fe<P> x = from_mont(x_mont);
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint32_t v = x.l[P::N - 1 - i];
        out[4 * i] = (uint8_t)(v >> 24);
        out[4 * i + 1] = (uint8_t)(v >> 16);
        out[4 * i + 2] = (uint8_t)(v >> 8);
        out[4 * i + 3] = (uint8_t)v;
    }
}
// y > (p-1)/2 on the canonical value (fp.Element.LexicographicallyLargest)
template <class P>
GA_HD bool fe_lex_largest(const Fe<P>& y_mont) {
    Fe<P> y = from_mont(y_mont);
    for (int i = P::N - 1; i >= 0; i--) {
        uint32_t half = (P::MOD[i] >> 1) | (i + 1 < P::N ? (P::MOD[i + 1] << 31) : 0);
        if (y.l[i] > half) return true;
        if (y.l[i] < half) return false;
    }
    return false;
}
template <class P>
GA_HD bool fe_lex_largest(const Fe2<P>& y) { return is_zero(y.c1) ? fe_lex_largest(y.c0) : fe_lex_largest(y.c1); }

// a^((p+1)/4): the square root when p = 3 mod 4 (both base fields); false if a is not a square
template <class P>
GA_HD_CALL bool fe_sqrt(const Fe<P>& a, Fe<P>* out) {
    uint32_t e[P::N];
    uint32_t carry = 1;   // (p >> 2) + 1 == (p + 1) / 4 because p = 3 mod 4
    for (int i = 0; i < P::N; i++) {
        uint32_t w = (P::MOD[i] >> 2) | (i + 1 < P::N ? (P::MOD[i + 1] << 30) : 0);
        uint32_t s = w + carry;
        carry = s < w ? 1 : 0;
        e[i] = s;
    }
    Fe<P> r = pow_words(a, e, P::N);
    *out = r;
    return eq(sqr(r), a);
}
// square root in Fp[u]/(u^2+1) (complex method): x0^2 = (a0 +- sqrt(a0^2 + a1^2))/2, x1 = a1/(2 x0)
template <class P>
GA_HD_CALL bool fe_sqrt(const Fe2<P>& a, Fe2<P>* out) {
    if (is_zero(a.c1)) {
        Fe<P> s;
        if (fe_sqrt(a.c0, &s)) {
            *out = {s, fe_zero<P>()};
            return true;
        }
        if (!fe_sqrt(neg(a.c0), &s)) return false;
        *out = {fe_zero<P>(), s};
        return true;
    }
    Fe<P> alpha;
    if (!fe_sqrt(add(sqr(a.c0), sqr(a.c1)), &alpha)) return false;
    const Fe<P> inv2 = inv(dbl(fe_one<P>()));
    Fe<P> x0;
    if (!fe_sqrt(mul(add(a.c0, alpha), inv2), &x0) && !fe_sqrt(mul(sub(a.c0, alpha), inv2), &x0)) return false;
    Fe<P> x1 = mul(a.c1, inv(dbl(x0)));
    Fe2<P> r{x0, x1};
    *out = r;
    return eq(sqr(r), a);
}

template <class C, int G> struct CurveB;
template <class C> struct CurveB<C, GA_G1> {
    typedef typename C::FpP P;
    GA_HD static Fe<P> get() { return fe_const<P>(P::B1); }
};
template <class C> struct CurveB<C, GA_G2> {
    typedef typename C::FpP P;
    GA_HD static Fe2<P> get() { return {fe_const<P>(P::B2_0), fe_const<P>(P::B2_1)}; }
};

// flag bits of the first byte.  Returns the number of bytes of the encoding (0 = malformed).
struct PointFlags {
    bool compressed, infinity, largest;
    uint8_t mask;   // bits of the first byte that belong to the coordinate
};
template <class C>
GA_HD bool point_flags(uint8_t b0, PointFlags* f) {
    if (C::ID == GA_BN254) {
        const uint8_t m = b0 >> 6;
        f->mask = 0x3F;
        f->compressed = m >= 2;
        f->infinity = m == 1;      // 0b01: infinity, compressed and uncompressed alike (the length comes from the context)
        f->largest = m == 3;
        return true;
    }
    const uint8_t m = b0 >> 5;     // ZCash: bit7 compressed, bit6 infinity, bit5 largest
    f->mask = 0x1F;
    f->compressed = (m & 4) != 0;
    f->infinity = (m & 2) != 0;
    f->largest = (m & 1) != 0;
    return !(f->infinity && f->largest) && !(!f->compressed && f->largest);
}

template <class P>
GA_HD bool coord_from_bytes(const uint8_t* b, uint8_t mask, Fe<P>* out) {
    bool ok;
    *out = fe_from_be_bytes<P>(b, mask, &ok);
    return ok;
}
template <class P>
GA_HD bool coord_from_bytes(const uint8_t* b, uint8_t mask, Fe2<P>* out) {   // A1 | A0
    bool ok1, ok0;
    out->c1 = fe_from_be_bytes<P>(b, mask, &ok1);
    out->c0 = fe_from_be_bytes<P>(b + 4 * P::N, 0xFF, &ok0);
    return ok0 && ok1;
}
template <class P>
GA_HD void coord_to_bytes(const Fe<P>& v, uint8_t* out) { fe_to_be_bytes(v, out); }
template <class P>
GA_HD void coord_to_bytes(const Fe2<P>& v, uint8_t* out) {
    fe_to_be_bytes(v.c1, out);
    fe_to_be_bytes(v.c0, out + 4 * P::N);
}

// one point from its wire encoding; `compressed` is the mode of the stream it belongs to
template <class C, int G>
GA_HD bool point_decode(const uint8_t* b, bool compressed, Affine<typename GroupField<C, G>::F>* out) {
    typedef typename GroupField<C, G>::F F;
    constexpr int CB = sizeof(F);   // bytes of one coordinate
    PointFlags f;
    if (!point_flags<C>(b[0], &f)) return false;
    if (f.infinity) {
        out->x = FieldTraits<F>::zero();
        out->y = FieldTraits<F>::zero();
        return true;
    }
    if (f.compressed != compressed) return false;
    if (!compressed) {   // an uncompressed point of all-zero bytes is accepted as infinity as well (lenient: the flagged form is 0x40 | 0...)
        bool all_zero = true;
        for (int i = 0; i < 2 * CB; i++) all_zero = all_zero && b[i] == 0;
        if (all_zero) {
            out->x = FieldTraits<F>::zero();
            out->y = FieldTraits<F>::zero();
            return true;
        }
    }
    F x, y;
    if (!coord_from_bytes(b, f.mask, &x)) return false;
    const F rhs = add(mul(sqr(x), x), CurveB<C, G>::get());
    if (compressed) {
        if (!fe_sqrt(rhs, &y)) return false;             // x is not the abscissa of a curve point
        if (fe_lex_largest(y) != f.largest) y = neg(y);
    } else {
        if (!coord_from_bytes(b + CB, 0xFF, &y)) return false;
        if (!eq(sqr(y), rhs)) return false;              // on-curve check (subgroup membership is NOT checked: UnsafeReadFrom semantics)
    }
    out->x = x;
    out->y = y;
    return true;
}
template <class C, int G>
GA_HD void point_encode(const Affine<typename GroupField<C, G>::F>& a, bool compressed, uint8_t* out) {
    typedef typename GroupField<C, G>::F F;
    constexpr int CB = sizeof(F);
    const int len = compressed ? CB : 2 * CB;
    for (int i = 0; i < len; i++) out[i] = 0;
    if (is_inf(a)) {
        out[0] = (C::ID == GA_BN254) ? 0x40 : (compressed ? 0xC0 : 0x40);
        return;
    }
    coord_to_bytes(a.x, out);
    if (!compressed) {
        coord_to_bytes(a.y, out + CB);
        return;
    }
    const bool largest = fe_lex_largest(a.y);
    if (C::ID == GA_BN254) out[0] |= largest ? 0xC0 : 0x80;
    else out[0] |= 0x80 | (largest ? 0x20 : 0);
}

template <class C, int G>
__global__ void key_decode_kernel(const uint8_t* __restrict__ in, uint64_t count, int compressed, void* __restrict__ out,
                                  uint32_t* __restrict__ bad) {
    typedef typename GroupField<C, G>::F F;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const size_t enc = compressed ? sizeof(F) : 2 * sizeof(F);
    Affine<F> p;
    if (!point_decode<C, G>(in + i * enc, compressed != 0, &p)) {
        atomicAdd(bad, 1u);
        p.x = FieldTraits<F>::zero();
        p.y = FieldTraits<F>::zero();
    }
    store_pod(reinterpret_cast<char*>(out) + i * sizeof(Affine<F>), p);
}
template <class C, int G>
__global__ void key_encode_kernel(const void* __restrict__ in, uint64_t count, int compressed, uint8_t* __restrict__ out) {
    typedef typename GroupField<C, G>::F F;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const size_t enc = compressed ? sizeof(F) : 2 * sizeof(F);
    Affine<F> p = load_pod<Affine<F>>(reinterpret_cast<const char*>(in) + i * sizeof(Affine<F>));
    point_encode<C, G>(p, compressed != 0, out + i * enc);
}

// ---- byte sources / sinks ----------------------------------------------------------------------------------------------------
struct ByteSource {
    const uint8_t* mem = nullptr;   // memory source ...
    size_t mem_len = 0, mem_pos = 0;
    int fd = -1;                    // ... or a file descriptor, read sequentially
    std::vector<uint8_t> ahead;     // bytes read from the fd but not consumed yet (peek)
    uint64_t consumed = 0;
    int fill(size_t n) {            // make `ahead` hold at least n bytes (fd mode)
        while (ahead.size() < n) {
            uint8_t tmp[4096];
            ssize_t r = ::read(fd, tmp, sizeof tmp);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) {
                set_error("key file: unexpected end of input (wanted %zu more bytes)", n - ahead.size());
                return GA_ERR_INVALID;
            }
            ahead.insert(ahead.end(), tmp, tmp + r);
        }
        return GA_OK;
    }
    // bytes left in the input, when that can be known (memory image; regular file): a length word read from the input is checked
    // against it BEFORE anything is allocated for it (the reader's fuzz test under ASAN: a damaged 32-bit length asked for 72 GB)
    uint64_t remaining() const {
        if (fd < 0) return mem_len - mem_pos;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return UINT64_MAX;
        const off_t at = lseek(fd, 0, SEEK_CUR);
        if (at < 0 || sb.st_size < at) return UINT64_MAX;
        return (uint64_t)(sb.st_size - at) + ahead.size();
    }
    int expect(uint64_t count, uint64_t bytes_each, const char* what) const {
        const uint64_t left = remaining();
        if (left != UINT64_MAX && (bytes_each == 0 || count > left / bytes_each)) {
            set_error("key file: unexpected end of input (%s announces %llu elements of at least %llu bytes, %llu bytes are left)", what, (unsigned long long)count,
                      (unsigned long long)bytes_each, (unsigned long long)left);
            return GA_ERR_INVALID;
        }
        return GA_OK;
    }
    int peek(uint8_t* dst, size_t n) {
        if (fd < 0) {
            if (mem_len - mem_pos < n) {
                set_error("key image: unexpected end of input");
                return GA_ERR_INVALID;
            }
            memcpy(dst, mem + mem_pos, n);
            return GA_OK;
        }
        GA_CHECK(fill(n));
        memcpy(dst, ahead.data(), n);
        return GA_OK;
    }
    int read(void* dstv, size_t n) {
        uint8_t* dst = static_cast<uint8_t*>(dstv);
        consumed += n;
        if (fd < 0) {
            if (mem_len - mem_pos < n) {
                set_error("key image: unexpected end of input");
                return GA_ERR_INVALID;
            }
            memcpy(dst, mem + mem_pos, n);
            mem_pos += n;
            return GA_OK;
        }
        size_t got = 0;
        if (!ahead.empty()) {
            got = ahead.size() < n ? ahead.size() : n;
            memcpy(dst, ahead.data(), got);
            ahead.erase(ahead.begin(), ahead.begin() + got);
        }
        while (got < n) {
            ssize_t r = ::read(fd, dst + got, n - got);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) {
                set_error("key file: unexpected end of input (wanted %zu more bytes)", n - got);
                return GA_ERR_INVALID;
            }
            got += (size_t)r;
        }
        return GA_OK;
    }
    int u64be(uint64_t* v) {
        uint8_t b[8];
        GA_CHECK(read(b, 8));
        *v = 0;
        for (int i = 0; i < 8; i++) *v = (*v << 8) | b[i];
        return GA_OK;
    }
    int u32be(uint32_t* v) {
        uint8_t b[4];
        GA_CHECK(read(b, 4));
        *v = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
        return GA_OK;
    }
    int u64le(uint64_t* v) {
        uint8_t b[8];
        GA_CHECK(read(b, 8));
        *v = 0;
        for (int i = 7; i >= 0; i--) *v = (*v << 8) | b[i];
        return GA_OK;
    }
};

struct ByteSink {
    int fd = -1;
    std::vector<uint8_t>* mem = nullptr;
    uint64_t written = 0;
    int write(const void* srcv, size_t n) {
        const uint8_t* src = static_cast<const uint8_t*>(srcv);
        written += n;
        if (mem) {
            mem->insert(mem->end(), src, src + n);
            return GA_OK;
        }
        size_t done = 0;
        while (done < n) {
            ssize_t r = ::write(fd, src + done, n - done);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) {
                set_error("key file: write failed (errno %d)", errno);
                return GA_ERR_INVALID;
            }
            done += (size_t)r;
        }
        return GA_OK;
    }
    int u64be(uint64_t v) {
        uint8_t b[8];
        for (int i = 7; i >= 0; i--, v >>= 8) b[i] = (uint8_t)v;
        return write(b, 8);
    }
    int u32be(uint32_t v) {
        uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
        return write(b, 4);
    }
    int u64le(uint64_t v) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++, v >>= 8) b[i] = (uint8_t)v;
        return write(b, 8);
    }
};

// two page-locked staging buffers: the file read of chunk k+1 overlaps the H2D copy (and the decode kernel) of chunk k
struct Staging {
    static constexpr size_t BYTES = 32u << 20;
    uint8_t* h[2] = {nullptr, nullptr};
    uint8_t* d_bytes = nullptr;    // device copy of an encoded chunk
    void* d_points = nullptr;      // decoded chunk
    uint32_t* d_bad = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int init() {
        for (int k = 0; k < 2; k++) {
            GA_HIP_CHECK(hipHostMalloc((void**)&h[k], BYTES, 0));
            GA_HIP_CHECK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
        GA_HIP_CHECK(device_malloc((void**)&d_bytes, BYTES));
        GA_HIP_CHECK(device_malloc(&d_points, 2 * BYTES));   // a compressed point doubles when decoded
        GA_HIP_CHECK(device_malloc((void**)&d_bad, 256));
        return GA_OK;
    }
    ~Staging() {
        for (int k = 0; k < 2; k++) {
            if (h[k]) hipHostFree(h[k]);
            if (ev[k]) hipEventDestroy(ev[k]);
        }
        hipFree(d_bytes);
        hipFree(d_points);
        hipFree(d_bad);
    }
};

}  // namespace ga
