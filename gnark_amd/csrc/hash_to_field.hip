// fr.Hash of gnark-crypto (hash-to-field with expand_message_xmd over SHA-256) -- host code.
//
// Used by the BSB22 commitment flow around the device MSMs: the solver hint turns a commitment point into a field element
// with hash_to_field.New("bsb22-commitment") (backend/groth16/bn254/prove.go:57-58,88-98) and the PoK fold challenge is
// fr.Hash(m, "G16-BSB22", 1) where m is the concatenation of the commitment WIRE VALUES sol.W[CommitmentIndex].Marshal() -- 32-byte
// big-endian fr elements, NOT the commitment points (prove.go:118-127).  gnark-crypto is not in the reference tree; the element-level code
// shape is visible in internal/smallfields/tinyfield/element.go:456-481 (L = 16 + Bytes pseudo-random bytes per element,
// big-endian, reduced mod r) and expand_message_xmd is RFC 9380 section 5.3.1, pinned by the 16 vectors of
// std/hash/expand/expand_test.go:52-140 (tests/test_oracle_fixtures.py).  A Go host would keep calling gnark-crypto; this
// file exists so that a C/C++ host of the library can run the whole commitment flow.
#include "common.hip.h"

namespace ga {

struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len = 0;
    size_t fill = 0;
    Sha256() {
        static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        memcpy(h, iv, sizeof(iv));
    }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
            0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
            0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
            0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
            0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
            0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
            0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            uint32_t ch = (e & f) ^ (~e & g);
            uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) {
            size_t k = std::min(n, (size_t)64 - fill);
            memcpy(buf + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void final(uint8_t out[32]) {
        uint64_t bits = len * 8;
        uint8_t pad = 0x80;
        update(&pad, 1);
        uint8_t z = 0;
        while (fill != 56) update(&z, 1);
        uint8_t lb[8];
        for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(lb, 8);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i];
        }
    }
};

// RFC 9380 5.3.1 expand_message_xmd with H = SHA-256 (b = 32 bytes, block 64 bytes)
static int expand_message_xmd(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, size_t n, std::vector<uint8_t>* out) {
    const size_t ell = (n + 31) / 32;
    if (ell > 255 || n > 65535 || dst_len > 255) {
        set_error("expand_message_xmd: invalid lengths (out %zu, dst %zu)", n, dst_len);
        return GA_ERR_INVALID;
    }
    const uint8_t dlen = (uint8_t)dst_len;
    uint8_t b0[32], bi[32];
    {
        Sha256 h;
        uint8_t zpad[64] = {0};
        h.update(zpad, 64);
        h.update(msg, msg_len);
        uint8_t lib[3] = {(uint8_t)(n >> 8), (uint8_t)n, 0};
        h.update(lib, 3);
        h.update(dst, dst_len);
        h.update(&dlen, 1);
        h.final(b0);
    }
    out->resize(ell * 32);
    for (size_t i = 1; i <= ell; i++) {
        Sha256 h;
        uint8_t x[32];
        for (int k = 0; k < 32; k++) x[k] = i == 1 ? b0[k] : (uint8_t)(b0[k] ^ bi[k]);
        h.update(x, 32);
        uint8_t ib = (uint8_t)i;
        h.update(&ib, 1);
        h.update(dst, dst_len);
        h.update(&dlen, 1);
        h.final(bi);
        memcpy(out->data() + (i - 1) * 32, bi, 32);
    }
    out->resize(n);
    return GA_OK;
}

template <class C>
static int hash_to_field(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint32_t count, void* out) {
    typedef typename C::FrP FrP;
    const size_t L = 16 + (FrP::BITS + 7) / 8;   // 48 for both scalar fields
    std::vector<uint8_t> bytes;
    GA_CHECK(expand_message_xmd(msg, msg_len, dst, dst_len, (size_t)count * L, &bytes));
    Fe<FrP> c256 = fe_zero<FrP>();
    c256.l[0] = 256;
    c256 = to_mont(c256);
    for (uint32_t e = 0; e < count; e++) {
        Fe<FrP> acc = fe_zero<FrP>();   // big-endian Horner, everything in Montgomery form
        for (size_t k = 0; k < L; k++) {
            Fe<FrP> d = fe_zero<FrP>();
            d.l[0] = bytes[e * L + k];
            acc = add(mul(acc, c256), to_mont(d));
        }
        memcpy((char*)out + (size_t)e * sizeof(acc.l), acc.l, sizeof(acc.l));
    }
    return GA_OK;
}

}  // namespace ga

using namespace ga;

extern "C" int ga_hash_to_field(int curve, const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint32_t count, void* out) try {
    GA_ABI_ENTRY();
    if ((!msg && msg_len) || (!dst && dst_len) || !out) {
        set_error("ga_hash_to_field: null argument");
        return GA_ERR_INVALID;
    }
    GA_DISPATCH_CURVE(curve, return hash_to_field<C>(msg, msg_len, dst, dst_len, count, out));
    return GA_OK;
} GA_ABI_CATCH

// exposed for the fixture test of the expand_test.go vectors
extern "C" int ga_expand_message_xmd(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, size_t n, uint8_t* out) try {
    GA_ABI_ENTRY();
    std::vector<uint8_t> v;
    GA_CHECK(expand_message_xmd(msg, msg_len, dst, dst_len, n, &v));
    memcpy(out, v.data(), n);
    return GA_OK;
} GA_ABI_CATCH
