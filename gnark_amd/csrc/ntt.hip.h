// Radix-2 NTT / iNTT over the scalar field Fr for gfx950, with gnark-crypto's ordering conventions.
//
// Replaces fft.Domain.FFT / FFTInverse as used by computeH (backend/groth16/bn254/prove.go:346-389) and the
// ICICLE ntt.Ntt calls (backend/accelerated/icicle/groth16/bn254/icicle.go:1425,1428,1474):
//   DIF: natural in -> bit-reversed out;  DIT: bit-reversed in -> natural out;
//   inverse includes 1/n;  OnCoset: forward pre-scales coefficient i by g^i, inverse post-scales by g^-i.
//
// Design (HBM-bound shape: 64 B algorithmic traffic per element per transform):
//   * log2(n) butterfly stages are split into passes; one pass = one kernel = one HBM round trip.  A pass owns
//     K consecutive stages [s_lo, s_lo+K); a workgroup stages a tile of 2^K strided rows x 2^lc contiguous
//     columns (x 2^g independent groups) in LDS, runs the K stages out of LDS and writes the tile back.
//   * global traffic is 16 B per lane, lanes walking consecutive 16-byte halves of consecutive elements
//     (2^lc * 32 B contiguous per row), so every pass streams at full width even for the strided top stages.
//   * LDS layout is two planes of uint4 (low / high 16 bytes of each element): a butterfly's ds_read_b128 /
//     ds_write_b128 are conflict-free for unit-stride lanes.
//   * coset scaling and the 1/n factor are fused into the first / last pass (two small power tables:
//     g^k for k < 2^12 and g^(k*2^12)), so OnCoset transforms cost no extra HBM pass.
#pragma once
#include "common.hip.h"
#include "field29.hip.h"
#include <stdlib.h>
#include <array>
#include <functional>
#include <mutex>
#include <utility>
#include <vector>

namespace ga {

constexpr int NTT_LG_TILE = 10;               // 1024 elements = 32 KiB of LDS per workgroup
constexpr int NTT_THREADS = 256;
constexpr int NTT_POW_LO_BITS = 12;

struct NttScale {
    int mode;                 // 0 none, 1 constant only, 2 power tables (lo*hi), constant folded into lo
    int bitrev;               // index the power by bitrev(i) instead of i
    const uint32_t* lo;       // [2^lo_bits] g^k (* const)
    const uint32_t* hi;       // [n >> lo_bits] g^(k << lo_bits)
    int lo_bits;
    uint32_t cst[8];          // mode 1: constant factor (Montgomery)
};

struct NttPass {
    int s_lo, K, lc;
};

// global element index of local slot l of tile `tile`
__device__ __forceinline__ uint64_t ntt_gidx(uint32_t l, uint64_t tile, int lg_tile, int s_lo, int K, int lc) {
    uint32_t col = l & ((1u << lc) - 1);
    uint32_t mid = (l >> lc) & ((1u << K) - 1);
    uint32_t grp = l >> (lc + K);
    uint64_t u = (tile << (lg_tile - lc - K)) | grp;
    uint64_t lo_chunk = u & ((1ull << (s_lo - lc)) - 1);
    uint64_t hi = u >> (s_lo - lc);
    return (hi << (s_lo + K)) | ((uint64_t)mid << s_lo) | (lo_chunk << lc) | col;
}

__device__ __forceinline__ uint64_t bitrev64(uint64_t i, int logn) {
    uint32_t lo = __brev((uint32_t)i), hi = __brev((uint32_t)(i >> 32));
    uint64_t r = ((uint64_t)lo << 32) | hi;
    return r >> (64 - logn);
}

template <class FrP>
__device__ __forceinline__ Fe<FrP> ntt_scale_factor(const NttScale& sc, uint64_t i, int logn) {
    if (sc.mode == 1) {
        Fe<FrP> f;
#pragma unroll
        for (int k = 0; k < 8; k++) f.l[k] = sc.cst[k];
        return f;
    }
    uint64_t idx = sc.bitrev ? bitrev64(i, logn) : i;
    Fe<FrP> a = load_fe_plain<FrP>(sc.lo + (idx & ((1ull << sc.lo_bits) - 1)) * 8);
    uint64_t h = idx >> sc.lo_bits;
    if (h == 0 && logn <= sc.lo_bits) return a;
    Fe<FrP> b = load_fe_plain<FrP>(sc.hi + h * 8);
    return mul(a, b);
}

// ---- one pass over stages [s_lo, s_lo+K) in the lazy unpacked representation (field29.hip.h) ----------------------------------
// Elements stay in gnark's Montgomery form x*2^256 but as 9 limbs of 29 bits, unreduced; twiddles and scale factors are
// tables of hat(w) = w*2^261, so  f29_mul(v, hat(w)) = v*w  is again in gnark's form -- no domain change at load/store.
// BOTH directions run on Cooley-Tukey butterflies (multiply, then add / subtract): a butterfly adds a fresh product (< 3p) to
// its partner, so values grow by < 3p per stage (< 33p after 10 stages) and need no reduction inside a pass; results are made
// canonical once, at the store.
//   * bit-reversed -> natural (gnark's DIT): the classic form, stages ascending, the butterfly at index bit s of element i takes
//     w^(x << (logn-1-s)), x = i mod 2^s -- its POSITION inside the block;
//   * natural -> bit-reversed (gnark's DIF): stages descending, the butterfly takes the twiddle of its BLOCK,
//     w^(bitrev_{logn-1-s}(i >> (s+1)) << s) (the "natural in, bit-reversed out, twiddles in bit-reversed order" form, as in
//     Longa-Naehrig's NTT^CT_{no->bo}) -- same function and output order as the Gentleman-Sande form round 2 used, without its
//     doubling sums (Barrett steps, 32p subtractions: 1728 -> 1434 instructions per radix-4 block).
// Twiddle tables are laid out so that BOTH forms read them densely (round 3; with one natural-order table w^e the
// twiddle-heavy stages of either form gathered 32-byte entries from distinct cache lines and the passes ran 25-30 % below
// their arithmetic, profiles/README.md):
//   TS ("stacked", n entries):      TS[2^s + x] = w^(x << (logn-1-s)), x < 2^s      -- DIT, stage s reads a contiguous block
//   TB ("bit-reversed", n/2):       TB[j] = w^bitrev_{logn-1}(j)                      -- natural -> bit-reversed: stage s reads the
//                                    prefix j < 2^(logn-1-s), block number = index
// For a quad on index bits (s, s+1) the three twiddles are TS[2^s + x], TS[2^(s+1) + x], TS[2^(s+1) + 2^s + x] (DIT) and
// TB[u], TB[2u], TB[2u+1] with u = i >> (s+2) (natural -> bit-reversed: the last two adjacent, all three shared by every lane
// with the same upper index bits).
// LDS tile: [NL][tile] words, one plane per limb.  Unit-stride lanes hit consecutive banks; the quads of the low stages do not
// (index bits lb < 5 leave a half-wave only 8 distinct banks: SQ_LDS_BANK_CONFLICT = 41 % of the LDS-array cycles).  Measured
// in round 3: an XOR swizzle of the slot index that makes every butterfly position conflict-free changes nothing (computeH
// 15.96 vs 15.99 ms, profiles/README.md) -- the LDS array is busy 17 % of the time, the conflicts hide behind the multiplies
// -- so the plain layout stays.
template <class FrP>
struct LdsTile29 {
    uint32_t* base;
    static constexpr int NL = Radix<FrP>::NL, STRIDE = 1 << NTT_LG_TILE;
    __device__ __forceinline__ F29<FrP> get(uint32_t l) const {
        F29<FrP> r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = base[i * STRIDE + l];
        return r;
    }
    __device__ __forceinline__ void put(uint32_t l, const F29<FrP>& v) const {
#pragma unroll
        for (int i = 0; i < NL; i++) base[i * STRIDE + l] = v.l[i];
    }
};

template <class FrP>
__device__ __forceinline__ F29<FrP> ntt_scale_factor29(const NttScale& sc, uint64_t i, int logn) {
    if (sc.mode == 1) {
        Fe<FrP> f;
#pragma unroll
        for (int k = 0; k < 8; k++) f.l[k] = sc.cst[k];
        return f29_unpack(f);
    }
    uint64_t idx = sc.bitrev ? bitrev64(i, logn) : i;
    F29<FrP> a = f29_unpack(load_fe_plain<FrP>(sc.lo + (idx & ((1ull << sc.lo_bits) - 1)) * 8));
    uint64_t h = idx >> sc.lo_bits;
    if (h == 0 && logn <= sc.lo_bits) return a;
    return f29_mul(a, f29_unpack(load_fe_plain<FrP>(sc.hi + h * 8)));   // hat(a)*hat(b)/R' = hat(a*b)
}

// entry k of a twiddle table as limbs (entries are stored unpacked: nine 29-bit limbs in 48 bytes, so that the pass kernels do
// no bit-field extraction on three twiddles per quad)
constexpr int NTT_TW_WORDS = 12;
template <class FrP>
__device__ __forceinline__ F29<FrP> ntt_twiddle29(const uint32_t* __restrict__ tw, uint64_t k) {
    static_assert(Radix<FrP>::NL == 9, "twiddle entries hold nine limbs");
    const uint32_t* p = tw + k * NTT_TW_WORDS;
    const u32x4 lo = reinterpret_cast<const u32x4*>(p)[0], hi = reinterpret_cast<const u32x4*>(p)[1];
    F29<FrP> r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
    r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    r.l[8] = p[8];
    return r;
}

// Two stages per LDS round trip (radix-4 in registers): a thread keeps a quad (index bits t and t+1) in registers across two
// stages, so there is one LDS read + write and one index computation per TWO stages, and the intermediate values are not
// carry-normalised: limb-wise sums stay below 2^32 and a 2^31-limb multiplicand still keeps the product columns below 2^64.
//
// Round 4: which rounds need the workgroup at all.  With a full tile (1024 slots, 256 threads) thread q of a round on slot bits
// (lb, lb+1) owns the quad whose other eight slot bits are q -- so as long as the round's bits lie below bit 8, the four waves
// work on the slots whose top two bits are their own number, round after round: the exchange between such rounds is traffic
// among the lanes of ONE wave, which the LDS serves in issue order -- no s_barrier, only a compiler fence (wave_lds_sync).
// The stage list is cut so that everything below slot bit 8 comes in such wave-local rounds (an odd count puts its single stage
// at the top of that part, bit 7) and bits 8, 9 form one radix-4 round of their own behind the only __syncthreads of the pass
// (two when the tile is also loaded / stored through LDS).  The first and the last round move their quads straight between
// registers and HBM when eight consecutive lanes still cover a 256-byte row (lb >= 3): bit-reversed -> natural upper passes
// (7 stages: bits 3..9) run  HBM -> (3,4) ~ (5,6) ~ (7) | (8,9) -> HBM  with ONE barrier (|) and three LDS exchanges instead
// of five barriers and five exchanges; the 10-stage pass  LDS <- HBM ~ (0,1) ~ (2,3) ~ (4,5) ~ (6,7) | (8,9) -> HBM.
// flags: bit 0 = the table may hold unit twiddles (not a coset table), bit 1 = wave-local rounds, bit 2 = direct first load /
// last store (GA_NTT_WAVE_LOCAL / GA_NTT_DIRECT, A/B knobs of round 4).
constexpr int NTT_F_UNIT = 1, NTT_F_WAVE_LOCAL = 2, NTT_F_DIRECT = 4;

#ifndef GA_NTT_MIN_WAVES   // (waves per SIMD the register allocation is sized for; 3 measured in round 6: tools/exp/r06_g.sh)
#define GA_NTT_MIN_WAVES 4
#endif
template <class FrP, bool DIT_>
__global__ void __launch_bounds__(NTT_THREADS, GA_NTT_MIN_WAVES)
ntt_pass29r4_kernel(uint32_t* data, const uint32_t* src, const uint32_t* __restrict__ tw, int logn, int lg_tile, int s_lo, int K,
                    int lc, NttScale pre, NttScale post, int flags) {
    // NTT_F_UNIT clear: the table is a coset table (every entry carries its stage's power of the coset generator): no twiddle is 1
    static_assert(FrP::N == 8, "Fr is 4x64-bit limbs on both curves");
    static_assert(NTT_THREADS == 256 && NTT_LG_TILE == 10, "the wave-local slot mapping below is written for 4 waves x 256 slots");
    GA_REQUIRE_WAVE64();   // a wave owns a 256-slot quarter of the tile between the barrier-free rounds
    typedef F29<FrP> E;
    __shared__ uint32_t lds[Radix<FrP>::NL << NTT_LG_TILE];
    LdsTile29<FrP> T{lds};
    const uint32_t tile_elems = 1u << lg_tile;
    const uint64_t tile = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const bool unit_ok = (flags & NTT_F_UNIT) != 0;
    const bool wl = lg_tile == NTT_LG_TILE && (flags & NTT_F_WAVE_LOCAL) != 0;
    const bool direct = lg_tile == NTT_LG_TILE && (flags & NTT_F_DIRECT) != 0;
    auto twid = [&](uint64_t k) { return ntt_twiddle29<FrP>(tw, k); };
    auto gidx = [&](uint32_t l) { return ntt_gidx(l, tile, lg_tile, s_lo, K, lc); };
    auto ld = [&](uint64_t i) {
        E v = f29_unpack(load_fe<FrP>(src + i * 8));   // src == data for an in-place pass
        if (pre.mode != 0) v = f29_mul(v, ntt_scale_factor29<FrP>(pre, i, logn));
        return v;
    };
    auto st = [&](uint64_t i, E v) {                   // v normalized
        if (post.mode != 0) v = f29_mul(v, ntt_scale_factor29<FrP>(post, i, logn));
        store_fe(data + i * 8, f29_pack_canonical(f29_reduce_3p(v)));
    };
    // slot of a thread's j-th element in the load / store phases: the wave's own quarter of the tile when rounds are wave-local
    auto own_slot = [&](uint32_t j) { return wl ? ((wv << 8) | (j << 6) | lane) : tid + j * NTT_THREADS; };
    // between two phases that exchange through LDS: inside the wave when both are wave-local, through the workgroup otherwise
    auto sync = [&](bool both_local) {
        if (both_local) wave_lds_sync();
        else __syncthreads();
    };

    // the rounds in ascending stage order, five bits each: first stage t | (stages - 1) << 4
    uint32_t plan = 0;
    int nr = 0;
    {
        const int below8 = 8 - lc < 0 ? 0 : 8 - lc;
        const int nlow = wl ? (K < below8 ? K : below8) : K;
        int t = 0;
        auto push = [&](int r) {
            plan |= (uint32_t)(t | ((r - 1) << 4)) << (5 * nr);
            nr++;
            t += r;
        };
        while (t + 2 <= nlow) push(2);
        if (t < nlow) push(1);
        while (t + 2 <= K) push(2);
        if (t < K) push(1);
    }
    auto round_of = [&](int k, int& t, int& r) {     // k-th round in execution order
        const uint32_t code = (plan >> (5 * (DIT_ ? k : nr - 1 - k))) & 31u;
        t = (int)(code & 15u);
        r = (int)(code >> 4) + 1;
    };
    auto direct_ok = [&](int t, int r) { return direct && r == 2 && lc + t >= 3; };
    int t0, r0, t9, r9;
    round_of(0, t0, r0);
    round_of(nr - 1, t9, r9);
    // (a pass that scales its input or output keeps the LDS phases, where the scale factors are applied: the register-resident
    // rounds stay lean -- with the factor tables in them the kernel needed 158 VGPRs, three waves per SIMD)
    const bool dload = nr > 0 && pre.mode == 0 && direct_ok(t0, r0), dstore = nr > 0 && post.mode == 0 && direct_ok(t9, r9);

    bool lds_live = false, prev_local = false;       // a previous phase left its results in LDS / it was wave-local
    if (!dload) {
        for (uint32_t j = 0; j * NTT_THREADS < tile_elems; j++) {
            const uint32_t l = own_slot(j);
            if (l < tile_elems) T.put(l, ld(gidx(l)));
        }
        lds_live = true;
        prev_local = wl;
    }

    for (int k = 0; k < nr; k++) {
        int t, r;
        round_of(k, t, r);
        const int lb = lc + t;
        const int s = s_lo + t;
        const bool local = wl && lb + r <= 8;
        const bool from_hbm = dload && k == 0, to_hbm = dstore && k == nr - 1;
        if (lds_live && !from_hbm) sync(prev_local && local);
        if (r == 2) {
        for (uint32_t q = tid; q < tile_elems / 4; q += NTT_THREADS) {
            const uint32_t l00 = ((q >> lb) << (lb + 2)) | (q & ((1u << lb) - 1));
            // (DIT: first stage on bit lb, second on bit lb+1; natural -> bit-reversed: the other way round, which is the same
            // code with the two middle elements of the quad exchanged)
            const uint32_t l01 = l00 | ((DIT_ ? 1u : 2u) << lb), l10 = l00 | ((DIT_ ? 2u : 1u) << lb), l11 = l00 | (3u << lb);
            const uint64_t i0 = gidx(l00);
            // (global neighbours of the quad: slot bit lb is element-index bit s)
            const uint64_t d01 = (DIT_ ? 1ull : 2ull) << s, d10 = (DIT_ ? 2ull : 1ull) << s, d11 = 3ull << s;
            uint64_t k1, k2, k3;      // table entries: first stage (both butterflies), second stage (.0) and (.1)
            bool unit;                // first-stage twiddle and the (.0) second-stage twiddle are 1
            if (DIT_) {
                const uint64_t x = i0 & ((1ull << s) - 1);
                k1 = (1ull << s) + x;
                k2 = (2ull << s) + x;
                k3 = k2 + (1ull << s);
                unit = unit_ok && x == 0;
            } else {
                const uint64_t u = i0 >> (s + 2);
                k1 = u;
                k2 = 2 * u;
                k3 = 2 * u + 1;
                unit = u == 0;
            }
            E a00, a01, a10, a11;
            if (from_hbm) {
                const uint32_t* p0 = src + i0 * 8;
                a00 = f29_unpack(load_fe<FrP>(p0));
                a01 = f29_unpack(load_fe<FrP>(p0 + d01 * 8));
                a10 = f29_unpack(load_fe<FrP>(p0 + d10 * 8));
                a11 = f29_unpack(load_fe<FrP>(p0 + d11 * 8));
            } else {
                a00 = T.get(l00);
                a01 = T.get(l01);
                a10 = T.get(l10);
                a11 = T.get(l11);
            }
            // first stage: (a00, a01) and (a10, a11); products (or, for w = 1, the operand itself) are brought below 3p
            E m0, m1;
            if (!unit) {
                const E w1 = twid(k1);
                m0 = f29_mul(a01, w1);
                m1 = f29_mul(a11, w1);
            } else {
                m0 = f29_reduce_3p(a01);
                m1 = f29_reduce_3p(a11);
            }
            E b00 = f29_add_raw(a00, m0), b01 = f29_sub_raw<4>(a00, m0);
            E b10 = f29_add_raw(a10, m1), b11 = f29_sub_raw<4>(a10, m1);
            // second stage: (b00, b10) with twiddle k2, (b01, b11) with twiddle k3; multiplicand limbs < 2^31
            E m2;
            if (!unit) {
                m2 = f29_mul(b10, twid(k2));
            } else {
                f29_normalize(b10);
                m2 = f29_reduce_3p(b10);
            }
            E m3 = f29_mul(b11, twid(k3));
            E c00 = f29_add_raw(b00, m2), c10 = f29_sub_raw<4>(b00, m2);
            E c01 = f29_add_raw(b01, m3), c11 = f29_sub_raw<4>(b01, m3);
            f29_normalize(c00);
            f29_normalize(c01);
            f29_normalize(c10);
            f29_normalize(c11);
            if (to_hbm) {
                uint32_t* p0 = data + gidx(l00) * 8;
                store_fe(p0, f29_pack_canonical(f29_reduce_3p(c00)));
                store_fe(p0 + d01 * 8, f29_pack_canonical(f29_reduce_3p(c01)));
                store_fe(p0 + d10 * 8, f29_pack_canonical(f29_reduce_3p(c10)));
                store_fe(p0 + d11 * 8, f29_pack_canonical(f29_reduce_3p(c11)));
            } else {
                T.put(l00, c00);
                T.put(l01, c01);
                T.put(l10, c10);
                T.put(l11, c11);
            }
        }
        } else {   // a single stage (odd stage count)
        for (uint32_t j = 0; j * NTT_THREADS < tile_elems / 2; j++) {
            // wave-local: the wave's number goes to slot bits 8, 9 (lb <= 7), the iteration to the bit below
            const uint32_t q = wl ? (lane | (j << 6) | (wv << 7)) : tid + j * NTT_THREADS;
            if (q >= tile_elems / 2) continue;
            uint32_t l0 = ((q >> lb) << (lb + 1)) | (q & ((1u << lb) - 1));
            uint32_t l1 = l0 | (1u << lb);
            uint64_t i0 = gidx(l0);
            uint64_t kk;
            bool unit;
            if (DIT_) {
                const uint64_t x = i0 & ((1ull << s) - 1);
                kk = (1ull << s) + x;
                unit = unit_ok && x == 0;
            } else {
                kk = i0 >> (s + 1);
                unit = kk == 0;
            }
            E x = T.get(l0), y = T.get(l1);
            E m = !unit ? f29_mul(y, twid(kk)) : f29_reduce_3p(y);
            T.put(l0, f29_add(x, m));
            T.put(l1, f29_sub<4>(x, m));
        }
        }
        lds_live = !to_hbm;
        prev_local = local;
    }

    if (lds_live) {
        sync(prev_local && wl);
        for (uint32_t j = 0; j * NTT_THREADS < tile_elems; j++) {
            const uint32_t l = own_slot(j);
            if (l < tile_elems) st(gidx(l), T.get(l));
        }
    }
}

// One twiddle table (unpacked hat(w^e) entries, NTT_TW_WORDS words each) from the table of w^(2^k):
//   stacked = 1:  out[2^s + x] = w^(x << (logn-1-s)), x < 2^s, s < logn   (entry 0 unused)      -- n entries
//   stacked = 0:  out[j] = w^bitrev_{logn-1}(j), j < n/2                                          -- n/2 entries
//   stage_cst != null (stacked only): entry [2^s + x] is additionally multiplied by stage_cst[s] -- the coset table TSg with
//   stage_cst[s] = g^(2^(logn-1-s)): a bit-reversed -> natural transform over it evaluates on the coset g*<w> without scaling its
//   input (A(g w^k) = A_even((g w^k)^2) + g w^k A_odd((g w^k)^2): the twiddle of the stage that merges halves of size 2^s is
//   g^(n/2^(s+1)) w^(x n/2^(s+1)), and the halves are transforms on the coset of g^2 -- recursively)
template <class FrP>
__global__ void ntt_twiddle_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ pow2, uint64_t count, int logn, int stacked,
                                   const uint32_t* __restrict__ stage_cst) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    uint64_t e = 0;
    int stage = 0;
    if (stacked) {
        if (k > 0) {
            int s = 63 - __clzll((long long)k);
            stage = s;
            e = (k - (1ull << s)) << (logn - 1 - s);
        }
    } else {
        e = logn > 1 ? bitrev64(k, logn - 1) : 0;
    }
    Fe<FrP> r = fe_one<FrP>();
    for (int b = 0; b + 1 < logn; b++)
        if ((e >> b) & 1) r = mul(r, load_fe<FrP>(pow2 + b * 8));
    if (stage_cst && k > 0) r = mul(r, load_fe<FrP>(stage_cst + stage * 8));
    const F29<FrP> u = f29_unpack(f29_hat_packed(r));   // the pass kernels multiply by hat(w) = w * 2^261
    for (int i = 0; i < NTT_TW_WORDS; i++) out[k * NTT_TW_WORDS + i] = i < Radix<FrP>::NL ? u.l[i] : 0u;
}

// a[i] = (a[i]*b[i] - c[i]*cn) * den      (prove.go:377-383; cn = n and den = 1/((g^n - 1) n^2) when a, b, c are the UNSCALED chains
// n * FFT_coset(iFFT(.)) of ntt_compute_h_chain: (n a)(n b) - n (n c) = n^2 (a b - c))
template <class FrP>
__global__ void ntt_pointwise_h_kernel(uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                       const uint32_t* __restrict__ c, uint64_t n, NttScale den, NttScale cn) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<FrP> d, e;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        d.l[k] = den.cst[k];
        e.l[k] = cn.cst[k];
    }
    Fe<FrP> x = load_fe<FrP>(a + i * 8), y = load_fe<FrP>(b + i * 8), z = load_fe<FrP>(c + i * 8);
    store_fe(a + i * 8, mul(sub(mul(x, y), mul(z, e)), d));
}

// ---- host side -----------------------------------------------------------------------------------

struct Domain {
    Ctx* ctx = nullptr;
    int curve = 0;
    uint64_t n = 0;
    int logn = 0;
    // twiddle tables of the pass kernels (layouts: ntt_twiddle_kernel): stacked per stage for bit-reversed -> natural transforms,
    // bit-reversed for natural -> bit-reversed ones; forward (w) and inverse (1/w), plus the forward coset table: 4n entries of
    // 48 bytes in all (3 GiB at 2^24)
    uint32_t* d_ts = nullptr;       // TS, w
    uint32_t* d_ts_inv = nullptr;   // TS, 1/w
    uint32_t* d_tb = nullptr;       // TB, w
    uint32_t* d_tb_inv = nullptr;   // TB, 1/w
    uint32_t* d_ts_g = nullptr;     // TS of w with the coset generator folded in (ntt_twiddle_kernel): computeH's coset FFTs scale nothing
    uint32_t* d_pow2 = nullptr;     // w^(2^k), k < 32, then (1/w)^(2^k): what ntt_twiddle_kernel builds tables from
    // coset tables built on demand (ntt_coset_table): (shift, table) pairs -- d_ts_g is the first, PLONK's rho cosets follow
    std::mutex coset_mu;
    std::vector<std::pair<std::array<uint32_t, 8>, uint32_t*>> coset_tabs;
    uint32_t den_n2[8];             // (g^n - 1)^-1 / n^2 and n (Montgomery): the point-wise step on the unscaled chains of computeH
    uint32_t n_mont[8];
    // coset power tables
    uint32_t* d_g_lo = nullptr;     // g^k
    uint32_t* d_g_hi = nullptr;
    uint32_t* d_gi_lo = nullptr;    // g^-k / n
    uint32_t* d_gi_hi = nullptr;
    static constexpr bool lazy = true;   // twiddle / scale tables are kept in the hat domain (w * 2^261) for the lazy passes
    uint32_t ninv[8];               // 1/n (Montgomery; hat-packed when lazy)
    uint32_t den[8];                // (g^n - 1)^-1 (Montgomery), prove.go:370-373
    std::vector<NttPass> passes;    // ascending stage order
};

// Passes of a size-2^logn transform, ascending stage order.  The first pass owns the contiguous low stages (up to a whole tile),
// every further pass K strided stages over rows of 2^lc contiguous elements, lc = min(3, NTT_LG_TILE - K) (256-byte rows, 128-byte
// ones for K = 8).  Among the splits with the fewest passes the one with the fewest ODD stage counts wins (an odd count ends
// with a radix-2 round: a whole LDS round trip and barrier for one stage), then the most balanced one, then the larger first pass.
// GA_NTT_PLAN="8,8,8" (read when a domain is created) forces a split -- experiments only.
inline std::vector<NttPass> ntt_plan(int logn) {
    std::vector<NttPass> v;
    if (logn == 0) return v;
    const int K0MAX = NTT_LG_TILE, LCMAX = 3, KUP = NTT_LG_TILE - 2;   // upper passes: K <= 8
    auto make = [&](const std::vector<int>& ks) {
        std::vector<NttPass> out;
        int s = 0;
        for (size_t p = 0; p < ks.size(); p++) {
            int lc = 0;
            if (p > 0) lc = NTT_LG_TILE - ks[p] < LCMAX ? NTT_LG_TILE - ks[p] : LCMAX;
            out.push_back({s, ks[p], lc});
            s += ks[p];
        }
        return out;
    };
    if (const char* e = getenv("GA_NTT_PLAN")) {
        std::vector<int> ks;
        int sum = 0;
        bool ok = true;
        for (const char* q = e; *q;) {
            char* end = nullptr;
            long k = strtol(q, &end, 10);
            if (end == q) break;
            ok = ok && k >= 1 && k <= (ks.empty() ? K0MAX : KUP);
            ks.push_back((int)k);
            sum += (int)k;
            q = *end == ',' ? end + 1 : end;
        }
        if (ok && sum == logn && !ks.empty()) return make(ks);
    }
    if (logn <= K0MAX) return make({logn});
    const int kup = 7;   // largest stage count of an upper pass in the default plan (8 = 128-byte rows: measured +-0.1 ms, GA_NTT_PLAN)
    const int np = 1 + (logn - K0MAX + kup - 1) / kup;
    std::vector<int> best, cur(np, 0);
    int best_odd = 1 << 30, best_min = 0;
    // enumerate k[0] <= K0MAX, k[p] <= kup, sum = logn (np <= 5 for every supported size)
    std::function<void(int, int)> rec = [&](int p, int left) {
        if (p == np - 1) {
            if (left < 1 || left > kup) return;
            cur[p] = left;
            int odd = 0, mn = 1 << 30;
            for (int k : cur) {
                odd += k & 1;
                mn = k < mn ? k : mn;
            }
            if (odd < best_odd || (odd == best_odd && (mn > best_min || (mn == best_min && cur[0] > best[0])))) {
                best_odd = odd;
                best_min = mn;
                best = cur;
            }
            return;
        }
        const int cap = p == 0 ? K0MAX : kup;
        for (int k = 1; k <= cap && k < left; k++) {
            cur[p] = k;
            rec(p + 1, left - k);
        }
    };
    rec(0, logn);
    return make(best);
}

template <class FrP>
int ntt_run(Domain* d, uint32_t* d_data, bool inverse, bool dit, const NttScale& pre_first, const NttScale& post_last,
            const uint32_t* d_src = nullptr, const uint32_t* coset_table = nullptr) {
    // coset_table != nullptr: a stacked table with the coset generator folded in (forward, bit-reversed -> natural only)
    // d_src != nullptr: out-of-place transform (the first pass reads d_src, every pass writes d_data; d_src is left untouched)
    Ctx* ctx = d->ctx;
    const uint32_t* tw = coset_table ? coset_table : dit ? (inverse ? d->d_ts_inv : d->d_ts) : (inverse ? d->d_tb_inv : d->d_tb);
    const int flags = (coset_table ? 0 : NTT_F_UNIT) | (ctx->tun.ntt_wave_local ? NTT_F_WAVE_LOCAL : 0) | (ctx->tun.ntt_direct ? NTT_F_DIRECT : 0);
    NttScale none;
    memset(&none, 0, sizeof(none));
    int np = (int)d->passes.size();
    for (int p = 0; p < np; p++) {
        const NttPass& ps = dit ? d->passes[p] : d->passes[np - 1 - p];
        int lg_tile = d->logn < NTT_LG_TILE ? d->logn : NTT_LG_TILE;
        uint64_t tiles = d->n >> lg_tile;
        const NttScale& pre = (p == 0) ? pre_first : none;
        const NttScale& post = (p == np - 1) ? post_last : none;
        const uint32_t* src = (p == 0 && d_src) ? d_src : d_data;
        StageTimer st(ctx, dit ? "ntt_pass_dit" : "ntt_pass_dif");
        if (dit)
            hipLaunchKernelGGL((ntt_pass29r4_kernel<FrP, true>), dim3((unsigned)tiles), dim3(NTT_THREADS), 0, ctx->work_stream(),
                               d_data, src, tw, d->logn, lg_tile, ps.s_lo, ps.K, ps.lc, pre, post, flags);
        else
            hipLaunchKernelGGL((ntt_pass29r4_kernel<FrP, false>), dim3((unsigned)tiles), dim3(NTT_THREADS), 0, ctx->work_stream(),
                               d_data, src, tw, d->logn, lg_tile, ps.s_lo, ps.K, ps.lc, pre, post, flags);
        GA_KERNEL_CHECK();
    }
    return GA_OK;
}

inline NttScale scale_none() {
    NttScale s;
    memset(&s, 0, sizeof(s));
    return s;
}
inline NttScale scale_const(const uint32_t* c) {
    NttScale s = scale_none();
    s.mode = 1;
    memcpy(s.cst, c, 32);
    return s;
}
inline NttScale scale_pow(const uint32_t* lo, const uint32_t* hi, bool bitrev) {
    NttScale s = scale_none();
    s.mode = 2;
    s.bitrev = bitrev ? 1 : 0;
    s.lo = lo;
    s.hi = hi;
    s.lo_bits = NTT_POW_LO_BITS;
    return s;
}

// gnark semantics wrapper: see header comment
template <class FrP>
int ntt_fft(Domain* d, uint32_t* d_data, int direction, int decimation, int on_coset) {
    if (d->logn == 0) return GA_OK;
    bool inverse = direction == GA_FFT_INVERSE;
    bool dit = decimation == GA_DIT;
    NttScale pre = scale_none(), post = scale_none();
    if (!inverse) {
        if (on_coset && dit && d->d_ts_g && d->ctx->tun.ntt_coset_fold)   // the coset lives in the twiddle table: no scaling pass
            return ntt_run<FrP>(d, d_data, false, true, pre, post, nullptr, d->d_ts_g);
        if (on_coset) pre = scale_pow(d->d_g_lo, d->d_g_hi, /*bitrev=*/dit);
    } else {
        if (on_coset) post = scale_pow(d->d_gi_lo, d->d_gi_hi, /*bitrev=*/!dit);
        else post = scale_const(d->ninv);
    }
    return ntt_run<FrP>(d, d_data, inverse, dit, pre, post);
}

// computeH in two pieces, so that a multi-GPU proof can run the three chains on different devices:
//   chain(v)        : v <- FFT_coset(iFFT(v))       per input vector a, b, c   (prove.go:362-368)
//   combine(a,b,c)  : a <- iFFT_coset((a*b - c) * den)   bit-reversed h        (prove.go:377-386)
// Buffers hold exactly n elements each (already zero-padded).
template <class FrP>
int ntt_compute_h_chain(Domain* d, uint32_t* d_v) {
    if (d->logn == 0) return GA_OK;   // n = 1: iFFT and coset FFT are identities
    // n * FFT_coset(iFFT(v)) without a single scaling multiplication: the inverse transform leaves its 1/n out (the point-wise step
    // of ntt_compute_h_combine absorbs it: the chains are only ever consumed there) and the forward one runs over the coset table
    GA_CHECK(ntt_run<FrP>(d, d_v, /*inverse=*/true, /*dit=*/false, scale_none(), scale_none()));
    if (!d->ctx->tun.ntt_coset_fold)   // (A/B: the round-2 form, coset powers and 1/n applied to the input of the forward transform; x n to match)
        return ntt_run<FrP>(d, d_v, /*inverse=*/false, /*dit=*/true, scale_pow(d->d_g_lo, d->d_g_hi, true), scale_none());
    return ntt_run<FrP>(d, d_v, /*inverse=*/false, /*dit=*/true, scale_none(), scale_none(), nullptr, d->d_ts_g);
}

template <class FrP>
int ntt_compute_h_combine(Domain* d, uint32_t* d_a, const uint32_t* d_b, const uint32_t* d_c) {
    Ctx* ctx = d->ctx;
    {
        StageTimer st(ctx, "h_pointwise");
        NttScale den = scale_const(d->den_n2), cn = scale_const(d->n_mont);   // a, b, c are n * (the reference's vectors)
        unsigned blocks = d->logn == 0 ? 1u : (unsigned)((d->n + 255) / 256);
        hipLaunchKernelGGL((ntt_pointwise_h_kernel<FrP>), dim3(blocks), dim3(d->logn == 0 ? 64 : 256), 0, ctx->work_stream(), d_a, d_b, d_c, d->n, den, cn);
        GA_KERNEL_CHECK();
    }
    if (d->logn == 0) return GA_OK;
    return ntt_fft<FrP>(d, d_a, GA_FFT_INVERSE, GA_DIF, 1);
}

// computeH on device buffers of exactly n elements each (already zero-padded); result in d_a (bit-reversed)
template <class FrP>
int ntt_compute_h(Domain* d, uint32_t* d_a, uint32_t* d_b, uint32_t* d_c) {
    uint32_t* v[3] = {d_a, d_b, d_c};
    for (int k = 0; k < 3; k++) GA_CHECK(ntt_compute_h_chain<FrP>(d, v[k]));
    return ntt_compute_h_combine<FrP>(d, d_a, d_b, d_c);
}

// The stacked forward table of the domain with the coset `shift` folded in (layout and derivation: ntt_twiddle_kernel): a
// bit-reversed -> natural transform over it evaluates on shift*<w> with no scaling pass.  Built once per (domain, shift) and kept:
// computeH's g, the rho cosets g*w_{rho n}^i of the PLONK quotient (n * 48 bytes each).
template <class FrP>
int ntt_coset_table(Domain* d, const Fe<FrP>& shift, const uint32_t** out) {
    typedef Fe<FrP> F;
    std::array<uint32_t, 8> key;
    memcpy(key.data(), shift.l, 32);
    std::lock_guard<std::mutex> g(d->coset_mu);
    for (auto& kv : d->coset_tabs)
        if (kv.first == key) {
            *out = kv.second;
            return GA_OK;
        }
    Ctx* ctx = d->ctx;
    std::vector<uint32_t> cs(64 * 8, 0);   // cs[s] = shift^(2^(logn-1-s))
    F c = shift;
    for (int sidx = d->logn - 1; sidx >= 0; sidx--) {
        memcpy(&cs[sidx * 8], c.l, 32);
        c = sqr(c);
    }
    struct DevTmp {
        void* p = nullptr;
        ~DevTmp() { hipFree(p); }
    } tmp;
    GA_HIP_CHECK(device_malloc(&tmp.p, cs.size() * 4));
    GA_HIP_CHECK(hipMemcpy(tmp.p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    uint32_t* tab = nullptr;
    GA_HIP_CHECK(device_malloc((void**)&tab, d->n * NTT_TW_WORDS * 4));
    hipLaunchKernelGGL((ntt_twiddle_kernel<FrP>), dim3((unsigned)((d->n + 255) / 256)), dim3(256), 0, ctx->work_stream(), tab,
                       (const uint32_t*)d->d_pow2, d->n, d->logn, 1, (const uint32_t*)tmp.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->work_stream());
    if (e != hipSuccess) {
        hipFree(tab);
        set_error("coset twiddle table: %s", hipGetErrorString(e));
        return GA_ERR_HIP;
    }
    d->coset_tabs.emplace_back(key, tab);
    *out = tab;
    return GA_OK;
}

template <class FrP>
int domain_init(Ctx* ctx, Domain* d, int curve, uint64_t n) {
    typedef Fe<FrP> F;
    d->ctx = ctx;
    d->curve = curve;
    d->n = n;
    d->logn = ilog2_u64(n);
    if ((1ull << d->logn) != n || d->logn > FrP::ADICITY) {
        set_error("domain cardinality %llu is not a power of two <= 2^%d", (unsigned long long)n, FrP::ADICITY);
        return GA_ERR_INVALID;
    }
    d->passes = ntt_plan(d->logn);
    // host: w = ROOT^(2^(adicity-logn)), inverse likewise; tables of w^(2^k)
    F w = fe_const<FrP>(FrP::ROOT), wi = fe_const<FrP>(FrP::ROOT_INV);
    for (int k = 0; k < FrP::ADICITY - d->logn; k++) {
        w = sqr(w);
        wi = sqr(wi);
    }
    F g = fe_const<FrP>(FrP::GEN), gi = fe_const<FrP>(FrP::GEN_INV);
    // 1/n = (1/2)^logn
    F two = add(fe_one<FrP>(), fe_one<FrP>());
    F half = inv(two);
    F ninv = fe_one<FrP>();
    for (int k = 0; k < d->logn; k++) ninv = mul(ninv, half);
    {
        F nv = d->lazy ? f29_hat_packed(ninv) : ninv;   // used as a scale factor by the pass kernels
        memcpy(d->ninv, nv.l, 32);
    }
    // den = (g^n - 1)^-1
    F gn = g;
    for (int k = 0; k < d->logn; k++) gn = sqr(gn);
    F den = inv(sub(gn, fe_one<FrP>()));
    memcpy(d->den, den.l, 32);
    {
        F nm = fe_one<FrP>();                       // n = 2^logn (Montgomery)
        for (int k = 0; k < d->logn; k++) nm = add(nm, nm);
        memcpy(d->n_mont, nm.l, 32);
        F dn2 = mul(den, mul(ninv, ninv));
        memcpy(d->den_n2, dn2.l, 32);
    }

    uint64_t half_n = n / 2;
    if (half_n > 0) {
        std::vector<uint32_t> p2(2 * 32 * 8, 0);
        F a = w, b = wi;
        for (int k = 0; k < 32; k++) {
            memcpy(&p2[k * 8], a.l, 32);
            memcpy(&p2[(32 + k) * 8], b.l, 32);
            a = sqr(a);
            b = sqr(b);
        }
        GA_HIP_CHECK(device_malloc((void**)&d->d_pow2, p2.size() * 4));
        GA_HIP_CHECK(hipMemcpy(d->d_pow2, p2.data(), p2.size() * 4, hipMemcpyHostToDevice));
        uint32_t** tabs[4] = {&d->d_ts, &d->d_ts_inv, &d->d_tb, &d->d_tb_inv};
        for (int t = 0; t < 4; t++) {
            const bool stacked = t < 2;
            const uint64_t count = stacked ? n : half_n;
            GA_HIP_CHECK(device_malloc((void**)tabs[t], count * NTT_TW_WORDS * 4));
            hipLaunchKernelGGL((ntt_twiddle_kernel<FrP>), dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->work_stream(), *tabs[t],
                               (const uint32_t*)d->d_pow2 + (t & 1) * 32 * 8, count, d->logn, stacked ? 1 : 0, (const uint32_t*)nullptr);
        }
        GA_KERNEL_CHECK();
        GA_HIP_CHECK(hipStreamSynchronize(ctx->work_stream()));
        const uint32_t* tg = nullptr;
        GA_CHECK(ntt_coset_table<FrP>(d, g, &tg));
        d->d_ts_g = const_cast<uint32_t*>(tg);
    }
    // coset power tables (host-computed: <= 2^12 + n/2^12 entries each)
    uint64_t nlo = 1ull << NTT_POW_LO_BITS;
    uint64_t nhi = (n >> NTT_POW_LO_BITS) ? (n >> NTT_POW_LO_BITS) : 1;
    auto build = [&](const F& base, const F& c0, uint32_t** dlo, uint32_t** dhi) -> int {
        std::vector<uint32_t> lo(nlo * 8), hi(nhi * 8);
        F acc = c0;
        for (uint64_t k = 0; k < nlo; k++) {
            F st = d->lazy ? f29_hat_packed(acc) : acc;
            memcpy(&lo[k * 8], st.l, 32);
            acc = mul(acc, base);
        }
        F step = base;
        for (int k = 0; k < NTT_POW_LO_BITS; k++) step = sqr(step);
        acc = fe_one<FrP>();
        for (uint64_t k = 0; k < nhi; k++) {
            F st = d->lazy ? f29_hat_packed(acc) : acc;
            memcpy(&hi[k * 8], st.l, 32);
            acc = mul(acc, step);
        }
        GA_HIP_CHECK(device_malloc((void**)dlo, nlo * 32));
        GA_HIP_CHECK(hipMemcpy(*dlo, lo.data(), nlo * 32, hipMemcpyHostToDevice));
        if (dhi) {
            GA_HIP_CHECK(device_malloc((void**)dhi, nhi * 32));
            GA_HIP_CHECK(hipMemcpy(*dhi, hi.data(), nhi * 32, hipMemcpyHostToDevice));
        }
        return GA_OK;
    };
    GA_CHECK(build(g, fe_one<FrP>(), &d->d_g_lo, &d->d_g_hi));
    GA_CHECK(build(gi, ninv, &d->d_gi_lo, &d->d_gi_hi));
    return GA_OK;
}

inline void domain_free(Domain* d) {
    hipFree(d->d_ts);
    hipFree(d->d_ts_inv);
    hipFree(d->d_tb);
    hipFree(d->d_tb_inv);
    hipFree(d->d_pow2);
    for (auto& kv : d->coset_tabs) hipFree(kv.second);   // (d_ts_g is one of them)
    d->coset_tabs.clear();
    hipFree(d->d_g_lo);
    hipFree(d->d_g_hi);
    hipFree(d->d_gi_lo);
    hipFree(d->d_gi_hi);
}

}  // namespace ga
