// Pippenger multi-scalar multiplication for G1/G2 of BN254 and BLS12-381 on gfx950.
//
// Replaces G1Jac.MultiExp / G2Jac.MultiExp (backend/groth16/bn254/prove.go:194,207,227,237,283) and the ICICLE
// msm.Msm / g2.G2Msm calls (backend/accelerated/icicle/groth16/bn254/icicle.go:362-467).
//
// Pipeline (all on the context's stream, no host synchronisation until a few hundred bytes of sums are copied back):
//   1. msm_digits_kernel   scalar -> signed c-bit digits (Montgomery reduction fused in); one (key, value) pair per
//                          (point, window): key = window*2^(c-1) + |digit|-1, value = point index | sign<<31.
//                          Zero digits get key = SKIP (sorts last), so zero scalars and the constant-0 wires of a
//                          witness cost nothing downstream.
//   2. radix sort          of the pairs by key (rocprim onesweep on the significant bits only, msm_sort_pairs); after it every
//                          bucket is a contiguous run of point indices.
//      From GA_MSM_FUSE_MIN (2^21) pairs up, 1 and 2 are ONE two-level sort of our own, fused with the digit extraction
//      (1b / 1c below): the pairs are written once and read once, the sorted keys are never materialised.
//   3. msm_offsets_tasks_kernel / msm_tasks_kernel  bucket boundaries (by binary search, or the fused sort's own per-key scan);
//                          buckets are split into tasks of at most SEG points so that a hot bucket (witness values 0/1 make
//                          bucket 1 of window 0 huge) is spread over many lanes; the task list is ordered by decreasing length
//                          (one 8-bit radix pass on the quantised length) so that the lanes of a wave finish together.
//   4. msm_accumulate29_kernel  one lane per task: gathers its bases from the window table (or the hat-domain copy of un-pinned
//                          bases), XYZZ mixed additions in the lazy 29 / 28-bit limb representation, accumulator in LDS.
//   5. msm_merge_kernel / msm_hot_kernel  partial sums -> one XYZZ sum per bucket (wave-level LDS tree for hot buckets).
//   6. msm_reduce_groups29_kernel + per-bit / segment sums  sum_k k*B_k per bucket set via per-group running sums and a wave
//                          tree; the last ~c doublings of that sum -- and, for un-pinned bases, the Horner step over the windows,
//                          merged into the same chain -- run on the host: a chain of sequential doublings is latency-bound on a
//                          GPU lane and free on a host core, and the multi-GPU window-sharded mode needs the window sums on the
//                          host anyway.
#pragma once
#include <cstring>
#include <hipcub/hipcub.hpp>
#include <rocprim/device/device_radix_sort.hpp>

#include <string>

#include "common.hip.h"
#include "field29.hip.h"

namespace ga {

constexpr int GA_ACC29_MINW = 4;      // waves per SIMD requested for the G1 bucket kernel (2 for Fp2 points: 72 KiB of LDS per workgroup)
#ifndef GA_ACC29_FP2_MINW             // (compile-time experiments only: tools/exp/r06_bls_g2_two_waves.sh)
#define GA_ACC29_FP2_MINW 2
#endif
constexpr uint32_t MSM_SIGN = 0x80000000u;
constexpr int MSM_HOT_TASKS = 16;     // buckets with more partials than this go to the wave-parallel merge
constexpr uint32_t MSM_VHOT_TASKS = 512;   // ... and with more than this, to the two-stage merge over MSM_VHOT_SPLIT blocks per bucket
constexpr uint32_t MSM_VHOT_SPLIT = 64;
constexpr int MSM_GROUP = 32;         // buckets per running-sum group in the window reduction

// Onesweep configuration for the bucket keys of large MSMs (17..22 significant bits at c = 18..22): two 11-bit passes instead of
// the library default's three 8-bit ones.  Measured on 12 x 2^24 pairs with 22-bit keys (tools/exp/sortbench.hip,
// profiles/r02_e_sort_configs.txt): default 4.34 ms, 1024 threads x 21 items with 11-bit digits 3.47 ms; 512-thread blocks,
// 12-bit digits (LDS) and more items per thread are slower or do not fit.
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 21>, rocprim::kernel_config<1024, 21>, 11,
                                                                       rocprim::block_radix_rank_algorithm::match>>
    MsmSortWide;
// The task list is sorted on 8 key bits (msm_prepare): ONE onesweep pass.  The library's default switches to a merge sort below 2^20
// items -- 20 launches, 0.14 ms, for the 1.04 M tasks of a 2^20-point MSM (profiles/r05_h_msm_2p20_raw_kernels_seg256.txt) -- so the
// limit is lowered to where a merge sort is really cheaper.
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768> MsmTaskSort;

// Sort (key, value) pairs on the low end_bit key bits, ping-ponging between the two buffer pairs (no third copy of the data); on
// return keys2/vals2 point at the sorted arrays and keys/vals at the other pair.  The library sort: small MSMs, and whatever the
// fused path (1b / 1c below) does not take.
inline int msm_sort_pairs(Ctx* ctx, const std::string& tmp_name, uint32_t*& keys, uint32_t*& keys2, uint32_t*& vals, uint32_t*& vals2, size_t m,
                          int end_bit, hipStream_t st) {
    rocprim::double_buffer<uint32_t> dk(keys, keys2), dv(vals, vals2);
    const bool wide = (end_bit + 10) / 11 < (end_bit + 7) / 8;   // fewer passes with 11-bit digits than with 8-bit ones
    size_t tmp_bytes = 0;
    void* tmp = nullptr;
    if (wide) GA_HIP_CHECK((rocprim::radix_sort_pairs<MsmSortWide>(nullptr, tmp_bytes, dk, dv, m, 0u, (unsigned)end_bit, st)));
    else GA_HIP_CHECK((rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, m, 0u, (unsigned)end_bit, st)));
    GA_CHECK(ctx->scratch_get(tmp_name.c_str(), tmp_bytes + 256, &tmp));
    if (wide) GA_HIP_CHECK((rocprim::radix_sort_pairs<MsmSortWide>(tmp, tmp_bytes, dk, dv, m, 0u, (unsigned)end_bit, st)));
    else GA_HIP_CHECK((rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, m, 0u, (unsigned)end_bit, st)));
    keys2 = dk.current();
    keys = dk.alternate();
    vals2 = dv.current();
    vals = dv.alternate();
    return GA_OK;
}

// ---- 1. digits ------------------------------------------------------------------------------------
// The signed c-bit digits of one scalar, least significant window first: (key, value) of window w.
// table mode: every window shares ONE bucket set (bucket set `key_base / half` of a batch of scalar vectors over the same
// table) and the value indexes the precomputed table [window][point]; skip = total bucket count (sorts last)
template <class FrP>
struct DigitWalk {
    Fe<FrP> s;
    uint32_t carry = 0;
    __device__ __forceinline__ void load(const uint32_t* __restrict__ scalars, uint64_t i, int mont) { set(load_fe<FrP>(scalars + i * 8), mont); }
    __device__ __forceinline__ void set(const Fe<FrP>& raw, int mont) {   // (the words may have been loaded ahead of time)
        s = raw;
        if (mont) s = from_mont(s);
        else {
            // canonical input may be any 256-bit integer (a caller's big.Int bytes): bring it below r, at most 2^256 / r < 6 steps,
            // so that only (BITS mod c) bits are live in the top window as the digit loop assumes
#pragma unroll 1
            for (int k = 0; k < 6; k++) reduce_once<FrP>(s.l);
        }
    }
    __device__ __forceinline__ void next(int c, int w, int win_lo, uint64_t n, uint64_t i, int table, uint32_t key_base, uint32_t skip,
                                         uint32_t& key, uint32_t& val) {
        const uint32_t half = 1u << (c - 1);
        const uint32_t mask = (1u << c) - 1;
        uint32_t d = (s.l[0] & mask) + carry;
        // s >>= c  (c < 32)
#pragma unroll
        for (int k = 0; k < 7; k++) s.l[k] = (s.l[k] >> c) | (s.l[k + 1] << (32 - c));
        s.l[7] >>= c;
        uint32_t neg = 0;
        if (d > half) {
            d = (1u << c) - d;
            neg = MSM_SIGN;
            carry = 1;
        } else {
            carry = 0;
        }
        key = d == 0 ? skip : key_base + (table ? 0u : (uint32_t)(w - win_lo) * half) + (d - 1);
        val = (table ? (uint32_t)((uint64_t)w * n + i) : (uint32_t)i) | neg;
    }
};

template <class FrP>
__global__ void msm_digits_kernel(const uint32_t* __restrict__ scalars, uint64_t n, int mont, int c, int nwin, int win_lo,
                                  int win_hi, int table, uint32_t key_base, uint32_t skip, uint32_t* __restrict__ keys,
                                  uint32_t* __restrict__ vals) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DigitWalk<FrP> D;
    D.load(scalars, i, mont);
    for (int w = 0; w < nwin; w++) {
        uint32_t k, v;
        D.next(c, w, win_lo, n, i, table, key_base, skip, k, v);
        if (w >= win_lo && w < win_hi) {
            uint64_t idx = (uint64_t)(w - win_lo) * n + i;
            keys[idx] = k;
            vals[idx] = v;
        }
    }
}

// ---- 1b. digits fused with the first level of the sort (large bucket sets) ------------------------------------------------------
// The plain sequence writes the (key, value) pairs in scalar order (1.6 GB at 12 x 2^24), reads the keys for the histograms and
// reads / scatters the pairs twice (two 11-bit onesweep passes).  Here the FIRST level -- a partition into at most 2^BITS groups of
// consecutive keys -- is made by the kernel that extracts the digits: a histogram of the groups straight from the scalars (digits
// are cheap to recompute: nothing is written), then a tile of <= 1024 scalars x windows is partitioned in LDS and leaves the CU as
// one run per (tile, group); the second level (1c below) finishes the grouping without the library.
// The order inside a group is not the input order (ranks come from LDS atomics) -- irrelevant for a first level.  Any key distribution
// works: a group's slice of the output is reserved with one global atomic per (tile, group).
// Measured at 12 x 2^24 pairs (tools/exp/partbench.hip, profiles/README.md round 3 batch ZZ2): digits 0.37 + sort 3.62 ms ->
// histogram 0.19 + digits/first pass 1.38 + one library pass for the other bits 1.87 ms (1c replaces that pass).
constexpr int MSM_P1_THREADS = 1024;
constexpr int MSM_P1_MAXW = 16;                       // windows a thread keeps in registers
constexpr uint32_t MSM_P1_ENTRIES = 1024 * 13;        // pairs staged per tile: 104 KB of LDS (+ 16 / 32 KB of bin tables)
constexpr uint32_t MSM_P2_SEG = 16384;                // pairs per second-level segment
constexpr uint32_t MSM_P2_HB = 4104;                  // capacity for the key parts the second level counts in LDS
constexpr uint32_t MSM_XCDS = 8;                      // XCDs of the device: block b is observed to run on XCD b % 8 (a speed assumption only)
// How a key splits between the two levels: the first level groups by key >> low (at most 2^BITS groups), the second level counts the
// 2^low <= 4096 low parts of a group's keys.  A group owns a CONTIGUOUS key range, hence a contiguous slice of the per-key counters,
// of the cursors and of the sorted output: a segment's atomics are consecutive words and its runs land inside the group's own slice.
// (Rounds 3 and 4 split the other way round -- first level on the low 11 / 12 key bits -- which spreads one segment's counters and
// runs 2^BITS keys apart: one memory transaction per (segment, key) three times over.  Same box, 2^24 points,
// profiles/r05_a_sort_ab_2p24.txt: second level 4.06 -> 1.38 ms on the 13 x 2^19 keys of un-pinned bases, 1.44 -> 1.08 ms on a
// table's 2^21 keys; with the XCD placement below 1.15 / 0.98 ms and the first level 1.80 -> 1.44 / 1.73 -> 1.28 ms.)
// BITS = 11 while 2^12 low parts suffice, else 12 (key spaces up to 2^24).
static inline int msm_p1_bits(uint64_t nb) { return (nb >> 12) + 1 <= 2048 ? 11 : 12; }
static inline bool msm_fused_fits(uint64_t nb) {
    const int b = msm_p1_bits(nb);
    return nb >= (1ull << b) && (nb >> 12) + 1 <= (1ull << b);
}
static inline int msm_key_low(uint64_t nb, int bits) {   // smallest low with (nb >> low) + 1 <= 2^bits groups
    int low = 0;
    while ((nb >> low) + 1 > (1ull << bits)) low++;
    return low;
}
static inline uint32_t msm_p1_tile_scalars(int nwl) {
    const uint32_t t = MSM_P1_ENTRIES / (uint32_t)nwl;
    return t < (uint32_t)MSM_P1_THREADS ? t : (uint32_t)MSM_P1_THREADS;
}
// In-place exclusive prefix sums of a[0, count) in LDS by a block of exactly 1024 threads (count <= 5 * 1024); a[count] receives the
// total, which is also returned.  wtot: 16 words of LDS.  The caller has synchronised the block on a[]; the block is synchronised on
// return.  (Round 3 scanned with two ping-pong arrays: 3 x 4 bytes per bin instead of 1 -- what kept 12-bit levels out of 160 KB.)
__device__ __forceinline__ uint32_t msm_block_excl_scan_1024(uint32_t* __restrict__ a, uint32_t count, uint32_t* __restrict__ wtot) {
    GA_REQUIRE_WAVE64();   // 16 waves of 64 lanes: lane 63 publishes the wave total, __shfl_up runs to distance 32
    constexpr int PER = 5;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t v[PER], s = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t idx = t * PER + k;
        const uint32_t x = idx < count ? a[idx] : 0;
        v[k] = s;
        s += x;
    }
    uint32_t inc = s;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
        const uint32_t x = wtot[w];
        if (w < wave) base += x;
        total += x;
    }
    base += inc - s;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t idx = t * PER + k;
        if (idx < count) a[idx] = base + v[k];
    }
    if (t == 0) a[count] = total;
    __syncthreads();
    return total;
}

// Histogram of the first-level groups straight from the scalars (digits recomputed, nothing written).  A block walks whole TILES of
// the first pass (tile t, t + grid, ...; the grid is a multiple of 8), so that with per-XCD slices (ncls = 8) the counts of class
// t % 8 -- the XCD the first pass's block t is expected on -- are kept apart: ghist[bin * ncls + class].
template <class FrP, int BITS>
__global__ void __launch_bounds__(256)
msm_digit_hist_kernel(const uint32_t* __restrict__ scalars, uint64_t n, int mont, int c, int nwin, int win_lo, int win_hi, int table,
                      uint32_t key_base, uint32_t skip, uint32_t tile_scalars, int low, uint32_t ncls, uint32_t* __restrict__ ghist) {
    constexpr uint32_t BINS = 1u << BITS;
    __shared__ uint32_t h[BINS];
    for (uint32_t b = threadIdx.x; b < BINS; b += blockDim.x) h[b] = 0;
    __syncthreads();
    const uint64_t ntiles = (n + tile_scalars - 1) / tile_scalars;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t i0 = tile * tile_scalars;
        const uint64_t i1 = i0 + tile_scalars < n ? i0 + tile_scalars : n;
        for (uint64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
            DigitWalk<FrP> D;
            D.load(scalars, i, mont);
            for (int w = 0; w < win_hi; w++) {
                uint32_t k, v;
                D.next(c, w, win_lo, n, i, table, key_base, skip, k, v);
                if (w >= win_lo) atomicAdd(&h[(k >> low)], 1u);
            }
        }
    }
    __syncthreads();
    const uint32_t cls = ncls > 1 ? (blockIdx.x % ncls) : 0;
    for (uint32_t b = threadIdx.x; b < BINS; b += blockDim.x)
        if (h[b]) atomicAdd(&ghist[b * ncls + cls], h[b]);
}

// exclusive scans of the group counts (one block): where each group's slice of the partitioned arrays starts (bin_off: kept, BINS + 1
// entries), where each (group, class) sub-slice starts (cursor: consumed by the first pass) and the number of MSM_P2_SEG-pair
// segments before each group (seg_off)
template <int BITS>
static __global__ void __launch_bounds__(1024) msm_p1_scan_kernel(const uint32_t* __restrict__ ghist, uint32_t ncls, uint32_t* __restrict__ cursor,
                                                                  uint32_t* __restrict__ bin_off, uint32_t* __restrict__ seg_off) {
    constexpr uint32_t BINS = 1u << BITS;
    __shared__ uint32_t a[BINS + 1], g[BINS + 1], wtot[16];
    for (uint32_t b = threadIdx.x; b < BINS; b += blockDim.x) {
        uint32_t tot = 0;
        for (uint32_t k = 0; k < ncls; k++) tot += ghist[b * ncls + k];
        a[b] = tot;
        g[b] = (tot + MSM_P2_SEG - 1) / MSM_P2_SEG;
    }
    __syncthreads();
    msm_block_excl_scan_1024(a, BINS, wtot);
    msm_block_excl_scan_1024(g, BINS, wtot);
    for (uint32_t b = threadIdx.x; b <= BINS; b += blockDim.x) {
        if (b < BINS) {
            uint32_t s = a[b];
            for (uint32_t k = 0; k < ncls; k++) {
                cursor[b * ncls + k] = s;
                s += ghist[b * ncls + k];
            }
        }
        bin_off[b] = a[b];
        seg_off[b] = g[b];
    }
}

// A block walks tiles blockIdx.x, blockIdx.x + grid, ... (the grid is a multiple of 8 whenever a block gets more than one tile, so a
// block's tiles share its XCD class) and loads the NEXT tile's scalars before it writes the current one out: the CU holds one
// workgroup (123 KB of LDS), so nothing else could hide that load.  Same box, 12 x 2^24 pairs (profiles/r05_s_sort_pipelined_ab.txt):
// histogram + first level 1.19 -> 1.13 ms with 256 / 512 / 1024 blocks (one tile per block in this loop form: 1.26).
template <class FrP, int BITS>
__global__ void __launch_bounds__(MSM_P1_THREADS)
msm_digits_pass1_kernel(const uint32_t* __restrict__ scalars, uint64_t n, int mont, int c, int nwin, int win_lo, int win_hi, int table,
                        uint32_t key_base, uint32_t skip, uint32_t tile_scalars, uint64_t ntiles, int low, uint32_t ncls,
                        uint32_t* __restrict__ cursor, uint16_t* __restrict__ out_keys, uint32_t* __restrict__ out_vals) {
    constexpr uint32_t BINS = 1u << BITS;
    __shared__ uint32_t stage_k[MSM_P1_ENTRIES], stage_v[MSM_P1_ENTRIES];
    __shared__ uint32_t start[BINS + 1], delta[BINS], wtot[16];   // start: counts, then (scanned in place) where a bin's run starts in the staging arrays
    const uint32_t t = threadIdx.x;
    const uint32_t cls = ncls > 1 ? (blockIdx.x % ncls) : 0;
    uint64_t i = (uint64_t)blockIdx.x * tile_scalars + t;
    bool live = t < tile_scalars && i < n;
    Fe<FrP> raw;
    if (live) raw = load_fe<FrP>(scalars + i * 8);
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (uint32_t b = t; b < BINS; b += blockDim.x) start[b] = 0;
        __syncthreads();
        uint32_t key[MSM_P1_MAXW], val[MSM_P1_MAXW], rank[MSM_P1_MAXW];
        if (live) {
            DigitWalk<FrP> D;
            D.set(raw, mont);
            for (int w = 0; w < win_lo; w++) {   // (windows below this device's share: only their carries matter)
                uint32_t k, v;
                D.next(c, w, win_lo, n, i, table, key_base, skip, k, v);
            }
#pragma unroll
            for (int q = 0; q < MSM_P1_MAXW; q++)
                if (win_lo + q < win_hi) {
                    D.next(c, win_lo + q, win_lo, n, i, table, key_base, skip, key[q], val[q]);
                    rank[q] = atomicAdd(&start[(key[q] >> low)], 1u);
                }
        }
        __syncthreads();
        const uint32_t total = msm_block_excl_scan_1024(start, BINS, wtot);
        // a bin's slice of the output is reserved with one global atomic per (tile, bin) -- with per-XCD slices (ncls = 8) inside the
        // sub-slice of this block's class, so that the runs one XCD's L2 collects are neighbours; delta = where the run goes - where
        // it is staged.  (The reservations are issued here and their results used only after the staging below, so that the atomics'
        // round trips run under the LDS writes: 1.26 -> 1.23 ms at 12 x 2^24 pairs, profiles/r05_n_sort_atomics_ab.txt.)
        constexpr int PER_T = (int)(BINS / MSM_P1_THREADS);
        uint32_t got[PER_T];
#pragma unroll
        for (int u = 0; u < PER_T; u++) {
            const uint32_t b = t + (uint32_t)u * MSM_P1_THREADS;
            const uint32_t cnt = start[b + 1] - start[b];
            got[u] = cnt ? atomicAdd(&cursor[b * ncls + cls], cnt) : 0u;
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < MSM_P1_MAXW; q++)
                if (win_lo + q < win_hi) {
                    const uint32_t at = start[(key[q] >> low)] + rank[q];
                    stage_k[at] = key[q];
                    stage_v[at] = val[q];
                }
        }
        // the next tile's scalars: requested now, needed after the write phase
        i += (uint64_t)gridDim.x * tile_scalars;
        live = tile + gridDim.x < ntiles && t < tile_scalars && i < n;
        if (live) raw = load_fe<FrP>(scalars + i * 8);
#pragma unroll
        for (int u = 0; u < PER_T; u++) {
            const uint32_t b = t + (uint32_t)u * MSM_P1_THREADS;
            delta[b] = got[u] - start[b];   // (bins without pairs: never looked up)
        }
        __syncthreads();
        for (uint32_t p = t; p < total; p += blockDim.x) {   // consecutive lanes write consecutive addresses inside a run
            const uint32_t k = stage_k[p];
            const uint32_t dst = p + delta[(k >> low)];
            out_keys[dst] = (uint16_t)(k & ((1u << low) - 1));   // the second level knows the group from its segment: only the part it counts travels
            out_vals[dst] = stage_v[p];
        }
        __syncthreads();   // (the staging arrays and the bin tables are rewritten by the next tile)
    }
}

// ---- 1c. the second level of the fused sort, in place of the library pass and the binary-search offsets ------------------------
// After the first pass the pairs are grouped; inside a group a pair's final place is off[key] + (any rank among the pairs with the
// same key): no stability is needed, only the per-key counts.  Segments of at most MSM_P2_SEG pairs of ONE group count their key
// parts in LDS and add them to a global per-key histogram (gcount[key]); an exclusive scan of that histogram IS the bucket-offset
// array `off`; then the same segments reserve one run per (segment, key) behind a global atomic and write the VALUES (the sorted
// keys are never materialised), LDS-staged so that a run leaves the CU as consecutive addresses.  Any key distribution works (a
// group of any size is just more segments).
// Measured at 12 x 2^24 pairs (tools/exp/partbench.hip variant C): 1.41 ms against the library pass + offsets kernel's 2.0 ms.
// swz: consecutive segments -- the segments of one group, whose runs are neighbours in the output when the groups are key ranges --
// go to ONE XCD (block b runs on XCD b % 8: it takes segment (b % 8) * ceil(S / 8) + b / 8 of the S the device counted), so that the
// partial lines they write meet in one L2.  Placement is a speed assumption only.
template <int BITS>
__device__ __forceinline__ bool msm_p2_segment(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ bin_off, int swz, uint32_t& bin,
                                               uint32_t& lo, uint32_t& hi) {
    constexpr uint32_t BINS = 1u << BITS;
    const uint32_t nseg = seg_off[BINS];
    uint32_t sidx = blockIdx.x;
    if (swz) {
        const uint32_t per = (nseg + MSM_XCDS - 1) / MSM_XCDS, j = blockIdx.x / MSM_XCDS;
        if (j >= per) return false;
        sidx = (blockIdx.x % MSM_XCDS) * per + j;
    }
    if (sidx >= nseg) return false;
    uint32_t l = 0, r = BINS;   // the last bin with seg_off[bin] <= sidx
    while (r - l > 1) {
        const uint32_t mid = (l + r) >> 1;
        if (seg_off[mid] <= sidx) l = mid;
        else r = mid;
    }
    bin = l;
    lo = bin_off[bin] + (sidx - seg_off[bin]) * MSM_P2_SEG;
    hi = bin_off[bin + 1];
    if (hi - lo > MSM_P2_SEG) hi = lo + MSM_P2_SEG;
    return true;
}
template <int BITS>
static __global__ void __launch_bounds__(1024)
msm_p2_count_kernel(const uint16_t* __restrict__ keys, const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ bin_off,
                    uint32_t hb, int low, int swz, uint32_t* __restrict__ gcount) {
    __shared__ uint32_t cnt[MSM_P2_HB];
    uint32_t bin, lo, hi;
    if (!msm_p2_segment<BITS>(seg_off, bin_off, swz, bin, lo, hi)) return;   // (uniform per block)
    for (uint32_t h = threadIdx.x; h < hb; h += blockDim.x) cnt[h] = 0;
    __syncthreads();
    constexpr int U = MSM_P2_SEG / 1024;
    uint32_t kk[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t p = lo + u * 1024 + threadIdx.x;
        kk[u] = p < hi ? (uint32_t)keys[p] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < U; u++)
        if (kk[u] != 0xFFFFFFFFu) atomicAdd(&cnt[kk[u]], 1u);
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < hb; h += blockDim.x)
        if (cnt[h]) atomicAdd(&gcount[(bin << low) | h], cnt[h]);
}
template <int BITS>
static __global__ void __launch_bounds__(1024)
msm_p2_scatter_kernel(const uint16_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ seg_off,
                      const uint32_t* __restrict__ bin_off, uint32_t hb, int low, int swz, uint32_t* __restrict__ cursor,
                      uint32_t* __restrict__ out_vals) {
    __shared__ uint32_t stage_v[MSM_P2_SEG];
    __shared__ uint16_t stage_h[MSM_P2_SEG];
    __shared__ uint32_t start[MSM_P2_HB + 1], delta[MSM_P2_HB], wtot[16];
    uint32_t bin, lo, hi;
    if (!msm_p2_segment<BITS>(seg_off, bin_off, swz, bin, lo, hi)) return;
    const uint32_t t = threadIdx.x;
    for (uint32_t h = t; h < hb; h += blockDim.x) start[h] = 0;
    __syncthreads();
    constexpr int U = MSM_P2_SEG / 1024;
    uint32_t kk[U], vv[U], rk[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint32_t p = lo + u * 1024 + t;
        kk[u] = p < hi ? (uint32_t)keys[p] : 0xFFFFFFFFu;
        vv[u] = p < hi ? vals[p] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; u++)
        if (kk[u] != 0xFFFFFFFFu) rk[u] = atomicAdd(&start[kk[u]], 1u);
    __syncthreads();
    msm_block_excl_scan_1024(start, hb, wtot);
    // (as in the first level: all of a thread's run reservations are issued before any result is used; no measurable change here,
    // 0.94-0.97 -> 0.94-0.95 ms)
    constexpr int PER_T = 4;   // hb <= 4096 key parts, 1024 threads
    uint32_t got[PER_T];
#pragma unroll
    for (int u = 0; u < PER_T; u++) {
        const uint32_t h = t + (uint32_t)u * 1024u;
        got[u] = 0;
        if (h < hb) {
            const uint32_t cnt = start[h + 1] - start[h];
            if (cnt) got[u] = atomicAdd(&cursor[(bin << low) | h], cnt);
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
        if (kk[u] != 0xFFFFFFFFu) {
            const uint32_t h = kk[u], at = start[h] + rk[u];
            stage_v[at] = vv[u];
            stage_h[at] = (uint16_t)h;
        }
#pragma unroll
    for (int u = 0; u < PER_T; u++) {
        const uint32_t h = t + (uint32_t)u * 1024u;
        if (h < hb) delta[h] = got[u] - start[h];
    }
    __syncthreads();
    const uint32_t total = hi - lo;
    for (uint32_t p = t; p < total; p += blockDim.x) out_vals[p + delta[stage_h[p]]] = stage_v[p];
}

// The fused sort of `batch` scalar vectors' (key, value) pairs: on return `off` holds the bucket offsets (nb + 2 entries) and vals2
// the values grouped by key.  keys / vals: the first level's output (scratch).
template <class FrP, int BITS>
int msm_fused_sort(Ctx* ctx, const std::string& sfx, hipStream_t st, const void* d_scalars, size_t n, bool scalars_mont, int c, int nwin,
                   int win_lo, int win_hi, bool table, int batch, uint32_t half, uint32_t nb, uint64_t m, uint16_t* keys, uint32_t* vals,
                   uint32_t* vals2, uint32_t* off, int xcd) {
    // xcd (GA_MSM_XCD, A/B knob): bit 0 per-XCD slices in the first level, bit 1 XCD swizzle of the second level's segments, bit 2
    // the slices at any size (tests)
    constexpr uint32_t BINS = 1u << BITS;
    auto key = [&](const char* k) { return std::string(k) + sfx; };
    const int nwl = win_hi - win_lo;
    const int low = msm_key_low(nb, BITS);
    // (per-XCD slices make the histogram and cursor tables 8 x as long: below 2^24 pairs they cost what they save,
    // profiles/r05_b_fuse_min_sweep.txt)
    const uint32_t ncls = ((xcd & 1) && (m >= (1ull << 24) || (xcd & 4))) ? MSM_XCDS : 1;
    const int swz = (xcd & 2) ? 1 : 0;
    uint32_t *ghist, *cursor, *bin_off, *seg_off, *gcount, *kcursor;
    GA_CHECK(ctx->scratch_get(key("msm_p1_hist").c_str(), BINS * MSM_XCDS * 4, (void**)&ghist));
    GA_CHECK(ctx->scratch_get(key("msm_p1_cursor").c_str(), BINS * MSM_XCDS * 4, (void**)&cursor));
    GA_CHECK(ctx->scratch_get(key("msm_p1_bin_off").c_str(), (BINS + 1) * 4, (void**)&bin_off));
    GA_CHECK(ctx->scratch_get(key("msm_p2_seg_off").c_str(), (BINS + 1) * 4, (void**)&seg_off));
    // the key parts the second level counts, and every key a (group, part) pair can form (>= nb + 1)
    const uint32_t hb = 1u << low;   // <= 4096 (msm_fused_fits): msm_p2_scatter_kernel reserves four runs per thread
    const uint64_t nkeys = (((uint64_t)nb >> low) + 1) << low;
    GA_CHECK(ctx->scratch_get(key("msm_p2_count").c_str(), nkeys * 4, (void**)&gcount));
    GA_CHECK(ctx->scratch_get(key("msm_p2_cursor").c_str(), nkeys * 4, (void**)&kcursor));
    auto vec = [&](int b) { return batch == 1 ? (const uint32_t*)d_scalars : reinterpret_cast<const uint32_t* const*>(d_scalars)[b]; };
    {
        StageTimer tm(ctx, "msm_digits_pass1", st);
        const uint32_t tile = msm_p1_tile_scalars(nwl);
        const uint64_t ntiles = (n + tile - 1) / tile;
        uint64_t hist_blocks = (ntiles + MSM_XCDS - 1) / MSM_XCDS * MSM_XCDS;   // a multiple of 8: tile t and the block that counts it agree on t % 8
        if (hist_blocks > 2048) hist_blocks = 2048;
        const uint64_t p1_grid = ctx->tun.msm_p1_grid.load(std::memory_order_relaxed);   // GA_MSM_P1_GRID (A/B knob; tests)
        uint64_t p1_blocks = ntiles <= p1_grid ? ntiles : p1_grid;
        if (ncls > 1 && p1_blocks < ntiles) p1_blocks = (p1_blocks + MSM_XCDS - 1) / MSM_XCDS * MSM_XCDS;   // a block's tiles must share t % 8 (the histogram's classes)
        GA_HIP_CHECK(hipMemsetAsync(ghist, 0, BINS * ncls * 4, st));
        for (int b = 0; b < batch; b++)   // (a batch: the vectors' bucket sets are stacked in ONE key space, key_base = b * 2^(c-1))
            hipLaunchKernelGGL((msm_digit_hist_kernel<FrP, BITS>), dim3((unsigned)hist_blocks), dim3(256), 0, st, vec(b), (uint64_t)n,
                               scalars_mont ? 1 : 0, c, nwin, win_lo, win_hi, table ? 1 : 0, (uint32_t)b * half, nb, tile, low, ncls, ghist);
        hipLaunchKernelGGL(msm_p1_scan_kernel<BITS>, dim3(1), dim3(1024), 0, st, (const uint32_t*)ghist, ncls, cursor, bin_off, seg_off);
        for (int b = 0; b < batch; b++)
            hipLaunchKernelGGL((msm_digits_pass1_kernel<FrP, BITS>), dim3((unsigned)p1_blocks), dim3(MSM_P1_THREADS), 0, st, vec(b),
                               (uint64_t)n, scalars_mont ? 1 : 0, c, nwin, win_lo, win_hi, table ? 1 : 0, (uint32_t)b * half, nb, tile, (uint64_t)ntiles, low,
                               ncls, cursor, keys, vals);
        GA_KERNEL_CHECK();
    }
    {
        StageTimer tm(ctx, "msm_sort", st);
        unsigned max_seg = (unsigned)(m / MSM_P2_SEG + BINS);
        if (swz) max_seg = (max_seg + MSM_XCDS - 1) / MSM_XCDS * MSM_XCDS + MSM_XCDS;   // ceil(S / 8) blocks per XCD for any S <= max_seg
        GA_HIP_CHECK(hipMemsetAsync(gcount, 0, nkeys * 4, st));
        hipLaunchKernelGGL(msm_p2_count_kernel<BITS>, dim3(max_seg), dim3(1024), 0, st, (const uint16_t*)keys, (const uint32_t*)seg_off,
                           (const uint32_t*)bin_off, hb, low, swz, gcount);
        GA_KERNEL_CHECK();
        size_t sb = 0;
        void* stmp;
        GA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, gcount, off, (int)(nb + 1), st));
        GA_CHECK(ctx->scratch_get(key("msm_p2_scan_tmp").c_str(), sb + 256, &stmp));
        GA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(stmp, sb, gcount, off, (int)(nb + 1), st));   // off[b], b = 0..nb (nb = SKIP)
        GA_HIP_CHECK(hipMemcpyAsync(kcursor, off, ((uint64_t)nb + 1) * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(msm_p2_scatter_kernel<BITS>, dim3(max_seg), dim3(1024), 0, st, (const uint16_t*)keys, (const uint32_t*)vals,
                           (const uint32_t*)seg_off, (const uint32_t*)bin_off, hb, low, swz, kcursor, vals2);
        GA_KERNEL_CHECK();
    }
    return GA_OK;
}

// ---- 3. bucket boundaries and tasks ---------------------------------------------------------------
// bucket boundaries by binary search + the number of tasks per bucket, one launch (library-sort path: small MSMs, where every launch
// is ~5 us of a ~2 ms call): a block shares its boundaries in LDS
static __global__ void __launch_bounds__(256) msm_offsets_tasks_kernel(const uint32_t* __restrict__ keys, uint64_t m, uint32_t nb, uint32_t seg,
                                                                       uint32_t* __restrict__ off, uint32_t* __restrict__ ntask) {
    __shared__ uint32_t sh[257];
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    auto lower = [&](uint32_t key) {   // first index with keys[idx] >= key
        uint64_t lo = 0, hi = m;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (keys[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        return (uint32_t)lo;
    };
    if (b <= nb) {
        sh[threadIdx.x] = lower(b);
        off[b] = sh[threadIdx.x];
        if (threadIdx.x == blockDim.x - 1 && b < nb) sh[blockDim.x] = lower(b + 1);
    }
    __syncthreads();
    if (b <= nb) ntask[b] = b < nb ? (sh[threadIdx.x + 1] - sh[threadIdx.x] + seg - 1) / seg : 0;
}

static __global__ void msm_tasks_kernel(const uint32_t* __restrict__ off, uint32_t nb, uint32_t seg, uint32_t* __restrict__ ntask) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    uint32_t sz = b < nb ? off[b + 1] - off[b] : 0;
    ntask[b] = (sz + seg - 1) / seg;
}

// task t of bucket b covers sorted pairs [start, start+len); key = SEG - len so that an ascending radix sort puts the
// longest tasks first and lanes of one wave get tasks of (nearly) equal length (bucket sizes are Poisson-distributed:
// without this a wave waits for its longest bucket, ~25 % of the lanes' time at 2^24).  Below 2^25 pairs the SORT key is the length
// quantised to 7 bits (qkey = key >> qshift; lanes of a wave then differ by < 2^qshift points): with the padding bit that is ONE
// 8-bit radix pass over the task list instead of two (2^20 raw MSM: task stage 0.18 -> 0.08 ms, profiles/r05_e); the exact key
// stays in task_key.  From 2^25 pairs up the sort key IS the exact key: the second pass costs ~0.02 ms, lanes that wait for a
// neighbour 1-3 points longer cost the bucket kernel 1.4 % (15.40 -> 15.62 ms at 12 x 2^24 pairs, same box against round 4's
// exact sort, profiles/r05_w_round4_vs_round5_same_box.txt).
// (one launch for what were two memsets and an iota: padding keys, the identity permutation the task sort starts from, the counter
// of the long-bucket queue)
static __global__ void msm_task_init_kernel(uint32_t* __restrict__ task_qkey, uint32_t* __restrict__ task_id, uint32_t n, uint32_t* __restrict__ long_count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        task_qkey[i] = 0xFFFFFFFFu;
        task_id[i] = i;
    }
    if (i == 0) *long_count = 0;
}

// Buckets with more than MSM_LONG_TASKS tasks (a boolean-heavy witness puts millions of points into the digit-1 bucket of window 0)
// are not written by their one lane: they are queued and written by msm_task_list_long_kernel, a block per bucket.
constexpr uint32_t MSM_LONG_TASKS = 64;
static __global__ void msm_task_list_kernel(const uint32_t* __restrict__ off, const uint32_t* __restrict__ task_off, uint32_t nb,
                                            uint32_t seg, int qshift, uint32_t* __restrict__ task_start, uint32_t* __restrict__ task_key,
                                            uint32_t* __restrict__ task_qkey, uint32_t* __restrict__ task_dest, uint32_t* __restrict__ long_list,
                                            uint32_t* __restrict__ long_count) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t t0 = task_off[b], t1 = task_off[b + 1];
    uint32_t start = off[b], end = off[b + 1];
    if (t1 - t0 > MSM_LONG_TASKS) {
        long_list[atomicAdd(long_count, 1u)] = b;
        return;
    }
    for (uint32_t t = t0; t < t1; t++) {
        uint32_t len = end - start < seg ? end - start : seg;
        task_start[t] = start;
        // where the task's sum goes in the array [nb bucket sums | partial sums]: the only task of a bucket writes the bucket
        // sum itself (the merge pass then has nothing to do for that bucket)
        task_dest[t] = (t1 - t0 == 1) ? b : nb + t;
        task_key[t] = seg - len;
        task_qkey[t] = (seg - len) >> qshift;
        start += len;
    }
}

static __global__ void msm_task_list_long_kernel(const uint32_t* __restrict__ off, const uint32_t* __restrict__ task_off, uint32_t nb, uint32_t seg,
                                                 int qshift, uint32_t* __restrict__ task_start, uint32_t* __restrict__ task_key,
                                                 uint32_t* __restrict__ task_qkey, uint32_t* __restrict__ task_dest,
                                                 const uint32_t* __restrict__ long_list, const uint32_t* __restrict__ long_count) {
    const uint32_t nl = *long_count;
    for (uint32_t h = blockIdx.x; h < nl; h += gridDim.x) {
        const uint32_t b = long_list[h];
        const uint32_t t0 = task_off[b], t1 = task_off[b + 1];
        const uint32_t start = off[b], end = off[b + 1];
        for (uint32_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) {   // every task but the last is full
            const uint32_t s0 = start + (t - t0) * seg;
            const uint32_t len = end - s0 < seg ? end - s0 : seg;
            task_start[t] = s0;
            task_dest[t] = nb + t;
            task_key[t] = seg - len;
            task_qkey[t] = (seg - len) >> qshift;
        }
    }
}

// ---- 4. accumulate over a table in the unpacked ("29-bit limb") format ---------------------------------------------------------
// Table entry = hat(x) | hat(y) as NL limbs each (field29.hip.h), padded to a multiple of 16 bytes; (0,0) = infinity.
// The bucket loop runs entirely in the lazy representation: per mixed addition 10 products of 2*NL^2 MADs + one
// shift/mask per column, limb-wise add/sub with a carry sweep, no unpacking, no conditional subtractions.  The
// exceptional cases of the addition law (doubling, P + (-P), accumulator at infinity) are not branched on: they all
// make ZZ == 0 (mod p) and 0 is absorbing, so ONE exact test when the task ends detects them; such tasks are queued for
// msm_accumulate29_redo_kernel, which repeats them with the complete formulas.
template <class F>
struct Table29 {
    static constexpr int NW = Lazy<F>::NW;                 // 32-bit registers per coordinate once unpacked
    // In HBM a table entry is the hat-domain point with each coordinate packed as an ordinary 32N-bit integer: 64 B
    // (BN254 G1, half a cache line, never straddling), 128 B (BN254 G2, one line), 96 / 192 B for BLS12-381.  Storing the
    // limbs unpacked (80 B for BN254 G1) made every third gather touch two lines: FETCH_SIZE 41 GB per 2^24 MSM.
    static constexpr int WORDS = sizeof(Affine<F>) / 4;
    // workgroup size: the LDS-resident accumulators (4*NW words per lane) must leave room for 2 workgroups per CU.  (Measured and
    // removed -- tools/exp/r04_pruned_knobs.patch brings the build knobs back: 128- / 64-lane workgroups for the 14-limb fields,
    // G1 30.41 / 30.92 / 30.44 ms per 2^24 launch, G2 95.6 at 128 lanes / 108.6 at 64, profiles/r03_v_bls_workgroup_size_ab.txt;
    // the next table entry requested one addition ahead and parked in registers: G2 97.1 -> 95.9 ms on BLS12-381, nothing on the
    // other three kernels, profiles/r04_a_prefetch_ab.txt.)
    static constexpr int THREADS = (4 * NW * 4 * 256 <= 72 * 1024) ? 256 : 128;
    static constexpr int MIN_WAVES = Lazy<F>::FP2 ? GA_ACC29_FP2_MINW : GA_ACC29_MINW;
};

// The lane's XYZZ accumulator lives in LDS, word-major (conflict-free): that is what keeps the G1 kernel at 128 VGPRs and four waves
// per SIMD (the accumulator in registers measured slower in round 3: 15.58 vs 15.04 ms per 2^24 launch, profiles/README.md).
template <class F>
struct LdsAcc29 {
    typedef typename Lazy<F>::T T;
    uint32_t* base;
    __device__ __forceinline__ explicit LdsAcc29(uint32_t* b) : base(b) {}
    static constexpr int NW = Lazy<F>::NW, STRIDE = Table29<F>::THREADS;
    __device__ __forceinline__ T get(int field) const {
        T r;
#pragma unroll
        for (int i = 0; i < NW; i++) Lazy<F>::set_word(r, i, base[(field * NW + i) * STRIDE]);
        return r;
    }
    __device__ __forceinline__ void put(int field, const T& v) const {
#pragma unroll
        for (int i = 0; i < NW; i++) base[(field * NW + i) * STRIDE] = Lazy<F>::word(v, i);
    }
};

template <class F>
__device__ __forceinline__ void load_point29(const uint32_t* __restrict__ table, uint32_t idx, typename Lazy<F>::T& x,
                                             typename Lazy<F>::T& y) {
    Affine<F> a = load_pod<Affine<F>>(table + (uint64_t)idx * Table29<F>::WORDS);
    x = Lazy<F>::unpack(a.x);
    y = Lazy<F>::unpack(a.y);
}

// acc += q in the lazy representation (madd-2008-s).  Subtraction constants and partial reductions come from the bound
// analysis in DESIGN.md ("lazy bounds"): G1 keeps every value < 2^257 (BN254) / 2^385 (BLS12-381) with no reduction at
// all; G2 (Karatsuba doubles the operand bounds) additionally applies f29_partial_reduce to P, R, PPP and X3.
template <class P>
__device__ __forceinline__ void madd29(const LdsAcc29<Fe<P>>& A, const F29<P>& qx, const F29<P>& qy) {
    F29<P> zz = A.get(2);
    F29<P> U2 = f29_mul(qx, zz);
    F29<P> ax = A.get(0);
    F29<P> Pp = f29_sub<8>(U2, ax);
    F29<P> zzz = A.get(3);
    F29<P> S2 = f29_mul(qy, zzz);
    F29<P> ay = A.get(1);
    F29<P> R = f29_sub<8>(S2, ay);
    F29<P> PP = f29_sqr(Pp);
    A.put(2, f29_mul(zz, PP));
    F29<P> PPP = f29_mul(Pp, PP);
    A.put(3, f29_mul(zzz, PPP));
    F29<P> Q = f29_mul(ax, PP);
    // X3 = R^2 - (PPP + 2Q): the sum stays un-normalized (limbs < 3*2^L) and is subtracted with a 4-unit loan: one carry
    // sweep instead of three; t = Q - X3 + 8p also stays raw (limbs < 3*2^L): a 2^31-limb multiplicand keeps the two product
    // columns of f29_mul_sub below 2^64
    F29<P> X3 = f29_sub_wide<4, 4>(f29_sqr(R), f29_add_raw(PPP, f29_add_raw(Q, Q)));
    A.put(0, X3);
    A.put(1, f29_mul_sub<8>(R, f29_sub_raw<8>(Q, X3), ay, PPP));   // Y3 = R*(Q - X3) - Y1*PPP, one reduction
}

template <class P>
__device__ __forceinline__ void madd29(const LdsAcc29<Fe2<P>>& A, const F29x2<P>& qx, const F29x2<P>& qy) {
    typedef F29x2<P> T;
    T zz = A.get(2);
    T U2 = f29_mul(qx, zz);
    T ax = A.get(0);
    T Pp = f29_sub<4>(U2, ax);
    T zzz = A.get(3);
    T S2 = f29_mul(qy, zzz);
    T ay = A.get(1);
    T R = f29_sub<4>(S2, ay);
    T PP = f29_sqr(Pp);
    A.put(2, f29_mul(zz, PP));
    T PPP = f29_mul(Pp, PP);
    A.put(3, f29_mul(zzz, PPP));
    T Q = f29_mul(ax, PP);
    T X3 = f29_partial_reduce(f29_sub_wide<4, 4>(f29_sqr(R), f29_add_raw(PPP, f29_add_raw(Q, Q))));
    A.put(0, X3);
    A.put(1, f29_mul_sub<P::FP2Z_K>(R, f29_sub<8>(Q, X3), ay, PPP));   // Y3 = R*(Q - X3) - Y1*PPP, two reductions instead of four
}

// acc = 2*(qx, qy) for an affine q in the lazy representation (mdbl-2008-s-1, a = 0); qy may be a negated 2p - y.  Bounds
// (tools/lazy_bounds.py check_mdbl): every output stays below the fixed-point bounds of the accumulator coordinates of madd29.
template <class F>
__device__ __forceinline__ void mdbl29(const LdsAcc29<F>& A, const typename Lazy<F>::T& qx, const typename Lazy<F>::T& qy) {
    typedef typename Lazy<F>::T T;
    typedef typename Lazy<F>::Params P;
    const T U = f29_add(qy, qy);
    const T V = f29_sqr(U);
    const T W = f29_mul(U, V);
    const T S = f29_mul(qx, V);
    const T xx = f29_sqr(qx);
    const T M = f29_add(f29_add(xx, xx), xx);
    T X3 = f29_sub<4>(f29_sqr(M), f29_add(S, S));
    if constexpr (Lazy<F>::FP2) X3 = f29_partial_reduce(X3);
    constexpr int KMS = Lazy<F>::FP2 ? P::FP2Z_K : 8;
    A.put(1, f29_mul_sub<KMS>(M, f29_sub<8>(S, X3), W, qy));   // Y3 = M*(S - X3) - W*y
    A.put(0, X3);
    A.put(2, V);
    A.put(3, W);
}

// acc += q with the exceptional cases of the addition law handled: same x and same y -> doubling, same x and opposite y -> the
// accumulator becomes the point at infinity (returns false: the caller restarts it with the next point).  One exact zero test of
// P = X2*ZZ1 - X1 per addition (~80 instructions on top of the ~2400 of madd29); R is only tested when P vanishes.
template <class F>
__device__ __forceinline__ bool madd29_complete(const LdsAcc29<F>& A, const typename Lazy<F>::T& qx, const typename Lazy<F>::T& qy) {
    typedef typename Lazy<F>::T T;
    typedef typename Lazy<F>::Params P;
    constexpr int KS = Lazy<F>::FP2 ? 4 : 8;
    const T Pp = f29_sub<KS>(f29_mul(qx, A.get(2)), A.get(0));
    if (f29_is_zero_mod_p(Pp)) {
        const T R = f29_sub<KS>(f29_mul(qy, A.get(3)), A.get(1));
        if (!f29_is_zero_mod_p(R)) return false;
        mdbl29<F>(A, qx, qy);
        return true;
    }
    madd29<P>(A, qx, qy);   // (recomputes P: the common path stays the code the bound analysis covers)
    return true;
}

// one task = the sorted pairs [start, end): its sum into the lane's LDS accumulator; returns whether the sum is a finite point
template <class F, bool COMPLETE>
__device__ __forceinline__ bool accumulate_task29(const LdsAcc29<F>& A, const uint32_t* __restrict__ table, const uint32_t* __restrict__ vals,
                                                  uint32_t start, uint32_t end) {
    typedef typename Lazy<F>::T T;
    typedef typename Lazy<F>::Params P;
    const T one = Lazy<F>::from_mem(FieldTraits<F>::one());
    bool have = false;
    uint32_t v = vals[start];
    uint32_t vn = v;
    for (uint32_t p = start; p < end; p++) {
        T qx, qy;
        vn = p + 1 < end ? vals[p + 1] : v;
        load_point29<F>(table, v & ~MSM_SIGN, qx, qy);
        if (!(f29_is_zero_limbs(qx) & f29_is_zero_limbs(qy))) {   // (0,0) = infinity: skip
            if (v & MSM_SIGN) qy = f29_sub<2>(Lazy<F>::from_mem(FieldTraits<F>::zero()), qy);   // 2p - y
            if (!have) {
                A.put(0, qx);
                A.put(1, qy);
                A.put(2, one);
                A.put(3, one);
                have = true;
            } else if constexpr (COMPLETE) {
                have = madd29_complete<F>(A, qx, qy);
            } else {
                madd29<P>(A, qx, qy);
            }
        }
        v = vn;
    }
    return have;
}

// the task's sum out of the LDS accumulator: false when an exceptional addition slipped through (ZZ == 0 mod p)
template <class F>
__device__ __forceinline__ bool store_task29(const LdsAcc29<F>& A, bool have, XYZZ<F>* __restrict__ dst) {
    XYZZ<F> acc = xyzz_inf<F>();
    if (have) {
        F zz = Lazy<F>::to_mem(A.get(2));
        if (is_zero(zz)) return false;
        acc.x = Lazy<F>::to_mem(A.get(0));
        acc.y = Lazy<F>::to_mem(A.get(1));
        acc.zz = zz;
        acc.zzz = Lazy<F>::to_mem(A.get(3));
    }
    store_pod(dst, acc);
    return true;
}

// COMPLETE = false: the fast loop (exceptional additions make ZZ == 0 and flag the task); true: the same loop with the exceptional
// cases handled in place -- used directly on tables that turned out degenerate (a DummySetup key: every base the same point).
// Multi-table pass (the Groth16 witness MSMs A, B1, K: ONE scalar vector, k wire-indexed tables of the same shape): blockIdx.y is
// the table; its sums live in the table's own slice of [k x nb bucket sums | k x max_tasks partial sums] and its flagged tasks in its
// own redo lists.  A single-table launch is the case k = 1, y = 0 of the same arithmetic.
struct MsmTables {
    const uint32_t* t[4];
    uint32_t k, nb, max_tasks;
};
__device__ __forceinline__ uint32_t msm_multi_dest(const MsmTables& mt, uint32_t dest) {
    return dest < mt.nb ? dest + blockIdx.y * mt.nb : dest + (mt.k - 1) * mt.nb + blockIdx.y * mt.max_tasks;
}

#ifdef GA_ACC29_NUM_VGPR   // (compile-time experiment: FORCE that many waves per SIMD on the bucket kernel -- the allocator must spill to get there; tools/exp/r06_bls_g2_two_waves.sh)
#define GA_ACC29_VGPR_ATTR __attribute__((amdgpu_waves_per_eu(GA_ACC29_NUM_VGPR, GA_ACC29_NUM_VGPR)))
#else
#define GA_ACC29_VGPR_ATTR
#endif
template <class F, bool COMPLETE>
__global__ void __launch_bounds__(Table29<F>::THREADS, Table29<F>::MIN_WAVES) GA_ACC29_VGPR_ATTR
msm_accumulate29_kernel(const MsmTables mt, const uint32_t* __restrict__ vals,
                        const uint32_t* __restrict__ task_start, const uint32_t* __restrict__ task_qkey_sorted,
                        const uint32_t* __restrict__ task_key_by_tid, const uint32_t* __restrict__ task_perm, uint32_t max_tasks, uint32_t seg,
                        const uint32_t* __restrict__ task_dest, XYZZ<F>* __restrict__ sums, uint32_t* __restrict__ redo_list,
                        uint32_t* __restrict__ redo_count) {
    constexpr int NW = Lazy<F>::NW;
    __shared__ uint32_t lds[4 * NW * Table29<F>::THREADS];
    // (Round 3 measured two ways of making room for kernels of the partner lane beside this one -- which fills 144 of the 160 KB
    // of LDS of a CU: the accumulator in registers instead of LDS (15.58 vs 15.04 ms, slower) and a cap of 3 resident waves per
    // SIMD through the register allocation (proof time unchanged, 139.4 vs 139.9 ms).  Neither stays.  Round 4: a resident grid
    // striding over the task list instead of one task per lane is slower on all four kernels, tools/exp/r04_resident_bucket_grid.patch,
    // profiles/r04_e_resident_bucket_grid_ab.txt.)
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= max_tasks) return;
    if (task_qkey_sorted[t] == 0xFFFFFFFFu) return;   // padding slot of the task list
    const uint32_t tid = task_perm[t];
    const uint32_t key = task_key_by_tid[tid];
    const uint32_t start = task_start[tid];
    LdsAcc29<F> A(lds + threadIdx.x);
    const bool have = accumulate_task29<F, COMPLETE>(A, mt.t[blockIdx.y], vals, start, start + (seg - key));
    if (!store_task29<F>(A, have, &sums[msm_multi_dest(mt, task_dest[tid])]))   // redo it
        redo_list[(uint64_t)blockIdx.y * (mt.max_tasks + 2) + atomicAdd(redo_count + 2 * blockIdx.y, 1u)] = tid;
}

// second chance for the tasks the fast loop flagged: the complete lazy loop over the redo list (grid-stride); what even that
// cannot finish (a base of order 2, never on these curves) goes to the exact kernel below through a second list
template <class F>
__global__ void __launch_bounds__(Table29<F>::THREADS, Table29<F>::MIN_WAVES)
msm_accumulate29_retry_kernel(const MsmTables mt, const uint32_t* __restrict__ vals,
                              const uint32_t* __restrict__ task_start, const uint32_t* __restrict__ task_key_by_tid, uint32_t seg,
                              const uint32_t* __restrict__ redo_list, const uint32_t* __restrict__ redo_count,
                              const uint32_t* __restrict__ task_dest, XYZZ<F>* __restrict__ sums, uint32_t* __restrict__ redo2_list,
                              uint32_t* __restrict__ redo2_count) {
    constexpr int NW = Lazy<F>::NW;
    __shared__ uint32_t lds[4 * NW * Table29<F>::THREADS];
    // (the lists and counters of table y: see msm_accumulate29_kernel)
    redo_list += (uint64_t)blockIdx.y * (mt.max_tasks + 2);
    redo2_list += (uint64_t)blockIdx.y * (mt.max_tasks + 2);
    const uint32_t nredo = redo_count[2 * blockIdx.y];
    LdsAcc29<F> A(lds + threadIdx.x);
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nredo; r += gridDim.x * blockDim.x) {
        const uint32_t tid = redo_list[r];
        const uint32_t start = task_start[tid];
        const bool have = accumulate_task29<F, true>(A, mt.t[blockIdx.y], vals, start, start + (seg - task_key_by_tid[tid]));
        if (!store_task29<F>(A, have, &sums[msm_multi_dest(mt, task_dest[tid])])) redo2_list[atomicAdd(redo2_count + 2 * blockIdx.y, 1u)] = tid;
    }
}

// exact re-run of the tasks the lazy kernel flagged (complete formulas; table points converted back to gnark's form)
template <class F>
__global__ void __launch_bounds__(64)
msm_accumulate29_redo_kernel(const MsmTables mt, const uint32_t* __restrict__ vals,
                             const uint32_t* __restrict__ task_start, const uint32_t* __restrict__ task_key_by_tid,
                             uint32_t seg, const uint32_t* __restrict__ redo_list, const uint32_t* __restrict__ redo_count,
                             const uint32_t* __restrict__ task_dest, XYZZ<F>* __restrict__ sums) {
    typedef typename Lazy<F>::T T;
    redo_list += (uint64_t)blockIdx.y * (mt.max_tasks + 2);
    const uint32_t nredo = redo_count[2 * blockIdx.y];
    const uint32_t* __restrict__ table = mt.t[blockIdx.y];
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nredo; r += gridDim.x * blockDim.x) {
        const uint32_t tid = redo_list[r];
        const uint32_t start = task_start[tid];
        const uint32_t end = start + (seg - task_key_by_tid[tid]);
        XYZZ<F> acc = xyzz_inf<F>();
        for (uint32_t p = start; p < end; p++) {
            const uint32_t v = vals[p];
            T qx, qy;
            load_point29<F>(table, v & ~MSM_SIGN, qx, qy);
            Affine<F> q{Lazy<F>::to_mem(qx), Lazy<F>::to_mem(qy)};
            if (v & MSM_SIGN) q.y = neg(q.y);
            acc = madd(acc, q);
        }
        store_pod(&sums[msm_multi_dest(mt, task_dest[tid])], acc);
    }
}

template <class F> struct BaseFieldOf;
template <class Pp> struct BaseFieldOf<Fe<Pp>> { typedef Pp P; static constexpr bool IS_FP = true; };
template <class Pp> struct BaseFieldOf<Fe2<Pp>> { typedef Pp P; static constexpr bool IS_FP = false; };

// (msm_table29_kernel follows the lazy point helpers below)

// ---- 5. merge partials (msm_merge_kernel itself follows the lazy helpers it uses, below) --------------
// 64-lane tree reduction through LDS; result valid in lane 0
template <class F>
__device__ __forceinline__ XYZZ<F> wave_tree_sum(XYZZ<F> acc, XYZZ<F>* sh) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t stride = 32; stride >= 1; stride >>= 1) {
        sh[lane] = acc;
        __syncthreads();
        if (lane < stride) acc = add(acc, sh[lane + stride]);
        __syncthreads();
    }
    return acc;
}

// (msm_hot_kernel and the very-hot-bucket kernels follow block_sum29 below)

// ---- 6. window reduction ----------------------------------------------------------------------------
// group g of window w covers digits k in [g*m+1, (g+1)*m]; out = sum_k k*B_k over the group
template <class F>
__global__ void __launch_bounds__(64)
msm_reduce_groups_kernel(const XYZZ<F>* __restrict__ bsum, uint32_t half, uint32_t m, uint32_t groups_per_win,
                         uint32_t total_groups, XYZZ<F>* __restrict__ gsum) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_groups) return;
    uint32_t w = gid / groups_per_win, g = gid % groups_per_win;
    const XYZZ<F>* B = bsum + (uint64_t)w * half + (uint64_t)g * m;   // B[j] = bucket of digit g*m + j + 1
    XYZZ<F> running = xyzz_inf<F>(), local = xyzz_inf<F>();
    for (int j = (int)m - 1; j >= 0; j--) {
        running = add(running, load_pod<XYZZ<F>>(&B[j]));
        local = add(local, running);
    }
    uint32_t base = g * m;
    if (base != 0) {
        // local += base * running
        XYZZ<F> r = xyzz_inf<F>();
        int top = 31 - __clz(base);
        for (int bit = top; bit >= 0; bit--) {
            r = dbl(r);
            if ((base >> bit) & 1) r = add(r, running);
        }
        local = add(local, r);
    }
    store_pod(&gsum[gid], local);
}

// ---- the same pass in the lazy representation ---------------------------------------------------------------------------
// General XYZZ + XYZZ addition (add-2008-s) on unreduced limbs: 14 limb products with 13 reductions (Y3 fused), no modular
// corrections.  Exceptional inputs (equal or opposite points) are NOT handled: they make ZZ3 = 0 (mod p), which sticks to every
// later sum, so the caller tests ZZ once at the end and falls back to the exact kernel.  Constants: tools/lazy_bounds.py
// check_add.
template <class F>
struct Lazy4 {
    typename Lazy<F>::T x, y, zz, zzz;
};
template <class F>
__device__ __forceinline__ Lazy4<F> lazy4_from_mem(const XYZZ<F>& p) {
    return {Lazy<F>::from_mem(p.x), Lazy<F>::from_mem(p.y), Lazy<F>::from_mem(p.zz), Lazy<F>::from_mem(p.zzz)};
}
template <class F>
__device__ __forceinline__ void add29(Lazy4<F>& a, const Lazy4<F>& b) {
    typedef typename Lazy<F>::T T;
    typedef typename Lazy<F>::Params P;
    constexpr int KMS = Lazy<F>::FP2 ? P::FP2Z_K : 8;
    T U1 = f29_mul(a.x, b.zz);
    T U2 = f29_mul(b.x, a.zz);
    T S1 = f29_mul(a.y, b.zzz);
    T S2 = f29_mul(b.y, a.zzz);
    T Pp = f29_sub<4>(U2, U1);
    T R = f29_sub<4>(S2, S1);
    T PP = f29_sqr(Pp);
    T PPP = f29_mul(Pp, PP);
    T Q = f29_mul(U1, PP);
    T X3 = f29_sub<4>(f29_sqr(R), f29_add(PPP, f29_add(Q, Q)));
    if constexpr (Lazy<F>::FP2) X3 = f29_partial_reduce(X3);
    a.y = f29_mul_sub<KMS>(R, f29_sub<8>(Q, X3), S1, PPP);
    a.x = X3;
    a.zz = f29_mul(f29_mul(a.zz, b.zz), PP);
    a.zzz = f29_mul(f29_mul(a.zzz, b.zzz), PPP);
}

// lsum[g] = sum_j (j+1)*B_j and rsum[g] = sum_j B_j over the m buckets of group g (no scalar multiplication: the term
// sum_g (g*m)*rsum[g] is assembled from per-bit tree sums, msm_bit_partial_kernel).  Groups in which an exceptional addition
// occurred (e.g. local + running when they are the same point because a bucket was empty) are appended to redo_list.
template <class F>
__global__ void __launch_bounds__(64)
msm_reduce_groups29_kernel(const XYZZ<F>* __restrict__ bsum, uint32_t half, uint32_t m, uint32_t groups_per_win,
                           uint32_t total_groups, XYZZ<F>* __restrict__ lsum, XYZZ<F>* __restrict__ rsum,
                           uint32_t* __restrict__ redo_list, uint32_t* __restrict__ redo_count) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_groups) return;
    uint32_t w = gid / groups_per_win, g = gid % groups_per_win;
    const XYZZ<F>* B = bsum + (uint64_t)w * half + (uint64_t)g * m;   // B[j] = bucket of digit g*m + j + 1
    Lazy4<F> running, local;
    bool r_inf = true, l_inf = true;
    for (int j = (int)m - 1; j >= 0; j--) {
        XYZZ<F> b = load_pod<XYZZ<F>>(&B[j]);
        if (!is_inf(b)) {
            Lazy4<F> lb = lazy4_from_mem<F>(b);
            if (r_inf) {
                running = lb;
                r_inf = false;
            } else {
                add29<F>(running, lb);
            }
        }
        if (!r_inf) {
            if (l_inf) {
                local = running;
                l_inf = false;
            } else {
                add29<F>(local, running);
            }
        }
    }
    XYZZ<F> lo = xyzz_inf<F>(), ro = xyzz_inf<F>();
    bool bad = false;
    if (!r_inf) {
        ro.zz = Lazy<F>::to_mem(running.zz);
        lo.zz = Lazy<F>::to_mem(local.zz);
        bad = is_zero(ro.zz) | is_zero(lo.zz);
        ro.x = Lazy<F>::to_mem(running.x);
        ro.y = Lazy<F>::to_mem(running.y);
        ro.zzz = Lazy<F>::to_mem(running.zzz);
        lo.x = Lazy<F>::to_mem(local.x);
        lo.y = Lazy<F>::to_mem(local.y);
        lo.zzz = Lazy<F>::to_mem(local.zzz);
    }
    if (bad) {
        redo_list[atomicAdd(redo_count, 1u)] = gid;
        return;
    }
    store_pod(&lsum[gid], lo);
    store_pod(&rsum[gid], ro);
}

// exact re-run (complete formulas) of the groups the lazy kernel flagged
template <class F>
__global__ void __launch_bounds__(64)
msm_reduce_groups_redo_kernel(const XYZZ<F>* __restrict__ bsum, uint32_t half, uint32_t m, uint32_t groups_per_win,
                              const uint32_t* __restrict__ redo_list, const uint32_t* __restrict__ redo_count,
                              XYZZ<F>* __restrict__ lsum, XYZZ<F>* __restrict__ rsum) {
    const uint32_t nredo = *redo_count;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nredo; r += gridDim.x * blockDim.x) {
        const uint32_t gid = redo_list[r];
        uint32_t w = gid / groups_per_win, g = gid % groups_per_win;
        const XYZZ<F>* B = bsum + (uint64_t)w * half + (uint64_t)g * m;
        XYZZ<F> running = xyzz_inf<F>(), local = xyzz_inf<F>();
        for (int j = (int)m - 1; j >= 0; j--) {
            running = add(running, load_pod<XYZZ<F>>(&B[j]));
            local = add(local, running);
        }
        store_pod(&lsum[gid], local);
        store_pod(&rsum[gid], running);
    }
}

// ---- precomputed tables: table29[w*n + i] = [2^(c*w)] P_i in the packed hat format --------------------------------------------
// 2 P for a general XYZZ point in the lazy representation (dbl-2008-s-1, a = 0); bounds: tools/lazy_bounds.py check_dbl (the fixed
// point of repeated doublings, the same subtraction constants as mdbl29)
template <class F>
__device__ __forceinline__ void dbl29(Lazy4<F>& a) {
    typedef typename Lazy<F>::T T;
    typedef typename Lazy<F>::Params P;
    constexpr int KMS = Lazy<F>::FP2 ? P::FP2Z_K : 8;
    const T U = f29_add(a.y, a.y);
    const T V = f29_sqr(U);
    const T W = f29_mul(U, V);
    const T S = f29_mul(a.x, V);
    const T xx = f29_sqr(a.x);
    const T M = f29_add(f29_add(xx, xx), xx);
    T X3 = f29_sub<4>(f29_sqr(M), f29_add(S, S));
    if constexpr (Lazy<F>::FP2) X3 = f29_partial_reduce(X3);
    const T Y3 = f29_mul_sub<KMS>(M, f29_sub<8>(S, X3), W, a.y);
    a.zz = f29_mul(V, a.zz);
    a.zzz = f29_mul(W, a.zzz);
    a.x = X3;
    a.y = Y3;
}

// A lane carries TableBatch<F>::K points through the doubling chain together, in the lazy representation (the chain is 22 doublings
// per window step: 9 products each, no reductions in between), and brings them back to affine with ONE field inversion per window
// step (Montgomery's trick on zz*zzz; lanes of a wave cannot share one -- SIMD: 64 inversions cost what one costs -- so the batch
// is inside the lane).  The affine coordinates leave the lane as canonical packed hat-domain words: the table's storage format.
// History (2^22 points, kernel time): exact arithmetic, one point per lane 0.248 s (BN254 G1) / 0.835 s (BLS12-381 G1) / 0.627 s
// (BN254 G2); exact arithmetic with 8 / 2 points per lane 0.158 / 0.392 / 0.594 s; this version: see profiles/r02_h notes.
template <class F> struct TableBatch { static constexpr int K = BaseFieldOf<F>::IS_FP ? (BaseFieldOf<F>::P::N <= 8 ? 4 : 2) : (BaseFieldOf<F>::P::N <= 8 ? 2 : 1); };

template <class F>
__global__ void __launch_bounds__(64)
msm_table29_kernel(const Affine<F>* __restrict__ bases, uint64_t n, int c, int nwin, uint32_t* __restrict__ table) {
    typedef typename Lazy<F>::T T;
    constexpr int K = TableBatch<F>::K;
    const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;   // point q of a lane is gid + q*lanes: nothing to do when even q = 0 is out of range
    const T one = Lazy<F>::from_mem(FieldTraits<F>::one());
    Lazy4<F> p[K];
    bool live[K], inf[K];
#pragma unroll
    for (int q = 0; q < K; q++) {
        const uint64_t i = gid + q * lanes;
        live[q] = i < n;
        Affine<F> a = live[q] ? load_pod<Affine<F>>(&bases[i]) : Affine<F>{FieldTraits<F>::zero(), FieldTraits<F>::zero()};
        inf[q] = is_inf(a);
        p[q].x = Lazy<F>::from_mem(a.x);
        p[q].y = Lazy<F>::from_mem(a.y);
        p[q].zz = one;
        p[q].zzz = one;
    }
    for (int w = 0; w < nwin; w++) {
        if (w > 0) {
#pragma unroll
            for (int q = 0; q < K; q++)
                if (!inf[q])
                    for (int k = 0; k < c; k++) dbl29<F>(p[q]);
            // batch to affine: t_q = zz_q * zzz_q (1 for a point at infinity, which stays (0,0)), one inversion of their product
            T t[K], pre[K];
#pragma unroll
            for (int q = 0; q < K; q++) {
                t[q] = inf[q] ? one : f29_mul(p[q].zz, p[q].zzz);
                pre[q] = q == 0 ? t[0] : f29_mul(pre[q - 1], t[q]);
            }
            T run = f29_inv(pre[K - 1]);
#pragma unroll
            for (int q = K - 1; q >= 0; q--) {
                const T it = q > 0 ? f29_mul(run, pre[q - 1]) : run;   // 1 / t_q
                if (q > 0) run = f29_mul(run, t[q]);
                if (!inf[q]) {
                    p[q].x = f29_mul(p[q].x, f29_mul(it, p[q].zzz));   // X / zz
                    p[q].y = f29_mul(p[q].y, f29_mul(it, p[q].zz));    // Y / zzz
                    p[q].zz = one;
                    p[q].zzz = one;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < K; q++) {
            if (!live[q]) continue;
            Affine<F> h{FieldTraits<F>::zero(), FieldTraits<F>::zero()};
            if (!inf[q]) {
                h.x = f29_pack_hat(p[q].x);
                h.y = f29_pack_hat(p[q].y);
                // restart the chain from the canonical coordinates: keeps the doublings' inputs at their smallest
                p[q].x = Lazy<F>::unpack(h.x);
                p[q].y = Lazy<F>::unpack(h.y);
            }
            store_pod(table + ((uint64_t)w * n + gid + q * lanes) * Table29<F>::WORDS, h);
        }
    }
}

// Un-pinned bases -> the packed hat format the bucket kernel gathers (a one-window "table"): S modular doublings per coordinate,
// one point per lane.  (Round 1-4 ran msm_table29_kernel with a single window for this -- a kernel shaped for chains of doublings,
// one wave per block: 83 us for 2^20 points, 1.3 ms for 2^24, of an HBM-bound conversion.)
template <class F>
__global__ void __launch_bounds__(256)
msm_hat_bases_kernel(const Affine<F>* __restrict__ bases, uint64_t n, uint32_t* __restrict__ hat) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> a = load_pod<Affine<F>>(&bases[i]);
    if (!is_inf(a)) {   // (0,0) = infinity stays (0,0): the bucket kernel skips it
        if constexpr (BaseFieldOf<F>::IS_FP) {
            a.x = f29_hat_packed(a.x);
            a.y = f29_hat_packed(a.y);
        } else {
            a.x = {f29_hat_packed(a.x.c0), f29_hat_packed(a.x.c1)};
            a.y = {f29_hat_packed(a.y.c0), f29_hat_packed(a.y.c1)};
        }
    }
    store_pod(hat + i * Table29<F>::WORDS, a);
}

// ---- block-wide sums of XYZZ points in the lazy representation ---------------------------------------------------------------
// The tree sums after the group pass (per-bit sums, segment sums) are LATENCY-bound: one wave per block adds a handful of points
// serially and then walks a 6-level tree, every step an addition in the exact packed arithmetic (~20 us G1, ~60 us G2 per dependent
// addition).  In the lazy representation a dependent addition is ~3x shorter.  Exceptional additions (equal or opposite points: never
// for sums of distinct random buckets, always for a degenerate key) leave ZZ == 0; the block then repeats its sum exactly.
template <class F>
struct LazyPt {
    Lazy4<F> v;
    uint32_t inf;
};
template <class F>
__device__ __forceinline__ void lazy_acc(LazyPt<F>& acc, const XYZZ<F>& p) {
    if (is_inf(p)) return;
    const Lazy4<F> b = lazy4_from_mem<F>(p);
    if (acc.inf) {
        acc.v = b;
        acc.inf = 0;
    } else {
        add29<F>(acc.v, b);
    }
}
template <class F>
__device__ __forceinline__ void lazy_acc(LazyPt<F>& acc, const LazyPt<F>& b) {
    if (b.inf) return;
    if (acc.inf) acc = b;
    else add29<F>(acc.v, b.v);
}
// sum over the block's 64 lanes (result in lane 0) of the points src(i), i = lane, lane + 64, ... < count; written to *dst.
// src(i) returns a pointer to the i-th input of this block.
template <class F, class Src>
__device__ __forceinline__ void block_sum29(uint32_t count, Src src, XYZZ<F>* dst, LazyPt<F>* sh, XYZZ<F>* shx, uint32_t* bad) {
    const uint32_t lane = threadIdx.x;
    LazyPt<F> acc;
    acc.inf = 1;
    for (uint32_t i = lane; i < count; i += 64) lazy_acc<F>(acc, load_pod<XYZZ<F>>(src(i)));
    for (uint32_t stride = 32; stride >= 1; stride >>= 1) {
        sh[lane] = acc;
        __syncthreads();
        if (lane < stride) lazy_acc<F>(acc, sh[lane + stride]);
        __syncthreads();
    }
    if (lane == 0) {
        XYZZ<F> out = xyzz_inf<F>();
        uint32_t b = 0;
        if (!acc.inf) {
            out.zz = Lazy<F>::to_mem(acc.v.zz);
            b = is_zero(out.zz) ? 1u : 0u;
            out.x = Lazy<F>::to_mem(acc.v.x);
            out.y = Lazy<F>::to_mem(acc.v.y);
            out.zzz = Lazy<F>::to_mem(acc.v.zzz);
        }
        if (!b) store_pod(dst, out);
        *bad = b;
    }
    __syncthreads();
    if (*bad) {   // an exceptional addition somewhere in this block's sum: once more with the complete formulas
        XYZZ<F> e = xyzz_inf<F>();
        for (uint32_t i = lane; i < count; i += 64) e = add(e, load_pod<XYZZ<F>>(src(i)));
        e = wave_tree_sum(e, shx);
        if (lane == 0) store_pod(dst, e);
    }
    __syncthreads();
}

// hot buckets (17..512 partial sums): one block per bucket
template <class F>
__global__ void __launch_bounds__(64)
msm_hot_kernel(const XYZZ<F>* __restrict__ partial, const uint32_t* __restrict__ task_off,
               const uint32_t* __restrict__ hot_list, const uint32_t* __restrict__ hot_count, XYZZ<F>* __restrict__ bsum,
               uint32_t bsum_stride, uint32_t part_stride) {
    __shared__ LazyPt<F> sh[64];
    __shared__ XYZZ<F> shx[64];
    __shared__ uint32_t bad;
    partial += (uint64_t)blockIdx.y * part_stride;   // (multi-table pass: table y's slices; the lists are the same for every table)
    bsum += (uint64_t)blockIdx.y * bsum_stride;
    const uint32_t nh = *hot_count;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t b = hot_list[h];
        const uint32_t t0 = task_off[b], t1 = task_off[b + 1];
        block_sum29<F>(t1 - t0, [&](uint32_t i) { return &partial[t0 + i]; }, &bsum[b], sh, shx, &bad);
    }
}

// Very hot buckets (thousands of partial sums: the digit-1 bucket of a boolean-heavy witness): stage 1 gives each of
// MSM_VHOT_SPLIT blocks a contiguous share of the bucket's partials (64 lanes strided + LDS tree), stage 2 sums the
// MSM_VHOT_SPLIT block results of a bucket.  One block per bucket (msm_hot_kernel) would add n/2/seg/64 partials serially per lane:
// measured 4.1 ms (G1) / 16.3 ms (G2) of merge at 2^24 with half the scalars equal to one.
template <class F>
__global__ void __launch_bounds__(64)
msm_vhot_stage1_kernel(const XYZZ<F>* __restrict__ partial, const uint32_t* __restrict__ task_off, const uint32_t* __restrict__ vhot_list,
                       const uint32_t* __restrict__ vhot_count, XYZZ<F>* __restrict__ vtmp, uint32_t part_stride, uint32_t vtmp_stride) {
    __shared__ LazyPt<F> sh[64];
    __shared__ XYZZ<F> shx[64];
    __shared__ uint32_t bad;
    partial += (uint64_t)blockIdx.y * part_stride;
    vtmp += (uint64_t)blockIdx.y * vtmp_stride;
    const uint32_t items = *vhot_count * MSM_VHOT_SPLIT;
    for (uint32_t id = blockIdx.x; id < items; id += gridDim.x) {
        const uint32_t h = id / MSM_VHOT_SPLIT, part = id % MSM_VHOT_SPLIT;
        const uint32_t b = vhot_list[h];
        const uint32_t t0 = task_off[b], t1 = task_off[b + 1];
        const uint32_t per = (t1 - t0 + MSM_VHOT_SPLIT - 1) / MSM_VHOT_SPLIT;
        const uint32_t lo = t0 + part * per < t1 ? t0 + part * per : t1;
        const uint32_t hi = lo + per < t1 ? lo + per : t1;
        block_sum29<F>(hi - lo, [&](uint32_t i) { return &partial[lo + i]; }, &vtmp[id], sh, shx, &bad);
    }
}
template <class F>
__global__ void __launch_bounds__(64)
msm_vhot_stage2_kernel(const XYZZ<F>* __restrict__ vtmp, const uint32_t* __restrict__ vhot_list, const uint32_t* __restrict__ vhot_count,
                       XYZZ<F>* __restrict__ bsum, uint32_t bsum_stride, uint32_t vtmp_stride) {
    static_assert(MSM_VHOT_SPLIT == 64, "one partial result per lane");
    __shared__ LazyPt<F> sh[64];
    __shared__ XYZZ<F> shx[64];
    __shared__ uint32_t bad;
    vtmp += (uint64_t)blockIdx.y * vtmp_stride;
    bsum += (uint64_t)blockIdx.y * bsum_stride;
    const uint32_t nv = *vhot_count;
    for (uint32_t h = blockIdx.x; h < nv; h += gridDim.x)
        block_sum29<F>(MSM_VHOT_SPLIT, [&](uint32_t i) { return &vtmp[h * MSM_VHOT_SPLIT + i]; }, &bsum[vhot_list[h]], sh, shx, &bad);
}

// ---- 5. merge partials ----------------------------------------------------------------------------
// one lane per bucket: nothing to do for single-task buckets, a serial sum of the 2..16 partial sums (lazy representation: the lane
// is latency-bound on dependent additions; exact re-run by the same lane in the exceptional case), the hot lists for the rest
template <class F>
__global__ void msm_merge_kernel(const XYZZ<F>* __restrict__ partial, const uint32_t* __restrict__ task_off, uint32_t nb,
                                 XYZZ<F>* __restrict__ bsum, uint32_t* __restrict__ hot_list, uint32_t* __restrict__ hot_count,
                                 uint32_t* __restrict__ vhot_list, uint32_t* __restrict__ vhot_count, uint32_t part_stride) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    // multi-table pass: table y's slices; which buckets are hot depends on the task list alone, so table 0's blocks write the lists
    partial += (uint64_t)blockIdx.y * part_stride;
    bsum += (uint64_t)blockIdx.y * nb;
    uint32_t t0 = task_off[b], t1 = task_off[b + 1];
    uint32_t nt = t1 - t0;
    if (nt == 1) return;   // its only task wrote bsum[b] directly (task_dest)
    if (nt > MSM_VHOT_TASKS) {
        if (blockIdx.y == 0) vhot_list[atomicAdd(vhot_count, 1u)] = b;
        return;
    }
    if (nt > MSM_HOT_TASKS) {
        if (blockIdx.y == 0) hot_list[atomicAdd(hot_count, 1u)] = b;
        return;
    }
    XYZZ<F> out = xyzz_inf<F>();
    bool exact = true;
    // (measured: the lazy sum pays for the 254-bit field -- merge 0.24 -> 0.17 ms G1 -- and loses for the 381-bit one, where the
    // eight conversions of a 14-limb point outweigh ten shorter additions: 0.23 -> 0.27 ms)
    if constexpr (BaseFieldOf<F>::P::N <= 8) {
        LazyPt<F> acc;
        acc.inf = 1;
        for (uint32_t t = t0; t < t1; t++) lazy_acc<F>(acc, load_pod<XYZZ<F>>(&partial[t]));
        exact = false;
        if (!acc.inf) {
            out.zz = Lazy<F>::to_mem(acc.v.zz);
            exact = is_zero(out.zz);   // an exceptional addition: once more with the complete formulas
            out.x = Lazy<F>::to_mem(acc.v.x);
            out.y = Lazy<F>::to_mem(acc.v.y);
            out.zzz = Lazy<F>::to_mem(acc.v.zzz);
        }
    }
    if (exact) {
        out = xyzz_inf<F>();
        for (uint32_t t = t0; t < t1; t++) out = add(out, load_pod<XYZZ<F>>(&partial[t]));
    }
    store_pod(&bsum[b], out);
}

// part[((w*nbits + b)*chunks + ch)] = sum of rsum[w][g] over the groups g of chunk ch (chunk_len groups, a power of two)
// whose index has bit b set.  grid = (chunks, nbits, nsets), one wave per block.
template <class F>
__global__ void __launch_bounds__(64)
msm_bit_partial_kernel(const XYZZ<F>* __restrict__ rsum, const XYZZ<F>* __restrict__ lsum, uint32_t groups_per_win,
                       uint32_t chunk_len, int log_chunk, XYZZ<F>* __restrict__ part) {
    // blockIdx.y == nbits - 1 (the last row of the grid) is not a bit: it sums the chunk of lsum, so that one launch and one
    // final segment sum produce every quantity the host needs
    __shared__ LazyPt<F> sh[64];
    __shared__ XYZZ<F> shx[64];
    __shared__ uint32_t bad;
    const uint32_t ch = blockIdx.x, b = blockIdx.y, w = blockIdx.z;
    const uint32_t nbits = gridDim.y, chunks = gridDim.x;
    const uint32_t base = ch * chunk_len;
    const XYZZ<F>* R = rsum + (uint64_t)w * groups_per_win;
    XYZZ<F>* dst = &part[((uint64_t)w * nbits + b) * chunks + ch];
    if (b == nbits - 1) {
        const XYZZ<F>* Lp = lsum + (uint64_t)w * groups_per_win;
        block_sum29<F>(chunk_len, [&](uint32_t i) { return &Lp[base + i]; }, dst, sh, shx, &bad);
    } else if ((int)b >= log_chunk) {
        // the whole chunk has the bit set, or none of it
        block_sum29<F>(((base >> b) & 1) ? chunk_len : 0u, [&](uint32_t i) { return &R[base + i]; }, dst, sh, shx, &bad);
    } else {
        // insert a 1 at bit position b of the local index
        block_sum29<F>(chunk_len / 2, [&](uint32_t i) { return &R[base + (((i >> b) << (b + 1)) | (1u << b) | (i & ((1u << b) - 1)))]; }, dst, sh,
                       shx, &bad);
    }
}

// the same per-segment sum in the lazy representation (window reduction of large bucket sets)
template <class F>
__global__ void __launch_bounds__(64)
msm_segment_sum29_kernel(const XYZZ<F>* __restrict__ in, uint32_t seg_len, XYZZ<F>* __restrict__ out) {
    __shared__ LazyPt<F> sh[64];
    __shared__ XYZZ<F> shx[64];
    __shared__ uint32_t bad;
    const uint64_t base = (uint64_t)blockIdx.x * seg_len;
    block_sum29<F>(seg_len, [&](uint32_t i) { return &in[base + i]; }, &out[blockIdx.x], sh, shx, &bad);
}

// out[b] = sum of in[b*seg_len .. (b+1)*seg_len): one wave per segment, strided partial sums + LDS tree
template <class F>
__global__ void __launch_bounds__(64)
msm_segment_sum_kernel(const XYZZ<F>* __restrict__ in, uint32_t seg_len, XYZZ<F>* __restrict__ out) {
    __shared__ XYZZ<F> sh[64];
    const uint64_t base = (uint64_t)blockIdx.x * seg_len;
    XYZZ<F> acc = xyzz_inf<F>();
    for (uint32_t g = threadIdx.x; g < seg_len; g += 64) acc = add(acc, load_pod<XYZZ<F>>(&in[base + g]));
    acc = wave_tree_sum(acc, sh);
    if (threadIdx.x == 0) store_pod(&out[blockIdx.x], acc);
}

// ---- precomputed tables (pinned keys): table[w*n + i] = [2^(c*w)] P_i, affine ----------------------------------
// With 288 GB of HBM a pinned key can afford windows x its size: all windows then share one bucket set (one
// reduction instead of `windows`, no Horner) and c can grow to 23 => 12 instead of 14 window passes over the scalars.
// (ICICLE exposes the same idea as MSMConfig.PrecomputeFactor, icicle.go:507-525.)
// ---- host driver ------------------------------------------------------------------------------------

// batch > 1 (table mode only): `d_scalars` is an array of `batch` device pointers, one scalar vector each, over the SAME table:
// one sort, one task list, one bucket set per vector (P->nsets = batch) -- the three wire commitments or the three quotient
// shards of a PLONK proof share every launch and every latency-bound tail.
template <class FrP>
int msm_prepare(Ctx* ctx, const void* d_scalars, size_t n, bool scalars_mont, int c, int win_lo, int win_hi, bool table,
                MsmPrepared* P, int slot = 0, int batch = 1) {
    hipStream_t st = ctx->work_stream();
    const std::string sfx = slot ? "#1" : "";
    auto key = [&](const char* k) { return std::string(k) + sfx; };
    const int nwin = FrP::BITS / c + 1;
    if (win_hi < 0) win_hi = nwin;
    if (win_lo < 0 || win_hi > nwin || win_lo >= win_hi || c < 2 || c > 24) {
        set_error("msm: bad window range [%d,%d) of %d (c=%d, table=%d)", win_lo, win_hi, nwin, c, (int)table);
        return GA_ERR_INVALID;
    }
    const int nwl = win_hi - win_lo;
    if (n == 0 || n >= (1ull << 31)) {
        set_error("msm: n=%zu outside [1, 2^31)", n);
        return GA_ERR_INVALID;
    }
    if (batch < 1 || (batch > 1 && !table)) {
        set_error("msm: a batch of scalar vectors needs a precomputed table (batch=%d, table=%d)", batch, (int)table);
        return GA_ERR_INVALID;
    }
    const uint32_t half = 1u << (c - 1);
    const uint64_t m = (uint64_t)batch * nwl * n;
    const uint64_t nb64 = table ? (uint64_t)batch * half : (uint64_t)nwl * half;
    // table mode: the value indexes the WHOLE table [window][point] even when only a window range is accumulated (multi-GPU
    // partition A on pinned bases: the 2^(c*w) factors are baked into the table, so partial results simply add)
    if (m >= (1ull << 31) || nb64 >= (1ull << 31) || (table && (uint64_t)nwin * n >= (1ull << 31))) {
        set_error("msm: %d windows x %zu points exceeds the 2^31 pair index space; shard the call", nwl, n);
        return GA_ERR_INVALID;
    }
    const uint32_t nb = (uint32_t)nb64;
    // task length: buckets up to 4x the mean size stay one task, unless that would leave fewer than ~2^20 tasks for the
    // 256 CUs x 16 waves x 64 lanes (few-bucket cases: small n, or table mode where all windows share 2^(c-1) buckets)
    uint64_t mean = m / nb + 1;
    const uint64_t min_seg = ctx->tun.msm_min_seg;
    uint64_t seg64 = mean * 4 < min_seg ? min_seg : mean * 4;
    if (nb < (1u << 19)) {
        uint64_t want = (m >> 19) + 1, lo = mean / 6 > 32 ? mean / 6 : 32;   // keep a bucket's partials <= ~MSM_HOT_TASKS
        if (want < lo) want = lo;
        if (want < seg64) seg64 = want;
    }
    const uint32_t seg = (uint32_t)seg64;
    const uint64_t max_tasks = nb + m / seg + 1;

    uint32_t *keys, *vals, *keys2, *vals2, *off, *ntask, *task_off, *task_start, *task_key, *task_key2, *task_id, *task_perm;
    void* tmp;
    GA_CHECK(ctx->scratch_get(key("msm_vals2").c_str(), m * 4, (void**)&vals2));
    GA_CHECK(ctx->scratch_get(key("msm_off").c_str(), ((uint64_t)nb + 2) * 4, (void**)&off));
    GA_CHECK(ctx->scratch_get(key("msm_ntask").c_str(), ((uint64_t)nb + 2) * 4, (void**)&ntask));
    GA_CHECK(ctx->scratch_get(key("msm_task_off").c_str(), ((uint64_t)nb + 2) * 4, (void**)&task_off));
    GA_CHECK(ctx->scratch_get(key("msm_task_start").c_str(), max_tasks * 4, (void**)&task_start));
    GA_CHECK(ctx->scratch_get(key("msm_task_key").c_str(), max_tasks * 4, (void**)&task_key));
    GA_CHECK(ctx->scratch_get(key("msm_task_key2").c_str(), max_tasks * 4, (void**)&task_key2));
    uint32_t* task_qkey;
    GA_CHECK(ctx->scratch_get(key("msm_task_qkey").c_str(), max_tasks * 4, (void**)&task_qkey));
    GA_CHECK(ctx->scratch_get(key("msm_task_id").c_str(), max_tasks * 4, (void**)&task_id));
    GA_CHECK(ctx->scratch_get(key("msm_task_perm").c_str(), max_tasks * 4, (void**)&task_perm));
    uint32_t* task_dest;
    GA_CHECK(ctx->scratch_get(key("msm_task_dest").c_str(), max_tasks * 4, (void**)&task_dest));

    int end_bit = 1;
    while ((1ull << end_bit) <= nb64) end_bit++;   // keys take values 0..nb (nb = SKIP)
    // digits fused with the first sort pass (1b / 1c): key spaces of 2^11 .. 2^24 keys, at most MSM_P1_MAXW windows per scalar, any
    // number of scalar vectors over one table, enough pairs for the saved traffic to matter (GA_MSM_FUSE_MIN)
    const int xcd = ctx->tun.msm_xcd.load(std::memory_order_relaxed);
    const bool fused = msm_fused_fits(nb64) && nwl <= MSM_P1_MAXW && m >= ctx->tun.msm_fuse_min.load(std::memory_order_relaxed);
    // Scratch of the sort.  Fused: key parts (16 bits) / vals are the first level's output, dead once the second level has run, and
    // the sorted keys are never materialised -- so the two sort slots of a lane SHARE them (stream order separates their uses) and
    // there is no keys2: 14 bytes per pair less per extra slot.  Library sort: ping-pong pairs, the result may live in either.
    // INVARIANT behind the sharing: scratch names are per LANE (Ctx::scratch_get appends "@lane") and a lane has exactly one work
    // stream, so both slots issue on the same stream; a buffer that has to grow is released with hipFree, which waits for the device.
    // A slot prepared on any other stream, or an asynchronous free (hipFreeAsync, a pool), would let one slot's first level overwrite
    // pairs the other slot's second level has not read yet: key these two buffers by stream before doing either.
    keys = keys2 = nullptr;
    uint16_t* key_parts = nullptr;   // fused: what the first level hands to the second per pair besides the value -- the <= 12 low key bits
    if (fused) {
        GA_CHECK(ctx->scratch_get("msm_keys_level1", m * 2, (void**)&key_parts));
        GA_CHECK(ctx->scratch_get("msm_vals_level1", m * 4, (void**)&vals));
    } else {
        GA_CHECK(ctx->scratch_get(key("msm_keys").c_str(), m * 4, (void**)&keys));
        GA_CHECK(ctx->scratch_get(key("msm_vals").c_str(), m * 4, (void**)&vals));
        GA_CHECK(ctx->scratch_get(key("msm_keys2").c_str(), m * 4, (void**)&keys2));
    }
    if (fused) {
        if (msm_p1_bits(nb64) == 11)
            GA_CHECK((msm_fused_sort<FrP, 11>(ctx, sfx, st, d_scalars, n, scalars_mont, c, nwin, win_lo, win_hi, table, batch, half, nb, m, key_parts, vals, vals2, off, xcd)));
        else
            GA_CHECK((msm_fused_sort<FrP, 12>(ctx, sfx, st, d_scalars, n, scalars_mont, c, nwin, win_lo, win_hi, table, batch, half, nb, m, key_parts, vals, vals2, off, xcd)));
    } else {
        {
            StageTimer tm(ctx, "msm_digits", st);
            for (int b = 0; b < batch; b++) {
                const uint32_t* sc = batch == 1 ? (const uint32_t*)d_scalars : reinterpret_cast<const uint32_t* const*>(d_scalars)[b];
                hipLaunchKernelGGL((msm_digits_kernel<FrP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sc, (uint64_t)n,
                                   scalars_mont ? 1 : 0, c, nwin, win_lo, win_hi, table ? 1 : 0, (uint32_t)b * half, (uint32_t)nb64,
                                   keys + (uint64_t)b * nwl * n, vals + (uint64_t)b * nwl * n);
            }
            GA_KERNEL_CHECK();
        }
        StageTimer tm(ctx, "msm_sort", st);
        GA_CHECK(msm_sort_pairs(ctx, key("msm_sort_tmp"), keys, keys2, vals, vals2, (size_t)m, end_bit, st));
    }
    {
        StageTimer tm(ctx, "msm_tasks", st);
        uint32_t *long_list, *long_count;
        GA_CHECK(ctx->scratch_get(key("msm_long").c_str(), (max_tasks / MSM_LONG_TASKS + 2) * 4, (void**)&long_list));
        GA_CHECK(ctx->scratch_get(key("msm_long_count").c_str(), 256, (void**)&long_count));
        // explicit task list, ordered by decreasing length (padding slots keep key = 0xFFFFFFFF >= seg)
        int kbits = 1;
        while ((1u << kbits) <= seg) kbits++;
        // (long lists: the exact length, two radix passes -- GA_MSM_TASK_EXACT_MIN, see msm_task_init_kernel)
        const bool exact = m >= ctx->tun.msm_task_exact_min.load(std::memory_order_relaxed);
        const int qbits = exact || kbits < 7 ? kbits : 7, qshift = kbits - qbits;
        hipLaunchKernelGGL(msm_task_init_kernel, dim3((unsigned)((max_tasks + 255) / 256)), dim3(256), 0, st, task_qkey, task_id, (uint32_t)max_tasks, long_count);
        if (!fused)   // (the fused sort produced `off` itself)
            hipLaunchKernelGGL(msm_offsets_tasks_kernel, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, (const uint32_t*)keys2, m, nb, seg, off, ntask);
        else
            hipLaunchKernelGGL(msm_tasks_kernel, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, (const uint32_t*)off, nb, seg, ntask);
        GA_KERNEL_CHECK();
        size_t tmp_bytes = 0;
        GA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, ntask, task_off, (int)(nb + 1), st));
        GA_CHECK(ctx->scratch_get(key("msm_scan_tmp").c_str(), tmp_bytes + 256, &tmp));
        GA_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, ntask, task_off, (int)(nb + 1), st));
        hipLaunchKernelGGL(msm_task_list_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, (const uint32_t*)off, (const uint32_t*)task_off,
                           nb, seg, qshift, task_start, task_key, task_qkey, task_dest, long_list, long_count);
        hipLaunchKernelGGL(msm_task_list_long_kernel, dim3(256), dim3(256), 0, st, (const uint32_t*)off, (const uint32_t*)task_off, nb, seg,
                           qshift, task_start, task_key, task_qkey, task_dest, (const uint32_t*)long_list, (const uint32_t*)long_count);
        GA_KERNEL_CHECK();
        // padding keys are all-ones: sort on qbits+1 bits so that they stay behind every real key (real keys < 2^qbits)
        size_t tb = 0;
        GA_HIP_CHECK((rocprim::radix_sort_pairs<MsmTaskSort>(nullptr, tb, task_qkey, task_key2, task_id, task_perm, (size_t)max_tasks, 0u, (unsigned)(qbits + 1), st)));
        GA_CHECK(ctx->scratch_get(key("msm_tasksort_tmp").c_str(), tb + 256, &tmp));
        GA_HIP_CHECK((rocprim::radix_sort_pairs<MsmTaskSort>(tmp, tb, task_qkey, task_key2, task_id, task_perm, (size_t)max_tasks, 0u, (unsigned)(qbits + 1), st)));
    }
    P->n = n;
    P->c = c;
    P->nwin = nwin;
    P->win_lo = win_lo;
    P->win_hi = win_hi;
    P->nsets = table ? batch : nwl;
    P->table = table;
    P->half = half;
    P->nb = nb;
    P->seg = seg;
    P->m = m;
    P->max_tasks = max_tasks;
    P->vals = vals2;
    P->task_off = task_off;
    P->task_start = task_start;
    P->task_key = task_key2;
    P->task_key_by_id = task_key;
    P->task_perm = task_perm;
    P->task_dest = task_dest;
    return GA_OK;
}

// Group-dependent half: bucket accumulation over `d_bases` (the affine bases, or the precomputed table in table mode),
// merge, per-set reduction.  Writes P.nsets XYZZ sums to host memory.

// horner_c > 0 (raw bases, every window of the call): instead of the P.nsets window sums, out[0] receives their combination
// sum_w 2^(horner_c * w) * set_w -- the window reduction's per-bit sums and the Horner step over the windows then share ONE chain of
// doublings on the host (see the end of this function).
//
// ntab > 1 (table mode, ONE scalar vector, `tables` = ntab tables of the same shape as d_bases -- the wire-indexed A, B1, K of a Groth16
// key): every kernel of the pipeline runs ONCE over all the tables -- the bucket kernel with the table as the grid's y dimension, the
// merge the same way, the window reduction over ntab bucket sets as it does for a batch of scalar vectors -- and out[i] receives
// table i's sum: one tail per kernel, one host synchronisation, ntab x the waves in the latency-bound reduction kernels.
template <class F>
int msm_accumulate_reduce(Ctx* ctx, const void* d_bases, const MsmPrepared& P, XYZZ<F>* out, int horner_c = 0, const void* const* tables = nullptr,
                          int ntab = 1) {
    const uint32_t nb = P.nb, half = P.half, seg = P.seg;
    if (ntab < 1 || ntab > 4 || (ntab > 1 && (!tables || !P.table || P.nsets != 1 || horner_c != 0))) {
        set_error("msm: a multi-table pass takes 2..4 precomputed tables over one prepared scalar vector (ntab=%d, table=%d, sets=%d)", ntab, (int)P.table, P.nsets);
        return GA_ERR_INVALID;
    }
    const int nsets = ntab > 1 ? ntab : P.nsets;
    // buckets per running-sum group: MSM_GROUP when there are plenty of buckets, smaller (down to 2) when a set has few so
    // that the reduction still spreads over >= 2^15 lanes (small n, or table mode's single bucket set)
    const int tuned_group = ctx->tun.msm_group.load(std::memory_order_relaxed);
    uint32_t m_groups = tuned_group ? (uint32_t)tuned_group : (uint32_t)MSM_GROUP;
    const uint64_t min_lanes = 32768;   // (65536 / 131072 measured in round 2: no change / slower)
    while (m_groups > 2 && (uint64_t)half * nsets / m_groups < min_lanes) m_groups >>= 1;
    if (m_groups > half) m_groups = half;
    const uint32_t groups_per_win = half / m_groups;
    const uint32_t total_groups = groups_per_win * nsets;
    uint32_t *hot_list, *hot_count;
    XYZZ<F>*partial, *bsum, *gsum, *gsum2, *wsum;
    GA_CHECK(ctx->scratch_get("msm_hot", ((uint64_t)nb + 2) * 4, (void**)&hot_list));
    GA_CHECK(ctx->scratch_get("msm_hot_count", 256, (void**)&hot_count));   // [0] hot buckets, [1] very hot buckets
    uint32_t* vhot_list;
    XYZZ<F>* vtmp;
    const uint64_t vhot_cap = P.max_tasks / MSM_VHOT_TASKS + 2;
    GA_CHECK(ctx->scratch_get("msm_vhot", vhot_cap * 4, (void**)&vhot_list));
    GA_CHECK(ctx->scratch_get("msm_vhot_tmp", (uint64_t)ntab * vhot_cap * MSM_VHOT_SPLIT * sizeof(XYZZ<F>), (void**)&vtmp));
    // one array [nb bucket sums | max_tasks partial sums]: tasks write at task_dest (see msm_task_list_kernel); a multi-table pass:
    // [ntab x nb | ntab x max_tasks], so that the bucket sums of the tables are the consecutive sets the reduction expects
    GA_CHECK(ctx->scratch_get("msm_bsum_partial", (uint64_t)ntab * ((uint64_t)nb + P.max_tasks) * sizeof(XYZZ<F>), (void**)&bsum));
    partial = bsum + (uint64_t)ntab * nb;
    const uint32_t part_stride = (uint32_t)P.max_tasks, vtmp_stride = (uint32_t)(vhot_cap * MSM_VHOT_SPLIT);
    GA_CHECK(ctx->scratch_get("msm_gsum", (uint64_t)total_groups * sizeof(XYZZ<F>), (void**)&gsum));
    GA_CHECK(ctx->scratch_get("msm_gsum2", ((uint64_t)total_groups / 1024 + 64) * sizeof(XYZZ<F>), (void**)&gsum2));
    GA_CHECK(ctx->scratch_get("msm_wsum", (uint64_t)nsets * sizeof(XYZZ<F>), (void**)&wsum));
    hipStream_t st = ctx->work_stream();
    GA_HIP_CHECK(hipMemsetAsync(hot_count, 0, 8, st));
    // bucket accumulation: the fast lazy loop, then the tasks it flagged (an exceptional addition: equal or opposite points met)
    // once more with the complete lazy loop, then whatever is left with the exact kernel.  A table on which most tasks were flagged
    // (a DummySetup key: every base the same point) is remembered and gets the complete loop directly from then on.
    uint32_t h_redo_stack = 0;
    uint32_t* h_redo = ctx->pinned_words();
    if (!h_redo) h_redo = &h_redo_stack;
    *h_redo = 0;
    // the count of flagged tasks travels back asynchronously -- into the lane's pinned words, so that the host keeps launching the merge
    // and reduction kernels while the bucket kernel runs (a stack variable, pageable, made that copy a host-side wait for the bucket
    // kernel and every launch after it start from an empty queue).  Every return path, the early error returns included, must leave
    // with that copy finished
    struct PendingRead {
        hipStream_t st;
        bool pending = false;
        ~PendingRead() {
            if (pending) hipStreamSynchronize(st);
        }
    } redo_read{ctx->work_stream()};
    const uint32_t* acc_table = (const uint32_t*)d_bases;
    {
        uint32_t *redo_list, *redo_count, *redo2_list;
        GA_CHECK(ctx->scratch_get("msm_redo", (uint64_t)ntab * (P.max_tasks + 2) * 4, (void**)&redo_list));
        GA_CHECK(ctx->scratch_get("msm_redo2", (uint64_t)ntab * (P.max_tasks + 2) * 4, (void**)&redo2_list));
        GA_CHECK(ctx->scratch_get("msm_redo_count", 256, (void**)&redo_count));   // per table: [0] flagged by the first loop, [1] by the retry
        GA_HIP_CHECK(hipMemsetAsync(redo_count, 0, 8 * ntab, st));
        StageTimer tm(ctx, "msm_accumulate");
        if (!P.table) {
            // raw (not precomputed) bases: one conversion pass to the packed hat-domain format (a one-window "table"), then the
            // same lazy bucket kernel as the table path (an exact packed-arithmetic kernel cost ~1.5x more per addition: dropped)
            uint32_t* hat;
            GA_CHECK(ctx->scratch_get("msm_hat_bases", (uint64_t)P.n * sizeof(Affine<F>) + 256, (void**)&hat));
            hipLaunchKernelGGL((msm_hat_bases_kernel<F>), dim3((unsigned)((P.n + 255) / 256)), dim3(256), 0, st, (const Affine<F>*)d_bases,
                               (uint64_t)P.n, hat);
            acc_table = hat;
        }
        constexpr unsigned AT = Table29<F>::THREADS;
        const dim3 grid((unsigned)((P.max_tasks + AT - 1) / AT), (unsigned)ntab);
        MsmTables mt;
        mt.k = (uint32_t)ntab;
        mt.nb = nb;
        mt.max_tasks = (uint32_t)P.max_tasks;
        for (int i = 0; i < 4; i++) mt.t[i] = ntab > 1 ? (const uint32_t*)tables[i < ntab ? i : 0] : acc_table;
        // (a multi-table pass is only started on tables none of which is known as degenerate: groth16.hip witness_msms)
        if (ntab == 1 && P.table && !ctx->tun.msm_exact_redo && ctx->is_degenerate(d_bases))
            hipLaunchKernelGGL((msm_accumulate29_kernel<F, true>), grid, dim3(AT), 0, st, mt, (const uint32_t*)P.vals,
                               (const uint32_t*)P.task_start, (const uint32_t*)P.task_key, (const uint32_t*)P.task_key_by_id, (const uint32_t*)P.task_perm, (uint32_t)P.max_tasks, seg,
                               (const uint32_t*)P.task_dest, bsum, redo2_list, redo_count + 1);
        else
            hipLaunchKernelGGL((msm_accumulate29_kernel<F, false>), grid, dim3(AT), 0, st, mt, (const uint32_t*)P.vals,
                               (const uint32_t*)P.task_start, (const uint32_t*)P.task_key, (const uint32_t*)P.task_key_by_id, (const uint32_t*)P.task_perm, (uint32_t)P.max_tasks, seg,
                               (const uint32_t*)P.task_dest, bsum, redo_list, redo_count);
        if (!ctx->tun.msm_exact_redo)   // (GA_MSM_EXACT_REDO=1: tests send the flagged tasks straight to the exact kernel below)
            hipLaunchKernelGGL((msm_accumulate29_retry_kernel<F>), dim3(2048, (unsigned)ntab), dim3(AT), 0, st, mt, (const uint32_t*)P.vals,
                               (const uint32_t*)P.task_start, (const uint32_t*)P.task_key_by_id, seg, (const uint32_t*)redo_list,
                               (const uint32_t*)redo_count, (const uint32_t*)P.task_dest, bsum, redo2_list, redo_count + 1);
        else
            hipLaunchKernelGGL((msm_accumulate29_redo_kernel<F>), dim3(1024, (unsigned)ntab), dim3(64), 0, st, mt, (const uint32_t*)P.vals,
                               (const uint32_t*)P.task_start, (const uint32_t*)P.task_key_by_id, seg, (const uint32_t*)redo_list,
                               (const uint32_t*)redo_count, (const uint32_t*)P.task_dest, bsum);
        hipLaunchKernelGGL((msm_accumulate29_redo_kernel<F>), dim3(1024, (unsigned)ntab), dim3(64), 0, st, mt, (const uint32_t*)P.vals,
                           (const uint32_t*)P.task_start, (const uint32_t*)P.task_key_by_id, seg, (const uint32_t*)redo2_list,
                           (const uint32_t*)(redo_count + 1), (const uint32_t*)P.task_dest, bsum);
        GA_KERNEL_CHECK();
        // read after the stream's final sync below (per table: [2 i] = tasks the fast loop flagged; a stack fallback holds table 0's only)
        GA_HIP_CHECK(hipMemcpyAsync(h_redo, redo_count, h_redo == &h_redo_stack ? 4 : 8 * (size_t)ntab, hipMemcpyDeviceToHost, st));
        redo_read.pending = true;
    }
    auto note_degenerate = [&]() {
        if (!P.table) return;
        if (ntab == 1) {
            if ((uint64_t)*h_redo * 4 > P.max_tasks) ctx->mark_degenerate(d_bases);
        } else if (h_redo != &h_redo_stack) {
            for (int i = 0; i < ntab; i++)
                if ((uint64_t)h_redo[2 * i] * 4 > P.max_tasks) ctx->mark_degenerate(tables[i]);
        }
    };
    {
        StageTimer tm(ctx, "msm_merge");
        const unsigned ny = (unsigned)ntab;
        hipLaunchKernelGGL((msm_merge_kernel<F>), dim3((nb + 255) / 256, ny), dim3(256), 0, st, (const XYZZ<F>*)partial,
                           (const uint32_t*)P.task_off, nb, bsum, hot_list, hot_count, vhot_list, hot_count + 1, part_stride);
        hipLaunchKernelGGL((msm_hot_kernel<F>), dim3(512, ny), dim3(64), 0, st, (const XYZZ<F>*)partial, (const uint32_t*)P.task_off,
                           (const uint32_t*)hot_list, (const uint32_t*)hot_count, bsum, nb, part_stride);
        hipLaunchKernelGGL((msm_vhot_stage1_kernel<F>), dim3(2048, ny), dim3(64), 0, st, (const XYZZ<F>*)partial, (const uint32_t*)P.task_off,
                           (const uint32_t*)vhot_list, (const uint32_t*)(hot_count + 1), vtmp, part_stride, vtmp_stride);
        hipLaunchKernelGGL((msm_vhot_stage2_kernel<F>), dim3(256, ny), dim3(64), 0, st, (const XYZZ<F>*)vtmp, (const uint32_t*)vhot_list,
                           (const uint32_t*)(hot_count + 1), bsum, nb, vtmp_stride);
        GA_KERNEL_CHECK();
    }
    // Window reduction.  Large bucket sets: lazy per-group pass without the per-lane scalar multiplication,
    //   set sum = sum_g lsum[g] + m * sum_b 2^b * T_b,   T_b = sum of rsum[g] over the groups whose index has bit b set,
    // the T_b being plain tree sums and the last line host arithmetic.  Tiny sets keep the exact kernel: empty buckets (which
    // the lazy formulas cannot add to themselves) are the rule there and every group would be redone.
    // buckets from which the lazy pass pays (measured: 2^20 points / 2^16 buckets 2.82 -> 2.53 ms); ctx->tun is read from the
    // environment once per entry point (GA_REDUCE_LAZY_MIN: tests force the lazy path on sparse bucket sets with 0)
    // ... unless the set is DENSE (a table's shared set: windows x n entries over 2^(c-1) buckets; >= 16 entries per bucket leave
    // e^-16 of them empty): a 2^14-constraint proof spent 1.1 ms per MSM in the exact kernel's per-lane scalar multiplications
    // (6.3 ms per proof against 3.2 ms at 2^16, profiles/README.md round 3 batch N)
    // P.m counts every (scalar, window) pair, the zero digits in the skip bucket included: a table over which a 0/1-heavy witness
    // runs looks dense by that count while most of its buckets are empty.  The pair count of the skip bucket is only known on the
    // device at this point, so the verdict comes from the previous call on the same table: when the lazy pass flagged more than a
    // quarter of the groups, the set is remembered as sparse and small sets take the exact kernel again.
    const bool big_set = (uint64_t)half * nsets >= ctx->tun.reduce_lazy_min;
    // (P.m and half describe ONE table's pairs and buckets; the tables of a multi-table pass share the scalar vector, hence the verdict)
    const bool dense_set = P.m >= 16ull * (uint64_t)half * (uint64_t)P.nsets && !(P.table && ctx->is_sparse_set(d_bases));
    const bool lazy_reduce = big_set || dense_set;
    if (!lazy_reduce) {
        StageTimer tm(ctx, "msm_reduce");
        hipLaunchKernelGGL((msm_reduce_groups_kernel<F>), dim3((total_groups + 63) / 64), dim3(64), 0, st, (const XYZZ<F>*)bsum,
                           half, m_groups, groups_per_win, total_groups, gsum);
        // set sum = sum of its group results; two levels when a set has many groups so that the first level spreads
        // over >= 16 waves per set instead of one
        const uint32_t sg = 1024;
        if (groups_per_win > 2 * sg) {
            const uint32_t nseg = groups_per_win / sg;   // powers of two: exact
            hipLaunchKernelGGL((msm_segment_sum_kernel<F>), dim3(nseg * nsets), dim3(64), 0, st, (const XYZZ<F>*)gsum, sg, gsum2);
            hipLaunchKernelGGL((msm_segment_sum_kernel<F>), dim3(nsets), dim3(64), 0, st, (const XYZZ<F>*)gsum2, nseg, wsum);
        } else {
            hipLaunchKernelGGL((msm_segment_sum_kernel<F>), dim3(nsets), dim3(64), 0, st, (const XYZZ<F>*)gsum, groups_per_win, wsum);
        }
        GA_KERNEL_CHECK();
        if (horner_c > 0) {
            std::vector<XYZZ<F>> ws((size_t)nsets);
            GA_HIP_CHECK(hipMemcpyAsync(ws.data(), wsum, (size_t)nsets * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
            GA_HIP_CHECK(hipStreamSynchronize(st));
            note_degenerate();
            XYZZ<F> acc = xyzz_inf<F>();
            for (int w = nsets - 1; w >= 0; w--) {
                for (int k = 0; k < horner_c; k++) acc = dbl(acc);
                acc = add(acc, ws[(size_t)w]);
            }
            out[0] = acc;
            return GA_OK;
        }
        GA_HIP_CHECK(hipMemcpyAsync(out, wsum, (size_t)nsets * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
        GA_HIP_CHECK(hipStreamSynchronize(st));
        note_degenerate();
        return GA_OK;
    }
    int nbits = 0;
    while ((1u << nbits) < groups_per_win) nbits++;
    // chunk of groups per wave of the per-bit sums: 1024 where the grid fills the device anyway (2^24: 128 chunks x 18 rows; 256 / 128
    // measured there: msm_reduce 1.05 -> 1.13 / 1.34 ms, the second-level sums grow).  Smaller bucket sets are latency-bound -- a
    // lane's 16 dependent additions + 6 tree levels at ~9 us each made this kernel as expensive as the bucket accumulation of a
    // 2^16 MSM (profiles/README.md round 3, batches K / L) -- so the chunk shrinks until the grid has ~2 waves per SIMD
    uint32_t sg = 1024;
    while (sg > 64 && (uint64_t)(groups_per_win / sg) * (uint64_t)(nbits + 1) * (uint64_t)nsets < 2048) sg >>= 1;
    const uint32_t chunk_len = groups_per_win > sg ? sg : groups_per_win;   // powers of two
    const uint32_t chunks = groups_per_win / chunk_len;
    int log_chunk = 0;
    while ((1u << log_chunk) < chunk_len) log_chunk++;
    XYZZ<F>*rsum = nullptr, *bpart = nullptr, *bits = nullptr;
    uint32_t *rg_list, *rg_count;
    GA_CHECK(ctx->scratch_get("msm_rsum", (uint64_t)total_groups * sizeof(XYZZ<F>), (void**)&rsum));
    GA_CHECK(ctx->scratch_get("msm_bpart", ((uint64_t)nsets * (nbits + 1) * chunks + 64) * sizeof(XYZZ<F>), (void**)&bpart));
    GA_CHECK(ctx->scratch_get("msm_bits", ((uint64_t)nsets * (nbits + 1) + 64) * sizeof(XYZZ<F>), (void**)&bits));
    GA_CHECK(ctx->scratch_get("msm_redo_groups", ((uint64_t)total_groups + 2) * 4, (void**)&rg_list));
    GA_CHECK(ctx->scratch_get("msm_redo_groups_count", 256, (void**)&rg_count));
    GA_HIP_CHECK(hipMemsetAsync(rg_count, 0, 4, st));
    {
        StageTimer tm(ctx, "msm_reduce");
        hipLaunchKernelGGL((msm_reduce_groups29_kernel<F>), dim3((total_groups + 63) / 64), dim3(64), 0, st, (const XYZZ<F>*)bsum,
                           half, m_groups, groups_per_win, total_groups, gsum, rsum, rg_list, rg_count);
        hipLaunchKernelGGL((msm_reduce_groups_redo_kernel<F>), dim3(256), dim3(64), 0, st, (const XYZZ<F>*)bsum, half, m_groups,
                           groups_per_win, (const uint32_t*)rg_list, (const uint32_t*)rg_count, gsum, rsum);
        // rows 0..nbits-1: per-bit sums of rsum; row nbits: sum of lsum
        hipLaunchKernelGGL((msm_bit_partial_kernel<F>), dim3(chunks, (unsigned)nbits + 1, (unsigned)nsets), dim3(64), 0, st,
                           (const XYZZ<F>*)rsum, (const XYZZ<F>*)gsum, groups_per_win, chunk_len, log_chunk, bpart);
        hipLaunchKernelGGL((msm_segment_sum29_kernel<F>), dim3((unsigned)(nsets * (nbits + 1))), dim3(64), 0, st, (const XYZZ<F>*)bpart,
                           chunks, bits);
        GA_KERNEL_CHECK();
    }
    const int rows = nbits + 1;
    std::vector<XYZZ<F>> hb((size_t)nsets * rows);
    uint32_t h_redo_groups = 0;
    GA_HIP_CHECK(hipMemcpyAsync(hb.data(), bits, (size_t)nsets * rows * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
    GA_HIP_CHECK(hipMemcpyAsync(&h_redo_groups, rg_count, 4, hipMemcpyDeviceToHost, st));
    GA_HIP_CHECK(hipStreamSynchronize(st));
    note_degenerate();
    if (P.table && !big_set) ctx->note_sparse_set(d_bases, (uint64_t)h_redo_groups * 4 > total_groups);
    int log_m = 0;
    while ((1u << log_m) < m_groups) log_m++;
    if (horner_c > 0) {
        // result = sum_w 2^(c w) [ L_w + 2^log_m * sum_b 2^b T_(w,b) ]: every term has a bit position (c w for L_w, c w + log_m + b for
        // T_(w,b)), and ONE descent over the positions -- a doubling per position, an addition per term -- replaces the per-set chains
        // (nbits + log_m doublings each) followed by the Horner step over the windows (c doublings each): 255 + 16 doublings instead
        // of 15 x 16 + 255 for a 2^20-point MSM (c = 17), 0.1 ms of the 0.35 ms the host spent per call.
        const int top = horner_c * (nsets - 1) + log_m + nbits - 1;
        std::vector<std::vector<const XYZZ<F>*>> at((size_t)top + 1);
        for (int w = 0; w < nsets; w++) {
            at[(size_t)(horner_c * w)].push_back(&hb[(size_t)w * rows + nbits]);
            for (int b = 0; b < nbits; b++) at[(size_t)(horner_c * w + log_m + b)].push_back(&hb[(size_t)w * rows + b]);
        }
        XYZZ<F> acc = xyzz_inf<F>();
        for (int pos = top; pos >= 0; pos--) {
            if (pos != top) acc = dbl(acc);
            for (const XYZZ<F>* t : at[(size_t)pos]) acc = add(acc, *t);
        }
        out[0] = acc;
        return GA_OK;
    }
    for (int w = 0; w < nsets; w++) {   // host: ~nbits + log2(m) doublings and nbits additions per set
        XYZZ<F> acc = xyzz_inf<F>();
        for (int b = nbits - 1; b >= 0; b--) acc = add(dbl(acc), hb[(size_t)w * rows + b]);
        for (int k = 0; k < log_m; k++) acc = dbl(acc);
        out[w] = add(hb[(size_t)w * rows + nbits], acc);
    }
    return GA_OK;
}

template <class C, int G>
int msm_windows_device(Ctx* ctx, const void* d_bases, const void* d_scalars, size_t n, bool scalars_mont, int c,
                       int win_lo, int win_hi, void* h_window_sums, bool combine) {
    typedef typename GroupField<C, G>::F F;
    XYZZ<F>* out = reinterpret_cast<XYZZ<F>*>(h_window_sums);
    const int nwin = C::FrP::BITS / c + 1;
    if (win_hi < 0) win_hi = nwin;
    if (n == 0) {
        for (int w = 0; w < (combine ? 1 : win_hi - win_lo); w++) out[w] = xyzz_inf<F>();
        return GA_OK;
    }
    MsmPrepared P;
    GA_CHECK(msm_prepare<typename C::FrP>(ctx, d_scalars, n, scalars_mont, c, win_lo, win_hi, false, &P));
    // combine: out[0] = sum_w 2^(c (w - win_lo)) W_w; a window share that does not start at window 0 is shifted by the caller
    return msm_accumulate_reduce<F>(ctx, d_bases, P, out, combine ? c : 0);
}

// MSM over a precomputed table: one XYZZ result (no Horner); windows [win_lo, win_hi) only (win_hi < 0 = all): the partial
// sums of disjoint window ranges add up to the full result
template <class C, int G>
int msm_table_device(Ctx* ctx, const void* d_table, const void* d_scalars, size_t n, bool scalars_mont, int c, void* h_sum, int win_lo,
                     int win_hi) {
    typedef typename GroupField<C, G>::F F;
    XYZZ<F>* out = reinterpret_cast<XYZZ<F>*>(h_sum);
    if (n == 0 || (win_hi >= 0 && win_hi <= win_lo)) {
        *out = xyzz_inf<F>();
        return GA_OK;
    }
    MsmPrepared P;
    GA_CHECK(msm_prepare<typename C::FrP>(ctx, d_scalars, n, scalars_mont, c, win_lo, win_hi, true, &P));
    return msm_accumulate_reduce<F>(ctx, d_table, P, out);
}

// `batch` scalar vectors (host array of device pointers) over one table: batch XYZZ results
template <class C, int G>
int msm_table_device_batch(Ctx* ctx, const void* d_table, const void* const* d_scalars, int batch, size_t n, bool scalars_mont, int c,
                           void* h_sums) {
    typedef typename GroupField<C, G>::F F;
    XYZZ<F>* out = reinterpret_cast<XYZZ<F>*>(h_sums);
    if (n == 0) {
        for (int b = 0; b < batch; b++) out[b] = xyzz_inf<F>();
        return GA_OK;
    }
    MsmPrepared P;
    // (msm_prepare takes the vector itself for batch == 1, the pointer array otherwise)
    GA_CHECK(msm_prepare<typename C::FrP>(ctx, batch == 1 ? d_scalars[0] : (const void*)d_scalars, n, scalars_mont, c, 0, -1, true, &P, 0, batch));
    return msm_accumulate_reduce<F>(ctx, d_table, P, out);
}

// prepared scalars (table mode) reused for another base table of the same length (G1.B / G2.B)
template <class C, int G>
int msm_table_device_reuse(Ctx* ctx, const void* d_table, const MsmPrepared& P, void* h_sum) {
    typedef typename GroupField<C, G>::F F;
    return msm_accumulate_reduce<F>(ctx, d_table, P, reinterpret_cast<XYZZ<F>*>(h_sum));
}

// ... and for SEVERAL tables of that shape in one pass (the wire-indexed G1.A, G1.B, G1.K of a Groth16 key over the one witness sort,
// prove.go:194,207,237): h_sums[i] = the MSM over tables[i].  2 <= ntab <= 4.
template <class C, int G>
int msm_table_device_reuse_multi(Ctx* ctx, const void* const* tables, int ntab, const MsmPrepared& P, void* h_sums) {
    typedef typename GroupField<C, G>::F F;
    return msm_accumulate_reduce<F>(ctx, tables[0], P, reinterpret_cast<XYZZ<F>*>(h_sums), 0, tables, ntab);
}

// group-independent preparation callable from translation units that do not include this header (groth16.hip)
template <class C>
int msm_prepare_table_scalars(Ctx* ctx, const void* d_scalars, size_t n, bool scalars_mont, int c, MsmPrepared* P, int slot, int win_lo,
                              int win_hi) {
    return msm_prepare<typename C::FrP>(ctx, d_scalars, n, scalars_mont, c, win_lo, win_hi, true, P, slot);
}

template <class C, int G>
size_t msm_table_point_bytes() {
    typedef typename GroupField<C, G>::F F;
    return Table29<F>::WORDS * 4;
}

template <class C, int G>
int msm_table_build(Ctx* ctx, const void* d_bases, size_t n, int c, void* d_table) {
    typedef typename GroupField<C, G>::F F;
    if (n == 0) return GA_OK;
    const int nwin = C::FrP::BITS / c + 1;
    StageTimer tm(ctx, "msm_table_build");
    hipLaunchKernelGGL((msm_table29_kernel<F>), dim3((unsigned)(((n + TableBatch<F>::K - 1) / TableBatch<F>::K + 63) / 64)), dim3(64), 0, ctx->work_stream(),
                       (const Affine<F>*)d_bases, (uint64_t)n, c, nwin, (uint32_t*)d_table);
    GA_KERNEL_CHECK();
    return GA_OK;
}

}  // namespace ga
