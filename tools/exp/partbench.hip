// Experiment (round 3, re-entry session): the MSM's "digits -> radix sort -> bucket offsets" stage as a fused MSD partition:
//   K0  histogram of the LOW 11 key bits straight from the scalars (digits recomputed, nothing written)
//   K2  digits + the first LSD pass in one kernel: a tile of 1024 scalars x 12 windows partitioned by the low 11 key bits
//       (LDS-staged, chunked writes) -- the (key, value) pairs are never written in scalar order and never re-read for a histogram
//   then ONE rocprim onesweep pass on key bits [11, 22) (stable, so the result is the fully sorted list) and the shipped offsets kernel
//   [variant B, kept for the record: K3 = one workgroup per coarse bin doing the second level itself; slower, see profiles/README.md]
// against the shipped sequence (msm_digits_kernel, rocprim onesweep with two 11-bit passes, binary-search offsets).
//   hipcc --offload-arch=gfx950 -O3 -w tools/exp/partbench.hip -o partbench && ./partbench
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr uint32_t SIGN = 0x80000000u;
constexpr int CB = 11;                       // coarse bin = the LOW CB key bits: the short top window (digits < 2^12 at c = 22: 2^24 extra
                                             // entries in buckets 0..4095) then spreads over every bin instead of filling bins 0 and 1
constexpr uint32_t NCB = 1u << CB;
constexpr uint32_t FINE = 2048;              // fine index = key >> CB, sorted inside a coarse bin by one workgroup

typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 21>, rocprim::kernel_config<1024, 21>, 11,
                                                                       rocprim::block_radix_rank_algorithm::match>>
    SortWide;

__global__ void gen_scalars(uint32_t* s, uint64_t n, int mode) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < 8; k++) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x += 0x632BE59BD9B4E019ull;
        s[i * 8 + k] = (uint32_t)x;
    }
    s[i * 8 + 7] &= 0x1FFFFFFFu;   // 253 bits: the top window is short, as for a field element
    if (mode == 1 && (i % 10) < 3) {   // "witness-like": 30 % of the scalars are 0 or 1
        for (int k = 0; k < 8; k++) s[i * 8 + k] = 0;
        s[i * 8] = (uint32_t)(i & 1);
    }
}

// the digit loop of msm_digits_kernel (table mode): window w of scalar i -> (key, val); key == skip for a zero digit
struct Digits {
    uint32_t l[8];
    uint32_t carry = 0;
    __device__ __forceinline__ void load(const uint32_t* s, uint64_t i) {
        const uint4 a = *reinterpret_cast<const uint4*>(s + i * 8), b = *reinterpret_cast<const uint4*>(s + i * 8 + 4);
        l[0] = a.x; l[1] = a.y; l[2] = a.z; l[3] = a.w; l[4] = b.x; l[5] = b.y; l[6] = b.z; l[7] = b.w;
    }
    __device__ __forceinline__ void next(int c, int w, uint64_t n, uint64_t i, uint32_t skip, uint32_t& key, uint32_t& val) {
        const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
        uint32_t d = (l[0] & mask) + carry;
#pragma unroll
        for (int k = 0; k < 7; k++) l[k] = (l[k] >> c) | (l[k + 1] << (32 - c));
        l[7] >>= c;
        uint32_t neg = 0;
        if (d > half) { d = (1u << c) - d; neg = SIGN; carry = 1; } else carry = 0;
        key = d == 0 ? skip : d - 1;
        val = (uint32_t)((uint64_t)w * n + i) | neg;
    }
};

__global__ void digits_kernel(const uint32_t* __restrict__ s, uint64_t n, int c, int nwin, uint32_t skip, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Digits D; D.load(s, i);
    for (int w = 0; w < nwin; w++) {
        uint32_t k, v; D.next(c, w, n, i, skip, k, v);
        keys[(uint64_t)w * n + i] = k; vals[(uint64_t)w * n + i] = v;
    }
}
__global__ void offsets_kernel(const uint32_t* __restrict__ keys, uint64_t m, uint32_t nb, uint32_t* __restrict__ off) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    uint64_t lo = 0, hi = m;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (keys[mid] < b) lo = mid + 1; else hi = mid; }
    off[b] = (uint32_t)lo;
}

// ---- K0: coarse histogram (ncp = coarse bins incl. the one that holds only the skip key) ----
__global__ void __launch_bounds__(256) part_hist_kernel(const uint32_t* __restrict__ s, uint64_t n, int c, int nwin, uint32_t skip, uint32_t ncp, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[2064];
    for (uint32_t b = threadIdx.x; b < ncp; b += blockDim.x) h[b] = 0;
    __syncthreads();
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Digits D; D.load(s, i);
        for (int w = 0; w < nwin; w++) {
            uint32_t k, v; D.next(c, w, n, i, skip, k, v);
            atomicAdd(&h[k & (NCB - 1)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < ncp; b += blockDim.x) if (h[b]) atomicAdd(&ghist[b], h[b]);
}
// exclusive scan of ncp <= 2049 counts by one block of 1024 threads; coff[ncp] = total; cursor = copy; stat[0] = largest non-skip bin
__global__ void __launch_bounds__(1024) part_scan_kernel(const uint32_t* __restrict__ ghist, uint32_t ncp, uint32_t* __restrict__ coff, uint32_t* __restrict__ cursor, uint32_t* __restrict__ stat) {
    __shared__ uint32_t a[2][2064];
    __shared__ uint32_t mx;
    if (threadIdx.x == 0) mx = 0;
    for (uint32_t b = threadIdx.x; b < 2064; b += 1024) a[0][b] = b < ncp ? ghist[b] : 0;
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < ncp; b += 1024) atomicMax(&mx, a[0][b]);
    int cur = 0;
    for (uint32_t d = 1; d < 2064; d <<= 1) {
        for (uint32_t b = threadIdx.x; b < 2064; b += 1024) a[cur ^ 1][b] = a[cur][b] + (b >= d ? a[cur][b - d] : 0);
        __syncthreads();
        cur ^= 1;
    }
    for (uint32_t b = threadIdx.x; b <= ncp; b += 1024) {
        const uint32_t ex = b ? a[cur][b - 1] : 0;
        coff[b] = ex;
        if (b < ncp) cursor[b] = ex;
    }
    if (threadIdx.x == 0) stat[0] = mx;
}

// ---- K2: digits + partition by the coarse key; one scalar per thread, nwin <= 16 entries per thread ----
constexpr int MAXW = 13;
template <int TILE>
__global__ void __launch_bounds__(TILE) part_scatter_kernel(const uint32_t* __restrict__ s, uint64_t n, int c, int nwin, uint32_t skip, uint32_t nc,
                                                              uint32_t* __restrict__ cursor, uint32_t* __restrict__ out_keys, uint32_t* __restrict__ out_vals) {
    extern __shared__ uint32_t stage[];                    // TILE * nwin keys, then TILE * nwin values
    uint32_t* const stage_v = stage + TILE * nwin;
    __shared__ uint32_t cnt[2048], loff[2][2048], gbase[2048];
    const uint32_t t = threadIdx.x;
    for (uint32_t b = t; b < 2048; b += TILE) cnt[b] = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * TILE + t;
    uint32_t key[MAXW], val[MAXW], rank[MAXW];
    if (i < n) {
        Digits D; D.load(s, i);
#pragma unroll
        for (int w = 0; w < MAXW; w++)
            if (w < nwin) {
                D.next(c, w, n, i, skip, key[w], val[w]);
                rank[w] = atomicAdd(&cnt[key[w] & (NCB - 1)], 1u);
            }
    }
    __syncthreads();
    // exclusive scan of cnt[0..nc) (nc <= 2048): Hillis-Steele over 2048 slots, two per thread
    for (uint32_t b = t; b < 2048; b += TILE) loff[0][b] = b < nc ? cnt[b] : 0;
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < 2048; d <<= 1) {
        for (uint32_t b = t; b < 2048; b += TILE) loff[cur ^ 1][b] = loff[cur][b] + (b >= d ? loff[cur][b - d] : 0);
        __syncthreads();
        cur ^= 1;
    }
    const uint32_t total = loff[cur][2047];
    for (uint32_t b = t; b < nc; b += TILE) gbase[b] = cnt[b] ? atomicAdd(&cursor[b], cnt[b]) : 0;
    // inclusive -> exclusive on the fly: start of bin b = incl[b] - cnt[b]
    if (i < n) {
#pragma unroll
        for (int w = 0; w < MAXW; w++)
            if (w < nwin) {
                const uint32_t b = key[w] & (NCB - 1), q = loff[cur][b] - cnt[b] + rank[w];
                stage[q] = key[w];
                stage_v[q] = val[w];
            }
    }
    __syncthreads();
    for (uint32_t p = t; p < total; p += TILE) {
        const uint32_t kk = stage[p];
        const uint32_t b = kk & (NCB - 1);
        const uint64_t dst = (uint64_t)gbase[b] + (p - (loff[cur][b] - cnt[b]));
        out_keys[dst] = kk;
        out_vals[dst] = stage_v[p];
    }
}


// ---- variant C: own second level in the NATURAL layout.  The pass-1 output is grouped by the low CB key bits; inside a group all
// pairs share them, so a pair's final place is off[key] + (any rank among the pairs of the same key): no stability needed.
// Segments of at most SEG pairs of ONE group: count the high key parts (LDS), add them into a global per-key histogram; an exclusive
// scan of that histogram IS the bucket-offset array; then the same segments reserve a run per (segment, key) and write the values.
constexpr uint32_t SEG = 16384, HB = 1032;   // HB >= 1025 high parts (skip key included)
__global__ void __launch_bounds__(1024) seg_plan_kernel(const uint32_t* __restrict__ ghist, uint32_t* __restrict__ seg_off) {
    __shared__ uint32_t a[2][2048];
    for (uint32_t b = threadIdx.x; b < 2048; b += 1024) a[0][b] = (ghist[b] + SEG - 1) / SEG;
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < 2048; d <<= 1) {
        for (uint32_t b = threadIdx.x; b < 2048; b += 1024) a[cur ^ 1][b] = a[cur][b] + (b >= d ? a[cur][b - d] : 0);
        __syncthreads();
        cur ^= 1;
    }
    for (uint32_t b = threadIdx.x; b <= 2048; b += 1024) seg_off[b] = b ? a[cur][b - 1] : 0;
}
__device__ __forceinline__ bool seg_range(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ coff, uint32_t& bin, uint32_t& lo, uint32_t& hi) {
    const uint32_t sidx = blockIdx.x;
    if (sidx >= seg_off[2048]) return false;
    uint32_t l = 0, r = 2048;                // last bin with seg_off[bin] <= sidx
    while (r - l > 1) { const uint32_t mid = (l + r) >> 1; if (seg_off[mid] <= sidx) l = mid; else r = mid; }
    bin = l;
    lo = coff[bin] + (sidx - seg_off[bin]) * SEG;
    hi = coff[bin + 1];
    if (hi - lo > SEG) hi = lo + SEG;
    return true;
}
__global__ void __launch_bounds__(1024) seg_count_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ coff,
                                                          uint32_t* __restrict__ gcount) {
    __shared__ uint32_t cnt[HB];
    uint32_t bin, lo, hi;
    if (!seg_range(seg_off, coff, bin, lo, hi)) return;
    for (uint32_t h = threadIdx.x; h < HB; h += 1024) cnt[h] = 0;
    __syncthreads();
    uint32_t kk[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { const uint32_t p = lo + u * 1024 + threadIdx.x; kk[u] = p < hi ? keys[p] : 0xFFFFFFFFu; }
#pragma unroll
    for (int u = 0; u < 16; u++) if (kk[u] != 0xFFFFFFFFu) atomicAdd(&cnt[kk[u] >> CB], 1u);
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < HB; h += 1024) if (cnt[h]) atomicAdd(&gcount[(h << CB) | bin], cnt[h]);
}
__global__ void __launch_bounds__(1024) seg_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ seg_off,
                                                            const uint32_t* __restrict__ coff, uint32_t* __restrict__ cursor, uint32_t* __restrict__ out_vals) {
    __shared__ uint32_t stage_v[SEG];
    __shared__ uint16_t stage_h[SEG];
    __shared__ uint32_t cnt[HB], incl[2][HB], gb[HB];
    uint32_t bin, lo, hi;
    if (!seg_range(seg_off, coff, bin, lo, hi)) return;
    const uint32_t t = threadIdx.x;
    for (uint32_t h = t; h < HB; h += 1024) cnt[h] = 0;
    __syncthreads();
    uint32_t kk[16], vv[16], rk[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const uint32_t p = lo + u * 1024 + t;
        kk[u] = p < hi ? keys[p] : 0xFFFFFFFFu;
        vv[u] = p < hi ? vals[p] : 0;
    }
#pragma unroll
    for (int u = 0; u < 16; u++) if (kk[u] != 0xFFFFFFFFu) rk[u] = atomicAdd(&cnt[kk[u] >> CB], 1u);
    __syncthreads();
    for (uint32_t h = t; h < HB; h += 1024) incl[0][h] = cnt[h];
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < HB; d <<= 1) {
        for (uint32_t h = t; h < HB; h += 1024) incl[cur ^ 1][h] = incl[cur][h] + (h >= d ? incl[cur][h - d] : 0);
        __syncthreads();
        cur ^= 1;
    }
    for (uint32_t h = t; h < HB; h += 1024) gb[h] = cnt[h] ? atomicAdd(&cursor[(h << CB) | bin], cnt[h]) : 0;
#pragma unroll
    for (int u = 0; u < 16; u++)
        if (kk[u] != 0xFFFFFFFFu) {
            const uint32_t h = kk[u] >> CB, at = incl[cur][h] - cnt[h] + rk[u];
            stage_v[at] = vv[u];
            stage_h[at] = (uint16_t)h;
        }
    __syncthreads();
    const uint32_t total = hi - lo;
    for (uint32_t p = t; p < total; p += 1024) {
        const uint32_t h = stage_h[p];
        out_vals[(uint64_t)gb[h] + (p - (incl[cur][h] - cnt[h]))] = stage_v[p];
    }
}

// per-bucket checksum of the values (order inside a bucket is free)
__global__ void bucket_sum_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ off, const uint32_t* __restrict__ off_end, uint32_t nb, uint64_t* __restrict__ sums) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint64_t s = 0;
    for (uint32_t p = off[b]; p < off_end[b]; p++) s += (uint64_t)vals[p] * 0x9E3779B97F4A7C15ull + (vals[p] ^ (vals[p] >> 7));
    sums[b] = s;
}

int main(int argc, char** argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 24;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const uint64_t n = 1ull << logn;
    const int c = 22, nwin = 12;
    const uint32_t half = 1u << (c - 1), nb = half, skip = nb;
    const uint64_t m = (uint64_t)nwin * n;
    const uint32_t nc = NCB, ncp = nc;
    hipStream_t st; hipStreamCreate(&st);
    uint32_t *s, *k, *k2, *v, *v2, *off_a, *off_b, *end_b, *vals_b = nullptr, *ghist, *coff, *cursor, *stat;
    uint64_t *sum_a, *sum_b;
    uint32_t *pk, *pk2, *pv, *pv2, *seg_off, *gcount, *cur2, *off_c, *vals_c;
    CK(hipMalloc(&s, n * 32)); CK(hipMalloc(&k, m * 4)); CK(hipMalloc(&k2, m * 4)); CK(hipMalloc(&v, m * 4)); CK(hipMalloc(&v2, m * 4));
    CK(hipMalloc(&off_a, (nb + 2) * 4ull)); CK(hipMalloc(&off_b, (nb + 2) * 4ull)); CK(hipMalloc(&end_b, (nb + 2) * 4ull)); CK(hipMalloc(&ghist, 4096 * 4)); CK(hipMalloc(&coff, 4096 * 4));
    CK(hipMalloc(&cursor, 4096 * 4)); CK(hipMalloc(&stat, 256)); CK(hipMalloc(&pk, m * 4)); CK(hipMalloc(&pk2, m * 4)); CK(hipMalloc(&pv, m * 4)); CK(hipMalloc(&pv2, m * 4)); CK(hipMalloc(&seg_off, 4096 * 4)); CK(hipMalloc(&gcount, (HB << CB) * 4ull)); CK(hipMalloc(&cur2, (HB << CB) * 4ull)); CK(hipMalloc(&off_c, (HB << CB) * 4ull + 64)); CK(hipMalloc(&vals_c, m * 4));
    CK(hipMalloc(&sum_a, nb * 8ull)); CK(hipMalloc(&sum_b, nb * 8ull));
    gen_scalars<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(s, n, mode);
    hipEvent_t e[10]; for (auto& x : e) hipEventCreate(&x);
    size_t tb = 0;
    rocprim::double_buffer<uint32_t> dk(k, k2), dv(v, v2);
    CK((rocprim::radix_sort_pairs<SortWide>(nullptr, tb, dk, dv, m, 0, 22, st)));
    void* tmp; CK(hipMalloc(&tmp, tb + 256));
    size_t tb2 = 0;
    { rocprim::double_buffer<uint32_t> a(pk, pk2), b(pv, pv2); CK((rocprim::radix_sort_pairs<SortWide>(nullptr, tb2, a, b, m, CB, 22, st))); }
    void* tmp2; CK(hipMalloc(&tmp2, tb2 + 256));
    const int TILE = argc > 3 ? atoi(argv[3]) : 1024;
    CK(hipFuncSetAttribute((const void*)part_scatter_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * nwin * 8));
    CK(hipFuncSetAttribute((const void*)part_scatter_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * nwin * 8));
    CK(hipFuncSetAttribute((const void*)part_scatter_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * nwin * 8));
    float t_dig = 0, t_sort = 0, t_off = 0, t_h = 0, t_sc = 0, t_f = 0, t_o2 = 0, t_c = 0;
    size_t scan_bytes = 0; CK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, gcount, off_c, (int)(nb + 1), st));
    void* scan_tmp; CK(hipMalloc(&scan_tmp, scan_bytes + 256));
    const int reps = 5;
    uint32_t *sk = nullptr, *sv = nullptr;
    for (int r = 0; r < reps + 1; r++) {
        // shipped sequence
        rocprim::double_buffer<uint32_t> ek(k, k2), ev(v, v2);
        hipEventRecord(e[0], st);
        digits_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(s, n, c, nwin, skip, k, v);
        hipEventRecord(e[1], st);
        CK((rocprim::radix_sort_pairs<SortWide>(tmp, tb, ek, ev, m, 0, 22, st)));
        hipEventRecord(e[2], st);
        sk = ek.current(); sv = ev.current();
        offsets_kernel<<<(nb + 1 + 255) / 256, 256, 0, st>>>(sk, m, nb, off_a);
        hipEventRecord(e[3], st);
        // fused partition
        hipMemsetAsync(ghist, 0, 4096 * 4, st);
        part_hist_kernel<<<2048, 256, 0, st>>>(s, n, c, nwin, skip, ncp, ghist);
        part_scan_kernel<<<1, 1024, 0, st>>>(ghist, ncp, coff, cursor, stat);
        hipEventRecord(e[4], st);
        if (TILE == 1024) part_scatter_kernel<1024><<<(unsigned)((n + 1023) / 1024), 1024, 1024 * nwin * 8, st>>>(s, n, c, nwin, skip, nc, cursor, pk, pv);
        else if (TILE == 512) part_scatter_kernel<512><<<(unsigned)((n + 511) / 512), 512, 512 * nwin * 8, st>>>(s, n, c, nwin, skip, nc, cursor, pk, pv);
        else part_scatter_kernel<256><<<(unsigned)((n + 255) / 256), 256, 256 * nwin * 8, st>>>(s, n, c, nwin, skip, nc, cursor, pk, pv);
        hipEventRecord(e[5], st);
        rocprim::double_buffer<uint32_t> fk(pk, pk2), fv(pv, pv2);
        CK((rocprim::radix_sort_pairs<SortWide>(tmp2, tb2, fk, fv, m, CB, 22, st)));
        hipEventRecord(e[6], st);
        vals_b = fv.current();
        offsets_kernel<<<(nb + 1 + 255) / 256, 256, 0, st>>>(fk.current(), m, nb, off_b);
        hipEventRecord(e[7], st);
        // variant C on the same pass-1 output (pk / pv were consumed by the library pass as `current`; they are unchanged: double buffers)
        hipEventRecord(e[8], st);
        const uint32_t nkeys = HB << CB;
        hipMemsetAsync(gcount, 0, nkeys * 4ull, st);
        seg_plan_kernel<<<1, 1024, 0, st>>>(ghist, seg_off);
        const unsigned max_seg = (unsigned)(m / SEG + 2048);
        seg_count_kernel<<<max_seg, 1024, 0, st>>>(pk, seg_off, coff, gcount);
        CK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, gcount, off_c, (int)(nb + 1), st));
        hipMemcpyAsync(cur2, off_c, (nb + 1) * 4ull, hipMemcpyDeviceToDevice, st);
        seg_scatter_kernel<<<max_seg, 1024, 0, st>>>(pk, pv, seg_off, coff, cur2, vals_c);
        hipEventRecord(e[9], st);
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        if (r) { float ms; hipEventElapsedTime(&ms, e[8], e[9]); t_c += ms; }
        if (r) {
            float ms;
            hipEventElapsedTime(&ms, e[0], e[1]); t_dig += ms; hipEventElapsedTime(&ms, e[1], e[2]); t_sort += ms; hipEventElapsedTime(&ms, e[2], e[3]); t_off += ms;
            hipEventElapsedTime(&ms, e[3], e[4]); t_h += ms; hipEventElapsedTime(&ms, e[4], e[5]); t_sc += ms; hipEventElapsedTime(&ms, e[5], e[6]); t_f += ms; hipEventElapsedTime(&ms, e[6], e[7]); t_o2 += ms;
        }
    }
    printf("tile=%d ", TILE); printf("n=2^%d mode=%d  shipped: digits %.3f + sort %.3f + offsets %.3f = %.3f ms   fused: hist+scan %.3f + digits/first pass %.3f + second pass %.3f + offsets %.3f = %.3f ms\n", logn, mode,
           t_dig / reps, t_sort / reps, t_off / reps, (t_dig + t_sort + t_off) / reps, t_h / reps, t_sc / reps, t_f / reps, t_o2 / reps, (t_h + t_sc + t_f + t_o2) / reps);
    printf("variant C (own second level: count + scan + scatter, offsets as a by-product): %.3f ms  -> fused total %.3f ms\n", t_c / reps, (t_h + t_sc + t_c) / reps);
    {
        uint64_t* sum_c; CK(hipMalloc(&sum_c, nb * 8ull));
        bucket_sum_kernel<<<(nb + 255) / 256, 256, 0, st>>>(vals_c, off_c, off_c + 1, nb, sum_c);
        std::vector<uint32_t> hc(nb + 1), h0(nb + 1); std::vector<uint64_t> sc_(nb), s0(nb);
        bucket_sum_kernel<<<(nb + 255) / 256, 256, 0, st>>>(sv, off_a, off_a + 1, nb, sum_a);
        CK(hipMemcpy(hc.data(), off_c, (nb + 1) * 4ull, hipMemcpyDeviceToHost)); CK(hipMemcpy(h0.data(), off_a, (nb + 1) * 4ull, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sc_.data(), sum_c, nb * 8ull, hipMemcpyDeviceToHost)); CK(hipMemcpy(s0.data(), sum_a, nb * 8ull, hipMemcpyDeviceToHost));
        uint64_t bo = 0, bs = 0;
        for (uint32_t b = 0; b <= nb; b++) bo += hc[b] != h0[b];
        for (uint32_t b = 0; b < nb; b++) bs += sc_[b] != s0[b];
        printf("variant C: offsets differing %llu, bucket checksums differing %llu\n", (unsigned long long)bo, (unsigned long long)bs);
    }
    // equality: offsets identical, per-bucket checksums of the values identical
    bucket_sum_kernel<<<(nb + 255) / 256, 256, 0, st>>>(sv, off_a, off_a + 1, nb, sum_a);
    bucket_sum_kernel<<<(nb + 255) / 256, 256, 0, st>>>(vals_b, off_b, off_b + 1, nb, sum_b);
    std::vector<uint32_t> ha(nb + 1), hb(nb + 1), he(nb + 1); std::vector<uint64_t> sa(nb), sb(nb); uint32_t hstat = 0;
    CK(hipMemcpy(ha.data(), off_a, (nb + 1) * 4ull, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), off_b, (nb + 1) * 4ull, hipMemcpyDeviceToHost)); CK(hipMemcpy(he.data(), end_b, (nb + 1) * 4ull, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sa.data(), sum_a, nb * 8ull, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), sum_b, nb * 8ull, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hstat, stat, 4, hipMemcpyDeviceToHost));
    uint64_t bad_off = 0, bad_sum = 0;
    for (uint32_t b = 0; b <= nb; b++) bad_off += ha[b] != hb[b];
    for (uint32_t b = 0; b < nb; b++) bad_sum += sa[b] != sb[b];
    printf("offsets differing: %llu of %u, bucket checksums differing: %llu; entries %u of %llu, largest coarse bin %u (mean %llu)\n", (unsigned long long)bad_off, nb + 1,
           (unsigned long long)bad_sum, ha[nb], (unsigned long long)m, hstat, (unsigned long long)(ha[nb] / nc));
    return (bad_off || bad_sum) ? 2 : 0;
}
