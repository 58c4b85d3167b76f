// Synthetic key material with known discrete logs, field dot products and gathers: what tests, bench.py and the
// Groth16 wire filtering (prove.go:147-168) need around the MSM/NTT kernels; plus the window-size planner.
#pragma once
#include "common.hip.h"

namespace ga {

// bases[i] = [k_i]G with k_i = splitmix64(seed, i) | 1  (64-bit, odd => never zero); dlogs[i] = k_i as canonical fr
template <class C, int G>
__global__ void __launch_bounds__(64)
gen_bases_kernel(uint64_t seed, uint64_t first, uint64_t n, void* __restrict__ bases, uint32_t* __restrict__ dlogs) {
    typedef typename GroupField<C, G>::F F;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = splitmix64(seed ^ splitmix64(first + i)) | 1ull;   // element `first + i` of the seed's sequence
    uint32_t kw[2] = {(uint32_t)k, (uint32_t)(k >> 32)};
    XYZZ<F> p = scalar_mul(to_xyzz(Generator<C, G>::get()), kw, 2);
    Affine<F> a = to_affine(p);
    store_pod(reinterpret_cast<char*>(bases) + i * sizeof(Affine<F>), a);
    if (dlogs) {
        uint32_t* o = dlogs + i * 8;
        o[0] = kw[0];
        o[1] = kw[1];
#pragma unroll
        for (int q = 2; q < 8; q++) o[q] = 0;
    }
}

// scalars[i] = 4 pseudo-random limbs masked to the field's bit length, minus r once if needed (uniform enough for
// benchmarking; the bytes are then *interpreted* as a Montgomery image, exactly like gnark's fr.Element memory)
template <class C>
__global__ void gen_scalars_kernel(uint64_t seed, uint64_t n, uint32_t* __restrict__ out) {
    typedef typename C::FrP P;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<P> s;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint64_t v = splitmix64(seed + 0x1000003ull * (uint64_t)q ^ splitmix64(i * 4 + q));
        s.l[2 * q] = (uint32_t)v;
        s.l[2 * q + 1] = (uint32_t)(v >> 32);
    }
    s.l[7] &= (1u << (P::BITS - 224)) - 1;
    reduce_once<P>(s.l);
    store_fe(out + i * 8, s);
}

// partial[block] = sum over a grid-stride slice of a_i (Montgomery) * b_i (canonical)  -> canonical
template <class C>
__global__ void __launch_bounds__(256)
fr_dot_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t n, uint32_t* __restrict__ partial) {
    typedef typename C::FrP P;
    __shared__ uint32_t sh[256 * 8];
    Fe<P> acc = fe_zero<P>();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        acc = add(acc, mul(load_fe<P>(a + i * 8), load_fe<P>(b + i * 8)));
    for (uint32_t stride = 128; stride >= 1; stride >>= 1) {
        store_fe(sh + threadIdx.x * 8, acc);
        __syncthreads();
        if (threadIdx.x < stride) acc = add(acc, load_fe<P>(sh + (threadIdx.x + stride) * 8));
        __syncthreads();
    }
    if (threadIdx.x == 0) store_fe(partial + blockIdx.x * 8, acc);
}

// out[i] = a[i] * b[i]  (fr.Vector.Mul; all Montgomery)
template <class C>
__global__ void fr_vec_mul_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t n, uint32_t* __restrict__ out) {
    typedef typename C::FrP P;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_fe(out + i * 8, mul(load_fe<P>(a + i * 8), load_fe<P>(b + i * 8)));
}

template <class C>
__global__ void gather_fr_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                 uint64_t n) {
    // two lanes per element, 16 bytes each
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i = t >> 1, half = t & 1;
    if (i >= n) return;
    const u32x4* s = reinterpret_cast<const u32x4*>(src);
    u32x4* d = reinterpret_cast<u32x4*>(dst);
    d[i * 2 + half] = s[(uint64_t)idx[i] * 2 + half];
}

template <class C, int G>
int util_gen_bases(Ctx* ctx, uint64_t seed, size_t n, void* d_bases, void* d_dlogs, uint64_t first) {
    if (n == 0) return GA_OK;
    StageTimer tm(ctx, "gen_bases");
    hipLaunchKernelGGL((gen_bases_kernel<C, G>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->work_stream(), seed, first, (uint64_t)n,
                       d_bases, (uint32_t*)d_dlogs);
    GA_KERNEL_CHECK();
    return GA_OK;
}

template <class C>
int util_gen_scalars(Ctx* ctx, uint64_t seed, size_t n, void* d_scalars) {
    if (n == 0) return GA_OK;
    hipLaunchKernelGGL((gen_scalars_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->work_stream(), seed, (uint64_t)n,
                       (uint32_t*)d_scalars);
    GA_KERNEL_CHECK();
    return GA_OK;
}

template <class C>
int util_fr_dot(Ctx* ctx, const void* d_a, const void* d_b, size_t n, void* h_out) {
    typedef typename C::FrP P;
    const unsigned blocks = 512;
    uint32_t* d_part;
    GA_CHECK(ctx->scratch_get("fr_dot_partial", blocks * 32, (void**)&d_part));
    hipLaunchKernelGGL((fr_dot_kernel<C>), dim3(blocks), dim3(256), 0, ctx->work_stream(), (const uint32_t*)d_a, (const uint32_t*)d_b,
                       (uint64_t)n, d_part);
    GA_KERNEL_CHECK();
    std::vector<uint32_t> part(blocks * 8);
    GA_HIP_CHECK(hipMemcpyAsync(part.data(), d_part, blocks * 32, hipMemcpyDeviceToHost, ctx->work_stream()));
    GA_HIP_CHECK(hipStreamSynchronize(ctx->work_stream()));
    Fe<P> acc = fe_zero<P>();
    for (unsigned b = 0; b < blocks; b++) {
        Fe<P> v;
        memcpy(v.l, &part[b * 8], 32);
        acc = add(acc, v);
    }
    memcpy(h_out, acc.l, 32);
    return GA_OK;
}

template <class C>
int util_fr_vec_mul(Ctx* ctx, const void* d_a, const void* d_b, size_t n, void* d_out) {
    if (n == 0) return GA_OK;
    hipLaunchKernelGGL((fr_vec_mul_kernel<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->work_stream(), (const uint32_t*)d_a,
                       (const uint32_t*)d_b, (uint64_t)n, (uint32_t*)d_out);
    GA_KERNEL_CHECK();
    return GA_OK;
}

template <class C>
int util_gather_fr(Ctx* ctx, void* d_dst, const void* d_src, const uint32_t* d_idx, size_t n) {
    if (n == 0) return GA_OK;
    StageTimer tm(ctx, "gather_fr");
    hipLaunchKernelGGL((gather_fr_kernel<C>), dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, ctx->work_stream(), (uint32_t*)d_dst,
                       (const uint32_t*)d_src, d_idx, (uint64_t)n);
    GA_KERNEL_CHECK();
    return GA_OK;
}

template <class C>
int msm_plan(int group, size_t n, int* c_out, int* nwin_out) {
    const int bits = C::FrP::BITS;
    // cost ~ windows * (n mixed adds + the per-bucket reduction work, ~2 mixed-add equivalents per bucket with the lazy
    // window reduction: measured best c = 17 at 2^20 and c = 20 at 2^24 on MI355X); G2 scales both terms alike
    double best = 1e300;
    int bc = 4;
    for (int c = 4; c <= 22; c++) {
        int nwin = bits / c + 1;
        double cost = (double)nwin * ((double)n + 2.0 * (double)(1u << (c - 1)));
        if (cost < best) {
            best = cost;
            bc = c;
        }
    }
    (void)group;
    *c_out = bc;
    *nwin_out = bits / bc + 1;
    return GA_OK;
}

template <class C>
int msm_plan_table(size_t n, int* c_out, int* nwin_out, bool batched) {
    const int bits = C::FrP::BITS;
    double best = 1e300;
    int bc = 4;
    // (c = 24 -- 11 windows over 2^23 buckets at 2^24 points -- was measured: the bucket kernel gains 6 %, the reduction grows with the
    // bucket count, 0.89 -> 3.08 ms: +1.5 ms per G1 MSM, +8.5 ms per proof, profiles/r05_t_table_c24_ab.txt)
    for (int c = 4; c <= 23; c++) {
        int nwin = bits / c + 1;
        if ((double)nwin * (double)n >= 2147483648.0) continue;   // table index must fit 31 bits
        if (table_c_override() && c != table_c_override()) continue;   // GA_TABLE_C (experiments; read with the other knobs)
        // one shared bucket set: its reduction costs ~6 mixed-add equivalents per bucket for mid-size inputs (latency-bound
        // kernels) and ~2.5 from 2^22 points up (measured: c = 17 best at 2^20, c = 22 best at 2^22 and 2^24)
        // batched: the table serves runs over k scalar vectors at once (PLONK's grouped commitments): k bucket sets widen the sort
        // keys and turn the reduction from latency- into issue-bound -- measured at 2^22 with batches of three: c = 20 beats c = 22
        // by 3.2 ms per PLONK proof (profiles/r04_l_plonk_table_c.txt) while a single run loses 0.05 ms
        const double per_bucket = (n >= (1u << 22) && !batched) ? 2.5 : 6.0;
        double cost = (double)nwin * (double)n + per_bucket * (double)(1u << (c - 1));
        // a top window of only 1-4 scalar bits puts its n additions into a handful of buckets, which the merge step then sums
        // almost serially (measured at c = 18 and 21: merge 0.2 -> 1.2-2.3 ms, profiles/r02_e_table_c_sweep.txt): avoid those widths
        const int top_bits = bits - (nwin - 1) * c;
        if (top_bits >= 1 && top_bits <= 4) cost *= 1.25;
        if (cost < best) {
            best = cost;
            bc = c;
        }
    }
    if (best == 1e300) {   // no window width keeps windows x n inside the 31-bit table index: the caller must shard the vector
        set_error("msm table plan: %zu points do not fit the 2^31 (window, point) index space; shard the vector", n);
        return GA_ERR_INVALID;
    }
    *c_out = bc;
    *nwin_out = bits / bc + 1;
    return GA_OK;
}

}  // namespace ga
