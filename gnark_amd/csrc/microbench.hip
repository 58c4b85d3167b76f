// Issue-rate microbenchmarks for the integer / FP64 instructions a big-integer multiplier can be built from on
// gfx950 ("microbenchmark v_mad_u64_u32 issue rate and report achieved MAD/s fraction"), plus the
// achieved throughput of this library's own Montgomery multiplication.
#include "common.hip.h"
#include "field29.hip.h"

namespace ga {

constexpr int MB_ITERS = 2048;

#define MB_KERNEL(name, decl, body)                                                      \
    __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) {          \
        decl;                                                                            \
        for (int it = 0; it < MB_ITERS; it++) {                                          \
            body                                                                         \
        }                                                                                \
        uint32_t r = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                  \
        if (r == 0x12345u) out[threadIdx.x] = r;                                         \
    }

#define MB_U64_DECL                                                                                            \
    uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t x = seed | 1u, y = (seed >> 3) | 1u
#define MB_U32_DECL                                                                                            \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t x = seed | 1u, y = (seed >> 3) | 1u
#define MB_F64_DECL                                                                                            \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double x = 1.0000001, y = 0.9999999
#define MB_F32_DECL                                                                                            \
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float x = 1.0000001f, y = 0.9999999f

#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

#define OP_MAD64(a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "vcc");
MB_KERNEL(mb_mad_u64_u32, MB_U64_DECL, REP8(OP_MAD64) REP8(OP_MAD64))
#define OP_MULLO(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(x));
MB_KERNEL(mb_mul_lo_u32, MB_U32_DECL, REP8(OP_MULLO) REP8(OP_MULLO))
#define OP_MULHI(a) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(x));
MB_KERNEL(mb_mul_hi_u32, MB_U32_DECL, REP8(OP_MULHI) REP8(OP_MULHI))
#define OP_MAD24(a) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
MB_KERNEL(mb_mad_u32_u24, MB_U32_DECL, REP8(OP_MAD24) REP8(OP_MAD24))
#define OP_ADD64(a) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a) : "v"(a7));
MB_KERNEL(mb_lshl_add_u64, MB_U64_DECL, REP8(OP_ADD64) REP8(OP_ADD64))
#define OP_ADDC(a) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(a) : "v"(x), "v"(y) : "vcc");
MB_KERNEL(mb_add_co_addc, MB_U32_DECL, REP8(OP_ADDC))
#define OP_MOV(a) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(x));
MB_KERNEL(mb_mov_b32, MB_U32_DECL, REP8(OP_MOV) REP8(OP_MOV))
#define OP_FMA64(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
MB_KERNEL(mb_fma_f64, MB_F64_DECL, REP8(OP_FMA64) REP8(OP_FMA64))
#define OP_FMA32(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
MB_KERNEL(mb_fma_f32, MB_F32_DECL, REP8(OP_FMA32) REP8(OP_FMA32))

// dependency distance and occupancy: DIST accumulators take the 16 multiply-adds of an iteration in turn (DIST = 1: every
// instruction waits for the previous one's result -- the shape of a product column, which accumulates in place); LDSPAD > 0 leaves
// room for ONE workgroup of 256 lanes per CU, i.e. one wave per SIMD (the occupancy of the BLS12-381 G2 bucket kernel)
template <int DIST, int LDSPAD>
__global__ void __launch_bounds__(256) mb_mad_chain(uint32_t* out, uint32_t seed) {
    __shared__ uint32_t pad[LDSPAD > 0 ? LDSPAD : 1];
    if (LDSPAD > 0) pad[threadIdx.x] = seed;
    uint64_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = seed + threadIdx.x + k;
    uint32_t x = seed | 1u, y = (seed >> 3) | 1u;
    for (int it = 0; it < MB_ITERS; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[k % DIST]) : "v"(x), "v"(y) : "vcc");
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r += (uint32_t)a[k];
    if (LDSPAD > 0) r += pad[(threadIdx.x + 1) & 255];
    if (r == 0x12345u) out[threadIdx.x] = r;
}

// Round 4: what a multiply-add costs INSIDE a product.  64 multiply-adds per loop trip (the branch of the 16-per-trip kernels above is
// ~5 % of their time), eight independent accumulators, in the variants a Montgomery product is made of:
//   0 all operands in VGPRs, carry-out to vcc            1 carry-out alternating between two SGPR pairs
//   2 one multiplicand in an SGPR (a modulus limb)        3 as 0 with a 64-bit shift + 64-bit add after every 8 multiply-adds (a column's carry)
//   4 as 3 with v_mul_lo_u32 + v_and_b32 as well (a whole reduction row: 9 MAD + mul_lo + and + shift + add)
template <int VARIANT>
__global__ void __launch_bounds__(256) mb_mad_mix(uint32_t* out, uint32_t seed) {
    uint64_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = seed + threadIdx.x + k;
    uint32_t x = seed | 1u, y = (seed >> 3) | 1u, m = seed ^ 0x5555u;
    const uint32_t sy = __builtin_amdgcn_readfirstlane(y);
    for (int it = 0; it < MB_ITERS / 4; it++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
            if (VARIANT == 1) {
                if (k & 1) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[k % 8]) : "v"(x), "v"(y) : "s10", "s11");
                else asm volatile("v_mad_u64_u32 %0, s[12:13], %1, %2, %0" : "+v"(a[k % 8]) : "v"(x), "v"(y) : "s12", "s13");
            } else if (VARIANT == 2) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[k % 8]) : "v"(x), "s"(sy) : "vcc");
            } else {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[k % 8]) : "v"(m), "v"(y) : "vcc");
                if (VARIANT >= 3 && (k % 8) == 7) {
                    const int j = (k / 8) % 8;
                    if (VARIANT == 4) {
                        asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(m) : "v"((uint32_t)a[j]), "v"(x));
                        asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(m));
                    }
                    uint64_t c;
                    asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(c) : "v"(a[j]));
                    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[(j + 1) % 8]) : "v"(c));
                }
            }
        }
    }
    uint32_t r = m;
#pragma unroll
    for (int k = 0; k < 8; k++) r += (uint32_t)a[k];
    if (r == 0x12345u) out[threadIdx.x] = r;
}

// Does the register BANK of a multiply-add's operands matter (four VGPR banks, index mod 4)?  The same 64 multiply-adds per trip with
// hand-picked registers: SAME = 1 puts both multiplicands and the low word of the accumulator in one bank (v2, v6, v[10:11], v[14:15], ...),
// SAME = 0 spreads them (v0, v1 against accumulators starting in banks 2 and 0).
template <int SAME>
__global__ void __launch_bounds__(256) mb_mad_banks(uint32_t* out, uint32_t seed) {
    uint32_t r = 0;
    if (SAME) {
        asm volatile(
            "v_mov_b32 v2, %1\n v_mov_b32 v6, %2\n"
            "v_mov_b32 v10, %1\n v_mov_b32 v11, 0\n v_mov_b32 v14, %2\n v_mov_b32 v15, 0\n v_mov_b32 v18, %1\n v_mov_b32 v19, 0\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0\n"
            "s_movk_i32 s20, 0x200\n"
            "1:\n"
            ".rept 16\n"
            "v_mad_u64_u32 v[10:11], vcc, v2, v6, v[10:11]\n v_mad_u64_u32 v[14:15], vcc, v2, v6, v[14:15]\n"
            "v_mad_u64_u32 v[18:19], vcc, v2, v6, v[18:19]\n v_mad_u64_u32 v[22:23], vcc, v2, v6, v[22:23]\n"
            ".endr\n"
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_u32 %0, v10, v14\n v_add_u32 %0, %0, v18\n v_add_u32 %0, %0, v22\n"
            : "=v"(r) : "v"(seed | 1u), "v"((seed >> 3) | 1u)
            : "v2", "v6", "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23", "s20", "vcc", "scc");
    } else {
        asm volatile(
            "v_mov_b32 v0, %1\n v_mov_b32 v1, %2\n"
            "v_mov_b32 v10, %1\n v_mov_b32 v11, 0\n v_mov_b32 v14, %2\n v_mov_b32 v15, 0\n v_mov_b32 v18, %1\n v_mov_b32 v19, 0\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0\n"
            "s_movk_i32 s20, 0x200\n"
            "1:\n"
            ".rept 16\n"
            "v_mad_u64_u32 v[10:11], vcc, v0, v1, v[10:11]\n v_mad_u64_u32 v[14:15], vcc, v0, v1, v[14:15]\n"
            "v_mad_u64_u32 v[18:19], vcc, v0, v1, v[18:19]\n v_mad_u64_u32 v[22:23], vcc, v0, v1, v[22:23]\n"
            ".endr\n"
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_u32 %0, v10, v14\n v_add_u32 %0, %0, v18\n v_add_u32 %0, %0, v22\n"
            : "=v"(r) : "v"(seed | 1u), "v"((seed >> 3) | 1u)
            : "v0", "v1", "v10", "v11", "v14", "v15", "v18", "v19", "v22", "v23", "s20", "vcc", "scc");
    }
    if (r == 0x12345u) out[threadIdx.x] = r;
}

// K independent Montgomery products per lane and trip (K = 1: the dependent chain of mb_f29_mul): how much of a product's cost
// is waiting for its own carries
template <class P, int K>
__global__ void __launch_bounds__(256) mb_f29_mul_ilp(uint32_t* out, uint32_t seed) {
    F29<P> a[K], b[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        a[k] = f29_from_mem(fe_const<P>(P::R2));
        b[k] = f29_from_mem(fe_one<P>());
        a[k].l[0] ^= (seed + threadIdx.x + k) & 0xFFFF;
        b[k].l[1] ^= (seed + 3 * k) & 0xFFFF;
    }
    for (int it = 0; it < MB_ITERS / 8; it++) {
#pragma unroll
        for (int k = 0; k < K; k++) a[k] = f29_mul(a[k], b[k]);
#pragma unroll
        for (int k = 0; k < K; k++) b[k] = f29_mul(b[k], a[k]);
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < K; k++) r += a[k].l[0] + b[k].l[0];
    if (r == 0x12345u) out[threadIdx.x] = r;
}

template <class P>
__global__ void __launch_bounds__(256) mb_field_mul(uint32_t* out, uint32_t seed) {
    Fe<P> a = fe_const<P>(P::R2), b = fe_one<P>();
    a.l[0] ^= seed + threadIdx.x;
    b.l[1] ^= seed;
    for (int it = 0; it < MB_ITERS / 8; it++) {
        a = mul(a, b);
        b = mul(b, a);
    }
    if (a.l[0] == 0x12345u && b.l[0] == 7) out[threadIdx.x] = a.l[1];
}

template <class P>
__global__ void __launch_bounds__(256) mb_f29_mul(uint32_t* out, uint32_t seed) {
    F29<P> a = f29_from_mem(fe_const<P>(P::R2)), b = f29_from_mem(fe_one<P>());
    a.l[0] ^= (seed + threadIdx.x) & 0xFFFF;
    b.l[1] ^= seed & 0xFFFF;
    for (int it = 0; it < MB_ITERS / 8; it++) {
        a = f29_mul(a, b);
        b = f29_mul(b, a);
    }
    if (a.l[0] == 0x12345u && b.l[0] == 7) out[threadIdx.x] = a.l[1];
}

template <class P>
__global__ void __launch_bounds__(256) mb_f29_addsub(uint32_t* out, uint32_t seed) {
    F29<P> a = f29_from_mem(fe_const<P>(P::R2)), b = f29_from_mem(fe_one<P>());
    a.l[0] ^= (seed + threadIdx.x) & 0xFFFF;
    for (int it = 0; it < MB_ITERS / 8; it++) {
        a = f29_sub<4>(f29_add(a, b), b);
        b = f29_sub<4>(f29_add(b, a), a);
    }
    if (a.l[0] == 0x12345u && b.l[0] == 7) out[threadIdx.x] = a.l[1];
}

template <class P>
__global__ void __launch_bounds__(256) mb_fe_addsub(uint32_t* out, uint32_t seed) {
    Fe<P> a = fe_const<P>(P::R2), b = fe_one<P>();
    a.l[0] ^= seed + threadIdx.x;
    for (int it = 0; it < MB_ITERS / 8; it++) {
        a = sub(add(a, b), b);
        b = sub(add(b, a), a);
    }
    if (a.l[0] == 0x12345u && b.l[0] == 7) out[threadIdx.x] = a.l[1];
}

// launches = 3: a burst of a few milliseconds (the boost clock); launches in the hundreds: the SUSTAINED rate of a kernel that runs
// for 0.1 s or more, which is what the bucket kernels see (round 4: the integer pipes at full load do not hold the burst clock)
template <class K>
static int run_one(Ctx* ctx, const char* name, K kernel, double ops_per_thread, std::string& out, uint32_t* d_out, int launches = 3) {
    const unsigned blocks = 256 * 8, threads = 256;
    hipEvent_t a, b;
    GA_HIP_CHECK(hipEventCreate(&a));
    GA_HIP_CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, ctx->stream, d_out, 12345u);   // warm-up
    GA_HIP_CHECK(hipEventRecord(a, ctx->stream));
    for (int r = 0; r < launches; r++) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, ctx->stream, d_out, 12345u + r);
    GA_HIP_CHECK(hipEventRecord(b, ctx->stream));
    GA_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    GA_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    double gops = (double)launches * ops_per_thread * blocks * threads / (ms * 1e-3) / 1e9;
    char tmp[128];
    snprintf(tmp, sizeof(tmp), "%s=%.1f;", name, gops);
    out += tmp;
    hipEventDestroy(a);
    hipEventDestroy(b);
    return GA_OK;
}

// One wave that watches the clocks for `ticks` ticks of the constant-rate counter (wall_clock64: 100 MHz): how many shader cycles
// (clock64 = s_memtime) went by.  Launched on a stream of its own beside whatever else the device is running, it reports the clock the
// chip HOLDS under that load -- the bucket kernels run for seconds of a proof stream, not for the milliseconds of a microbenchmark.
__global__ void __launch_bounds__(64) mb_clock_probe(uint64_t ticks, uint64_t* out) {
    const uint64_t w0 = wall_clock64(), c0 = clock64();
    uint64_t w = w0;
    while (w - w0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        w = wall_clock64();
    }
    if (threadIdx.x == 0) {
        out[0] = clock64() - c0;
        out[1] = w - w0;
    }
}

int util_clock_probe(Ctx* ctx, uint32_t micros, double* mhz_out) {
    // Runs BESIDE whatever holds the context, from another thread: nothing here may synchronise the device or touch the allocator
    // (hipFree waits for every stream -- the probe would block until the load it measures had drained -- and a raw hipMalloc would
    // bypass the GA_HBM_RESERVE_MB accounting of device_malloc).  The kernel writes its two counters straight into pinned host
    // memory, which a kernel can address as is; the probe's own stream is the only thing waited for.
    hipSetDevice(ctx->device);
    hipStream_t st;
    GA_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint64_t* h_out = nullptr;
    hipError_t e = hipHostMalloc((void**)&h_out, 16, hipHostMallocMapped);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        hipStreamDestroy(st);
        set_error("ga_clock_probe: hipHostMalloc failed: %s", hipGetErrorString(e));
        return GA_ERR_NOMEM;
    }
    h_out[0] = h_out[1] = 0;
    int wall_khz = 100000;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device);
    hipLaunchKernelGGL(mb_clock_probe, dim3(1), dim3(64), 0, st, (uint64_t)micros * (uint64_t)wall_khz / 1000ull, h_out);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    const uint64_t cycles = h_out[0], ticks = h_out[1];
    hipHostFree(h_out);
    hipStreamDestroy(st);
    if (e != hipSuccess || ticks == 0) {
        set_error("ga_clock_probe failed: %s", hipGetErrorString(e));
        return GA_ERR_HIP;
    }
    *mhz_out = (double)cycles / (double)ticks * (double)wall_khz / 1000.0;
    return GA_OK;
}

int util_microbench(Ctx* ctx, char* buf, size_t cap) {
    uint32_t* d_out;
    GA_CHECK(ctx->scratch_get("microbench", 4096, (void**)&d_out));
    std::string out;
    const double n16 = 16.0 * MB_ITERS, n8 = 8.0 * MB_ITERS;
    GA_CHECK(run_one(ctx, "v_mad_u64_u32_Gops", mb_mad_u64_u32, n16, out, d_out));
    GA_CHECK(run_one(ctx, "v_mul_lo_u32_Gops", mb_mul_lo_u32, n16, out, d_out));
    GA_CHECK(run_one(ctx, "v_mul_hi_u32_Gops", mb_mul_hi_u32, n16, out, d_out));
    GA_CHECK(run_one(ctx, "v_mad_u32_u24_Gops", mb_mad_u32_u24, n16, out, d_out));
    GA_CHECK(run_one(ctx, "v_lshl_add_u64_Gops", mb_lshl_add_u64, n16, out, d_out));
    GA_CHECK(run_one(ctx, "v_add_co_addc_pair_Gops", mb_add_co_addc, n8, out, d_out));
    GA_CHECK(run_one(ctx, "v_mov_b32_Gops", mb_mov_b32, n16, out, d_out));
    GA_CHECK(run_one(ctx, "v_fma_f64_Gops", mb_fma_f64, n16, out, d_out));
    GA_CHECK(run_one(ctx, "v_fma_f32_Gops", mb_fma_f32, n16, out, d_out));
    GA_CHECK(run_one(ctx, "fieldmul_bn254_fr_Gmul", mb_field_mul<BN254_Fr>, 2.0 * (MB_ITERS / 8), out, d_out));
    GA_CHECK(run_one(ctx, "fieldmul_bls12381_fp_Gmul", mb_field_mul<BLS12_381_Fp>, 2.0 * (MB_ITERS / 8), out, d_out));
    GA_CHECK(run_one(ctx, "f29mul_bn254_Gmul", mb_f29_mul<BN254_Fp>, 2.0 * (MB_ITERS / 8), out, d_out));
    GA_CHECK(run_one(ctx, "f29mul_bls12381_fp_Gmul", mb_f29_mul<BLS12_381_Fp>, 2.0 * (MB_ITERS / 8), out, d_out));
    GA_CHECK(run_one(ctx, "f29addsub_bn254_Gop", mb_f29_addsub<BN254_Fp>, 4.0 * (MB_ITERS / 8), out, d_out));
    GA_CHECK(run_one(ctx, "fe_addsub_bn254_Gop", mb_fe_addsub<BN254_Fp>, 4.0 * (MB_ITERS / 8), out, d_out));
    constexpr int ONE_WG = 22 * 1024;   // 88 KB of LDS per workgroup: one workgroup (4 waves) per CU
    GA_CHECK(run_one(ctx, "mad_dist1_Gops", mb_mad_chain<1, 0>, n16, out, d_out));
    GA_CHECK(run_one(ctx, "mad_dist2_Gops", mb_mad_chain<2, 0>, n16, out, d_out));
    GA_CHECK(run_one(ctx, "mad_dist4_Gops", mb_mad_chain<4, 0>, n16, out, d_out));
    GA_CHECK(run_one(ctx, "mad_dist8_Gops", mb_mad_chain<8, 0>, n16, out, d_out));
    GA_CHECK(run_one(ctx, "mad_dist1_one_wave_per_simd_Gops", mb_mad_chain<1, ONE_WG>, n16, out, d_out));
    GA_CHECK(run_one(ctx, "mad_dist2_one_wave_per_simd_Gops", mb_mad_chain<2, ONE_WG>, n16, out, d_out));
    GA_CHECK(run_one(ctx, "mad_dist8_one_wave_per_simd_Gops", mb_mad_chain<8, ONE_WG>, n16, out, d_out));
    const double n64 = 64.0 * (MB_ITERS / 4);
    GA_CHECK(run_one(ctx, "mad64_unroll64_Gops", mb_mad_mix<0>, n64, out, d_out, 20));
    GA_CHECK(run_one(ctx, "mad64_two_sdst_Gops", mb_mad_mix<1>, n64, out, d_out, 20));
    GA_CHECK(run_one(ctx, "mad64_sgpr_operand_Gops", mb_mad_mix<2>, n64, out, d_out, 20));
    GA_CHECK(run_one(ctx, "mad64_with_carry_per8_Gmad", mb_mad_mix<3>, n64, out, d_out, 20));
    GA_CHECK(run_one(ctx, "mad64_reduction_row_Gmad", mb_mad_mix<4>, n64, out, d_out, 20));
    GA_CHECK(run_one(ctx, "mad64_operands_spread_over_banks_Gops", mb_mad_banks<0>, 64.0 * 512, out, d_out, 20));
    GA_CHECK(run_one(ctx, "mad64_operands_in_one_bank_Gops", mb_mad_banks<1>, 64.0 * 512, out, d_out, 20));
    GA_CHECK(run_one(ctx, "f29mul_bn254_ilp1_Gmul", (mb_f29_mul_ilp<BN254_Fp, 1>), 2.0 * (MB_ITERS / 8), out, d_out, 20));
    GA_CHECK(run_one(ctx, "f29mul_bn254_ilp2_Gmul", (mb_f29_mul_ilp<BN254_Fp, 2>), 4.0 * (MB_ITERS / 8), out, d_out, 20));
    GA_CHECK(run_one(ctx, "f29mul_bn254_ilp3_Gmul", (mb_f29_mul_ilp<BN254_Fp, 3>), 6.0 * (MB_ITERS / 8), out, d_out, 20));
    // the same instruction streams held for ~0.2 s each
    GA_CHECK(run_one(ctx, "v_mad_u64_u32_sustained_Gops", mb_mad_u64_u32, n16, out, d_out, 400));
    GA_CHECK(run_one(ctx, "v_mov_b32_sustained_Gops", mb_mov_b32, n16, out, d_out, 800));
    GA_CHECK(run_one(ctx, "f29mul_bn254_sustained_Gmul", mb_f29_mul<BN254_Fp>, 2.0 * (MB_ITERS / 8), out, d_out, 100));
    GA_CHECK(run_one(ctx, "v_mad_u64_u32_after_Gops", mb_mad_u64_u32, n16, out, d_out));
    snprintf(buf, cap, "%s", out.c_str());
    return GA_OK;
}

}  // namespace ga
