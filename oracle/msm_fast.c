/* msm_fast.c -- the CPU BASELINE bench.py quotes beside the GPU figure.  TEST INFRASTRUCTURE ONLY (same rule as oracle.c:
 * only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load liboracle.so).
 *
 * What it stands in for: the G1 MultiExp calls of backend/groth16/bn254/prove.go:189-283, which live in gnark-crypto
 * v0.21.0 (go.mod:10; not in /root/reference, no Go toolchain here).  It restates the ALGORITHM gnark-crypto publishes for
 * its CPU MultiExp rather than the simplest Pippenger oracle.c's oracle_msm is:
 *   - signed c-bit digits (2^(c-1) buckets per window);
 *   - batch-affine bucket accumulation: up to 1024 additions into distinct buckets share ONE field inversion
 *     (Montgomery's trick), conflicting additions wait in a queue;
 *   - (window x point-range) tasks on a thread pool, so that every core works when there are more cores than windows
 *     (prove.go sizes its MSMs with runtime.NumCPU(); oracle_msm stops at one thread per window);
 *   - fixed-size limb arithmetic the compiler unrolls (-O3), 64x64->128 products (mulx/adx where the target has them).
 * It is still plain C, not gnark-crypto's assembly: kind "port-batch-affine".  Checked against oracle_msm and the
 * known-dlog closed form in tests/test_oracle.py. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_constants.h"

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define N 4
#define FN(x) CAT(x, _bn254)
#define P_MOD BN254_FP_MOD
#define P_INV 0x87d20782e4866389ull
#define P_ONE BN254_FP_ONE
#include "msm_fast_body.inc"
#undef N
#undef FN
#undef P_MOD
#undef P_INV
#undef P_ONE

#define N 6
#define FN(x) CAT(x, _bls12381)
#define P_MOD BLS12_381_FP_MOD
#define P_INV 0x89f3fffcfffcfffdull
#define P_ONE BLS12_381_FP_ONE
#include "msm_fast_body.inc"
#undef N
#undef FN
#undef P_MOD
#undef P_INV
#undef P_ONE

/* scalar field (4 limbs on both curves): Montgomery -> canonical = one CIOS product by 1 */
static void fr_from_mont(const u64* mod, u64 inv, u64* r, const u64* a) {
    u64 t[6] = {a[0], a[1], a[2], a[3], 0, 0};
    for (int i = 0; i < 4; i++) {
        u64 m = t[0] * inv;
        u128 s = (u128)m * mod[0] + t[0];
        u64 c = (u64)(s >> 64);
        for (int j = 1; j < 4; j++) {
            s = (u128)m * mod[j] + t[j] + c;
            t[j - 1] = (u64)s;
            c = (u64)(s >> 64);
        }
        s = (u128)t[4] + c;
        t[3] = (u64)s;
        t[4] = (u64)(s >> 64);
    }
    /* t < 2r here; one conditional subtraction */
    int ge = 1;
    for (int i = 3; i >= 0; i--) {
        if (t[i] > mod[i]) break;
        if (t[i] < mod[i]) {
            ge = 0;
            break;
        }
    }
    if (ge) {
        u64 br = 0;
        for (int i = 0; i < 4; i++) {
            u128 d = (u128)t[i] - mod[i] - br;
            t[i] = (u64)d;
            br = (u64)(d >> 64) & 1;
        }
    }
    memcpy(r, t, 32);
}

typedef struct {
    const u64* scalars;
    const u64* mod;
    u64 inv;
    int32_t* digits;
    size_t n, lo, hi;
    int c, nwin, mont;
} digit_job;

static void* digit_worker(void* p) {
    digit_job* j = (digit_job*)p;
    const int c = j->c;
    for (size_t i = j->lo; i < j->hi; i++) {
        u64 s[4];
        if (j->mont) fr_from_mont(j->mod, j->inv, s, j->scalars + 4 * i);
        else memcpy(s, j->scalars + 4 * i, 32);
        int carry = 0;
        for (int w = 0; w < j->nwin; w++) {
            const int bit = w * c;
            u64 raw = 0;
            if (bit < 256) {
                const int word = bit >> 6, off = bit & 63;
                raw = s[word] >> off;
                if (off + c > 64 && word + 1 < 4) raw |= s[word + 1] << (64 - off);
                raw &= ((u64)1 << c) - 1;
            }
            int dg = (int)raw + carry;
            if (dg > (1 << (c - 1))) {
                dg -= (1 << c);
                carry = 1;
            } else
                carry = 0;
            j->digits[(size_t)w * j->n + i] = dg;
        }
    }
    return NULL;
}

/* window width and range splits: the makespan of ceil(nwin*splits / threads) waves of tasks, a task = n/splits batch-affine
 * additions (~7 products each) + 2^(c-1) x 2 Jacobian additions (~14) */
static void fast_plan(int bits, size_t n, int threads, int* c_out, int* splits_out) {
    double best = 1e300;
    int bc = 2, bs = 1;
    for (int c = 2; c <= 16; c++) {
        const int nwin = bits / c + 1;
        const double nb = (double)(1u << (c - 1));
        for (int s = 1; s <= (threads > 1 ? threads : 1); s++) {
            const double per = (double)n / s;
            const double task = 7.0 * per + 28.0 * nb;
            const int waves = (nwin * s + threads - 1) / threads;
            const double cost = waves * task + 1e-3 * nwin * s; /* (ties: fewer tasks) */
            if (cost < best) {
                best = cost;
                bc = c;
                bs = s;
            }
        }
    }
    *c_out = bc;
    *splits_out = bs;
}

/* what oracle_msm_fast will do for (curve, n, nthreads): out3 = {window bits, windows, range splits} */
void oracle_msm_fast_plan(int curve, size_t n, int nthreads, int* out3) {
    const int bits = curve == 0 ? BN254_FR_BITS : BLS12_381_FR_BITS;
    int c, s;
    fast_plan(bits, n ? n : 1, nthreads < 1 ? 1 : nthreads, &c, &s);
    out3[0] = c;
    out3[1] = bits / c + 1;
    out3[2] = s;
}

/* sum scalars[i] * points[i] over G1 -> Jacobian image (same contract as oracle_msm with group = 0).
 * force_c / force_splits > 0 override the plan (tests). */
int oracle_msm_fast(int curve, const u64* points, const u64* scalars, size_t n, int mont, u64* out_jac, int nthreads, int force_c,
                    int force_splits) {
    if (curve != 0 && curve != 1) return -1;
    if (nthreads < 1) nthreads = 1;
    const int bits = curve == 0 ? BN254_FR_BITS : BLS12_381_FR_BITS;
    int c, splits;
    fast_plan(bits, n ? n : 1, nthreads, &c, &splits);
    if (force_c > 1 && force_c <= 16) c = force_c;
    if (force_splits > 0) splits = force_splits;
    if ((size_t)splits > n) splits = n ? (int)n : 1;
    const int nwin = bits / c + 1;
    int32_t* digits = (int32_t*)malloc(sizeof(int32_t) * (size_t)nwin * (n ? n : 1));
    {
        int T = nthreads > 64 ? 64 : nthreads;
        if (n < 4096) T = 1;
        digit_job jobs[64];
        pthread_t th[64];
        for (int t = 0; t < T; t++) {
            jobs[t] = (digit_job){scalars, curve == 0 ? BN254_FR_MOD : BLS12_381_FR_MOD, curve == 0 ? 0xc2e1f593efffffffull : 0xfffffffeffffffffull,
                                  digits, n, n * (size_t)t / T, n * (size_t)(t + 1) / T, c, nwin, mont};
            if (T == 1) digit_worker(&jobs[t]);
            else pthread_create(&th[t], NULL, digit_worker, &jobs[t]);
        }
        if (T > 1)
            for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    }
    int rc = curve == 0 ? msm_fast_bn254(points, digits, n, c, nwin, splits, nthreads, out_jac)
                        : msm_fast_bls12381(points, digits, n, c, nwin, splits, nthreads, out_jac);
    free(digits);
    return rc;
}
