/* oracle.c -- CPU restatement of the Groth16 prover hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load liboracle.so; the product
 * (gnark_amd/, libgnark_amd.so) never links or calls it.
 *
 * What it restates (reference = /root/reference, gnark v0.16.0):
 *   - computeH ................. backend/groth16/bn254/prove.go:346-389
 *   - Prove (no commitments) ... backend/groth16/bn254/prove.go:130-315
 *   - MultiExp / FFT ........... live in the external module github.com/consensys/gnark-crypto v0.21.0 (go.mod:10),
 *     whose source is NOT in /root/reference.  Restated here from the published algorithms gnark-crypto
 *     implements: Montgomery CIOS multiplication on 64-bit limbs; Jacobian / mixed addition (EFD
 *     "dbl-2009-l", "madd-2007-bl", "add-2007-bl"); bucket-method (Pippenger) MSM with signed c-bit digits,
 *     one thread per window, running-sum bucket reduction, Horner window combination;
 *     in-place radix-2 DIF / DIT NTT with the 1/n and coset conventions of fft.Domain.
 *
 * Parity pinning: this file is checked in tests/test_oracle.py against oracle/pyref.py (independent big-integer
 * code), which in turn is pinned to the fixtures the reference ships (tests/test_oracle_fixtures.py: the
 * EIP-4844 KZG setup relations, the [2^65]G2 literals, serialized verifying keys).  gnark itself cannot be run in
 * the build image (no Go toolchain), so no vector produced BY gnark's prover exists: prover outputs are pinned
 * mathematically (an MSM result / a polynomial is unique), not by golden bytes.  See oracle/README.md.
 *
 * Build: oracle/build.sh  ->  oracle/liboracle.so      (gcc -O2 -shared -fPIC -pthread)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_constants.h"

typedef unsigned __int128 u128;
typedef uint64_t u64;
#define MAXN 6

typedef struct {
    int n;
    int bits;
    const u64* mod;
    u64 inv;
    const u64* one;
    const u64* r2;
} field_t;

typedef struct {
    field_t fp, fr;
    const u64 *g1x, *g1y, *g2x0, *g2x1, *g2y0, *g2y1;
    const u64 *fr_root, *fr_root_inv, *fr_gen, *fr_gen_inv;
    int adicity;
} curve_t;

static const curve_t CURVES[2] = {
    {{BN254_FP_N, BN254_FP_BITS, BN254_FP_MOD, 0x87d20782e4866389ull, BN254_FP_ONE, BN254_FP_R2},
     {BN254_FR_N, BN254_FR_BITS, BN254_FR_MOD, 0xc2e1f593efffffffull, BN254_FR_ONE, BN254_FR_R2},
     BN254_FP_G1X, BN254_FP_G1Y, BN254_FP_G2X0, BN254_FP_G2X1, BN254_FP_G2Y0, BN254_FP_G2Y1,
     BN254_FR_ROOT, BN254_FR_ROOT_INV, BN254_FR_GEN, BN254_FR_GEN_INV, BN254_FR_ADICITY},
    {{BLS12_381_FP_N, BLS12_381_FP_BITS, BLS12_381_FP_MOD, 0x89f3fffcfffcfffdull, BLS12_381_FP_ONE, BLS12_381_FP_R2},
     {BLS12_381_FR_N, BLS12_381_FR_BITS, BLS12_381_FR_MOD, 0xfffffffeffffffffull, BLS12_381_FR_ONE, BLS12_381_FR_R2},
     BLS12_381_FP_G1X, BLS12_381_FP_G1Y, BLS12_381_FP_G2X0, BLS12_381_FP_G2X1, BLS12_381_FP_G2Y0, BLS12_381_FP_G2Y1,
     BLS12_381_FR_ROOT, BLS12_381_FR_ROOT_INV, BLS12_381_FR_GEN, BLS12_381_FR_GEN_INV, BLS12_381_FR_ADICITY},
};

/* ---------------- prime field, Montgomery form, 64-bit limbs (gnark-crypto fp/fr.Element layout) -------- */

static int fe_is_zero(const field_t* f, const u64* a) {
    u64 o = 0;
    for (int i = 0; i < f->n; i++) o |= a[i];
    return o == 0;
}
static int fe_eq(const field_t* f, const u64* a, const u64* b) { return memcmp(a, b, 8 * f->n) == 0; }
static void fe_copy(const field_t* f, u64* r, const u64* a) { memcpy(r, a, 8 * f->n); }
static void fe_zero(const field_t* f, u64* r) { memset(r, 0, 8 * f->n); }

static int geq(const field_t* f, const u64* a) {
    for (int i = f->n - 1; i >= 0; i--) {
        if (a[i] > f->mod[i]) return 1;
        if (a[i] < f->mod[i]) return 0;
    }
    return 1;
}
static void sub_mod_raw(const field_t* f, u64* a) {
    u64 borrow = 0;
    for (int i = 0; i < f->n; i++) {
        u128 d = (u128)a[i] - f->mod[i] - borrow;
        a[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}
static void fe_add(const field_t* f, u64* r, const u64* a, const u64* b) {
    u64 c = 0;
    for (int i = 0; i < f->n; i++) {
        u128 s = (u128)a[i] + b[i] + c;
        r[i] = (u64)s;
        c = (u64)(s >> 64);
    }
    if (c || geq(f, r)) sub_mod_raw(f, r);
}
static void fe_sub(const field_t* f, u64* r, const u64* a, const u64* b) {
    u64 borrow = 0;
    for (int i = 0; i < f->n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    if (borrow) {
        u64 c = 0;
        for (int i = 0; i < f->n; i++) {
            u128 s = (u128)r[i] + f->mod[i] + c;
            r[i] = (u64)s;
            c = (u64)(s >> 64);
        }
    }
}
static void fe_neg(const field_t* f, u64* r, const u64* a) {
    if (fe_is_zero(f, a)) {
        fe_zero(f, r);
        return;
    }
    u64 z[MAXN] = {0};
    fe_sub(f, r, z, a);
}
/* CIOS Montgomery multiplication (Koc et al.), t has n+2 words */
static void fe_mul(const field_t* f, u64* r, const u64* a, const u64* b) {
    const int n = f->n;
    u64 t[MAXN + 2] = {0};
    for (int i = 0; i < n; i++) {
        u64 c = 0;
        for (int j = 0; j < n; j++) {
            u128 s = (u128)a[j] * b[i] + t[j] + c;
            t[j] = (u64)s;
            c = (u64)(s >> 64);
        }
        u128 s = (u128)t[n] + c;
        t[n] = (u64)s;
        t[n + 1] = (u64)(s >> 64);
        u64 m = t[0] * f->inv;
        s = (u128)m * f->mod[0] + t[0];
        c = (u64)(s >> 64);
        for (int j = 1; j < n; j++) {
            s = (u128)m * f->mod[j] + t[j] + c;
            t[j - 1] = (u64)s;
            c = (u64)(s >> 64);
        }
        s = (u128)t[n] + c;
        t[n - 1] = (u64)s;
        t[n] = t[n + 1] + (u64)(s >> 64);
    }
    if (t[n] || geq(f, t)) sub_mod_raw(f, t);
    memcpy(r, t, 8 * n);
}
static void fe_from_mont(const field_t* f, u64* r, const u64* a) {
    u64 one[MAXN] = {1};
    fe_mul(f, r, a, one);
}
static void fe_pow(const field_t* f, u64* r, const u64* a, const u64* e, int ewords) {
    u64 acc[MAXN], base[MAXN];
    fe_copy(f, acc, f->one);
    fe_copy(f, base, a);
    for (int i = ewords - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            fe_mul(f, acc, acc, acc);
            if ((e[i] >> b) & 1) fe_mul(f, acc, acc, base);
        }
    fe_copy(f, r, acc);
}
static void fe_inv(const field_t* f, u64* r, const u64* a) {
    u64 e[MAXN];
    memcpy(e, f->mod, 8 * f->n);
    e[0] -= 2; /* all moduli here end in ...01/...47/...ab: no borrow */
    fe_pow(f, r, a, e, f->n);
}

/* ---------------- tower element: degree 1 (Fp) or 2 (Fp2 = Fp[u]/(u^2+1)) ------------------------------ */
typedef struct {
    u64 c[2][MAXN];
} el;
typedef struct {
    const field_t* f;
    int deg;
} tower_t;

static void el_add(const tower_t* t, el* r, const el* a, const el* b) {
    for (int k = 0; k < t->deg; k++) fe_add(t->f, r->c[k], a->c[k], b->c[k]);
}
static void el_sub(const tower_t* t, el* r, const el* a, const el* b) {
    for (int k = 0; k < t->deg; k++) fe_sub(t->f, r->c[k], a->c[k], b->c[k]);
}
static void el_neg(const tower_t* t, el* r, const el* a) {
    for (int k = 0; k < t->deg; k++) fe_neg(t->f, r->c[k], a->c[k]);
}
static void el_dbl(const tower_t* t, el* r, const el* a) { el_add(t, r, a, a); }
static int el_is_zero(const tower_t* t, const el* a) {
    for (int k = 0; k < t->deg; k++)
        if (!fe_is_zero(t->f, a->c[k])) return 0;
    return 1;
}
static int el_eq(const tower_t* t, const el* a, const el* b) {
    for (int k = 0; k < t->deg; k++)
        if (!fe_eq(t->f, a->c[k], b->c[k])) return 0;
    return 1;
}
static void el_mul(const tower_t* t, el* r, const el* a, const el* b) {
    if (t->deg == 1) {
        fe_mul(t->f, r->c[0], a->c[0], b->c[0]);
        return;
    }
    /* schoolbook: (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u */
    u64 t0[MAXN], t1[MAXN], t2[MAXN], t3[MAXN];
    fe_mul(t->f, t0, a->c[0], b->c[0]);
    fe_mul(t->f, t1, a->c[1], b->c[1]);
    fe_mul(t->f, t2, a->c[0], b->c[1]);
    fe_mul(t->f, t3, a->c[1], b->c[0]);
    fe_sub(t->f, r->c[0], t0, t1);
    fe_add(t->f, r->c[1], t2, t3);
}
static void el_sqr(const tower_t* t, el* r, const el* a) { el_mul(t, r, a, a); }
static void el_inv(const tower_t* t, el* r, const el* a) {
    if (t->deg == 1) {
        fe_inv(t->f, r->c[0], a->c[0]);
        return;
    }
    u64 n0[MAXN], n1[MAXN], d[MAXN];
    fe_mul(t->f, n0, a->c[0], a->c[0]);
    fe_mul(t->f, n1, a->c[1], a->c[1]);
    fe_add(t->f, d, n0, n1);
    fe_inv(t->f, d, d);
    fe_mul(t->f, r->c[0], a->c[0], d);
    fe_mul(t->f, n1, a->c[1], d);
    fe_neg(t->f, r->c[1], n1);
}
static void el_one(const tower_t* t, el* r) {
    memset(r, 0, sizeof(*r));
    fe_copy(t->f, r->c[0], t->f->one);
}

/* ---------------- points: affine (inf = 0,0) and Jacobian (x = X/Z^2, y = Y/Z^3) ------------------------ */
typedef struct {
    el x, y;
} aff_t;
typedef struct {
    el x, y, z;
} jac_t;

static void jac_set_inf(const tower_t* t, jac_t* p) {
    el_one(t, &p->x);
    el_one(t, &p->y);
    memset(&p->z, 0, sizeof(el));
}
static int aff_is_inf(const tower_t* t, const aff_t* p) { return el_is_zero(t, &p->x) && el_is_zero(t, &p->y); }

/* dbl-2009-l (a = 0) */
static void jac_dbl(const tower_t* t, jac_t* r, const jac_t* p) {
    if (el_is_zero(t, &p->z)) {
        *r = *p;
        return;
    }
    el A, B, C, D, E, F, X3, Y3, Z3, tmp;
    el_sqr(t, &A, &p->x);
    el_sqr(t, &B, &p->y);
    el_sqr(t, &C, &B);
    el_add(t, &tmp, &p->x, &B);
    el_sqr(t, &tmp, &tmp);
    el_sub(t, &tmp, &tmp, &A);
    el_sub(t, &tmp, &tmp, &C);
    el_dbl(t, &D, &tmp);
    el_dbl(t, &E, &A);
    el_add(t, &E, &E, &A);
    el_sqr(t, &F, &E);
    el_dbl(t, &tmp, &D);
    el_sub(t, &X3, &F, &tmp);
    el_sub(t, &tmp, &D, &X3);
    el_mul(t, &Y3, &E, &tmp);
    el_dbl(t, &tmp, &C);
    el_dbl(t, &tmp, &tmp);
    el_dbl(t, &tmp, &tmp);
    el_sub(t, &Y3, &Y3, &tmp);
    el_mul(t, &Z3, &p->y, &p->z);
    el_dbl(t, &Z3, &Z3);
    r->x = X3;
    r->y = Y3;
    r->z = Z3;
}
/* add-2007-bl with the doubling / inverse cases handled */
static void jac_add(const tower_t* t, jac_t* r, const jac_t* p, const jac_t* q) {
    if (el_is_zero(t, &p->z)) {
        *r = *q;
        return;
    }
    if (el_is_zero(t, &q->z)) {
        *r = *p;
        return;
    }
    el Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, X3, Y3, Z3, tmp;
    el_sqr(t, &Z1Z1, &p->z);
    el_sqr(t, &Z2Z2, &q->z);
    el_mul(t, &U1, &p->x, &Z2Z2);
    el_mul(t, &U2, &q->x, &Z1Z1);
    el_mul(t, &S1, &p->y, &q->z);
    el_mul(t, &S1, &S1, &Z2Z2);
    el_mul(t, &S2, &q->y, &p->z);
    el_mul(t, &S2, &S2, &Z1Z1);
    if (el_eq(t, &U1, &U2)) {
        if (el_eq(t, &S1, &S2)) {
            jac_dbl(t, r, p);
            return;
        }
        jac_set_inf(t, r);
        return;
    }
    el_sub(t, &H, &U2, &U1);
    el_dbl(t, &I, &H);
    el_sqr(t, &I, &I);
    el_mul(t, &J, &H, &I);
    el_sub(t, &rr, &S2, &S1);
    el_dbl(t, &rr, &rr);
    el_mul(t, &V, &U1, &I);
    el_sqr(t, &X3, &rr);
    el_sub(t, &X3, &X3, &J);
    el_dbl(t, &tmp, &V);
    el_sub(t, &X3, &X3, &tmp);
    el_sub(t, &tmp, &V, &X3);
    el_mul(t, &Y3, &rr, &tmp);
    el_mul(t, &tmp, &S1, &J);
    el_dbl(t, &tmp, &tmp);
    el_sub(t, &Y3, &Y3, &tmp);
    el_add(t, &Z3, &p->z, &q->z);
    el_sqr(t, &Z3, &Z3);
    el_sub(t, &Z3, &Z3, &Z1Z1);
    el_sub(t, &Z3, &Z3, &Z2Z2);
    el_mul(t, &Z3, &Z3, &H);
    r->x = X3;
    r->y = Y3;
    r->z = Z3;
}
static void jac_from_aff(const tower_t* t, jac_t* r, const aff_t* a) {
    if (aff_is_inf(t, a)) {
        jac_set_inf(t, r);
        return;
    }
    r->x = a->x;
    r->y = a->y;
    el_one(t, &r->z);
}
static void jac_add_mixed(const tower_t* t, jac_t* r, const jac_t* p, const aff_t* a, int negate) {
    jac_t q;
    jac_from_aff(t, &q, a);
    if (negate) el_neg(t, &q.y, &q.y);
    jac_add(t, r, p, &q);
}
static void jac_to_aff(const tower_t* t, aff_t* r, const jac_t* p) {
    if (el_is_zero(t, &p->z)) {
        memset(r, 0, sizeof(*r));
        return;
    }
    el zi, zi2, zi3;
    el_inv(t, &zi, &p->z);
    el_sqr(t, &zi2, &zi);
    el_mul(t, &zi3, &zi2, &zi);
    el_mul(t, &r->x, &p->x, &zi2);
    el_mul(t, &r->y, &p->y, &zi3);
}
/* [k]P, k little-endian canonical words */
static void jac_scalar_mul(const tower_t* t, jac_t* r, const jac_t* p, const u64* k, int kwords) {
    jac_t acc;
    jac_set_inf(t, &acc);
    for (int i = kwords - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            jac_dbl(t, &acc, &acc);
            if ((k[i] >> b) & 1) jac_add(t, &acc, &acc, p);
        }
    *r = acc;
}

/* ---------------- packing to / from gnark memory images ------------------------------------------------- */
static void load_el(const tower_t* t, el* r, const u64* p) {
    memset(r, 0, sizeof(*r));
    for (int k = 0; k < t->deg; k++) memcpy(r->c[k], p + k * t->f->n, 8 * t->f->n);
}
static void store_el(const tower_t* t, u64* p, const el* a) {
    for (int k = 0; k < t->deg; k++) memcpy(p + k * t->f->n, a->c[k], 8 * t->f->n);
}
static int el_words(const tower_t* t) { return t->deg * t->f->n; }
static void load_aff(const tower_t* t, aff_t* r, const u64* p) {
    load_el(t, &r->x, p);
    load_el(t, &r->y, p + el_words(t));
}
static void store_aff(const tower_t* t, u64* p, const aff_t* a) {
    store_el(t, p, &a->x);
    store_el(t, p + el_words(t), &a->y);
}
static void load_jac(const tower_t* t, jac_t* r, const u64* p) {
    load_el(t, &r->x, p);
    load_el(t, &r->y, p + el_words(t));
    load_el(t, &r->z, p + 2 * el_words(t));
}
static void store_jac(const tower_t* t, u64* p, const jac_t* a) {
    store_el(t, p, &a->x);
    store_el(t, p + el_words(t), &a->y);
    store_el(t, p + 2 * el_words(t), &a->z);
}
static tower_t tower_of(int curve, int group) {
    tower_t t = {&CURVES[curve].fp, group == 0 ? 1 : 2};
    return t;
}

/* ---------------- MSM: bucket method, signed digits, one thread per window ------------------------------ */
typedef struct {
    const tower_t* t;
    const u64* points;
    const int32_t* digits; /* [nwin][n] */
    size_t n;
    int c, w;
    jac_t result;
} win_job;

static void* msm_window(void* arg) {
    win_job* j = (win_job*)arg;
    const tower_t* t = j->t;
    const int nb = 1 << (j->c - 1);
    const int aw = 2 * el_words(t);
    jac_t* buckets = (jac_t*)malloc(sizeof(jac_t) * nb);
    for (int b = 0; b < nb; b++) jac_set_inf(t, &buckets[b]);
    const int32_t* d = j->digits + (size_t)j->w * j->n;
    for (size_t i = 0; i < j->n; i++) {
        int32_t dg = d[i];
        if (dg == 0) continue;
        aff_t a;
        load_aff(t, &a, j->points + i * aw);
        if (aff_is_inf(t, &a)) continue;
        if (dg > 0) jac_add_mixed(t, &buckets[dg - 1], &buckets[dg - 1], &a, 0);
        else jac_add_mixed(t, &buckets[-dg - 1], &buckets[-dg - 1], &a, 1);
    }
    jac_t running, total;
    jac_set_inf(t, &running);
    jac_set_inf(t, &total);
    for (int b = nb - 1; b >= 0; b--) {
        jac_add(t, &running, &running, &buckets[b]);
        jac_add(t, &total, &total, &running);
    }
    j->result = total;
    free(buckets);
    return NULL;
}

static int msm_best_c(int bits, size_t n) {
    double best = 1e300;
    int bc = 2;
    for (int c = 2; c <= 16; c++) {
        int nwin = bits / c + 1;
        double cost = (double)nwin * ((double)n + 2.0 * (double)(1u << (c - 1)));
        if (cost < best) {
            best = cost;
            bc = c;
        }
    }
    return bc;
}

/* number of windows (= max useful threads) oracle_msm uses for n points */
int oracle_msm_windows(int curve, size_t n) {
    const int bits = CURVES[curve].fr.bits;
    return bits / msm_best_c(bits, n ? n : 1) + 1;
}

/* sum scalars[i]*points[i] -> Jacobian image.  scalars: fr images (Montgomery if mont). */
int oracle_msm(int curve, int group, const u64* points, const u64* scalars, size_t n, int mont, u64* out_jac,
               int nthreads) {
    const curve_t* cv = &CURVES[curve];
    tower_t t = tower_of(curve, group);
    const int bits = cv->fr.bits;
    const int c = msm_best_c(bits, n ? n : 1);
    const int nwin = bits / c + 1;
    int32_t* digits = (int32_t*)malloc(sizeof(int32_t) * (size_t)nwin * (n ? n : 1));
    for (size_t i = 0; i < n; i++) {
        u64 s[4];
        if (mont) fe_from_mont(&cv->fr, s, scalars + 4 * i);
        else memcpy(s, scalars + 4 * i, 32);
        int carry = 0;
        for (int w = 0; w < nwin; w++) {
            int bit = w * c;
            u64 raw = 0;
            if (bit < 256) {
                int word = bit >> 6, off = bit & 63;
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
            digits[(size_t)w * n + i] = dg;
        }
    }
    win_job* jobs = (win_job*)calloc(nwin, sizeof(win_job));
    pthread_t* th = (pthread_t*)calloc(nwin, sizeof(pthread_t));
    if (nthreads < 1) nthreads = 1;
    for (int w0 = 0; w0 < nwin; w0 += nthreads) {
        int cnt = nwin - w0 < nthreads ? nwin - w0 : nthreads;
        for (int k = 0; k < cnt; k++) {
            win_job* j = &jobs[w0 + k];
            j->t = &t;
            j->points = points;
            j->digits = digits;
            j->n = n;
            j->c = c;
            j->w = w0 + k;
            if (cnt == 1) msm_window(j);
            else pthread_create(&th[w0 + k], NULL, msm_window, j);
        }
        if (cnt > 1)
            for (int k = 0; k < cnt; k++) pthread_join(th[w0 + k], NULL);
    }
    jac_t acc;
    jac_set_inf(&t, &acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) jac_dbl(&t, &acc, &acc);
        jac_add(&t, &acc, &acc, &jobs[w].result);
    }
    store_jac(&t, out_jac, &acc);
    free(jobs);
    free(th);
    free(digits);
    return 0;
}

/* naive double-and-add MSM (second, independent path for small n) */
int oracle_msm_naive(int curve, int group, const u64* points, const u64* scalars, size_t n, int mont, u64* out_jac) {
    const curve_t* cv = &CURVES[curve];
    tower_t t = tower_of(curve, group);
    const int aw = 2 * el_words(&t);
    jac_t acc;
    jac_set_inf(&t, &acc);
    for (size_t i = 0; i < n; i++) {
        u64 s[4];
        if (mont) fe_from_mont(&cv->fr, s, scalars + 4 * i);
        else memcpy(s, scalars + 4 * i, 32);
        aff_t a;
        jac_t p, q;
        load_aff(&t, &a, points + i * aw);
        jac_from_aff(&t, &p, &a);
        jac_scalar_mul(&t, &q, &p, s, 4);
        jac_add(&t, &acc, &acc, &q);
    }
    store_jac(&t, out_jac, &acc);
    return 0;
}

int oracle_jac_to_affine(int curve, int group, const u64* jac, u64* aff) {
    tower_t t = tower_of(curve, group);
    jac_t p;
    aff_t a;
    load_jac(&t, &p, jac);
    jac_to_aff(&t, &a, &p);
    store_aff(&t, aff, &a);
    return 0;
}
int oracle_jac_add(int curve, int group, const u64* a, const u64* b, u64* out) {
    tower_t t = tower_of(curve, group);
    jac_t p, q, r;
    load_jac(&t, &p, a);
    load_jac(&t, &q, b);
    jac_add(&t, &r, &p, &q);
    store_jac(&t, out, &r);
    return 0;
}
/* [k]G, k canonical little-endian 4 words */
int oracle_generator_mul(int curve, int group, const u64* k, u64* out_jac) {
    const curve_t* cv = &CURVES[curve];
    tower_t t = tower_of(curve, group);
    aff_t g;
    memset(&g, 0, sizeof(g));
    if (group == 0) {
        fe_copy(&cv->fp, g.x.c[0], cv->g1x);
        fe_copy(&cv->fp, g.y.c[0], cv->g1y);
    } else {
        fe_copy(&cv->fp, g.x.c[0], cv->g2x0);
        fe_copy(&cv->fp, g.x.c[1], cv->g2x1);
        fe_copy(&cv->fp, g.y.c[0], cv->g2y0);
        fe_copy(&cv->fp, g.y.c[1], cv->g2y1);
    }
    jac_t p, r;
    jac_from_aff(&t, &p, &g);
    jac_scalar_mul(&t, &r, &p, k, 4);
    store_jac(&t, out_jac, &r);
    return 0;
}
/* bases[i] = [k_i]G (affine images) for 64-bit k_i -- known-dlog key material for tests */
int oracle_gen_bases(int curve, int group, const u64* ks, size_t n, u64* out_aff) {
    tower_t t = tower_of(curve, group);
    const int aw = 2 * el_words(&t);
    for (size_t i = 0; i < n; i++) {
        u64 k[4] = {ks[i], 0, 0, 0};
        u64 j[6 * MAXN];
        oracle_generator_mul(curve, group, k, j);
        oracle_jac_to_affine(curve, group, j, out_aff + i * aw);
    }
    return 0;
}

/* ---------------- Fr vector helpers ----------------------------------------------------------------------- */
/* sum a_i (Montgomery) * b_i (canonical) -> canonical */
int oracle_fr_dot(int curve, const u64* a, const u64* b, size_t n, u64* out) {
    const field_t* f = &CURVES[curve].fr;
    u64 acc[4] = {0}, p[4];
    for (size_t i = 0; i < n; i++) {
        fe_mul(f, p, a + 4 * i, b + 4 * i);
        fe_add(f, acc, acc, p);
    }
    memcpy(out, acc, 32);
    return 0;
}
int oracle_fr_from_mont(int curve, const u64* a, size_t n, u64* out) {
    const field_t* f = &CURVES[curve].fr;
    for (size_t i = 0; i < n; i++) fe_from_mont(f, out + 4 * i, a + 4 * i);
    return 0;
}
int oracle_fr_mul(int curve, const u64* a, const u64* b, size_t n, u64* out) {
    const field_t* f = &CURVES[curve].fr;
    for (size_t i = 0; i < n; i++) fe_mul(f, out + 4 * i, a + 4 * i, b + 4 * i);
    return 0;
}

int oracle_fr_add(int curve, const u64* a, const u64* b, size_t n, u64* out) {
    const field_t* f = &CURVES[curve].fr;
    for (size_t i = 0; i < n; i++) fe_add(f, out + 4 * i, a + 4 * i, b + 4 * i);
    return 0;
}
int oracle_fr_sub(int curve, const u64* a, const u64* b, size_t n, u64* out) {
    const field_t* f = &CURVES[curve].fr;
    for (size_t i = 0; i < n; i++) fe_sub(f, out + 4 * i, a + 4 * i, b + 4 * i);
    return 0;
}
/* out[i] = first * base^i (all Montgomery) */
int oracle_fr_powers(int curve, const u64* base, const u64* first, size_t n, u64* out) {
    const field_t* f = &CURVES[curve].fr;
    u64 cur[4];
    fe_copy(f, cur, first);
    for (size_t i = 0; i < n; i++) {
        fe_copy(f, out + 4 * i, cur);
        fe_mul(f, cur, cur, base);
    }
    return 0;
}

/* p(x) by Horner; coefficients and x in Montgomery form, result Montgomery */
int oracle_fr_horner(int curve, const u64* coeffs, size_t n, const u64* x, u64* out) {
    const field_t* f = &CURVES[curve].fr;
    u64 acc[4] = {0};
    for (size_t i = n; i-- > 0;) {
        fe_mul(f, acc, acc, x);
        fe_add(f, acc, acc, coeffs + 4 * i);
    }
    memcpy(out, acc, 32);
    return 0;
}

/* ---------------- NTT (fft.Domain conventions) ------------------------------------------------------------- */
static int ilog2(u64 n) {
    int l = 0;
    while (((u64)1 << l) < n) l++;
    return l;
}
static u64 bitrev_u64(u64 i, int logn) {
    u64 r = 0;
    for (int k = 0; k < logn; k++) {
        r = (r << 1) | (i & 1);
        i >>= 1;
    }
    return r;
}
static void root_of_unity(const curve_t* cv, u64 n, int inverse, u64* w) {
    fe_copy(&cv->fr, w, inverse ? cv->fr_root_inv : cv->fr_root);
    for (int k = 0; k < cv->adicity - ilog2(n); k++) fe_mul(&cv->fr, w, w, w);
}
/* ---- parallel-for over [0, total) on `nthreads` pthreads (the checker may use the host's cores for the BASELINE-size
 * cases; nthreads <= 1 runs inline) ---- */
typedef void (*par_body)(void* ctx, u64 lo, u64 hi, int tid);
typedef struct {
    par_body fn;
    void* ctx;
    u64 lo, hi;
    int tid;
} par_arg;
static void* par_tramp(void* p) {
    par_arg* a = (par_arg*)p;
    a->fn(a->ctx, a->lo, a->hi, a->tid);
    return NULL;
}
static void par_for(int nthreads, u64 total, par_body fn, void* ctx) {
    if (nthreads > 64) nthreads = 64;
    if (nthreads <= 1 || total < 4096) {
        fn(ctx, 0, total, 0);
        return;
    }
    pthread_t th[64];
    par_arg args[64];
    u64 chunk = (total + nthreads - 1) / nthreads;
    int used = 0;
    for (int t = 0; t < nthreads; t++) {
        u64 lo = (u64)t * chunk, hi = lo + chunk > total ? total : lo + chunk;
        if (lo >= hi) break;
        args[t] = (par_arg){fn, ctx, lo, hi, t};
        pthread_create(&th[t], NULL, par_tramp, &args[t]);
        used++;
    }
    for (int t = 0; t < used; t++) pthread_join(th[t], NULL);
}

typedef struct {
    const field_t* f;
    u64* a;
    const u64* tw;
    int logn, s, dit;
} fft_stage_ctx;
static void fft_stage_body(void* p, u64 lo, u64 hi, int tid) {
    (void)tid;
    fft_stage_ctx* c = (fft_stage_ctx*)p;
    const field_t* f = c->f;
    const int s = c->s;
    const u64 h = (u64)1 << s;
    for (u64 t = lo; t < hi; t++) { /* butterfly t: pair (base + j, base + j + h) */
        u64 j = t & (h - 1), base = (t >> s) << (s + 1);
        u64 *x = c->a + 4 * (base + j), *y = c->a + 4 * (base + j + h);
        const u64* w = c->tw + 4 * (j << (c->logn - 1 - s));
        if (!c->dit) {
            u64 d[4];
            fe_sub(f, d, x, y);
            fe_add(f, x, x, y);
            fe_mul(f, y, d, w);
        } else {
            u64 v[4], x0[4];
            fe_mul(f, v, y, w);
            fe_copy(f, x0, x);
            fe_add(f, x, x0, v);
            fe_sub(f, y, x0, v);
        }
    }
}
typedef struct {
    const field_t* f;
    u64* a;
    const u64 *g, *ginv; /* ratio of the geometric progression and its inverse */
    const u64* first;    /* value at exponent 0 */
    int logn, bitrev;
} fft_scale_ctx;
/* a[i] *= first * g^(i or bitrev(i)) */
static void fft_scale_body(void* p, u64 lo, u64 hi, int tid) {
    (void)tid;
    fft_scale_ctx* c = (fft_scale_ctx*)p;
    const field_t* f = c->f;
    if (!c->bitrev) {
        u64 cur[4], e[1] = {lo};
        fe_pow(f, cur, c->g, e, 1);
        fe_mul(f, cur, cur, c->first);
        for (u64 i = lo; i < hi; i++) {
            fe_mul(f, c->a + 4 * i, c->a + 4 * i, cur);
            fe_mul(f, cur, cur, c->g);
        }
    } else {
        /* slot i holds natural index rev(i).  i -> i+1 flips t trailing ones to zero and sets bit t, i.e. in rev() it clears
         * the top t bits and sets bit logn-1-t: g^rev(i+1) = g^rev(i) * D[t], D[t] = g^(2^(logn-1-t)) * prod_{k<t} g^-(2^(logn-1-k)) */
        u64 pw[64][4], ipw[64][4], D[64][4], cur[4];
        fe_copy(f, pw[0], c->g);
        fe_copy(f, ipw[0], c->ginv);
        for (int k = 1; k < c->logn; k++) {
            fe_mul(f, pw[k], pw[k - 1], pw[k - 1]);
            fe_mul(f, ipw[k], ipw[k - 1], ipw[k - 1]);
        }
        for (int t = 0; t < c->logn; t++) {
            fe_copy(f, D[t], pw[c->logn - 1 - t]);
            for (int k = 0; k < t; k++) fe_mul(f, D[t], D[t], ipw[c->logn - 1 - k]);
        }
        u64 j = bitrev_u64(lo, c->logn);
        fe_copy(f, cur, c->first);
        for (int k = 0; k < c->logn; k++)
            if ((j >> k) & 1) fe_mul(f, cur, cur, pw[k]);
        for (u64 i = lo; i < hi; i++) {
            fe_mul(f, c->a + 4 * i, c->a + 4 * i, cur);
            int t = 0;
            while (t < c->logn && ((i >> t) & 1)) t++;
            if (t < c->logn) fe_mul(f, cur, cur, D[t]);
        }
    }
}
typedef struct {
    const field_t* f;
    u64* tw;
    const u64* w;
} fft_tw_ctx;
static void fft_tw_body(void* p, u64 lo, u64 hi, int tid) {
    (void)tid;
    fft_tw_ctx* c = (fft_tw_ctx*)p;
    u64 cur[4], e[1] = {lo};
    fe_pow(c->f, cur, c->w, e, 1);
    for (u64 i = lo; i < hi; i++) {
        fe_copy(c->f, c->tw + 4 * i, cur);
        fe_mul(c->f, cur, cur, c->w);
    }
}

/* a: n Montgomery elements, in place.  direction 0 fwd / 1 inv; decimation 0 DIF / 1 DIT; on_coset.
 * In-place radix-2 with the conventions of gnark-crypto's fft.Domain: DIF natural -> bit-reversed,
 * DIT bit-reversed -> natural, FFTInverse includes 1/n, OnCoset pre-multiplies coefficient j by g^j (forward) and
 * post-multiplies by g^-j (inverse).  Every stage's butterflies are independent: nthreads > 1 spreads them. */
int oracle_fft_mt(int curve, u64* a, u64 n, int direction, int decimation, int on_coset, int nthreads) {
    const curve_t* cv = &CURVES[curve];
    const field_t* f = &cv->fr;
    const int logn = ilog2(n);
    if (((u64)1 << logn) != n || logn > cv->adicity) return -1;
    if (n == 1) return 0; /* the size-1 transform is the identity in every mode (g^0 = 1, 1/n = 1) */
    u64 w[4];
    root_of_unity(cv, n, direction, w);
    u64* tw = (u64*)malloc(32 * (n / 2 + 1));
    fft_tw_ctx tc = {f, tw, w};
    par_for(nthreads, n / 2, fft_tw_body, &tc);
    if (direction == 0 && on_coset) { /* pre-scale coefficient j by g^j; DIT input sits at slot bitrev(j) */
        fft_scale_ctx sc = {f, a, cv->fr_gen, cv->fr_gen_inv, f->one, logn, decimation == 1};
        par_for(nthreads, n, fft_scale_body, &sc);
    }
    if (decimation == 0) { /* DIF: natural -> bit-reversed */
        for (int s = logn - 1; s >= 0; s--) {
            fft_stage_ctx st = {f, a, tw, logn, s, 0};
            par_for(nthreads, n / 2, fft_stage_body, &st);
        }
    } else { /* DIT: bit-reversed -> natural */
        for (int s = 0; s < logn; s++) {
            fft_stage_ctx st = {f, a, tw, logn, s, 1};
            par_for(nthreads, n / 2, fft_stage_body, &st);
        }
    }
    if (direction == 1) {
        u64 two[4], ninv[4];
        fe_add(f, two, f->one, f->one);
        fe_inv(f, two, two);
        fe_copy(f, ninv, f->one);
        for (int k = 0; k < logn; k++) fe_mul(f, ninv, ninv, two);
        /* 1/n, and on a coset g^-j with j the natural index of the slot (DIF output is bit-reversed) */
        fft_scale_ctx sc = {f, a, on_coset ? cv->fr_gen_inv : f->one, on_coset ? cv->fr_gen : f->one, ninv, logn, on_coset && decimation == 0};
        par_for(nthreads, n, fft_scale_body, &sc);
    }
    free(tw);
    return 0;
}
int oracle_fft(int curve, u64* a, u64 n, int direction, int decimation, int on_coset) {
    return oracle_fft_mt(curve, a, n, direction, decimation, on_coset, 1);
}

/* ---- O(n) evaluation checkers for the BASELINE-size cases (SURVEY 8c "oracle self-checks") ---------------------------------
 * P(x) from the values of P on the size-n domain, barycentric form:  P(x) = (x^n - 1)/n * sum_i v_i * w^i / (x - w^i),
 * x outside the domain.  k vectors share the weights.  All values Montgomery. */
typedef struct {
    const field_t* f;
    const u64* const* vecs;
    int k;
    const u64 *w, *x;
    u64* partial; /* [64][k][4] */
} bary_ctx;
static void bary_body(void* p, u64 lo, u64 hi, int tid) {
    bary_ctx* c = (bary_ctx*)p;
    const field_t* f = c->f;
    enum { BLK = 1024 };
    u64 (*wp)[4] = malloc(sizeof(u64[4]) * BLK), (*den)[4] = malloc(sizeof(u64[4]) * BLK), (*pre)[4] = malloc(sizeof(u64[4]) * BLK);
    u64 cur[4], e[1] = {lo};
    fe_pow(f, cur, c->w, e, 1);
    u64* acc = c->partial + (size_t)tid * c->k * 4;
    memset(acc, 0, 32 * c->k);
    for (u64 b0 = lo; b0 < hi; b0 += BLK) {
        const u64 cnt = hi - b0 < BLK ? hi - b0 : BLK;
        u64 run[4], inv[4];
        fe_copy(f, run, f->one);
        for (u64 i = 0; i < cnt; i++) { /* w^i, x - w^i, prefix products */
            fe_copy(f, wp[i], cur);
            fe_sub(f, den[i], c->x, cur);
            fe_copy(f, pre[i], run);
            fe_mul(f, run, run, den[i]);
            fe_mul(f, cur, cur, c->w);
        }
        fe_inv(f, inv, run);
        for (u64 i = cnt; i-- > 0;) { /* Montgomery's trick backwards */
            u64 di[4], t[4];
            fe_mul(f, di, inv, pre[i]);      /* 1/(x - w^i) */
            fe_mul(f, inv, inv, den[i]);
            fe_mul(f, di, di, wp[i]);        /* w^i/(x - w^i) */
            for (int v = 0; v < c->k; v++) {
                fe_mul(f, t, di, c->vecs[v] + 4 * (b0 + i));
                fe_add(f, acc + 4 * v, acc + 4 * v, t);
            }
        }
    }
    free(wp); free(den); free(pre);
}
int oracle_fr_eval_lagrange(int curve, const u64* const* vecs, int k, u64 n, const u64* x, u64* out, int nthreads) {
    const curve_t* cv = &CURVES[curve];
    const field_t* f = &cv->fr;
    const int logn = ilog2(n);
    if (((u64)1 << logn) != n || logn > cv->adicity || k < 1 || k > 16) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    u64 w[4];
    root_of_unity(cv, n, 0, w);
    u64* partial = (u64*)calloc((size_t)64 * k * 4, 8);
    bary_ctx c = {f, vecs, k, w, x, partial};
    par_for(nthreads, n, bary_body, &c);
    /* (x^n - 1)/n */
    u64 xn[4], ninv[4], two[4], e[1] = {n};
    fe_pow(f, xn, x, e, 1);
    fe_sub(f, xn, xn, f->one);
    fe_add(f, two, f->one, f->one);
    fe_inv(f, two, two);
    fe_copy(f, ninv, f->one);
    for (int b = 0; b < logn; b++) fe_mul(f, ninv, ninv, two);
    fe_mul(f, xn, xn, ninv);
    for (int v = 0; v < k; v++) {
        u64 acc[4] = {0};
        for (int t = 0; t < 64; t++) fe_add(f, acc, acc, partial + ((size_t)t * k + v) * 4);
        fe_mul(f, out + 4 * v, acc, xn);
    }
    free(partial);
    return 0;
}
/* P(x) for coefficients stored in BIT-REVERSED order (what computeH returns and what pk.G1.Z is indexed by): slot j holds the
 * coefficient of x^rev(j) and rev(j' + (n/2) b) = b + 2 rev'(j'), so  P(x) = sum_j' (c[j'] + x c[j' + n/2]) (x^2)^rev'(j'):
 * fold the upper half into the lower one and square x, log n times; n - 1 multiplications in all. */
int oracle_fr_eval_bitrev(int curve, const u64* coeffs, u64 n, const u64* x, u64* out) {
    const field_t* f = &CURVES[curve].fr;
    if (n == 0 || (n & (n - 1))) return -1;
    u64* v = (u64*)malloc(32 * n);
    memcpy(v, coeffs, 32 * n);
    u64 xs[4], t[4];
    fe_copy(f, xs, x);
    for (u64 h = n / 2; h >= 1; h >>= 1) {
        for (u64 j = 0; j < h; j++) {
            fe_mul(f, t, v + 4 * (j + h), xs);
            fe_add(f, v + 4 * j, v + 4 * j, t);
        }
        fe_mul(f, xs, xs, xs);
    }
    memcpy(out, v, 32);
    free(v);
    return 0;
}

/* ---------------- PLONK quotient: computeNumerator + divideByZH (backend/plonk/bn254/prove.go:841-1123,1287-1350) ----------
 * polys: np = 12 + 2*nb_bsb canonical coefficient vectors of n elements each, order L R O Z Ql Qr Qm Qo Qk S1 S2 S3
 * (Qcp_i, Pi2_i)...; bl/br/bo: 2 coefficients, bz: 3; everything Montgomery.  h_out: rho*n canonical coefficients.
 * Structure follows the reference: for each coset g*w1^i scale the coefficients, FFT on the small domain (:1033-1058), apply
 * allConstraints pointwise (:950-981), place at the bit-reversed slot (:1073); then divideByZH's factor and the inverse coset
 * transform on the big domain (:1311-1319). */
int oracle_plonk_quotient(int curve, u64 n, int nb_bsb, const u64* const* polys, const u64* bl, const u64* br, const u64* bo,
                          const u64* bz, const u64* alpha, const u64* beta, const u64* gamma, u64* h_out) {
    const curve_t* cv = &CURVES[curve];
    const field_t* f = &cv->fr;
    const u64 rho = n < 6 ? 8 : 4, N = rho * n;
    const int logn = ilog2(n), logN = ilog2(N), np = 12 + 2 * nb_bsb;
    if (((u64)1 << logn) != n || n < 2) return -1;
    u64 w0[4], w1[4], cs[4], css[4], ninv[4], two[4], one[4];
    fe_copy(f, one, f->one);
    root_of_unity(cv, n, 0, w0);
    root_of_unity(cv, N, 0, w1);
    fe_copy(f, cs, cv->fr_gen);
    fe_mul(f, css, cs, cs);
    fe_add(f, two, one, one);
    fe_inv(f, two, two);
    fe_copy(f, ninv, one);
    for (int k = 0; k < logn; k++) fe_mul(f, ninv, ninv, two);
    u64* tw = (u64*)malloc(32 * n);         /* twiddles0: w0^j */
    fe_copy(f, tw, one);
    for (u64 j = 1; j < n; j++) fe_mul(f, tw + 4 * j, tw + 4 * (j - 1), w0);
    u64* ev = (u64*)malloc(32 * n * (size_t)np);
    u64* den = (u64*)malloc(32 * n);
    u64* pre = (u64*)malloc(32 * n);
    u64* cres = (u64*)malloc(32 * N);
    u64 coset[4];
    fe_copy(f, coset, one);
    for (u64 i = 0; i < rho; i++) {
        fe_mul(f, coset, coset, i == 0 ? cv->fr_gen : w1);           /* shifters, :936-941,998 */
        u64 cexp[4], e[1] = {n};
        fe_pow(f, cexp, coset, e, 1);
        fe_sub(f, cexp, cexp, one);                                  /* coset^n - 1, :999-1000 */
        /* evaluations of every polynomial on coset*H, natural order: scale by coset^k, DIF, undo the bit reversal */
        for (int p = 0; p < np; p++) {
            u64* dst = ev + 4 * n * (size_t)p;
            u64 acc[4];
            fe_copy(f, acc, one);
            for (u64 k = 0; k < n; k++) {
                fe_mul(f, pre + 4 * k, polys[p] + 4 * k, acc);
                fe_mul(f, acc, acc, coset);
            }
            if (oracle_fft(curve, pre, n, 0, 0, 0) != 0) return -1;
            for (u64 k = 0; k < n; k++) fe_copy(f, dst + 4 * bitrev_u64(k, logn), pre + 4 * k);
        }
        /* 1/(x_j - 1) by Montgomery's trick (batchInvert, :1134-1147) */
        for (u64 j = 0; j < n; j++) {
            fe_mul(f, den + 4 * j, coset, tw + 4 * j);
            fe_sub(f, den + 4 * j, den + 4 * j, one);
        }
        fe_copy(f, pre, den);
        for (u64 j = 1; j < n; j++) fe_mul(f, pre + 4 * j, pre + 4 * (j - 1), den + 4 * j);
        u64 inv[4];
        fe_inv(f, inv, pre + 4 * (n - 1));
        for (u64 j = n - 1; j > 0; j--) {
            u64 t[4];
            fe_mul(f, t, inv, pre + 4 * (j - 1));
            fe_mul(f, inv, inv, den + 4 * j);
            fe_copy(f, den + 4 * j, t);
        }
        fe_copy(f, den, inv);
        u64 zhinv[4];
        fe_inv(f, zhinv, cexp);                                      /* evaluateXnMinusOneDomainBigCoset, :1327-1350 */
        for (u64 j = 0; j < n; j++) {
#define EV(p, idx) (ev + 4 * n * (size_t)(p) + 4 * (idx))
            u64 x[4], xw[4], t[4], u[4], l[4], r[4], o[4], z[4], zs[4];
            fe_mul(f, x, coset, tw + 4 * j);
            fe_mul(f, xw, x, w0);
            /* blinded wires: p + b(x) * (coset^n - 1)  (:957-973; the reference pre-scales the blinding coefficients) */
            fe_mul(f, t, bl + 4, x); fe_add(f, t, t, bl); fe_mul(f, t, t, cexp); fe_add(f, l, EV(0, j), t);
            fe_mul(f, t, br + 4, x); fe_add(f, t, t, br); fe_mul(f, t, t, cexp); fe_add(f, r, EV(1, j), t);
            fe_mul(f, t, bo + 4, x); fe_add(f, t, t, bo); fe_mul(f, t, t, cexp); fe_add(f, o, EV(2, j), t);
            fe_mul(f, t, bz + 8, x); fe_add(f, t, t, bz + 4); fe_mul(f, t, t, x); fe_add(f, t, t, bz); fe_mul(f, t, t, cexp);
            fe_add(f, z, EV(3, j), t);
            fe_mul(f, t, bz + 8, xw); fe_add(f, t, t, bz + 4); fe_mul(f, t, t, xw); fe_add(f, t, t, bz); fe_mul(f, t, t, cexp);
            fe_add(f, zs, EV(3, (j + 1) % n), t);
            /* gate (:868-885) */
            u64 gate[4];
            fe_mul(f, gate, EV(4, j), l);
            fe_mul(f, t, EV(5, j), r); fe_add(f, gate, gate, t);
            fe_mul(f, t, EV(6, j), l); fe_mul(f, t, t, r); fe_add(f, gate, gate, t);
            fe_mul(f, t, EV(7, j), o); fe_add(f, gate, gate, t);
            fe_add(f, gate, gate, EV(8, j));
            for (int k = 0; k < nb_bsb; k++) {
                fe_mul(f, t, EV(12 + 2 * k, j), EV(13 + 2 * k, j));
                fe_add(f, gate, gate, t);
            }
            /* ordering (:898-923) */
            u64 id[4], a[4], b[4], c[4], rr[4], ll[4];
            fe_mul(f, id, x, beta);
            fe_add(f, a, gamma, l); fe_add(f, a, a, id);
            fe_mul(f, b, id, cs); fe_add(f, b, b, r); fe_add(f, b, b, gamma);
            fe_mul(f, c, id, css); fe_add(f, c, c, o); fe_add(f, c, c, gamma);
            fe_mul(f, rr, a, b); fe_mul(f, rr, rr, c); fe_mul(f, rr, rr, z);
            fe_mul(f, a, EV(9, j), beta); fe_add(f, a, a, l); fe_add(f, a, a, gamma);
            fe_mul(f, b, EV(10, j), beta); fe_add(f, b, b, r); fe_add(f, b, b, gamma);
            fe_mul(f, c, EV(11, j), beta); fe_add(f, c, c, o); fe_add(f, c, c, gamma);
            fe_mul(f, ll, a, b); fe_mul(f, ll, ll, c); fe_mul(f, ll, ll, zs);
            fe_sub(f, ll, ll, rr);
            /* local (:926-934, :380-385) */
            fe_mul(f, t, cexp, ninv); fe_mul(f, t, t, den + 4 * j);
            fe_sub(f, u, z, one); fe_mul(f, u, u, t);
            /* ((local*alpha) + ordering)*alpha + gate, times 1/(x^n - 1) on this coset */
            fe_mul(f, u, u, alpha); fe_add(f, u, u, ll); fe_mul(f, u, u, alpha); fe_add(f, u, u, gate);
            fe_mul(f, u, u, zhinv);
            fe_copy(f, cres + 4 * bitrev_u64(rho * j + i, logN), u);   /* :1073 */
#undef EV
        }
    }
    /* a.ToCanonical(bigDomain).ToRegular() from LagrangeCoset / BitReverse (:1319): inverse DIT on the coset */
    int rc = oracle_fft(curve, cres, N, 1, 1, 1);
    memcpy(h_out, cres, 32 * N);
    free(tw); free(ev); free(den); free(pre); free(cres);
    return rc;
}

/* computeH, prove.go:346-389.  a,b,c: m elements; h_out: n elements (bit-reversed coefficient order). */
int oracle_compute_h_mt(int curve, const u64* a, const u64* b, const u64* c, u64 m, u64 n, u64* h_out, int nthreads) {
    const curve_t* cv = &CURVES[curve];
    const field_t* f = &cv->fr;
    u64* v[3];
    const u64* src[3] = {a, b, c};
    for (int k = 0; k < 3; k++) {
        v[k] = (u64*)calloc(n, 32);
        memcpy(v[k], src[k], 32 * m);
        oracle_fft_mt(curve, v[k], n, 1, 0, 0, nthreads); /* FFTInverse(DIF) */
    }
    for (int k = 0; k < 3; k++) oracle_fft_mt(curve, v[k], n, 0, 1, 1, nthreads); /* FFT(DIT, OnCoset) */
    u64 den[4], e[1] = {n};
    fe_pow(f, den, cv->fr_gen, e, 1);
    fe_sub(f, den, den, f->one);
    fe_inv(f, den, den);
    for (u64 i = 0; i < n; i++) {
        u64 t[4];
        fe_mul(f, t, v[0] + 4 * i, v[1] + 4 * i);
        fe_sub(f, t, t, v[2] + 4 * i);
        fe_mul(f, v[0] + 4 * i, t, den);
    }
    oracle_fft_mt(curve, v[0], n, 1, 0, 1, nthreads); /* FFTInverse(DIF, OnCoset) */
    memcpy(h_out, v[0], 32 * n);
    for (int k = 0; k < 3; k++) free(v[k]);
    return 0;
}

int oracle_compute_h(int curve, const u64* a, const u64* b, const u64* c, u64 m, u64 n, u64* h_out) {
    return oracle_compute_h_mt(curve, a, b, c, m, n, h_out, 1);
}

/* ---------------- Groth16 Prove (no commitments), prove.go:130-315 ------------------------------------------- */
typedef struct {
    int curve;
    u64 n; /* domain cardinality */
    const u64 *alpha1, *beta1, *delta1;
    const u64* A;
    u64 len_a;
    const u64* B;
    u64 len_b;
    const u64* Z;
    u64 len_z;
    const u64* K;
    u64 len_k;
    const u64 *beta2, *delta2;
    const u64* B2;
    u64 len_b2;
    const uint8_t *inf_a, *inf_b;
    u64 nb_wires;
} oracle_pk;

int oracle_groth16_prove(const oracle_pk* pk, const u64* w, const u64* a, const u64* b, const u64* c, u64 m, u64 nb_public,
                         const u64* r_mont, const u64* s_mont, u64* proof_out, int nthreads) {
    const int cu = pk->curve;
    const curve_t* cv = &CURVES[cu];
    const field_t* fr = &cv->fr;
    tower_t t1 = tower_of(cu, 0), t2 = tower_of(cu, 1);
    u64* h = (u64*)malloc(32 * pk->n);
    oracle_compute_h_mt(cu, a, b, c, m, pk->n, h, nthreads);
    u64* wa = (u64*)malloc(32 * (pk->len_a + 1));
    u64* wb = (u64*)malloc(32 * (pk->len_b + 1));
    u64 ja = 0, jb = 0;
    for (u64 i = 0; i < pk->nb_wires; i++) { /* prove.go:147-168 */
        if (!pk->inf_a[i]) memcpy(wa + 4 * ja++, w + 4 * i, 32);
        if (!pk->inf_b[i]) memcpy(wb + 4 * jb++, w + 4 * i, 32);
    }
    u64 kr[4], rc[4], sc[4], krc[4];
    fe_mul(fr, kr, r_mont, s_mont);
    fe_neg(fr, kr, kr);
    fe_from_mont(fr, rc, r_mont);
    fe_from_mont(fr, sc, s_mont);
    fe_from_mont(fr, krc, kr);
    u64 buf[6 * MAXN];
    jac_t ar, bs1, krs, krs2, bs2, d1, d2, tmp, p;
    aff_t af;
    oracle_msm(cu, 0, pk->A, wa, pk->len_a, 1, buf, nthreads);
    load_jac(&t1, &ar, buf);
    oracle_msm(cu, 0, pk->B, wb, pk->len_b, 1, buf, nthreads);
    load_jac(&t1, &bs1, buf);
    oracle_msm(cu, 1, pk->B2, wb, pk->len_b2, 1, buf, nthreads);
    load_jac(&t2, &bs2, buf);
    oracle_msm(cu, 0, pk->K, w + 4 * nb_public, pk->len_k, 1, buf, nthreads);
    load_jac(&t1, &krs, buf);
    oracle_msm(cu, 0, pk->Z, h, pk->len_z, 1, buf, nthreads);
    load_jac(&t1, &krs2, buf);
    /* deltas = [r]delta, [s]delta, [kr]delta (prove.go:185) */
    load_aff(&t1, &af, pk->delta1);
    jac_from_aff(&t1, &d1, &af);
    jac_t dr, ds, dkr;
    jac_scalar_mul(&t1, &dr, &d1, rc, 4);
    jac_scalar_mul(&t1, &ds, &d1, sc, 4);
    jac_scalar_mul(&t1, &dkr, &d1, krc, 4);
    load_aff(&t1, &af, pk->beta1); /* bs1 += beta + s*delta (199-200) */
    jac_add_mixed(&t1, &bs1, &bs1, &af, 0);
    jac_add(&t1, &bs1, &bs1, &ds);
    load_aff(&t1, &af, pk->alpha1); /* ar += alpha + r*delta (212-213) */
    jac_add_mixed(&t1, &ar, &ar, &af, 0);
    jac_add(&t1, &ar, &ar, &dr);
    jac_add(&t1, &krs, &krs, &dkr); /* 241-266 */
    jac_add(&t1, &krs, &krs, &krs2);
    jac_scalar_mul(&t1, &p, &ar, sc, 4);
    jac_add(&t1, &krs, &krs, &p);
    jac_scalar_mul(&t1, &p, &bs1, rc, 4);
    jac_add(&t1, &krs, &krs, &p);
    load_aff(&t2, &af, pk->delta2); /* Bs += s*delta2 + beta2 (287-290) */
    jac_from_aff(&t2, &d2, &af);
    jac_scalar_mul(&t2, &tmp, &d2, sc, 4);
    jac_add(&t2, &bs2, &bs2, &tmp);
    load_aff(&t2, &af, pk->beta2);
    jac_add_mixed(&t2, &bs2, &bs2, &af, 0);
    const int w1 = 2 * el_words(&t1), w2 = 2 * el_words(&t2);
    jac_to_aff(&t1, &af, &ar);
    store_aff(&t1, proof_out, &af);
    jac_to_aff(&t2, &af, &bs2);
    store_aff(&t2, proof_out + w1, &af);
    jac_to_aff(&t1, &af, &krs);
    store_aff(&t1, proof_out + w1 + w2, &af);
    free(h);
    free(wa);
    free(wb);
    return 0;
}
