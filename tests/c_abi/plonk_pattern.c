/* The call sequence of the PLONK prover once go/backend/accelerated/mi355x/plonk/bn254/prove.patch is applied
 * (backend/plonk/bn254/prove.go with the Accelerator hooks of hooks.go), replayed from plain C so that it runs without a Go
 * toolchain.  Order and arguments are the patched prover's:
 *
 *   NewDevice            pin pk.Kzg.G1 and pk.KzgLagrange.G1 as window tables, domains n and 4n        (hooks.go NewDevice)
 *   commitToLRO          ONE batched MSM over the Lagrange SRS for L, R, O                               (prove.go:404-489)
 *   buildRatioCopyConstraint   Z from L, R, O and the permutation; commit Z                              (prove.go:636-668)
 *   PinTrace + computeQuotient   circuit constants pinned once; h = divideByZH(computeNumerator())      (prove.go:558-633)
 *   commitToQuotient     ONE batched MSM over the monomial SRS for h1, h2, h3                            (prove.go:1263-1285)
 *   openZ                kzg.Open(blindedZ, zeta*omega)                                                  (prove.go:670-689)
 *   batchOpening         fold the six polynomials with powers of gamma, kzg.Open(folded, zeta)           (prove.go:796-838)
 *
 * Every input buffer is a transient copy that is POISONED and freed as soon as its call returns (cgo lets C use a Go pointer only
 * for the duration of the call), the two struct-of-pointers arguments live in C heap with their pointer arrays (what the shim does
 * under runtime.Pinner), and the whole proof is replayed a second time from fresh copies: identical outputs, so nothing was
 * retained.  Cross-checks inside one replay: batched commitments == single commitments, pinned quotient == un-pinned quotient,
 * claimed values == ga_fr_poly_evaluate, fold == a second fold in two halves.
 *
 *   gcc -std=gnu99 -O2 -I include tests/c_abi/plonk_pattern.c -L gnark_amd -lgnark_amd -Wl,-rpath,$PWD/gnark_amd -o plonk_pattern
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnark_amd.h"

#ifndef PLONK_LOGN
#define PLONK_LOGN 10
#endif
#ifndef PLONK_CURVE
#define PLONK_CURVE GA_BN254
#endif
#define FP_BYTES (PLONK_CURVE == GA_BN254 ? 32 : 48)
#define G1_AFF (2 * FP_BYTES)
#define G1_JAC (3 * FP_BYTES)

#define CHECK(x)                                                              \
    do {                                                                      \
        int rc_ = (x);                                                        \
        if (rc_ != GA_OK) {                                                   \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, ga_last_error());      \
            exit(1);                                                          \
        }                                                                     \
    } while (0)
#define REQUIRE(cond, what)                          \
    do {                                             \
        if (!(cond)) {                               \
            fprintf(stderr, "FAILED: %s\n", what);   \
            exit(1);                                 \
        }                                            \
    } while (0)

static ga_ctx* ctx;

/* a transient copy of a "Go slice": handed to one call, then poisoned and freed */
static void* transient(const void* src, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    memcpy(p, src, bytes);
    return p;
}
static void poison(void* p, size_t bytes) {
    memset(p, 0xA5, bytes);
    free(p);
}

static void* gen_fr(uint64_t seed, size_t count) {
    void* d = NULL;
    CHECK(ga_malloc(ctx, count * 32, &d));
    CHECK(ga_gen_scalars(ctx, PLONK_CURVE, seed, count, d));
    void* h = malloc(count * 32);
    CHECK(ga_copy_to_host(ctx, h, d, count * 32));
    CHECK(ga_free(ctx, d));
    return h;
}
static ga_msm_table* pin_srs(uint64_t seed, size_t count) {
    void* d = NULL;
    CHECK(ga_malloc(ctx, count * G1_AFF, &d));
    CHECK(ga_gen_bases(ctx, PLONK_CURVE, GA_G1, seed, count, d, NULL));
    void* h = malloc(count * G1_AFF); /* "pk.Kzg.G1", a Go slice */
    CHECK(ga_copy_to_host(ctx, h, d, count * G1_AFF));
    CHECK(ga_free(ctx, d));
    ga_msm_table* t = NULL;
    CHECK(ga_msm_table_create(ctx, PLONK_CURVE, GA_G1, h, count, GA_TABLE_BATCHED, &t)); /* hooks.go: NewTable(..., batched = true) */
    poison(h, count * G1_AFF);
    return t;
}

typedef struct {
    uint64_t n;
    void *l, *r, *o, *ql, *qr, *qm, *qo, *qk, *s1, *s2, *s3, *qcp0, *pi20; /* n fr each: the prover's polynomials */
    int64_t* perm;
    uint8_t bl[64], br[64], bo[64], bz[96], alpha[32], beta[32], gamma[32], zeta[32];
} witness_t;

typedef struct {
    uint8_t lro[3 * 3 * 48], z[3 * 48], h[3 * 3 * 48], zopen_h[3 * 48], zopen_v[32], bopen_h[3 * 48], bopen_v[32];
    void *zpoly, *hpoly;
} proof_t;

/* one commitment over a pinned table; the polynomial is zero-padded to the table's size as hooks.go does */
static void commit1(ga_msm_table* t, size_t tn, const void* p, size_t len, void* out_jac) {
    void* s = calloc(tn, 32);
    memcpy(s, p, len * 32);
    CHECK(ga_msm_table_run(t, s, GA_SCALARS_MONTGOMERY, out_jac));
    poison(s, tn * 32);
}
static void commit_batch(ga_msm_table* t, size_t tn, const void* const* ps, const size_t* lens, int k, void* out_jacs) {
    void** arr = malloc(k * sizeof(void*)); /* the pointer array in C memory */
    for (int i = 0; i < k; i++) {
        arr[i] = calloc(tn, 32);
        memcpy(arr[i], ps[i], lens[i] * 32);
    }
    CHECK(ga_msm_table_run_batch(t, (const void* const*)arr, (uint32_t)k, GA_SCALARS_MONTGOMERY, out_jacs));
    for (int i = 0; i < k; i++) poison(arr[i], tn * 32);
    poison(arr, k * sizeof(void*));
}

static void fill_quotient_in(ga_plonk_quotient_in* in, const witness_t* w, const void* z, const void** qcp_arr, const void** pi2_arr, void** owned, int* nowned) {
    const uint64_t nb = w->n * 32;
    memset(in, 0, sizeof(*in));
    const void* src[12] = {w->l, w->r, w->o, z, w->ql, w->qr, w->qm, w->qo, w->qk, w->s1, w->s2, w->s3};
    const void** dst[12] = {&in->l, &in->r, &in->o, &in->z, &in->ql, &in->qr, &in->qm, &in->qo, &in->qk, &in->s1, &in->s2, &in->s3};
    for (int k = 0; k < 12; k++) {
        void* c = transient(src[k], nb);
        owned[(*nowned)++] = c;
        *dst[k] = c;
    }
    in->nb_bsb = 1;
    qcp_arr[0] = owned[(*nowned)++] = transient(w->qcp0, nb);
    pi2_arr[0] = owned[(*nowned)++] = transient(w->pi20, nb);
    in->qcp = qcp_arr;
    in->pi2 = pi2_arr;
    in->lagrange_mask = 0x1ull | 0x2ull | 0x4ull | 0x8ull | (1ull << 8) | (1ull << 13); /* L R O Z Qk Pi2_0 in Lagrange form, the trace canonical */
    in->bl = owned[(*nowned)++] = transient(w->bl, 64);
    in->br = owned[(*nowned)++] = transient(w->br, 64);
    in->bo = owned[(*nowned)++] = transient(w->bo, 64);
    in->bz = owned[(*nowned)++] = transient(w->bz, 96);
    in->alpha = owned[(*nowned)++] = transient(w->alpha, 32);
    in->beta = owned[(*nowned)++] = transient(w->beta, 32);
    in->gamma = owned[(*nowned)++] = transient(w->gamma, 32);
}

static void prove(const witness_t* w, ga_msm_table* srs, size_t srs_n, ga_msm_table* srs_lag, ga_domain* d0, ga_domain* d1, ga_plonk_pk** trace,
                  proof_t* out) {
    const uint64_t n = w->n, nb = n * 32;
    /* commitToLRO: one pass */
    {
        const void* ps[3] = {w->l, w->r, w->o};
        const size_t lens[3] = {n, n, n};
        commit_batch(srs_lag, n, ps, lens, 3, out->lro);
        uint8_t single[3 * 48];
        for (int i = 0; i < 3; i++) {
            commit1(srs_lag, n, ps[i], n, single);
            uint8_t a[2 * 48], b[2 * 48];
            CHECK(ga_jac_to_affine(PLONK_CURVE, GA_G1, out->lro + i * G1_JAC, a));
            CHECK(ga_jac_to_affine(PLONK_CURVE, GA_G1, single, b));
            REQUIRE(memcmp(a, b, G1_AFF) == 0, "batched commitment != single commitment");
        }
    }
    /* buildRatioCopyConstraint */
    out->zpoly = malloc(nb);
    {
        void *l = transient(w->l, nb), *r = transient(w->r, nb), *o = transient(w->o, nb);
        int64_t* perm = transient(w->perm, 3 * n * sizeof(int64_t));
        void *beta = transient(w->beta, 32), *gamma = transient(w->gamma, 32);
        CHECK(ga_plonk_build_z(d0, l, r, o, perm, beta, gamma, 0, out->zpoly));
        poison(l, nb), poison(r, nb), poison(o, nb), poison(perm, 3 * n * sizeof(int64_t)), poison(beta, 32), poison(gamma, 32);
        commit1(srs_lag, n, out->zpoly, n, out->z);
    }
    /* PinTrace (once per key) and computeQuotient: the struct and its pointer arrays live in C heap */
    out->hpoly = malloc(4 * nb);
    {
        ga_plonk_quotient_in* in = malloc(sizeof(*in));
        const void** qcp_arr = malloc(sizeof(void*));
        const void** pi2_arr = malloc(sizeof(void*));
        void* owned[32];
        int nowned = 0;
        fill_quotient_in(in, w, out->zpoly, qcp_arr, pi2_arr, owned, &nowned);
        if (!*trace) CHECK(ga_plonk_pk_create(d0, d1, in, trace));
        CHECK(ga_plonk_quotient_pinned(*trace, in, out->hpoly));
        void* h2 = malloc(4 * nb);
        CHECK(ga_plonk_quotient(d0, d1, in, h2)); /* the same quotient with nothing pinned */
        REQUIRE(memcmp(h2, out->hpoly, 4 * nb) == 0, "pinned quotient != un-pinned quotient");
        free(h2);
        for (int k = 0; k < nowned; k++) poison(owned[k], 32);
        poison(qcp_arr, sizeof(void*)), poison(pi2_arr, sizeof(void*)), poison(in, sizeof(*in));
    }
    /* commitToQuotient: h1, h2, h3 = slices of n+2 coefficients, one pass over the monomial SRS */
    {
        const char* h = out->hpoly;
        const void* ps[3] = {h, h + (n + 2) * 32, h + 2 * (n + 2) * 32};
        const size_t lens[3] = {n + 2, n + 2, n + 2};
        commit_batch(srs, srs_n, ps, lens, 3, out->h);
    }
    /* openZ: blindedZ has n + 3 coefficients (Z in canonical form + the blinding terms); here Z's evaluations stand in */
    {
        void* bz = calloc(n + 3, 32);
        memcpy(bz, out->zpoly, nb);
        memcpy((char*)bz + nb, w->bz, 96);
        void* pt = transient(w->zeta, 32);
        CHECK(ga_kzg_open(srs, bz, n + 3, GA_SCALARS_MONTGOMERY, pt, out->zopen_v, out->zopen_h));
        uint8_t v[32];
        CHECK(ga_fr_poly_evaluate(ctx, PLONK_CURVE, bz, n + 3, pt, v, 0));
        REQUIRE(memcmp(v, out->zopen_v, 32) == 0, "claimed value of the Z opening != polynomial evaluation");
        poison(bz, (n + 3) * 32), poison(pt, 32);
    }
    /* batchOpening: fold six polynomials with 1, gamma, gamma^2, ... (any six scalars do for the replay) and open the fold */
    {
        const void* polys[6] = {out->hpoly, w->l, w->r, w->o, w->s1, w->s2};
        void** arr = malloc(6 * sizeof(void*));
        for (int i = 0; i < 6; i++) arr[i] = transient(polys[i], nb);
        void* sc = gen_fr(0x600D, 6);
        void* folded = malloc(nb);
        CHECK(ga_fr_linear_combination(ctx, PLONK_CURVE, n, 6, (const void* const*)arr, sc, folded, 0));
        /* the same fold in two halves (first three, last three): their evaluations at a point add up to the fold's */
        void *half = malloc(nb), *rest = malloc(nb);
        CHECK(ga_fr_linear_combination(ctx, PLONK_CURVE, n, 3, (const void* const*)arr, sc, half, 0));
        CHECK(ga_fr_linear_combination(ctx, PLONK_CURVE, n, 3, (const void* const*)(arr + 3), (char*)sc + 3 * 32, rest, 0));
        uint8_t vf[32], vh[32], vr[32];
        void* pt = transient(w->zeta, 32);
        CHECK(ga_fr_poly_evaluate(ctx, PLONK_CURVE, folded, n, pt, vf, 0));
        CHECK(ga_fr_poly_evaluate(ctx, PLONK_CURVE, half, n, pt, vh, 0));
        CHECK(ga_fr_poly_evaluate(ctx, PLONK_CURVE, rest, n, pt, vr, 0));
        {   /* vf == 1*vh + 1*vr; the Montgomery image of 1 is Z[0] of the grand product */
            const void* two[2] = {vh, vr};
            uint8_t ones[64], sum[32];
            memcpy(ones, out->zpoly, 32);
            memcpy(ones + 32, out->zpoly, 32);
            CHECK(ga_fr_linear_combination(ctx, PLONK_CURVE, 1, 2, two, ones, sum, 0));
            REQUIRE(memcmp(sum, vf, 32) == 0, "fold in one pass != fold in two halves");
        }
        CHECK(ga_kzg_open(srs, folded, n, GA_SCALARS_MONTGOMERY, pt, out->bopen_v, out->bopen_h));
        REQUIRE(memcmp(out->bopen_v, vf, 32) == 0, "claimed value of the batch opening != evaluation of the folded polynomial");
        for (int i = 0; i < 6; i++) poison(arr[i], nb);
        poison(arr, 6 * sizeof(void*)), poison(pt, 32);
        free(sc), free(folded), free(half), free(rest);
    }
}

int main(void) {
    CHECK(ga_ctx_create(0, &ctx));
    const uint64_t n = 1ull << PLONK_LOGN;
    witness_t w;
    memset(&w, 0, sizeof(w));
    w.n = n;
    void** polys[13] = {&w.l, &w.r, &w.o, &w.ql, &w.qr, &w.qm, &w.qo, &w.qk, &w.s1, &w.s2, &w.s3, &w.qcp0, &w.pi20};
    for (int k = 0; k < 13; k++) *polys[k] = gen_fr(0x9100 + k, n);
    w.perm = malloc(3 * n * sizeof(int64_t));
    for (uint64_t i = 0; i < 3 * n; i++) w.perm[i] = (int64_t)((i * 7 + 3) % (3 * n)); /* 7 is coprime to 3 * 2^k: a permutation */
    void* ch = gen_fr(0xC4A1, 16);
    memcpy(w.bl, ch, 64), memcpy(w.br, (char*)ch + 64, 64), memcpy(w.bo, (char*)ch + 128, 64), memcpy(w.bz, (char*)ch + 192, 96);
    memcpy(w.alpha, (char*)ch + 288, 32), memcpy(w.beta, (char*)ch + 320, 32), memcpy(w.gamma, (char*)ch + 352, 32), memcpy(w.zeta, (char*)ch + 384, 32);
    free(ch);

    const size_t srs_n = n + 3;
    ga_msm_table *srs = pin_srs(0x5125, srs_n), *srs_lag = pin_srs(0x1A62, n);
    ga_domain *d0 = NULL, *d1 = NULL;
    CHECK(ga_domain_create(ctx, PLONK_CURVE, n, &d0));
    CHECK(ga_domain_create(ctx, PLONK_CURVE, 4 * n, &d1));
    ga_plonk_pk* trace = NULL;

    proof_t p1, p2;
    memset(&p1, 0, sizeof(p1)), memset(&p2, 0, sizeof(p2));
    prove(&w, srs, srs_n, srs_lag, d0, d1, &trace, &p1);
    prove(&w, srs, srs_n, srs_lag, d0, d1, &trace, &p2); /* second proof: the pinned trace is reused, every input is a fresh copy */
    REQUIRE(memcmp(p1.lro, p2.lro, sizeof(p1.lro)) == 0 && memcmp(p1.z, p2.z, sizeof(p1.z)) == 0 && memcmp(p1.h, p2.h, sizeof(p1.h)) == 0 &&
                memcmp(p1.zopen_h, p2.zopen_h, sizeof(p1.zopen_h)) == 0 && memcmp(p1.zopen_v, p2.zopen_v, 32) == 0 &&
                memcmp(p1.bopen_h, p2.bopen_h, sizeof(p1.bopen_h)) == 0 && memcmp(p1.bopen_v, p2.bopen_v, 32) == 0 &&
                memcmp(p1.hpoly, p2.hpoly, 4 * n * 32) == 0 && memcmp(p1.zpoly, p2.zpoly, n * 32) == 0,
            "second replay differs from the first: something was retained across calls");
    ga_plonk_pk_destroy(trace);
    ga_domain_destroy(d1);
    ga_domain_destroy(d0);
    ga_msm_table_destroy(srs_lag);
    ga_msm_table_destroy(srs);
    ga_ctx_destroy(ctx);
    printf("PLONK_PATTERN_OK n=%llu\n", (unsigned long long)n);
    return 0;
}
