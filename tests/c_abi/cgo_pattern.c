/* The exact call pattern of the Go shim (go/backend/accelerated/mi355x/groth16/<curve>/mi355x.go), replayed from plain C so that
 * it can run where there is no Go toolchain:
 *
 *   1. key pinning through ga_g16_builder_*: every call gets ONE flat pointer into a buffer that is poisoned and freed as soon
 *      as the call returns (cgo lets C use a Go pointer only for the duration of the call);
 *   2. key pinning through ga_g16_pk_create with the ga_g16_key struct in C heap and the arrays poisoned right after the call
 *      (what the shim does under runtime.Pinner);
 *   3. ga_g16_prove with W, A, B, C poisoned after it returns, then again from fresh copies: same proof, so nothing was retained;
 *   4. the guarantees of icicle.go:77-86,821-823 (one proof at a time per device, any goroutine may call): two host threads
 *      driving ONE context (proofs interleaved with ga_fft on the same context), and a second context on the same device proving
 *      concurrently -- every result equal to the single-threaded one.
 *
 *   gcc -std=c99 -O2 -pthread -I include tests/c_abi/cgo_pattern.c -L gnark_amd -lgnark_amd -Wl,-rpath,$PWD/gnark_amd -o cgo_pattern
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnark_amd.h"

#ifndef CGO_LOGN
#define CGO_LOGN 10
#endif
#ifndef CGO_THREAD_ROUNDS
#define CGO_THREAD_ROUNDS 3
#endif

#define CHECK(x)                                                              \
    do {                                                                      \
        int rc_ = (x);                                                        \
        if (rc_ != GA_OK) {                                                   \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, ga_last_error());      \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

static void* host_copy(ga_ctx* ctx, const void* dev, size_t bytes) {
    void* h = malloc(bytes ? bytes : 1);
    CHECK(ga_copy_to_host(ctx, h, dev, bytes));
    return h;
}

/* synthetic key material (SURVEY 8d config 3 shape): bases with distinct discrete logs, generated on the device */
typedef struct {
    uint64_t n, nw, nb_public;
    void *A, *B, *Z, *K, *B2, *misc1, *misc2; /* host copies, "the Go slices" */
    uint64_t len_a, len_b, len_z, len_k;
    uint8_t *inf_a, *inf_b;
} host_key;

static void* gen_points(ga_ctx* ctx, int group, uint64_t seed, size_t count) {
    const size_t psz = group == GA_G1 ? 64 : 128;
    void* d = NULL;
    CHECK(ga_malloc(ctx, count * psz, &d));
    CHECK(ga_gen_bases(ctx, GA_BN254, group, seed, count, d, NULL));
    void* h = host_copy(ctx, d, count * psz);
    CHECK(ga_free(ctx, d));
    return h;
}
static void* gen_fr(ga_ctx* ctx, uint64_t seed, size_t count) {
    void* d = NULL;
    CHECK(ga_malloc(ctx, count * 32, &d));
    CHECK(ga_gen_scalars(ctx, GA_BN254, seed, count, d));
    void* h = host_copy(ctx, d, count * 32);
    CHECK(ga_free(ctx, d));
    return h;
}

static void make_key(ga_ctx* ctx, host_key* k) {
    k->n = 1ull << CGO_LOGN;
    k->nw = k->n;
    k->nb_public = 2;
    k->inf_a = calloc(k->nw, 1);
    k->inf_b = calloc(k->nw, 1);
    k->inf_a[1] = k->inf_a[k->nw - 1] = 1;
    k->inf_b[0] = k->inf_b[k->nw - 2] = 1;
    k->len_a = k->nw - 2;
    k->len_b = k->nw - 2;
    k->len_z = k->n - 1;
    k->len_k = k->nw - k->nb_public;
    k->A = gen_points(ctx, GA_G1, 101, k->len_a);
    k->B = gen_points(ctx, GA_G1, 102, k->len_b);
    k->Z = gen_points(ctx, GA_G1, 103, k->len_z);
    k->K = gen_points(ctx, GA_G1, 104, k->len_k);
    k->B2 = gen_points(ctx, GA_G2, 105, k->len_b);
    k->misc1 = gen_points(ctx, GA_G1, 106, 3);
    k->misc2 = gen_points(ctx, GA_G2, 107, 2);
}

/* one call with a pointer that dies right after it */
static void append_transient(ga_g16_builder* b, int which, const void* src, uint64_t count, size_t psz) {
    void* tmp = malloc(count * psz);
    memcpy(tmp, src, count * psz);
    CHECK(ga_g16_builder_append(b, which, tmp, count));
    memset(tmp, 0xA5, count * psz);
    free(tmp);
}

static ga_g16_pk* pin_staged_pre(ga_ctx* ctx, const host_key* k, int32_t precompute);
static ga_g16_pk* pin_staged(ga_ctx* ctx, const host_key* k) { return pin_staged_pre(ctx, k, 0); }

static ga_g16_pk* pin_staged_pre(ga_ctx* ctx, const host_key* k, int32_t precompute) {
    ga_g16_builder* b = NULL;
    CHECK(ga_g16_builder_create(ctx, GA_BN254, k->n, k->nw, 0, 1, &b));
    const void* vec[5] = {k->A, k->B, k->Z, k->K, k->B2};
    const uint64_t len[5] = {k->len_a, k->len_b, k->len_z, k->len_k, k->len_b};
    for (int w = 0; w < 5; w++) {
        const size_t psz = w == GA_KEY_G2_B ? 128 : 64;
        const uint64_t chunk = 300; /* ragged on purpose */
        CHECK(ga_g16_builder_reserve(b, w, len[w]));
        for (uint64_t lo = 0; lo < len[w]; lo += chunk)
            append_transient(b, w, (const char*)vec[w] + lo * psz, len[w] - lo < chunk ? len[w] - lo : chunk, psz);
    }
    CHECK(ga_g16_builder_set_point(b, GA_KEY_G1_ALPHA, k->misc1));
    CHECK(ga_g16_builder_set_point(b, GA_KEY_G1_BETA, (const char*)k->misc1 + 64));
    CHECK(ga_g16_builder_set_point(b, GA_KEY_G1_DELTA, (const char*)k->misc1 + 128));
    CHECK(ga_g16_builder_set_point(b, GA_KEY_G2_BETA, k->misc2));
    CHECK(ga_g16_builder_set_point(b, GA_KEY_G2_DELTA, (const char*)k->misc2 + 128));
    CHECK(ga_g16_builder_set_infinity(b, 0, k->inf_a, k->nw));
    CHECK(ga_g16_builder_set_infinity(b, 1, k->inf_b, k->nw));
    ga_g16_pk* pk = NULL;
    CHECK(ga_g16_builder_finish(b, precompute, &pk));
    return pk;
}

static ga_g16_pk* pin_struct(ga_ctx* ctx, const host_key* k) {
    /* everything the struct points to is a private copy that is poisoned when ga_g16_pk_create returns */
    ga_g16_key* key = calloc(1, sizeof *key); /* the struct itself lives in C heap, as under cgo */
    const size_t bytes[5] = {k->len_a * 64, k->len_b * 64, k->len_z * 64, k->len_k * 64, k->len_b * 128};
    const void* src[5] = {k->A, k->B, k->Z, k->K, k->B2};
    void* cp[5];
    for (int w = 0; w < 5; w++) {
        cp[w] = malloc(bytes[w]);
        memcpy(cp[w], src[w], bytes[w]);
    }
    uint8_t* ia = malloc(k->nw);
    uint8_t* ib = malloc(k->nw);
    memcpy(ia, k->inf_a, k->nw);
    memcpy(ib, k->inf_b, k->nw);
    key->curve = GA_BN254;
    key->domain_cardinality = k->n;
    key->g1_alpha = k->misc1;
    key->g1_beta = (const char*)k->misc1 + 64;
    key->g1_delta = (const char*)k->misc1 + 128;
    key->g1_a = cp[0]; key->len_a = k->len_a;
    key->g1_b = cp[1]; key->len_b = k->len_b;
    key->g1_z = cp[2]; key->len_z = k->len_z;
    key->g1_k = cp[3]; key->len_k = k->len_k;
    key->g2_beta = k->misc2;
    key->g2_delta = (const char*)k->misc2 + 128;
    key->g2_b = cp[4]; key->len_b2 = k->len_b;
    key->infinity_a = ia;
    key->infinity_b = ib;
    key->nb_wires = k->nw;
    key->nb_infinity_a = 2;
    key->nb_infinity_b = 2;
    ga_g16_pk* pk = NULL;
    CHECK(ga_g16_pk_create(ctx, key, &pk));
    for (int w = 0; w < 5; w++) {
        memset(cp[w], 0xA5, bytes[w]);
        free(cp[w]);
    }
    memset(ia, 0xA5, k->nw);
    memset(ib, 0xA5, k->nw);
    free(ia);
    free(ib);
    memset(key, 0xA5, sizeof *key);
    free(key);
    return pk;
}

typedef struct {
    uint64_t n, nw, nb_public;
    void *W, *A, *B, *C, *rs;
} solution;

/* prove from transient copies of the solution, poisoned after the call */
static void prove_transient(ga_g16_pk* pk, const solution* s, uint64_t* proof /* 32 u64: G1 | G2 | G1 */) {
    const size_t bw = s->nw * 32, bc = s->n * 32;
    void *w = malloc(bw), *a = malloc(bc), *b = malloc(bc), *c = malloc(bc);
    memcpy(w, s->W, bw);
    memcpy(a, s->A, bc);
    memcpy(b, s->B, bc);
    memcpy(c, s->C, bc);
    CHECK(ga_g16_prove(pk, w, a, b, c, s->n, s->nb_public, s->rs, (const char*)s->rs + 32, proof));
    memset(w, 0xA5, bw); memset(a, 0xA5, bc); memset(b, 0xA5, bc); memset(c, 0xA5, bc);
    free(w); free(a); free(b); free(c);
}

typedef struct {
    ga_g16_pk* pk;
    const solution* sol;
    const uint64_t* want;
    ga_domain* dom;        /* optional: interleave transforms on the same context */
    const void* fft_in;
    const void* fft_want;
    uint64_t n;
    int failures;
} worker;

static void* prove_worker(void* p) {
    worker* w = (worker*)p;
    for (int r = 0; r < CGO_THREAD_ROUNDS; r++) {
        uint64_t proof[32];
        prove_transient(w->pk, w->sol, proof);
        if (memcmp(proof, w->want, sizeof proof)) w->failures++;
    }
    return NULL;
}
/* like prove_worker, but a refused call (the key is being destroyed) is not a failure; a WRONG proof is */
static void* prove_worker_tolerant(void* p) {
    worker* w = (worker*)p;
    uint64_t proof[32];
    const solution* s = w->sol;   /* ONE call: after the destroy has returned the handle is gone and must not be used again */
    int rc = ga_g16_prove(w->pk, s->W, s->A, s->B, s->C, s->n, s->nb_public, s->rs, (const char*)s->rs + 32, proof);
    if (rc != GA_ERR_STATE && (rc != GA_OK || memcmp(proof, w->want, sizeof proof))) w->failures++;
    return NULL;
}
static void* fft_worker(void* p) {
    worker* w = (worker*)p;
    void* buf = malloc(w->n * 32);
    for (int r = 0; r < 2 * CGO_THREAD_ROUNDS; r++) {
        memcpy(buf, w->fft_in, w->n * 32);
        CHECK(ga_fft(w->dom, buf, GA_FFT_FORWARD, GA_DIF, 1, 0));
        if (memcmp(buf, w->fft_want, w->n * 32)) w->failures++;
    }
    free(buf);
    return NULL;
}

int main(void) {
    ga_ctx *ctx = NULL, *ctx2 = NULL;
    CHECK(ga_ctx_create(0, &ctx));
    CHECK(ga_ctx_create(0, &ctx2)); /* a second context on the SAME device */
    host_key k;
    make_key(ctx, &k);
    solution s;
    s.n = k.n;
    s.nw = k.nw;
    s.nb_public = k.nb_public;
    s.W = gen_fr(ctx, 201, s.nw);
    s.A = gen_fr(ctx, 202, s.n);
    s.B = gen_fr(ctx, 203, s.n);
    s.C = malloc(s.n * 32);
    CHECK(ga_fr_vec_mul(ctx, GA_BN254, s.A, s.B, s.n, s.C, 0));
    s.rs = gen_fr(ctx, 204, 2);

    ga_g16_pk* pk_staged = pin_staged(ctx, &k);
    ga_g16_pk* pk_struct = pin_struct(ctx, &k);
    ga_g16_pk* pk_ctx2 = pin_staged(ctx2, &k);
    uint64_t p1[32], p2[32], p3[32], p4[32];
    prove_transient(pk_staged, &s, p1);
    prove_transient(pk_struct, &s, p2);
    prove_transient(pk_staged, &s, p3); /* again: scratch reuse, nothing retained from the poisoned buffers */
    prove_transient(pk_ctx2, &s, p4);
    if (memcmp(p1, p2, sizeof p1) || memcmp(p1, p3, sizeof p1) || memcmp(p1, p4, sizeof p1)) {
        fprintf(stderr, "staged / struct / repeated / second-context proofs differ\n");
        return 1;
    }
    uint8_t bytes[256];
    size_t blen = 0;
    CHECK(ga_g16_proof_marshal(GA_BN254, p1, bytes, sizeof bytes, &blen));
    if (blen != 164) {
        fprintf(stderr, "proof is %zu bytes, expected 164\n", blen);
        return 1;
    }

    /* ---- the shim's DEFAULT path: a key that is not kept on the device (PinToGPU false, opts.go) is uploaded as plain vectors
     * (precompute -1: no window tables are built just to be freed after the proof), proves to the same bytes and gives its memory
     * back.  On a GPU the free-memory readings bound its footprint: the plain key and nothing like the 12x of the tables. */
    {
        uint64_t total = 0, free0 = 0, free1 = 0, free2 = 0;
        char name[128];
        CHECK(ga_device_info(ctx, name, sizeof name, &total, &free0));
        ga_g16_pk* pk_once = pin_staged_pre(ctx, &k, -1);
        CHECK(ga_device_info(ctx, name, sizeof name, &total, &free1));
        uint64_t p5[32];
        prove_transient(pk_once, &s, p5);
        ga_g16_pk_destroy(pk_once);
        CHECK(ga_device_info(ctx, name, sizeof name, &total, &free2));
        const uint64_t plain = (k.len_a + k.len_b + k.len_z + k.len_k) * 64 + k.len_b * 128 + 3 * k.nw * 4;
        if (memcmp(p5, p1, sizeof p1)) {
            fprintf(stderr, "the un-pinned (plain vectors) key proves to different bytes\n");
            return 1;
        }
        if (free0 > free1 && free0 - free1 > 3 * plain + (64u << 20)) {
            fprintf(stderr, "un-pinned key took %llu bytes of HBM for %llu bytes of plain vectors: tables were built\n",
                    (unsigned long long)(free0 - free1), (unsigned long long)plain);
            return 1;
        }
        printf("un-pinned key: %llu bytes of plain vectors, HBM delta on pin %lld, after destroy %lld\n", (unsigned long long)plain,
               (long long)(free0 - free1), (long long)(free0 - free2));
    }
    /* ---- ... and the same default as ONE call (round 6, what mi355x.go proveOneShot does): ga_g16_prove_oneshot with the ga_g16_key in C
     * heap, every array it points to and the whole solution private copies that are poisoned the moment the call returns -- the key
     * goes up while the proof runs, so a pointer used after the return would read 0xA5 garbage -- twice (the second call takes the
     * context's spare buffers and NTT domain), same bytes as the pinned proof. */
    for (int round = 0; round < 2; round++) {
        ga_g16_key* key = calloc(1, sizeof *key);
        const size_t bytes[5] = {k.len_a * 64, k.len_b * 64, k.len_z * 64, k.len_k * 64, k.len_b * 128};
        const void* src[5] = {k.A, k.B, k.Z, k.K, k.B2};
        void* cp[5];
        for (int w = 0; w < 5; w++) {
            cp[w] = malloc(bytes[w]);
            memcpy(cp[w], src[w], bytes[w]);
        }
        uint8_t *ia = malloc(k.nw), *ib = malloc(k.nw);
        memcpy(ia, k.inf_a, k.nw);
        memcpy(ib, k.inf_b, k.nw);
        key->curve = GA_BN254;
        key->domain_cardinality = k.n;
        key->g1_alpha = k.misc1;
        key->g1_beta = (const char*)k.misc1 + 64;
        key->g1_delta = (const char*)k.misc1 + 128;
        key->g1_a = cp[0]; key->len_a = k.len_a;
        key->g1_b = cp[1]; key->len_b = k.len_b;
        key->g1_z = cp[2]; key->len_z = k.len_z;
        key->g1_k = cp[3]; key->len_k = k.len_k;
        key->g2_beta = k.misc2;
        key->g2_delta = (const char*)k.misc2 + 128;
        key->g2_b = cp[4]; key->len_b2 = k.len_b;
        key->infinity_a = ia;
        key->infinity_b = ib;
        key->nb_wires = k.nw;
        key->nb_infinity_a = 2;
        key->nb_infinity_b = 2;
        key->precompute = 1;   /* ignored: a one-shot key is plain vectors */
        const size_t bw = s.nw * 32, bc = s.n * 32;
        void *w = malloc(bw), *a = malloc(bc), *b = malloc(bc), *c = malloc(bc);
        memcpy(w, s.W, bw);
        memcpy(a, s.A, bc);
        memcpy(b, s.B, bc);
        memcpy(c, s.C, bc);
        uint64_t p6[32];
        CHECK(ga_g16_prove_oneshot(ctx, key, w, a, b, c, s.n, s.nb_public, s.rs, (const char*)s.rs + 32, p6));
        for (int v = 0; v < 5; v++) {
            memset(cp[v], 0xA5, bytes[v]);
            free(cp[v]);
        }
        memset(ia, 0xA5, k.nw); memset(ib, 0xA5, k.nw);
        free(ia); free(ib);
        memset(w, 0xA5, bw); memset(a, 0xA5, bc); memset(b, 0xA5, bc); memset(c, 0xA5, bc);
        free(w); free(a); free(b); free(c);
        memset(key, 0xA5, sizeof *key);
        free(key);
        if (memcmp(p6, p1, sizeof p1)) {
            fprintf(stderr, "ga_g16_prove_oneshot (round %d) proves to different bytes\n", round);
            return 1;
        }
    }
    printf("ga_g16_prove_oneshot: same proof bytes, twice\n");
    /* ---- FreeGPUResources from one thread while another is still proving on the key (a Go `defer pk.FreeGPUResources()` beside a
     * second goroutine's Prove): ga_g16_pk_destroy waits for the proof in flight; a call that arrives after the key started dying is
     * refused with GA_ERR_STATE instead of touching freed memory ---- */
    {
        ga_g16_pk* pk_tmp = pin_staged(ctx, &k);
        worker wd;
        memset(&wd, 0, sizeof wd);
        wd.pk = pk_tmp; wd.sol = &s; wd.want = p1;
        pthread_t td;
        uint64_t st0[6], st1[6];
        CHECK(ga_g16_lane_stats(ctx, st0));
        pthread_create(&td, NULL, prove_worker_tolerant, &wd);
        do {   /* wait until the call is registered on the key (the counters move after the key's use count has been taken) */
            CHECK(ga_g16_lane_stats(ctx, st1));
        } while (st1[0] + st1[1] + st1[2] == st0[0] + st0[1] + st0[2]);
        ga_g16_pk_destroy(pk_tmp);   /* while that proof is in flight */
        pthread_join(td, NULL);
        if (wd.failures) {
            fprintf(stderr, "a proof that overlapped ga_g16_pk_destroy came back wrong\n");
            return 1;
        }
    }

    /* ---- concurrency ---- */
    ga_domain* dom = NULL;
    CHECK(ga_domain_create(ctx, GA_BN254, s.n, &dom));
    void* fft_want = malloc(s.n * 32);
    memcpy(fft_want, s.A, s.n * 32);
    CHECK(ga_fft(dom, fft_want, GA_FFT_FORWARD, GA_DIF, 1, 0));
    worker w[4];
    memset(w, 0, sizeof w);
    w[0].pk = pk_staged; w[0].sol = &s; w[0].want = p1;                 /* thread 0: proofs on ctx            */
    w[1].pk = pk_struct; w[1].sol = &s; w[1].want = p1;                 /* thread 1: proofs on the SAME ctx   */
    w[2].dom = dom; w[2].fft_in = s.A; w[2].fft_want = fft_want; w[2].n = s.n;   /* thread 2: transforms on the same ctx */
    w[3].pk = pk_ctx2; w[3].sol = &s; w[3].want = p1;                   /* thread 3: proofs on a second ctx   */
    pthread_t th[4];
    pthread_create(&th[0], NULL, prove_worker, &w[0]);
    pthread_create(&th[1], NULL, prove_worker, &w[1]);
    pthread_create(&th[2], NULL, fft_worker, &w[2]);
    pthread_create(&th[3], NULL, prove_worker, &w[3]);
    int bad = 0;
    for (int t = 0; t < 4; t++) {
        pthread_join(th[t], NULL);
        bad += w[t].failures;
    }
    if (bad) {
        fprintf(stderr, "%d results differed under concurrent use\n", bad);
        return 1;
    }
    ga_domain_destroy(dom);
    ga_g16_pk_destroy(pk_staged);
    ga_g16_pk_destroy(pk_struct);
    ga_g16_pk_destroy(pk_ctx2);
    ga_ctx_destroy(ctx2);
    ga_ctx_destroy(ctx);
    printf("CGO_PATTERN_OK proofs identical (staged, struct, repeated, 2 contexts, 4 threads x %d rounds)\n", CGO_THREAD_ROUNDS);
    return 0;
}
