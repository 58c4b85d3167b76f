/* Plain-C client of include/gnark_amd.h -- what a cgo binding compiles against (no C++, no Python, no torch).
 * (also: PLONK grand product / batch inversion / hash-to-field entry points)
 * Builds known-discrete-log bases on the device, runs ga_msm and ga_msm_table_run, checks MSM(s, [k_i]G) == [sum s_i k_i]G
 * with the library's own host helpers, runs an NTT round trip, and prints "ABI_CLIENT_OK".
 *   gcc -O2 -I include tests/c_abi/abi_client.c -L gnark_amd -lgnark_amd -Wl,-rpath,$PWD/gnark_amd -o abi_client */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnark_amd.h"

#define CHECK(x)                                                              \
    do {                                                                      \
        int rc_ = (x);                                                        \
        if (rc_ != GA_OK) {                                                   \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, ga_last_error());      \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main(void) {
    ga_ctx* ctx = NULL;
    CHECK(ga_ctx_create(0, &ctx));
    char name[128];
    uint64_t total = 0, freeb = 0;
    CHECK(ga_device_info(ctx, name, sizeof name, &total, &freeb));
    printf("device: %s, %.0f GiB\n", name, total / 1073741824.0);

#ifndef ABI_CLIENT_N
#define ABI_CLIENT_N (1u << 16)
#endif
    const size_t n = ABI_CLIENT_N;
    void *bases = NULL, *dlogs = NULL, *scalars = NULL;
    CHECK(ga_malloc(ctx, n * 64, &bases));
    CHECK(ga_malloc(ctx, n * 32, &dlogs));
    CHECK(ga_malloc(ctx, n * 32, &scalars));
    CHECK(ga_gen_bases(ctx, GA_BN254, GA_G1, 7, n, bases, dlogs));
    CHECK(ga_gen_scalars(ctx, GA_BN254, 8, n, scalars));

    uint64_t jac[12], jac_t[12], want[12], aff[8], aff_t[8], aff_w[8], dot[4];
    CHECK(ga_msm(ctx, GA_BN254, GA_G1, bases, scalars, n, GA_BASES_ON_DEVICE | GA_SCALARS_ON_DEVICE | GA_SCALARS_MONTGOMERY, jac));
    ga_msm_table* t = NULL;
    CHECK(ga_msm_table_create(ctx, GA_BN254, GA_G1, bases, n, GA_BASES_ON_DEVICE, &t));
    CHECK(ga_msm_table_run(t, scalars, GA_SCALARS_ON_DEVICE | GA_SCALARS_MONTGOMERY, jac_t));
    ga_msm_table_destroy(t);
    CHECK(ga_fr_dot(ctx, GA_BN254, scalars, dlogs, n, dot));
    CHECK(ga_generator_mul(GA_BN254, GA_G1, dot, want));
    CHECK(ga_jac_to_affine(GA_BN254, GA_G1, jac, aff));
    CHECK(ga_jac_to_affine(GA_BN254, GA_G1, jac_t, aff_t));
    CHECK(ga_jac_to_affine(GA_BN254, GA_G1, want, aff_w));
    if (memcmp(aff, aff_w, sizeof aff) || memcmp(aff_t, aff_w, sizeof aff)) {
        fprintf(stderr, "MSM mismatch\n");
        return 1;
    }

    /* NTT round trip on host memory: FFT(DIF, coset) then FFTInverse(DIT, coset) */
    ga_domain* d = NULL;
    CHECK(ga_domain_create(ctx, GA_BN254, n, &d));
    uint64_t* a = malloc(n * 32);
    uint64_t* b = malloc(n * 32);
    CHECK(ga_copy_to_host(ctx, a, scalars, n * 32));
    memcpy(b, a, n * 32);
    CHECK(ga_fft(d, b, GA_FFT_FORWARD, GA_DIF, 1, 0));
    if (!memcmp(a, b, n * 32)) {
        fprintf(stderr, "FFT did nothing\n");
        return 1;
    }
    CHECK(ga_fft(d, b, GA_FFT_INVERSE, GA_DIT, 1, 0));
    if (memcmp(a, b, n * 32)) {
        fprintf(stderr, "NTT round trip mismatch\n");
        return 1;
    }
    /* PLONK grand product with the identity permutation: every ratio is 1, so Z is the constant 1 (Montgomery one = a[..]
     * of ga_generator... no helper needed: Z[0] is 1 by definition and all entries must equal it) */
    {
        int64_t* perm = malloc(3 * n * sizeof(int64_t));
        for (size_t i = 0; i < 3 * n; i++) perm[i] = (int64_t)i;
        uint64_t* z = malloc(n * 32);
        CHECK(ga_plonk_build_z(d, a, a, a, perm, a, a + 4, 0, z));   /* l = r = o = the random vector, beta/gamma from it */
        for (size_t i = 1; i < n; i++)
            if (memcmp(z, z + 4 * i, 32)) {
                fprintf(stderr, "grand product with the identity permutation is not constant\n");
                return 1;
            }
        /* fr.BatchInvert is an involution (and keeps zeros) */
        memcpy(b, a, n * 32);
        memset(b, 0, 32);
        CHECK(ga_fr_batch_invert(ctx, GA_BN254, b, n, 0));
        CHECK(ga_fr_batch_invert(ctx, GA_BN254, b, n, 0));
        memset(a, 0, 32);
        if (memcmp(a, b, n * 32)) {
            fprintf(stderr, "batch inversion is not an involution\n");
            return 1;
        }
        free(perm);
        free(z);
    }
    /* hash-to-field host code: first expand_message_xmd vector of the reference (std/hash/expand/expand_test.go:52-56) */
    {
        static const char dst[] = "QUUX-V01-CS02-with-expander-SHA256-128";
        static const uint8_t want32[4] = {0x68, 0xa9, 0x85, 0xb8};
        uint8_t out[32];
        CHECK(ga_expand_message_xmd((const uint8_t*)"", 0, (const uint8_t*)dst, sizeof dst - 1, 32, out));
        if (memcmp(out, want32, 4)) {
            fprintf(stderr, "expand_message_xmd mismatch\n");
            return 1;
        }
    }
    ga_domain_destroy(d);
    free(a);
    free(b);
    CHECK(ga_free(ctx, bases));
    CHECK(ga_free(ctx, dlogs));
    CHECK(ga_free(ctx, scalars));
    ga_ctx_destroy(ctx);
    printf("ABI_CLIENT_OK %s\n", ga_version());
    return 0;
}
