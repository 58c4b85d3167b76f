/* Plain-C client of include/gnark_amd.h -- what a cgo binding compiles against (no C++, no Python, no torch).
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

    const size_t n = 1u << 16;
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
