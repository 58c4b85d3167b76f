// Explicit instantiation: Fr NTT / computeH, bls12381 (see ntt.hip.h).
#include "ntt.hip.h"
namespace ga {
template <>
int ntt_domain_new<Bls12381>(Ctx* ctx, uint64_t n, Domain** out) {
    Domain* d = new Domain();
    int rc = domain_init<Bls12381::FrP>(ctx, d, Bls12381::ID, n);
    if (rc != GA_OK) {
        domain_free(d);
        delete d;
        return rc;
    }
    *out = d;
    return GA_OK;
}
template <>
int ntt_domain_fft<Bls12381>(Domain* d, void* d_data, int direction, int decimation, int on_coset) {
    return ntt_fft<Bls12381::FrP>(d, (uint32_t*)d_data, direction, decimation, on_coset);
}
template <>
int ntt_domain_h_chain<Bls12381>(Domain* d, void* d_v) {
    return ntt_compute_h_chain<Bls12381::FrP>(d, (uint32_t*)d_v);
}
template <>
int ntt_domain_h_combine<Bls12381>(Domain* d, void* d_a, const void* d_b, const void* d_c) {
    return ntt_compute_h_combine<Bls12381::FrP>(d, (uint32_t*)d_a, (const uint32_t*)d_b, (const uint32_t*)d_c);
}
template <>
int ntt_domain_compute_h<Bls12381>(Domain* d, void* d_a, void* d_b, void* d_c) {
    return ntt_compute_h<Bls12381::FrP>(d, (uint32_t*)d_a, (uint32_t*)d_b, (uint32_t*)d_c);
}
}  // namespace ga
