// Explicit instantiation: PLONK quotient / grand product / batch inversion, bls12381 (see plonk.hip.h).
#include "plonk.hip.h"
namespace ga {
template <>
int plonk_domain_quotient<Bls12381>(Domain* d0, Domain* d1, const PlonkQuotientArgs& args, void* h_out) {
    return plonk_quotient<Bls12381::FrP>(d0, d1, args, h_out);
}
template <>
int plonk_domain_fixed_create<Bls12381>(Domain* d0, Domain* d1, const PlonkQuotientArgs& args, PlonkFixed** out) {
    return plonk_fixed_create<Bls12381::FrP>(d0, d1, args, out);
}
template <>
int plonk_domain_quotient_pinned<Bls12381>(PlonkFixed* fx, const PlonkQuotientArgs& args, void* h_out) {
    return plonk_quotient<Bls12381::FrP>(fx->d0, fx->d1, args, h_out, 2, fx);
}
template <>
int plonk_domain_build_z<Bls12381>(Domain* d0, const void* L, const void* R, const void* O, const int64_t* perm, const void* beta,
                                const void* gamma, bool on_device, void* z_out) {
    return plonk_build_z<Bls12381::FrP>(d0, L, R, O, perm, beta, gamma, on_device, z_out);
}
template <>
int fr_vec_batch_inverse<Bls12381>(Ctx* ctx, void* v, uint64_t n, bool on_device) {
    return fr_batch_inverse<Bls12381::FrP>(ctx, v, n, on_device);
}
template <>
int kzg_domain_divide<Bls12381>(Ctx* ctx, const void* d_poly, uint64_t n, const void* z_mont, void* d_quot, void* value_out) {
    return kzg_divide_by_linear<Bls12381::FrP>(ctx, (const uint32_t*)d_poly, n, z_mont, (uint32_t*)d_quot, value_out);
}
template <>
int fr_vec_lincomb<Bls12381>(Ctx* ctx, uint64_t n, int k, const void* const* vecs, const void* scalars, void* out, bool on_device) {
    return fr_lincomb<Bls12381::FrP>(ctx, n, k, vecs, scalars, out, on_device);
}
}  // namespace ga
