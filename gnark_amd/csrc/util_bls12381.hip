// Explicit instantiation: synthetic-input / checking kernels, bls12381 (see util.hip.h).
#include "util.hip.h"
namespace ga {
template int util_gen_bases<Bls12381, GA_G1>(Ctx*, uint64_t, size_t, void*, void*, uint64_t);
template int util_gen_bases<Bls12381, GA_G2>(Ctx*, uint64_t, size_t, void*, void*, uint64_t);
template int util_gen_scalars<Bls12381>(Ctx*, uint64_t, size_t, void*);
template int util_fr_dot<Bls12381>(Ctx*, const void*, const void*, size_t, void*);
template int util_fr_vec_mul<Bls12381>(Ctx*, const void*, const void*, size_t, void*);
template int util_gather_fr<Bls12381>(Ctx*, void*, const void*, const uint32_t*, size_t);
template int msm_plan<Bls12381>(int, size_t, int*, int*);
template int msm_plan_table<Bls12381>(size_t, int*, int*, bool);
}  // namespace ga
