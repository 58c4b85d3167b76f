// Explicit instantiation: synthetic-input / checking kernels, bn254 (see util.hip.h).
#include "util.hip.h"
namespace ga {
template int util_gen_bases<Bn254, GA_G1>(Ctx*, uint64_t, size_t, void*, void*, uint64_t);
template int util_gen_bases<Bn254, GA_G2>(Ctx*, uint64_t, size_t, void*, void*, uint64_t);
template int util_gen_scalars<Bn254>(Ctx*, uint64_t, size_t, void*);
template int util_fr_dot<Bn254>(Ctx*, const void*, const void*, size_t, void*);
template int util_fr_vec_mul<Bn254>(Ctx*, const void*, const void*, size_t, void*);
template int util_gather_fr<Bn254>(Ctx*, void*, const void*, const uint32_t*, size_t);
template int msm_plan<Bn254>(int, size_t, int*, int*);
template int msm_plan_table<Bn254>(size_t, int*, int*, bool);
}  // namespace ga
