// Explicit instantiation: Pippenger MSM, bls12381 G2 (see msm.cuh).
#include "msm.cuh"
namespace ga {
template int msm_windows_device<Bls12381, GA_G2>(Ctx*, const void*, const void*, size_t, bool, int, int, int, void*);
}  // namespace ga
