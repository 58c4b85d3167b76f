// Explicit instantiation: Pippenger MSM, bn254 G1 (see msm.cuh).
#include "msm.cuh"
namespace ga {
template int msm_windows_device<Bn254, GA_G1>(Ctx*, const void*, const void*, size_t, bool, int, int, int, void*);
}  // namespace ga
