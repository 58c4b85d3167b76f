// Explicit instantiation: Pippenger MSM, bls12381 G2 (see msm.hip.h).
#include "msm.hip.h"
namespace ga {
template int msm_windows_device<Bls12381, GA_G2>(Ctx*, const void*, const void*, size_t, bool, int, int, int, void*, bool);
template int msm_table_device<Bls12381, GA_G2>(Ctx*, const void*, const void*, size_t, bool, int, void*, int, int);
template int msm_table_device_reuse<Bls12381, GA_G2>(Ctx*, const void*, const MsmPrepared&, void*);
template int msm_table_device_reuse_multi<Bls12381, GA_G2>(Ctx*, const void* const*, int, const MsmPrepared&, void*);
template int msm_table_device_batch<Bls12381, GA_G2>(Ctx*, const void*, const void* const*, int, size_t, bool, int, void*);
template int msm_table_build<Bls12381, GA_G2>(Ctx*, const void*, size_t, int, void*);
template size_t msm_table_point_bytes<Bls12381, GA_G2>();
}  // namespace ga
