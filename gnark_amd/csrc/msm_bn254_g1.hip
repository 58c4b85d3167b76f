// Explicit instantiation: Pippenger MSM, bn254 G1 (see msm.hip.h).
#include "msm.hip.h"
namespace ga {
template int msm_windows_device<Bn254, GA_G1>(Ctx*, const void*, const void*, size_t, bool, int, int, int, void*, bool);
template int msm_table_device<Bn254, GA_G1>(Ctx*, const void*, const void*, size_t, bool, int, void*, int, int);
template int msm_table_device_reuse<Bn254, GA_G1>(Ctx*, const void*, const MsmPrepared&, void*);
template int msm_table_device_reuse_multi<Bn254, GA_G1>(Ctx*, const void* const*, int, const MsmPrepared&, void*);
template int msm_table_device_batch<Bn254, GA_G1>(Ctx*, const void*, const void* const*, int, size_t, bool, int, void*);
template int msm_table_build<Bn254, GA_G1>(Ctx*, const void*, size_t, int, void*);
template size_t msm_table_point_bytes<Bn254, GA_G1>();
template int msm_prepare_table_scalars<Bn254>(Ctx*, const void*, size_t, bool, int, MsmPrepared*, int, int, int);
}  // namespace ga
