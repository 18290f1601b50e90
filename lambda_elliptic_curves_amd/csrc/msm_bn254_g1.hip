// Bn254G1 instantiation of the Pippenger MSM (msm_core.cuh).
#include "msm_core.cuh"

namespace lw {
LW_MSM_INSTANTIATE(Bn254G1, bn254_g1)
}  // namespace lw
