// Bn254G2 instantiation of the Pippenger MSM (msm_core.cuh).
#include "msm_core.cuh"

namespace lw {
LW_MSM_INSTANTIATE(Bn254G2, bn254_g2)
}  // namespace lw
