// Bls12381G2 instantiation of the Pippenger MSM (msm_core.cuh).
#include "msm_core.cuh"

namespace lw {
LW_MSM_INSTANTIATE(Bls12381G2, bls12381_g2)
}  // namespace lw
