// Bls12381G1 instantiation of the Pippenger MSM (msm_core.cuh).
#include "msm_core.cuh"

namespace lw {
LW_MSM_INSTANTIATE(Bls12381G1, bls12381_g1)
}  // namespace lw
