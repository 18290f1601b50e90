// Bls12381G2 instantiation of the Pippenger MSM (msm_core.cuh).
#include "msm_core.cuh"

namespace lw {
int msm_run_bls12381_g2(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_points, size_t n, void *out) {
    MsmRunner<Bls12381G2> r{c, s, 0};
    return r.run(d_scalars, d_points, n, out);
}
}  // namespace lw
