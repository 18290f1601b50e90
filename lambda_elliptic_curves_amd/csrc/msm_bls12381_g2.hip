// Bls12381G2 instantiation of the Pippenger MSM (msm_core.cuh).
#include "msm_core.cuh"

namespace lw {
int msm_run_bls12381_g2(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_points, size_t n, void *out, int affine,
                      hipEvent_t points_ready) {
    MsmRunner<Bls12381G2> r{c, s, 0};
    r.affine = affine != 0;
    r.points_ready = points_ready;
    return r.run(d_scalars, d_points, n, out);
}
int msm_normalize_bls12381_g2(Context &c, hipStream_t s, const void *d_in, size_t n, void *d_out) {
    MsmRunner<Bls12381G2> r{c, s, 0};
    return r.normalize(d_in, n, d_out);
}
}  // namespace lw
