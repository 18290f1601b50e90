// One FRI commit-phase layer on the device (SURVEY 8f "next" #4): the body of the loop in commit_phase
// (provers/stark/src/fri/mod.rs:44-58):
//     p'        = 2 * fold_polynomial(p, zeta)          fold: p'_i = c_2i + zeta * c_2i+1   (fri/fri_functions.rs:7-30)
//     layer     = new_fri_layer(p', offset, domain)     (fri/mod.rs:115-141):
//                   evaluation = evaluate_offset_fft(p', 1, Some(domain), offset), bit-reverse permuted,
//                   leaves     = chunks of 2 consecutive evaluations, BatchedMerkleTree (Keccak-256)
// The transcript (zeta sampling, root absorption) stays on the host; offset^2 and domain/2 are the caller's.
// A leaf [e_br[2k], e_br[2k+1]] of the bit-reversed vector is [e[j], e[j + N/2]] with j = bitrev_{L-1}(k): the same
// commitment kernel as the trace columns, with two "columns" that are the two halves of the evaluation vector.
#include "context.h"
#include "ntt_kernels.cuh"

namespace lw {

int ntt_device_locked(Context &c, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                      uint32_t log2n, uint32_t batch, size_t stride, const void *coset, hipStream_t stream, uint32_t in_log2);
int merkle_commit_device(Context &c, const void *d_cols, uint32_t n_cols, uint64_t col_stride, uint32_t log2n, int bit_reverse,
                         void *d_nodes, hipStream_t stream, uint32_t elem_bytes = 32);
int bitrev_device(size_t elem_bytes, const void *d_in, void *d_out, uint32_t log2n, hipStream_t stream);

// out[i] = 2 * (c[2i] + zeta * c[2i+1]) for i < n_out; zeros up to `padded`
struct FriZeta {
    uint32_t w[8];   // the challenge, internal limbs (8 x u32, least significant first): a kernel argument, nothing to upload
};
template <class F>
__global__ void fri_fold_kernel(const uint4 *in, uint64_t n, FriZeta zeta_words, uint4 *out, uint64_t n_out, uint64_t padded) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= padded) return;
    Fe<F> r = Fe<F>::zero();
    if (i < n_out) {
        Fe<F> z;
#pragma unroll
        for (int k = 0; k < 8; k++) z.v[k] = zeta_words.w[k];
        r = unpack_mem<F>(in[4 * i], in[4 * i + 1]);
        if (2 * i + 1 < n) r = fe_add<F>(r, fe_mul<F>(z, unpack_mem<F>(in[4 * i + 2], in[4 * i + 3])));
        r = fe_add<F>(r, r);
    }
    uint4 q0, q1;
    pack_mem<F>(r, q0, q1);
    out[2 * i] = q0;
    out[2 * i + 1] = q1;
}

// d_coeffs: n coefficients; d_poly: padded block (power of two >= max(ceil(n/2), 2)); d_eval / d_eval_br: domain elements;
// d_nodes: (domain - 1) * 32 bytes; zeta: 8 u32 (internal limbs, host).  d_eval == nullptr: fold only (the last step of
// commit_phase, fri/mod.rs:61-63, has no layer).
int fri_layer_device(Context &c, lw_field_t field, const void *d_coeffs, uint64_t n, const uint32_t *zeta, const void *offset_ref,
                     uint32_t log2_domain, void *d_poly, uint32_t log2_block, void *d_eval, void *d_eval_br, void *d_nodes,
                     hipStream_t stream) {
    const uint64_t n_out = (n + 1) / 2, padded = 1ull << log2_block;
    FriZeta d_zeta;
    for (int k = 0; k < 8; k++) d_zeta.w[k] = zeta[k];
    dim3 grid((uint32_t)((padded + 255) / 256));
    hipEvent_t pe = c.prof_begin(stream);
    if (field == LW_FIELD_STARK252)
        hipLaunchKernelGGL((fri_fold_kernel<Stark252>), grid, dim3(256), 0, stream, (const uint4 *)d_coeffs, n, d_zeta, (uint4 *)d_poly, n_out, padded);
    else
        hipLaunchKernelGGL((fri_fold_kernel<Fr381>), grid, dim3(256), 0, stream, (const uint4 *)d_coeffs, n, d_zeta, (uint4 *)d_poly, n_out, padded);
    c.prof_end("fri_fold_kernel", pe, stream);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    if (!d_eval) return LW_OK;
    int rc = ntt_device_locked(c, field, LW_LAYOUT_U64_LIMBS_MS_FIRST, LW_DIR_FORWARD, d_poly, d_eval, log2_domain, 1, 0, offset_ref, stream,
                               log2_block);
    if (rc) return rc;
    rc = merkle_commit_device(c, d_eval, 2, 1ull << (log2_domain - 1), log2_domain - 1, 1, d_nodes, stream);
    if (rc) return rc;
    return d_eval_br ? bitrev_device(32, d_eval, d_eval_br, log2_domain, stream) : LW_OK;
}

}  // namespace lw
