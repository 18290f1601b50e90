// The two remaining operations of the reference's GPU seam (math/src/fft/gpu/cuda/ops.rs:45-77):
//   gen_twiddles(order, config)   -> 2^order / 2 powers of the primitive root, natural or bit-reversed, w or w^-1
//                                    (math/src/fft/cpu/roots_of_unity.rs:13-75)
//   bitrev_permutation(input)     -> out[i] = in[bitrev(i)]  (math/src/fft/cpu/bit_reversing.rs:2-18)
// Both are served from the library's cached bit-reversed twiddle table / a plain gather kernel.
#include "context.h"
#include "ntt_kernels.cuh"

namespace lw {

const uint4 *ntt256_twiddle_table(Context &c, int field, lw_dir_t dir, uint32_t log2n, hipStream_t stream, int *rc);
const uint32_t *ntt_bb_twiddle_table(Context &c, lw_dir_t dir, uint32_t log2n, hipStream_t stream, int *rc);

// out (reference layout) [i] = T[natural ? bitrev(i) : i]   (T[g] = w^bitrev(g), internal layout)
template <class F>
__global__ void twiddle_export_kernel(const uint4 *tw, uint4 *out, uint32_t bits, uint64_t count, int natural) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fe<F> x = tw_load<F>(tw, natural ? bitrev_bits((uint32_t)i, bits) : i);
    uint4 q0, q1;
    pack_mem<F>(x, q0, q1);
    out[2 * i] = q0;
    out[2 * i + 1] = q1;
}
template <bool W64>
__global__ void bb_twiddle_export_kernel(const uint32_t *tw, void *out, uint32_t bits, uint64_t count, int natural) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t v = tw[natural ? bitrev_bits((uint32_t)i, bits) : i];
    if (W64) reinterpret_cast<uint64_t *>(out)[i] = bb_to_r64(v);
    else reinterpret_cast<uint32_t *>(out)[i] = v;
}

// out[i] = in[bitrev(i)], elements of WORDS x 4 bytes
template <int WORDS>
__global__ void bitrev_gather_kernel(const uint32_t *in, uint32_t *out, uint32_t bits, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t j = bitrev_bits((uint32_t)i, bits);   // bits <= 32: checked by bitrev_device
    if (WORDS == 8) {
        const uint4 *s = reinterpret_cast<const uint4 *>(in) + 2 * j;
        uint4 *d = reinterpret_cast<uint4 *>(out) + 2 * i;
        d[0] = s[0];
        d[1] = s[1];
    } else if (WORDS == 2) {
        reinterpret_cast<uint64_t *>(out)[i] = reinterpret_cast<const uint64_t *>(in)[j];
    } else {
        out[i] = in[j];
    }
}

int gen_twiddles_device(Context &c, lw_field_t field, lw_layout_t layout, uint32_t order, int config, void *d_out, hipStream_t stream) {
    const uint64_t count = (1ull << order) / 2;
    if (count == 0) return LW_OK;
    const lw_dir_t dir = (config == 1 || config == 3) ? LW_DIR_INVERSE : LW_DIR_FORWARD;
    const int natural = (config == 0 || config == 1) ? 1 : 0;
    const uint32_t bits = order - 1;
    int rc = LW_OK;
    dim3 grid((uint32_t)((count + 255) / 256));
    if (field == LW_FIELD_BABYBEAR) {
        const uint32_t *tw = ntt_bb_twiddle_table(c, dir, order, stream, &rc);
        if (rc) return rc;
        if (layout == LW_LAYOUT_BABYBEAR_U32_R32)
            hipLaunchKernelGGL((bb_twiddle_export_kernel<false>), grid, dim3(256), 0, stream, tw, d_out, bits, count, natural);
        else
            hipLaunchKernelGGL((bb_twiddle_export_kernel<true>), grid, dim3(256), 0, stream, tw, d_out, bits, count, natural);
    } else {
        const uint4 *tw = ntt256_twiddle_table(c, (int)field, dir, order, stream, &rc);
        if (rc) return rc;
        if (field == LW_FIELD_STARK252)
            hipLaunchKernelGGL((twiddle_export_kernel<Stark252>), grid, dim3(256), 0, stream, tw, (uint4 *)d_out, bits, count, natural);
        else
            hipLaunchKernelGGL((twiddle_export_kernel<Fr381>), grid, dim3(256), 0, stream, tw, (uint4 *)d_out, bits, count, natural);
    }
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

// out[b * out_stride + i] = in[b] for i < n: the low-degree extension of a constant polynomial (every evaluation is c_0,
// on any coset: c_0 * h^0), elements of WORDS x 4 bytes
template <int WORDS>
__global__ void broadcast_kernel(const uint32_t *in, uint32_t *out, uint64_t n, uint64_t out_stride) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *s = in + (size_t)blockIdx.y * WORDS;
    uint32_t *d = out + ((size_t)blockIdx.y * out_stride + i) * WORDS;
#pragma unroll
    for (int k = 0; k < WORDS; k++) d[k] = s[k];
}
int broadcast_device(size_t elem_bytes, const void *d_in, void *d_out, uint64_t n, uint32_t batch, uint64_t out_stride, hipStream_t stream) {
    dim3 grid((uint32_t)((n + 255) / 256), batch);
    if (elem_bytes == 32)
        hipLaunchKernelGGL((broadcast_kernel<8>), grid, dim3(256), 0, stream, (const uint32_t *)d_in, (uint32_t *)d_out, n, out_stride);
    else if (elem_bytes == 8)
        hipLaunchKernelGGL((broadcast_kernel<2>), grid, dim3(256), 0, stream, (const uint32_t *)d_in, (uint32_t *)d_out, n, out_stride);
    else
        hipLaunchKernelGGL((broadcast_kernel<1>), grid, dim3(256), 0, stream, (const uint32_t *)d_in, (uint32_t *)d_out, n, out_stride);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

int bitrev_device(size_t elem_bytes, const void *d_in, void *d_out, uint32_t log2n, hipStream_t stream) {
    if (log2n > 32) {   // the index reversal is 32 bits wide (FFTError::OrderError territory: 2^33 elements are >= 32 GiB)
        set_error("bit-reverse permutation of 2^%u elements: at most 2^32 are supported", log2n);
        return LW_ERR_ORDER_TOO_LARGE;
    }
    const uint64_t n = 1ull << log2n;
    dim3 grid((uint32_t)((n + 255) / 256));
    if (elem_bytes == 32)
        hipLaunchKernelGGL((bitrev_gather_kernel<8>), grid, dim3(256), 0, stream, (const uint32_t *)d_in, (uint32_t *)d_out, log2n, n);
    else if (elem_bytes == 8)
        hipLaunchKernelGGL((bitrev_gather_kernel<2>), grid, dim3(256), 0, stream, (const uint32_t *)d_in, (uint32_t *)d_out, log2n, n);
    else
        hipLaunchKernelGGL((bitrev_gather_kernel<1>), grid, dim3(256), 0, stream, (const uint32_t *)d_in, (uint32_t *)d_out, log2n, n);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

}  // namespace lw
