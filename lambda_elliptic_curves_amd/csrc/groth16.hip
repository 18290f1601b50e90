// Groth16 quotient pipeline on the device (SURVEY 8f "next" #3): QuadraticArithmeticProgram::calculate_h_coefficients
// (provers/groth16/src/qap.rs:15-39) =
//     l, r, o  = evaluate_offset_fft(L|R|O, 1, Some(2g), 7)          three coset LDEs of g coefficients to 2g points
//     t_i      = (7 w^i)^g - 1 = 7^g * (-1)^i - 1                    (w^g = -1): two values, inverted on the host
//     h_eval_i = (l_i * r_i - o_i) * t_i^-1                          pointwise
//     h        = interpolate_offset_fft(h_eval, 7)
// composed from the NTT kernels plus one elementwise kernel; the reference makes four host round trips.
#include "context.h"
#include "ntt_kernels.cuh"

namespace lw {

int ntt_device_locked(Context &c, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                      uint32_t log2n, uint32_t batch, size_t stride, const void *coset, hipStream_t stream, uint32_t in_log2);

struct QapParams {
    const uint4 *l, *r, *o;
    uint4 *out;
    uint64_t n;
    uint32_t t_even_inv[8], t_odd_inv[8];
};

__global__ void groth16_quotient_kernel(QapParams p) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    using F = Fr381;
    Fe<F> l = unpack_mem<F>(p.l[2 * i], p.l[2 * i + 1]);
    Fe<F> r = unpack_mem<F>(p.r[2 * i], p.r[2 * i + 1]);
    Fe<F> o = unpack_mem<F>(p.o[2 * i], p.o[2 * i + 1]);
    Fe<F> t;
#pragma unroll
    for (int k = 0; k < 8; k++) t.v[k] = (i & 1) ? p.t_odd_inv[k] : p.t_even_inv[k];
    Fe<F> h = fe_mul<F>(fe_sub<F>(fe_mul<F>(l, r), o), t);
    uint4 q0, q1;
    pack_mem<F>(h, q0, q1);
    p.out[2 * i] = q0;
    p.out[2 * i + 1] = q1;
}

// d_l/d_r/d_o: max(2, 2^log2_gates) coefficients each, zero padded (reference layout); d_tmp: 3 * 2^(log2_gates+1) elements; d_out: 2^(log2_gates+1)
int groth16_h_device(Context &c, const void *d_l, const void *d_r, const void *d_o, uint32_t log2_gates, void *d_out, void *d_tmp,
                     hipStream_t stream) {
    using F = Fr381;
    const uint32_t L = log2_gates + 1;
    const uint64_t n = 1ull << L;
    // ORDER_R_MINUS_1_ROOT_UNITY = 7 (provers/groth16/src/common.rs:26), as an FrElement in reference layout
    Fe<F> seven = fe_from_u64<F>(7);
    uint32_t off_ref[8];
    for (int k = 0; k < 8; k++) off_ref[2 * (3 - k / 2) + (k & 1)] = seven.v[k];
    char *ev = (char *)d_tmp;
    const void *src[3] = {d_l, d_r, d_o};
    for (int k = 0; k < 3; k++) {
        int rc = ntt_device_locked(c, LW_FIELD_BLS12_381_FR, LW_LAYOUT_U64_LIMBS_MS_FIRST, LW_DIR_FORWARD, src[k], ev + (size_t)k * n * 32,
                                   L, 1, 0, off_ref, stream, log2_gates >= 1 ? log2_gates : L);
        if (rc) return rc;
    }
    // t_i = 7^g * (-1)^i - 1
    Fe<F> pw = fe_pow_u64<F>(seven, 1ull << log2_gates);
    Fe<F> te = fe_sub<F>(pw, Fe<F>::one()), to = fe_sub<F>(fe_neg<F>(pw), Fe<F>::one());
    if (te.is_zero() || to.is_zero()) { set_error("vanishing polynomial has a root on the coset"); return LW_ERR_INV_ZERO; }
    te = fe_inv<F>(te);
    to = fe_inv<F>(to);
    QapParams p{};
    p.l = (const uint4 *)ev;
    p.r = (const uint4 *)(ev + n * 32);
    p.o = (const uint4 *)(ev + 2 * n * 32);
    p.out = (uint4 *)ev;   // in place over l
    p.n = n;
    for (int k = 0; k < 8; k++) { p.t_even_inv[k] = te.v[k]; p.t_odd_inv[k] = to.v[k]; }
    hipEvent_t pe = c.prof_begin(stream);
    hipLaunchKernelGGL(groth16_quotient_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, p);
    c.prof_end("groth16_quotient_kernel", pe, stream);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return ntt_device_locked(c, LW_FIELD_BLS12_381_FR, LW_LAYOUT_U64_LIMBS_MS_FIRST, LW_DIR_INVERSE, ev, d_out, L, 1, 0, off_ref, stream, L);
}

// Polynomial::new strips trailing zero coefficients (math/src/polynomial/mod.rs:19-31): *d_len = 1 + index of the last
// non-zero 32-byte element, 0 for the zero polynomial
__global__ void stripped_length_kernel(const uint4 *e, uint64_t n, unsigned long long *len) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 a = e[2 * i], b = e[2 * i + 1];
    if (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) atomicMax(len, (unsigned long long)(i + 1));
}
int stripped_length_device(const void *d_elems, uint64_t n, uint64_t *d_len, hipStream_t stream) {
    LW_HIP_CHECK(hipMemsetAsync(d_len, 0, 8, stream), LW_ERR_LAUNCH);
    if (n) hipLaunchKernelGGL(stripped_length_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, (const uint4 *)d_elems, n,
                              (unsigned long long *)d_len);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

}  // namespace lw
