// Cross-shard step of the multi-GPU NTT (Bailey four-step with N1 = number of GPUs G).
//
// A length-N vector is block-distributed over G ranks, M = N/G elements each: x[j1*M + j2], j1 = owning rank.
// With k = k1 + G*k2:   X[k] = sum_j2 w_M^(j2 k2) * { w_N^(j2 k1) * sum_j1 w_G^(j1 k1) x[j1*M + j2] }.
// After the first all-to-all a rank holds, for its slice of j2, all G values of j1; this kernel evaluates the
// braces (a G-point transform across the received chunks, then the inter-step twiddle).  The second all-to-all
// hands row k1 to rank k1, which runs an ordinary local M-point NTT (lw_hip_ntt_device).
// The reference has no multi-device path; the result is the same vector Polynomial::evaluate_fft
// returns (math/src/fft/polynomial.rs:25-68), sharded.
#include <vector>
#include "context.h"
#include "ntt_kernels.cuh"

namespace lw {

int ntt256_power_tables(Context &c, int field, int slot, const uint32_t *base_words, bool invert, uint32_t hbits,
                        uint32_t hi_bits, hipStream_t stream, const uint4 **lo, const uint4 **hi);
int ntt256_root_words(int field, uint32_t order, bool inverse, uint32_t *words);
int ntt256_inv_u64_words(int field, uint64_t v, uint32_t *words);
const uint4 *ntt256_twiddle_table(Context &c, int field, lw_dir_t dir, uint32_t log2n, hipStream_t stream, int *rc);

struct CrossParams {
    const uint4 *in;
    uint4 *out;
    const uint4 *tw;          // bit-reversed table (prefix serves the G-point transform)
    const uint4 *plo, *phi;   // power tables of w_N: w_N^e = plo[e & mask] * phi[e >> hbits]
    uint32_t hbits;
    uint64_t chunk_stride;    // elements between consecutive j1 (in) / k1 (out) chunks
    uint64_t batch_stride;
    uint64_t j2_begin, slice_len;
    uint32_t scale;           // inverse: multiply by G^-1
    uint32_t sc[8];
};

template <class F, int LG>
__global__ __launch_bounds__(128) void ntt_cross_kernel(CrossParams p) {
    constexpr int G = 1 << LG;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.slice_len) return;
    const uint4 *gin = p.in + 2 * ((uint64_t)blockIdx.y * p.batch_stride + t);
    uint4 *gout = p.out + 2 * ((uint64_t)blockIdx.y * p.batch_stride + t);
    Fe<F> x[G];
#pragma unroll
    for (int j = 0; j < G; j++) x[j] = unpack_mem<F>(gin[2 * j * p.chunk_stride], gin[2 * j * p.chunk_stride + 1]);
    // G-point NR-DIT (math/src/fft/cpu/fft.rs:20-55) in registers; outputs land bit-reversed
#pragma unroll
    for (int u = 0; u < LG; u++) {
        const int half = 1 << (LG - 1 - u);
#pragma unroll
        for (int jt = 0; jt < (1 << u); jt++) {
            Fe<F> tw = tw_load<F>(p.tw, (uint64_t)jt);
#pragma unroll
            for (int jl = 0; jl < half; jl++) {
                const int j = (jt << (LG - u)) | jl;
                Fe<F> wb = fe_mul<F>(tw, x[j + half]);
                Fe<F> a = x[j];
                x[j] = fe_add<F>(a, wb);
                x[j + half] = fe_sub<F>(a, wb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // inter-step twiddle w_N^(j2*k1): running power of w_N^j2
    const uint64_t j2 = p.j2_begin + t;
    Fe<F> base = fe_mul<F>(tw_load<F>(p.plo, j2 & ((1ull << p.hbits) - 1)), tw_load<F>(p.phi, j2 >> p.hbits));
    Fe<F> pw;
#pragma unroll
    for (int i = 0; i < 8; i++) pw.v[i] = p.sc[i];
#pragma unroll
    for (int k1 = 0; k1 < G; k1++) {
        // natural k1 sits at NR position bitrev(k1)
        int q = 0;
#pragma unroll
        for (int b = 0; b < LG; b++) q |= ((k1 >> b) & 1) << (LG - 1 - b);
        Fe<F> y = x[q];
        if (k1 > 0 || p.scale) y = fe_mul<F>(y, pw);
        if (k1 + 1 < G) pw = (k1 == 0 && !p.scale) ? base : fe_mul<F>(pw, base);
        uint4 q0, q1;
        pack_mem<F>(y, q0, q1);
        gout[2 * k1 * p.chunk_stride] = q0;
        gout[2 * k1 * p.chunk_stride + 1] = q1;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- BabyBear ----
struct CrossParamsBb {
    const void *in;
    void *out;
    const uint32_t *tw;
    uint32_t wN;              // w_N (or its inverse), R = 2^32 domain
    uint32_t lgV;
    uint64_t chunk_stride, batch_stride;   // in words
    uint64_t j2_begin, slice_words;
    uint32_t scale, sc;
};

template <int LG, bool W64>
__global__ __launch_bounds__(256) void bb_cross_kernel(CrossParamsBb p) {
    constexpr int G = 1 << LG;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.slice_words) return;
    const uint64_t off = (uint64_t)blockIdx.y * p.batch_stride + t;
    uint32_t x[G];
#pragma unroll
    for (int j = 0; j < G; j++) {
        if (W64) x[j] = bb_from_r64(reinterpret_cast<const uint64_t *>(p.in)[off + j * p.chunk_stride]);
        else x[j] = reinterpret_cast<const uint32_t *>(p.in)[off + j * p.chunk_stride];
    }
#pragma unroll
    for (int u = 0; u < LG; u++) {
        const int half = 1 << (LG - 1 - u);
#pragma unroll
        for (int jt = 0; jt < (1 << u); jt++) {
            const uint32_t tw = p.tw[jt];
#pragma unroll
            for (int jl = 0; jl < half; jl++) {
                const int j = (jt << (LG - u)) | jl;
                const uint32_t wb = bb_mul(tw, x[j + half]);
                const uint32_t a = x[j];
                x[j] = bb_add(a, wb);
                x[j + half] = bb_sub(a, wb);
            }
        }
    }
    const uint64_t j2 = p.j2_begin + (t >> p.lgV);
    const uint32_t base = bb_pow(p.wN, j2);
    uint32_t pw = p.sc;
#pragma unroll
    for (int k1 = 0; k1 < G; k1++) {
        int q = 0;
#pragma unroll
        for (int b = 0; b < LG; b++) q |= ((k1 >> b) & 1) << (LG - 1 - b);
        uint32_t y = x[q];
        if (k1 > 0 || p.scale) y = bb_mul(y, pw);
        if (k1 + 1 < G) pw = (k1 == 0 && !p.scale) ? base : bb_mul(pw, base);
        if (W64) reinterpret_cast<uint64_t *>(p.out)[off + k1 * p.chunk_stride] = bb_to_r64(y);
        else reinterpret_cast<uint32_t *>(p.out)[off + k1 * p.chunk_stride] = y;
    }
}

const uint32_t *ntt_bb_twiddle_table(Context &c, lw_dir_t dir, uint32_t log2n, hipStream_t stream, int *rc);
uint32_t ntt_bb_root(uint32_t order, bool inverse);

template <class F>
static int cross256(Context &c, int field, lw_dir_t dir, const void *d_in, void *d_out, uint32_t log2_total, uint32_t lg,
                    uint64_t j2_begin, uint64_t slice_len, uint64_t chunk_stride, uint32_t batch, uint64_t batch_stride,
                    hipStream_t stream) {
    int rc = LW_OK;
    CrossParams p{};
    p.tw = ntt256_twiddle_table(c, field, dir, lg, stream, &rc);   // only T[0..G/2) is read: any cached table serves (prefix property)
    if (rc) return rc;
    uint32_t wN[8];
    rc = ntt256_root_words(field, log2_total, dir == LW_DIR_INVERSE, wN);
    if (rc) return rc;
    const uint32_t mbits = log2_total - lg;             // j2 < 2^mbits
    const uint32_t hbits = (mbits + 1) / 2;
    rc = ntt256_power_tables(c, field, 2, wN, false, hbits, mbits - hbits, stream, &p.plo, &p.phi);
    if (rc) return rc;
    p.hbits = hbits;
    p.in = (const uint4 *)d_in;
    p.out = (uint4 *)d_out;
    p.chunk_stride = chunk_stride;
    p.batch_stride = batch_stride;
    p.j2_begin = j2_begin;
    p.slice_len = slice_len;
    p.scale = dir == LW_DIR_INVERSE ? 1 : 0;
    if (p.scale) {
        rc = ntt256_inv_u64_words(field, 1ull << lg, p.sc);
        if (rc) return rc;
    }
    dim3 grid((uint32_t)((slice_len + 127) / 128), batch);
    hipEvent_t pe = c.prof_begin(stream);
    switch (lg) {
        case 1: hipLaunchKernelGGL((ntt_cross_kernel<F, 1>), grid, dim3(128), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((ntt_cross_kernel<F, 2>), grid, dim3(128), 0, stream, p); break;
        default: hipLaunchKernelGGL((ntt_cross_kernel<F, 3>), grid, dim3(128), 0, stream, p); break;
    }
    c.prof_end("ntt_cross_kernel", pe, stream);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

template <bool W64>
static int cross_bb(Context &c, lw_dir_t dir, uint32_t lgV, const void *d_in, void *d_out, uint32_t log2_total, uint32_t lg,
                    uint64_t j2_begin, uint64_t slice_len, uint64_t chunk_stride, uint32_t batch, uint64_t batch_stride,
                    hipStream_t stream) {
    int rc = LW_OK;
    CrossParamsBb p{};
    p.tw = ntt_bb_twiddle_table(c, dir, lg, stream, &rc);   // only T[0..G/2) is read
    if (rc) return rc;
    p.wN = ntt_bb_root(log2_total, dir == LW_DIR_INVERSE);
    p.lgV = lgV;
    p.in = d_in;
    p.out = d_out;
    p.chunk_stride = chunk_stride << lgV;
    p.batch_stride = batch_stride << lgV;
    p.j2_begin = j2_begin;
    p.slice_words = slice_len << lgV;
    p.scale = dir == LW_DIR_INVERSE ? 1 : 0;
    p.sc = p.scale ? bb_inv(bb_mul((uint32_t)(1u << lg), BabyBear::R2)) : BabyBear::ONE;
    dim3 grid((uint32_t)((p.slice_words + 255) / 256), batch);
    hipEvent_t pe = c.prof_begin(stream);
    switch (lg) {
        case 1: hipLaunchKernelGGL((bb_cross_kernel<1, W64>), grid, dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((bb_cross_kernel<2, W64>), grid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((bb_cross_kernel<3, W64>), grid, dim3(256), 0, stream, p); break;
    }
    c.prof_end("bb_cross_kernel", pe, stream);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

int ntt_cross_device(Context &c, lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                     uint32_t log2_total, uint32_t log2_g, uint64_t j2_begin, uint64_t slice_len, uint64_t chunk_stride,
                     uint32_t batch, uint64_t batch_stride, hipStream_t stream) {
    if (log2_g < 1 || log2_g > 3 || log2_g > log2_total) {
        set_error("cross step supports 2, 4 or 8 shards (got 2^%u of 2^%u)", log2_g, log2_total);
        return LW_ERR_BAD_ARG;
    }
    if (field == LW_FIELD_STARK252)
        return cross256<Stark252>(c, field, dir, d_in, d_out, log2_total, log2_g, j2_begin, slice_len, chunk_stride, batch, batch_stride, stream);
    if (field == LW_FIELD_BLS12_381_FR)
        return cross256<Fr381>(c, field, dir, d_in, d_out, log2_total, log2_g, j2_begin, slice_len, chunk_stride, batch, batch_stride, stream);
    switch (layout) {
        case LW_LAYOUT_BABYBEAR_U32_R32: return cross_bb<false>(c, dir, 0, d_in, d_out, log2_total, log2_g, j2_begin, slice_len, chunk_stride, batch, batch_stride, stream);
        case LW_LAYOUT_BABYBEAR_U64_R64: return cross_bb<true>(c, dir, 0, d_in, d_out, log2_total, log2_g, j2_begin, slice_len, chunk_stride, batch, batch_stride, stream);
        case LW_LAYOUT_EXT4_INTERLEAVED: return cross_bb<true>(c, dir, 2, d_in, d_out, log2_total, log2_g, j2_begin, slice_len, chunk_stride, batch, batch_stride, stream);
        default: set_error("bad layout"); return LW_ERR_BAD_ARG;
    }
}

}  // namespace lw
