// BabyBear (p = 2^31 - 2^27 + 1) NTT on gfx950 for the reference's three memory shapes:
//   * U32_R32          one u32 per element, Montgomery R = 2^32  (babybear_u32.rs:6)
//   * U64_R64          one u64 per element, Montgomery R = 2^64  (babybear.rs:19-20)
//   * EXT4_INTERLEAVED Degree4BabyBearExtensionField values (4 x u64 per element) over a base-field domain:
//                      F x E multiplication is four independent base products (quartic_babybear.rs:155-166), so
//                      the transform is four interleaved base-field transforms (V = 4 components per index).
// Same NR-DIT dataflow and pass structure as ntt_kernels.cuh (math/src/fft/cpu/fft.rs:20-55 +
// bit_reversing.rs:2-18); registers and LDS hold u32 values in the R = 2^32 domain, the u64 shapes are
// converted on load/store (field.cuh bb_from_r64 / bb_to_r64), which returns the same canonical residues.
#include <stdlib.h>
#include <vector>
#include "context.h"
#include "field.cuh"

namespace lw {

uint32_t ntt_get_debug();

constexpr int BB_TILE_LOG = 13;            // 8192 u32 = 32 KiB of LDS
constexpr int BB_TILE = 1 << BB_TILE_LOG;
constexpr int BB_THREADS = 256;
constexpr int BB_KMAX = 4;                 // radix-16 register steps

struct BbPassParams {
    const void *in;
    void *out;
    const uint32_t *tw;        // T[g] = w^bitrev(g), R = 2^32 domain
    const uint2 *dd;           // dd[g] = (T[2g] * T[g], -T[2g+1] * T[g]): the composite twiddles of two fused stages
    uint64_t in_batch_stride;  // in memory words
    uint64_t out_batch_stride;
    uint32_t L;                // log2 N (transform length)
    uint32_t lgV;              // log2 components per index (0, or 2 for EXT4)
    uint32_t s0, r, logC;      // logC counts columns x components
    uint32_t nsteps;
    uint32_t k[8];
    uint32_t scale, sc;        // N^-1 (R = 2^32 domain) on the last pass of an inverse transform
    // coset transforms (evaluate_offset_fft / interpolate_offset_fft): element i is multiplied by lo[i & mask] * hi[i >> hbits]
    // = h^i while the first pass loads it (cos_in), or by h^-i * N^-1 while the last pass stores it (cos_out, N^-1 folded into hi)
    const uint32_t *cos_lo, *cos_hi;
    uint32_t cos_hbits, cos_in, cos_out;
    // low-degree extension (first pass only): word g of the zero-padded input is read from in[g & in_mask] — after the
    // log2(blow-up) stages that only pair data with padding the vector is the coefficient block replicated, so those
    // stages are skipped (s0 starts there) and the padding is never materialised.  Plain transforms: all ones.
    uint32_t in_mask;
    uint32_t dbg;              // ablation builds only (-DLW_HIP_ABLATION): bit0 skip butterflies, bit1 skip loads, bit2 skip stores, bit3 old last-pass mapping
};

__device__ __forceinline__ uint32_t bb_bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

__device__ __forceinline__ uint32_t bb_coset_factor(const BbPassParams &p, uint32_t i) {
    return bb_mul(p.cos_lo[i & ((1u << p.cos_hbits) - 1)], p.cos_hi[i >> p.cos_hbits]);
}
template <bool W64>
__device__ __forceinline__ uint32_t bb_load_word(const void *base, uint32_t idx) {
    if (W64) return bb_from_r64(reinterpret_cast<const uint64_t *>(base)[idx]);
    return reinterpret_cast<const uint32_t *>(base)[idx];
}
template <bool W64>
__device__ __forceinline__ void bb_store_word(void *base, uint32_t idx, uint32_t v) {
    if (W64) reinterpret_cast<uint64_t *>(base)[idx] = bb_to_r64(v);
    else reinterpret_cast<uint32_t *>(base)[idx] = v;
}

// RX: 0 = tile shape from the parameters; 8 / 7 / 6 = a full-size tile of r = RX stages x 2^(13 - r) columns x components in
// two register steps (4+4, 4+3, 3+3 stages) with lgV = VX (0 or 2) — all compile-time constants, so that every shift,
// mask, bit reversal and swizzle of the index arithmetic folds (every pass of a transform of 2^18 words and more; LDS
// addressing, loads and stores were a quarter of the kernel time with run-time shapes).
template <int K, bool LAST, bool IN64, int RX = 0, int VX = 0>
__device__ __forceinline__ void bb_item(const BbPassParams &p, uint32_t *lds, const uint32_t *ltw, const uint32_t *ld1, const uint32_t *ld2,
                                        const void *gin, uint32_t w,
                                        uint32_t step, uint32_t t0, uint32_t base, uint32_t lgS, uint32_t hi_uniform,
                                        uint32_t hi_low, bool last_step) {
    constexpr int E = 1 << K;
    const uint32_t r = RX ? (uint32_t)RX : p.r, logC = RX ? (uint32_t)(BB_TILE_LOG - RX) : p.logC, L = p.L, lgV = RX ? (uint32_t)VX : p.lgV;
    const uint32_t logCh = logC - lgV;               // columns proper (distinct tiles in the last pass)
    const uint32_t sh = r - t0 - K;
    // Last pass: work-items walk rows fastest in every step.  In its first step that matches the memory order of the
    // input (column, row, component); in the later ones it keeps the lanes of a wave inside a few columns, whose
    // twiddles T[(hi_c << t) | x] are then neighbours in the table (columns-fastest made every lane fetch a different
    // cache line).  LDS slots are XOR-swizzled by row bits (lds_slot) so rows-fastest accesses spread over the banks.
    const bool rows_fastest = LAST && (step == 0 || !(LW_DBG(p) & 8));
    uint32_t c, mr;
    if (rows_fastest) {
        const uint32_t comp = w & ((1u << lgV) - 1);
        mr = (w >> lgV) & ((1u << (r - K)) - 1);
        c = ((w >> (lgV + r - K)) << lgV) | comp;
    } else {
        c = w & ((1u << logC) - 1);
        mr = w >> logC;
    }
    const uint32_t swz = (LAST && !(LW_DBG(p) & 8)) ? ((1u << logC) - 1) : 0u;
    const uint32_t m_low = mr & ((1u << sh) - 1);
    const uint32_t m_high = mr >> sh;
    const uint32_t mbase = (m_high << (sh + K)) | m_low;
    uint32_t hi_c = hi_uniform;
    if (LAST) hi_c = (bb_bitrev(c >> lgV, logCh) << (L - r - logCh)) | hi_low;

    uint32_t x[E];
    if (step == 0) {   // branch hoisted out of the element loop: the E global loads are issued back to back
#pragma unroll
        for (int j = 0; j < E; j++) {
            const uint32_t m = mbase | ((uint32_t)j << sh);
            // word indices fit 32 bits: two-adicity 24 (babybear.rs:29) caps a transform at 2^26 words
            uint32_t g;
            if (LAST) g = ((((hi_c << r) | m)) << lgV) | (c & ((1u << lgV) - 1));
            else g = base + (m << lgS) + c;
            x[j] = (LW_DBG(p) & 2) ? g : bb_load_word<IN64>(gin, g & p.in_mask);
        }
        if (p.cos_in) {   // c_i * h^i, fused into the first pass's load (its own loop: the loads above stay back to back)
#pragma unroll
            for (int j = 0; j < E; j++) {
                const uint32_t m = mbase | ((uint32_t)j << sh);
                uint32_t g;
                if (LAST) g = ((((hi_c << r) | m)) << lgV) | (c & ((1u << lgV) - 1));
                else g = base + (m << lgS) + c;
                x[j] = bb_mul(x[j], bb_coset_factor(p, (g & p.in_mask) >> lgV));
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; j++) {
            const uint32_t m = mbase | ((uint32_t)j << sh);
            x[j] = lds[(m << logC) | (c ^ ((m ^ (m >> 4)) & swz))];
        }
    }
    // Two stages at a time where the step has them (K = 4: twice), as one radix-4 butterfly whose products are grouped by
    // output, not by stage: with a, c, b, d = x[j], x[j+q], x[j+2q], x[j+3q], stage twiddle w0 = T[G] and next-stage
    // twiddles w1 = T[2G], w2 = T[2G+1],
    //   S = w0 b,  U = w1 c + (w1 w0) d,  V = w2 c - (w2 w0) d,
    //   a" = (a + S) + U,  c" = (a + S) - U,  b" = (a - S) + V,  d" = (a - S) - V
    // are exactly the radix-2 results (same canonical residues), but U and V are two-term dot products with ONE
    // Montgomery reduction each (2 p^2 + 2^32 p < 2^64): 5 v_mad_u64_u32 + 3 reductions + 6 additions = 35 VALU
    // instructions where two radix-2 stages take 4 x 5 + 8 x 3 = 44.  A 32-bit Montgomery product is dominated by its
    // reduction (4 of 5 instructions), which is why this pays here and not for the 256-bit fields.  The composite
    // twiddles w1 w0 and -w2 w0 come from the table dd (same indexing as T).  Non-last passes only (their twiddles are
    // staged in LDS): in the last pass every work-item fetches its own twiddles from the global tables and the two extra
    // values per butterfly cost more than the instructions saved (measured: last pass 0.227 -> 0.235 ms, others 0.165 -> 0.150).
    if (!(LW_DBG(p) & 1)) {
        int u = 0;
#pragma unroll
        for (; !LAST && u + 1 < K; u += 2) {
            const int half = 1 << (K - 1 - u), q = half >> 1;
            const uint32_t gt = (hi_c << (t0 + u)) | (m_high << u);
#pragma unroll
            for (int jt = 0; jt < (1 << u); jt++) {
                uint32_t w0, w1, w2, e1, e2;
                if (LAST) {
                    const uint32_t G = gt | (uint32_t)jt;
                    const uint2 w12 = reinterpret_cast<const uint2 *>(p.tw)[G], e12 = p.dd[G];   // one 8-byte load each
                    w0 = p.tw[G]; w1 = w12.x; w2 = w12.y; e1 = e12.x; e2 = e12.y;
                } else {   // the tile's twiddles sit in LDS, stage t group x at slot 2^t - 1 + x
                    const uint32_t xg = (m_high << u) | (uint32_t)jt;
                    const uint32_t s0 = (1u << (t0 + u)) - 1 + xg, s1 = (2u << (t0 + u)) - 1 + 2 * xg;
                    w0 = ltw[s0]; w1 = ltw[s1]; w2 = ltw[s1 + 1]; e1 = ld1[s0]; e2 = ld2[s0];
                }
#pragma unroll
                for (int jl = 0; jl < q; jl++) {
                    const int j = (jt << (K - u)) | jl;
                    const uint32_t a = x[j], cc = x[j + q], b = x[j + half], d = x[j + half + q];
                    const uint32_t S = bb_mul(w0, b);
                    const uint32_t U = bb_reduce((uint64_t)w1 * cc + (uint64_t)e1 * d);
                    const uint32_t V = bb_reduce((uint64_t)w2 * cc + (uint64_t)e2 * d);
                    const uint32_t ap = bb_add(a, S), am = bb_sub(a, S);
                    x[j] = bb_add(ap, U);
                    x[j + q] = bb_sub(ap, U);
                    x[j + half] = bb_add(am, V);
                    x[j + half + q] = bb_sub(am, V);
                }
            }
        }
#pragma unroll
        for (; u < K; u++) {   // odd K: one radix-2 stage is left
            const int half = 1 << (K - 1 - u);
            const uint32_t gt = (hi_c << (t0 + u)) | (m_high << u);
            // last pass: the 2^u twiddles of this stage are consecutive table entries starting at a multiple of 2^u — one
            // 8- or 16-byte load (two for u = 3) instead of 2^u dword loads: 5 load instructions per radix-16 step, not 15
            uint32_t twv[K > 1 ? (1 << (K - 1)) : 2];
            if (LAST) {
                if (u == 0) twv[0] = p.tw[gt];
                else if (u == 1) { const uint2 v = *reinterpret_cast<const uint2 *>(p.tw + gt); twv[0] = v.x; twv[1] = v.y; }
                else {
#pragma unroll
                    for (int q4 = 0; q4 < (1 << u) / 4; q4++) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(p.tw + gt + 4 * q4);
                        twv[4 * q4] = v.x; twv[4 * q4 + 1] = v.y; twv[4 * q4 + 2] = v.z; twv[4 * q4 + 3] = v.w;
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < (1 << u); jt++) {
                const uint32_t tw = LAST ? twv[jt] : ltw[(1u << (t0 + u)) - 1 + ((m_high << u) | (uint32_t)jt)];
#pragma unroll
                for (int jl = 0; jl < half; jl++) {
                    const int j = (jt << (K - u)) | jl;
                    const uint32_t wb = bb_mul(tw, x[j + half]);
                    const uint32_t a = x[j];
                    x[j] = bb_add(a, wb);
                    x[j + half] = bb_sub(a, wb);
                }
            }
        }
    }
    if (last_step && p.scale) {
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = bb_mul(x[j], p.sc);
    }
#pragma unroll
    for (int j = 0; j < E; j++) {
        const uint32_t m = mbase | ((uint32_t)j << sh);
        lds[(m << logC) | (c ^ ((m ^ (m >> 4)) & swz))] = x[j];
    }
}

// IN64 / OUT64: word type of the pass's source / destination.  Only the caller's buffers hold u64 words (R = 2^64): the
// intermediate vector between two passes is always written as u32 in the R = 2^32 domain, so a three-pass transform of
// the u64 shapes moves (8+4) + (4+4) + (4+8) bytes per word instead of 3 x 16.
template <bool LAST, bool IN64, bool OUT64, int RX = 0, int VX = 0>
__global__ __launch_bounds__(BB_THREADS) void bb_pass_kernel(BbPassParams p) {
    __shared__ uint32_t lds[BB_TILE];
    __shared__ uint32_t ltw[LAST ? 1 : 256];
    __shared__ uint32_t ld1[LAST ? 1 : 128], ld2[LAST ? 1 : 128];   // composite twiddles of the stages 0 .. r-2
    const uint32_t tid = threadIdx.x;
    const uint32_t r = RX ? (uint32_t)RX : p.r, logC = RX ? (uint32_t)(BB_TILE_LOG - RX) : p.logC, L = p.L, lgV = RX ? (uint32_t)VX : p.lgV;
    const uint32_t tile_log = r + logC;
    const char *gin = (const char *)p.in + (uint64_t)blockIdx.y * p.in_batch_stride * (IN64 ? 8 : 4);
    char *gout = (char *)p.out + (uint64_t)blockIdx.y * p.out_batch_stride * (OUT64 ? 8 : 4);
    const uint32_t b = blockIdx.x;
    const uint32_t Lw = L + lgV;      // index bits of the word array

    uint32_t base = 0;
    uint32_t lgS = 0, hi_uniform = 0, hi_low = 0;
    if (!LAST) {
        lgS = Lw - p.s0 - r;          // row stride in words
        const uint32_t lo_bits = lgS - logC;
        const uint32_t lo_blk = b & ((1u << lo_bits) - 1);
        hi_uniform = b >> lo_bits;
        base = (hi_uniform << (Lw - p.s0)) + (lo_blk << logC);
        for (uint32_t i = tid; i + 1 < (1u << r); i += BB_THREADS) {   // r <= 8: at most 255 twiddles
            const uint32_t t = 31 - __clz(i + 1), xg = i + 1 - (1u << t);
            ltw[i] = p.tw[(hi_uniform << t) | xg];
            if (t + 1 < r) {
                const uint2 e12 = p.dd[(hi_uniform << t) | xg];
                ld1[i] = e12.x;
                ld2[i] = e12.y;
            }
        }
        __syncthreads();
    } else {
        hi_low = bb_bitrev(b, L - r - (logC - lgV));
    }
    if constexpr (RX != 0) {   // two register steps, everything about the tile shape known at compile time
        static_assert(RX == 8 || RX == 7 || RX == 6, "full-size tiles of 8, 7 or 6 stages");
        constexpr int K0 = RX >= 7 ? 4 : 3, K1 = RX - K0;
#define LW_BB_STEP(KK, STEP, T0, LASTSTEP)                                                                                                  \
    do {                                                                                                                                     \
        bb_item<KK, LAST, IN64, RX, VX>(p, lds, ltw, ld1, ld2, gin, tid, STEP, T0, base, lgS, hi_uniform, hi_low, LASTSTEP);                 \
        bb_item<KK, LAST, IN64, RX, VX>(p, lds, ltw, ld1, ld2, gin, tid + BB_THREADS, STEP, T0, base, lgS, hi_uniform, hi_low, LASTSTEP);    \
        if constexpr (KK == 3) {   /* 2^(13-3) = 1024 items: four per work-item */                                                           \
            bb_item<KK, LAST, IN64, RX, VX>(p, lds, ltw, ld1, ld2, gin, tid + 2 * BB_THREADS, STEP, T0, base, lgS, hi_uniform, hi_low, LASTSTEP); \
            bb_item<KK, LAST, IN64, RX, VX>(p, lds, ltw, ld1, ld2, gin, tid + 3 * BB_THREADS, STEP, T0, base, lgS, hi_uniform, hi_low, LASTSTEP); \
        }                                                                                                                                    \
    } while (0)
        static_assert(BB_TILE == 16 * 2 * BB_THREADS, "two radix-16 items (four radix-8 items) per work-item and step");
        LW_BB_STEP(K0, 0u, 0u, false);
        __syncthreads();
        LW_BB_STEP(K1, 1u, (uint32_t)K0, true);
#undef LW_BB_STEP
    } else {
    uint32_t t0 = 0;
    for (uint32_t step = 0; step < p.nsteps; step++) {
        const uint32_t k = p.k[step];
        const uint32_t nitems = 1u << (tile_log - k);
        const bool last_step = (step + 1 == p.nsteps);
        if (step) __syncthreads();
        for (uint32_t w = tid; w < nitems; w += BB_THREADS) {
            if (k == 4) bb_item<4, LAST, IN64>(p, lds, ltw, ld1, ld2, gin, w, step, t0, base, lgS, hi_uniform, hi_low, last_step);
            else if (k == 3) bb_item<3, LAST, IN64>(p, lds, ltw, ld1, ld2, gin, w, step, t0, base, lgS, hi_uniform, hi_low, last_step);
            else if (k == 2) bb_item<2, LAST, IN64>(p, lds, ltw, ld1, ld2, gin, w, step, t0, base, lgS, hi_uniform, hi_low, last_step);
            else bb_item<1, LAST, IN64>(p, lds, ltw, ld1, ld2, gin, w, step, t0, base, lgS, hi_uniform, hi_low, last_step);
        }
        t0 += k;
    }
    }
    __syncthreads();
    const uint32_t logCh = logC - lgV;
    auto store_one = [&](uint32_t e) {
        const uint32_t c = e & ((1u << logC) - 1);
        const uint32_t m = e >> logC;
        uint32_t g;
        if (!LAST) g = base + (m << lgS) + c;
        else g = (((bb_bitrev(m, r) << (L - r)) + (b << logCh) + (c >> lgV)) << lgV) | (c & ((1u << lgV) - 1));
        const uint32_t swz = (LAST && !(LW_DBG(p) & 8)) ? ((1u << logC) - 1) : 0u;   // same slot mapping as bb_item
        uint32_t v = lds[(m << logC) | (c ^ ((m ^ (m >> 4)) & swz))];
        if (p.cos_out) v = bb_mul(v, bb_coset_factor(p, g >> lgV));   // h^-i * N^-1, fused into the last pass's store
        if (!(LW_DBG(p) & 4)) bb_store_word<OUT64>(gout, g, v);
    };
    if constexpr (RX != 0) {
#pragma unroll
        for (int q = 0; q < BB_TILE / BB_THREADS; q++) store_one(tid + (uint32_t)q * BB_THREADS);   // 32 stores, index arithmetic folded
    } else {
        const uint32_t total = 1u << tile_log;
        for (uint32_t e = tid; e < total; e += BB_THREADS) store_one(e);
    }
}

__global__ void bb_twiddle_fill_kernel(uint32_t *tw, uint32_t root, uint32_t bits, uint64_t count) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count) return;
    tw[g] = bb_pow(root, bb_bitrev((uint32_t)g, bits));
}

// composite twiddles of two fused stages (bb_item): dd[g] = (T[2g] * T[g], -T[2g+1] * T[g]), g < count / 2
__global__ void bb_twiddle_pairs_kernel(const uint32_t *tw, uint2 *dd, uint64_t half_count) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= half_count) return;
    dd[g] = make_uint2(bb_mul(tw[2 * g], tw[g]), bb_sub(0u, bb_mul(tw[2 * g + 1], tw[g])));
}

// two-level power tables of a coset offset: lo[j] = h^j (j < 2^hbits), hi[j] = scale * h^(j << hbits) (j < n_hi)
__global__ void bb_power_tables_kernel(uint32_t *lo, uint32_t *hi, uint32_t h, uint32_t hbits, uint32_t n_hi, uint32_t scale) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (1u << hbits)) lo[t] = bb_pow(h, t);
    if (t < n_hi) hi[t] = bb_mul(scale, bb_pow(h, (uint64_t)t << hbits));
}

// ---------------------------------------------------------------- host
static uint32_t bb_host_root(uint32_t order, bool inverse) {   // traits.rs:82-94
    if (order == 0) return BabyBear::ONE;
    uint32_t g = bb_mul(BabyBear::ROOT, BabyBear::R2);
    for (uint32_t i = 0; i < BabyBear::TWO_ADICITY - order; i++) g = bb_mul(g, g);
    if (inverse) g = bb_inv(g);
    return g;
}

static int bb_ensure_twiddles(Context &c, lw_dir_t dir, uint32_t log2n, hipStream_t stream) {
    TwiddleTable &t = c.tw[LW_FIELD_BABYBEAR][dir];
    if (t.valid && t.log_n >= log2n) return LW_OK;
    if (log2n < 1) return LW_OK;
    ExclusiveScope excl(c);   // shared tables: rebuild with every other call out of the library (ntt256.hip ensure_twiddles)
    if (t.valid && t.log_n >= log2n) return LW_OK;
    uint32_t L = log2n < 16 ? 16 : log2n;
    const uint32_t bits = L - 1;
    const uint64_t count = 1ull << bits;
    if (t.buf.ensure(count * 8)) return LW_ERR_ALLOC;   // T[count] | dd[count / 2] (pairs)
    const uint32_t w = bb_host_root(L, dir == LW_DIR_INVERSE);
    uint32_t *T = (uint32_t *)t.buf.p;
    hipLaunchKernelGGL(bb_twiddle_fill_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, stream, T, w, bits, count);
    hipLaunchKernelGGL(bb_twiddle_pairs_kernel, dim3((uint32_t)((count / 2 + 255) / 256)), dim3(256), 0, stream, (const uint32_t *)T,
                       reinterpret_cast<uint2 *>(T + count), count / 2);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipStreamSynchronize(stream), LW_ERR_LAUNCH);
    t.log_n = L;
    t.valid = true;
    return LW_OK;
}

// in_log2 < log2n: low-degree extension of dense blocks of 2^in_log2 coefficients (forward, d_in != d_out)
template <bool W64>
static int bb_run(Context &c, lw_dir_t dir, uint32_t lgV, const void *d_in, void *d_out, uint32_t log2n, uint32_t batch,
                  uint64_t stride_elems, const void *coset, hipStream_t stream, uint32_t in_log2) {
    const uint64_t n = 1ull << log2n;
    const bool lde = in_log2 < log2n;
    const uint32_t skip = lde ? log2n - in_log2 : 0;   // leading stages that only replicate the block
    const uint64_t nwords = n << lgV;
    const uint64_t wbytes = W64 ? 8 : 4;
    uint64_t stride = (stride_elems ? stride_elems : n) << lgV;   // in words
    if (log2n == 0) {
        if (d_in != d_out)
            LW_HIP_CHECK(hipMemcpy2DAsync(d_out, stride * wbytes, d_in, stride * wbytes, nwords * wbytes, batch,
                                          hipMemcpyDeviceToDevice, stream), LW_ERR_LAUNCH);
        return LW_OK;
    }
    int rc = bb_ensure_twiddles(c, dir, log2n, stream);
    if (rc) return rc;
    uint32_t h = 0;
    if (coset) {
        // offset is one base-field element in the layout's word type
        h = W64 ? bb_from_r64(*(const uint64_t *)coset) : *(const uint32_t *)coset;
        if (h == 0) { set_error("coset offset is zero"); return LW_ERR_INV_ZERO; }
    }

    // pass plan: r <= 8 stages per pass; logC columns x components fill the tile
    const uint32_t max_r = 8;
    const uint32_t nstages = log2n - skip;
    int npass = (int)((nstages + max_r - 1) / max_r);
    if (npass < 1) npass = 1;
    const bool need_scratch = npass > 1 || d_in == d_out;
    if (need_scratch && c.scratch.ensure((size_t)nwords * batch * wbytes)) return LW_ERR_ALLOC;
    // coset factors: h^i fused into the first pass's load (forward), h^-i * N^-1 into the last pass's store (inverse) —
    // a separate kernel with one exponentiation per element cost more than the transform (0.59 ms against 0.48 at 4 x 2^24)
    // (a low-degree extension scales the 2^in_log2 coefficients only: zero padding happens after Polynomial::scale)
    const uint32_t cos_len = lde ? in_log2 : log2n;
    const uint32_t cos_hbits = cos_len < 12 ? cos_len : 12, cos_nhi = 1u << (cos_len - cos_hbits);
    uint32_t *cos_lo = nullptr, *cos_hi = nullptr;
    if (coset) {
        if (c.bb_coset.ensure(4 * ((size_t)(1u << cos_hbits) + cos_nhi))) return LW_ERR_ALLOC;
        cos_lo = (uint32_t *)c.bb_coset.p;
        cos_hi = cos_lo + (1u << cos_hbits);
        const bool inv = dir == LW_DIR_INVERSE;
        const uint32_t base_h = inv ? bb_inv(h) : h;
        const uint32_t hi_scale = inv ? bb_inv(bb_mul((uint32_t)(n % BabyBear::P), BabyBear::R2)) : BabyBear::ONE;
        const uint32_t cnt = (1u << cos_hbits) > cos_nhi ? (1u << cos_hbits) : cos_nhi;
        hipLaunchKernelGGL(bb_power_tables_kernel, dim3((cnt + 255) / 256), dim3(256), 0, stream, cos_lo, cos_hi, base_h, cos_hbits, cos_nhi,
                           hi_scale);
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    }

    const void *src = d_in;
    uint64_t src_stride = lde ? ((uint64_t)1 << (in_log2 + lgV)) : stride;
    bool src64 = W64;   // word type of `src`: the layout's in the caller's buffers, u32 in every intermediate this function writes
    if (npass == 1 && src == d_out) {
        LW_HIP_CHECK(hipMemcpy2DAsync(c.scratch.p, nwords * wbytes, d_in, stride * wbytes, nwords * wbytes, batch,
                                      hipMemcpyDeviceToDevice, stream), LW_ERR_LAUNCH);
        src = c.scratch.p;
        src_stride = nwords;
    }
    uint32_t base = nstages / npass, extra = nstages % npass, s = skip;
    for (int i = 0; i < npass; i++) {
        const bool last = (i == npass - 1);
        BbPassParams p{};
        p.tw = (const uint32_t *)c.tw[LW_FIELD_BABYBEAR][dir].buf.p;
        {
            const uint64_t tcount = 1ull << (c.tw[LW_FIELD_BABYBEAR][dir].log_n - 1);   // the cached table may be larger than this transform's
            p.dd = reinterpret_cast<const uint2 *>(p.tw + tcount);
        }
        p.L = log2n;
        p.lgV = lgV;
        p.dbg = ntt_get_debug();
        p.s0 = s;
        p.r = base + ((uint32_t)i < extra ? 1 : 0);
        const uint32_t room = BB_TILE_LOG - p.r;
        const uint32_t avail = last ? (log2n - p.r + lgV) : (log2n + lgV - s - p.r);
        p.logC = room < avail ? room : avail;
        uint32_t nsteps = (p.r + BB_KMAX - 1) / BB_KMAX, left = p.r;
        p.nsteps = nsteps;
        for (uint32_t q = 0; q < nsteps; q++) {
            uint32_t k = (left + (nsteps - q) - 1) / (nsteps - q);
            p.k[q] = k;
            left -= k;
        }
        p.in = src;
        p.in_batch_stride = src_stride;
        p.cos_lo = cos_lo;
        p.cos_hi = cos_hi;
        p.cos_hbits = cos_hbits;
        p.in_mask = (lde && i == 0) ? (uint32_t)(((uint64_t)1 << (in_log2 + lgV)) - 1) : 0xffffffffu;
        p.cos_in = (coset && dir == LW_DIR_FORWARD && i == 0) ? 1u : 0u;
        p.cos_out = (coset && dir == LW_DIR_INVERSE && last) ? 1u : 0u;
        if (last) {
            if (src == d_out) { set_error("internal: last NTT pass would run in place"); return LW_ERR_BAD_ARG; }
            p.out = d_out;
            p.out_batch_stride = stride;
            if (dir == LW_DIR_INVERSE && !coset) {   // with a coset offset N^-1 rides in the hi table
                p.scale = 1;
                p.sc = bb_inv(bb_mul((uint32_t)(n % BabyBear::P), BabyBear::R2));   // FieldElement::from(n).inv()
            }
        } else {
            p.out = c.scratch.p;
            p.out_batch_stride = nwords;
        }
        const uint32_t blocks = 1u << (log2n + lgV - p.r - p.logC);
        dim3 grid(blocks, batch);
        hipEvent_t pe = c.prof_begin(stream);
        // (a 4-columns-per-lane variant of this kernel measured no faster — the pass is bound by butterfly issue and
        // LDS exchange, not by the width of its memory instructions)
        // full-size tiles (every pass of a transform of 2^18 words and more) take the kernels with the tile shape compiled in
        const bool full = p.r >= 6 && p.r <= 8 && p.logC == BB_TILE_LOG - p.r && p.nsteps == 2 && p.k[0] == (p.r >= 7 ? 4u : 3u) &&
                          p.k[1] == p.r - p.k[0] && (lgV == 0 || lgV == 2);
#define LW_BB_LAUNCH_R(LASTV, INV, OUTV, R)                                                                                          \
    do {                                                                                                                            \
        if (lgV == 0) hipLaunchKernelGGL((bb_pass_kernel<LASTV, INV, OUTV, R, 0>), grid, dim3(BB_THREADS), 0, stream, p);            \
        else hipLaunchKernelGGL((bb_pass_kernel<LASTV, INV, OUTV, R, 2>), grid, dim3(BB_THREADS), 0, stream, p);                     \
    } while (0)
#define LW_BB_LAUNCH(LASTV, INV, OUTV)                                                                                              \
    do {                                                                                                                            \
        if (full && p.r == 8) LW_BB_LAUNCH_R(LASTV, INV, OUTV, 8);                                                                  \
        else if (full && p.r == 7) LW_BB_LAUNCH_R(LASTV, INV, OUTV, 7);                                                             \
        else if (full) LW_BB_LAUNCH_R(LASTV, INV, OUTV, 6);                                                                         \
        else hipLaunchKernelGGL((bb_pass_kernel<LASTV, INV, OUTV, 0, 0>), grid, dim3(BB_THREADS), 0, stream, p);                    \
    } while (0)
        if (last) {
            if (src64) LW_BB_LAUNCH(true, W64, W64);
            else LW_BB_LAUNCH(true, false, W64);
        } else {
            if (src64) LW_BB_LAUNCH(false, W64, false);
            else LW_BB_LAUNCH(false, false, false);
        }
#undef LW_BB_LAUNCH
#undef LW_BB_LAUNCH_R
        c.prof_end(last ? "bb_pass_kernel<last>" : "bb_pass_kernel", pe, stream);
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
        src = p.out;
        src_stride = p.out_batch_stride;
        src64 = last ? W64 : false;
        s += p.r;
    }
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

// get_powers_of_primitive_root[_coset] for the BabyBear shapes: out[i] = scale * w^e(i) in the layout's base word
template <bool W64>
__global__ void bb_powers_export_kernel(void *out, uint32_t root, uint32_t scale, uint32_t bitrev, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t e = bitrev ? bb_bitrev((uint32_t)i, bitrev) : i;
    bb_store_word<W64>(out, (uint32_t)i, bb_mul(scale, bb_pow(root, e)));
}
int ntt_bb_gen_powers(lw_layout_t layout, uint32_t order, uint64_t count, uint32_t bitrev, bool inverse, const void *scale, void *d_out,
                      hipStream_t stream) {
    const bool w64 = layout != LW_LAYOUT_BABYBEAR_U32_R32;
    const uint32_t root = bb_host_root(order, inverse);
    const uint32_t sc = scale ? (w64 ? bb_from_r64(*(const uint64_t *)scale) : *(const uint32_t *)scale) : BabyBear::ONE;
    dim3 grid((uint32_t)((count + 255) / 256));
    if (w64) hipLaunchKernelGGL((bb_powers_export_kernel<true>), grid, dim3(256), 0, stream, d_out, root, sc, bitrev, count);
    else hipLaunchKernelGGL((bb_powers_export_kernel<false>), grid, dim3(256), 0, stream, d_out, root, sc, bitrev, count);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    LW_HIP_CHECK(hipStreamSynchronize(stream), LW_ERR_LAUNCH);
    return LW_OK;
}

const uint32_t *ntt_bb_twiddle_table(Context &c, lw_dir_t dir, uint32_t log2n, hipStream_t stream, int *rc) {
    *rc = bb_ensure_twiddles(c, dir, log2n, stream);
    return (const uint32_t *)c.tw[LW_FIELD_BABYBEAR][dir].buf.p;
}
uint32_t ntt_bb_root(uint32_t order, bool inverse) { return bb_host_root(order, inverse); }

int ntt_bb_device(Context &c, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out, uint32_t log2n, uint32_t batch,
                  uint64_t stride, const void *coset_offset, hipStream_t stream, uint32_t in_log2) {
    if (in_log2 > log2n || in_log2 == 0) in_log2 = log2n;   // (in_log2 == 0 < log2n never arrives: ntt_device_locked broadcasts a constant)
    switch (layout) {
        case LW_LAYOUT_BABYBEAR_U32_R32: return bb_run<false>(c, dir, 0, d_in, d_out, log2n, batch, stride, coset_offset, stream, in_log2);
        case LW_LAYOUT_BABYBEAR_U64_R64: return bb_run<true>(c, dir, 0, d_in, d_out, log2n, batch, stride, coset_offset, stream, in_log2);
        case LW_LAYOUT_EXT4_INTERLEAVED: return bb_run<true>(c, dir, 2, d_in, d_out, log2n, batch, stride, coset_offset, stream, in_log2);
        default: set_error("bad BabyBear layout %d", (int)layout); return LW_ERR_BAD_ARG;
    }
}

}  // namespace lw
