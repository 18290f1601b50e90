// Radix-2 NTT passes for 256-bit Montgomery fields (Stark252, BLS12-381 Fr) on gfx950.
//
// Dataflow = the reference's NR decimation-in-time transform (math/src/fft/cpu/fft.rs:20-55) followed by
// the bit-reverse permutation (math/src/fft/cpu/bit_reversing.rs:2-18): stage s pairs elements N/2^(s+1)
// apart and group g uses twiddle T[g] = w^bitrev(g) (math/src/fft/cpu/roots_of_unity.rs:26-45).  Because
// every field op returns the canonical residue, results are byte-identical to the reference however
// the stages are scheduled; here they are scheduled for the GPU:
//
//   * log2(N) stages are cut into passes of r <= 8 stages (the staged twiddles ltw[2][256] and the 17p lazy bound
//     both need r <= 8; ntt_set_max_pass_stages enforces it).  One workgroup owns a tile of 2^r "rows"
//     (the index bits the pass's stages touch) x C adjacent "columns" (contiguous elements), stages it in
//     LDS once and runs all r stages there: a pass costs one HBM read + one HBM write of the vector,
//     versus one round trip per stage in the reference's CUDA path (math/src/fft/gpu/cuda/ops.rs:28-38).
//   * inside a pass a work-item keeps 2^k (k <= 3) elements in VGPRs and runs k stages register-only
//     (radix-8 = 12 Montgomery products per 8 elements), exchanging through LDS between groups of k stages.
//   * the last pass folds the bit-reverse permutation into its store addresses: a workgroup takes the C
//     tiles whose bit-reversed tile ids are consecutive, so natural-order output leaves as C*32-byte runs.
//   * the INTT's N^-1 scaling (math/src/fft/polynomial.rs:172-173) is fused into the last pass.
#pragma once
#include "field.cuh"

namespace lw {

// Kernel geometry: a tile of 2^TILE_LOG elements (32 B each) in LDS per workgroup, THREADS threads, and at
// most KMAX stages per register step (2^KMAX elements per work-item).
template <int TILE_LOG_, int THREADS_, int KMAX_>
struct NttCfg {
    static constexpr int TILE_LOG = TILE_LOG_;
    static constexpr int TILE = 1 << TILE_LOG_;
    static constexpr int THREADS = THREADS_;
    static constexpr int KMAX = KMAX_;
    // workgroups per CU by LDS, waves per SIMD that follow (launch bound for the register allocator)
    static constexpr int WG_PER_CU = (160 * 1024) / (TILE * 32 + 8192) > 8 ? 8 : (160 * 1024) / (TILE * 32 + 8192);   // + staged twiddles
    static constexpr int WAVES_PER_SIMD = (WG_PER_CU * THREADS / 256) > 8 ? 8 : (WG_PER_CU * THREADS / 256) < 1 ? 1 : (WG_PER_CU * THREADS / 256);
};
using NttCfgA = NttCfg<11, 512, 2>;   // 64 KiB tile, 2 workgroups/CU, 4 waves/SIMD
// (measured and dropped: 32 KiB tile / 256 threads -> 128-byte runs, 5.4 G elem/s; 64 KiB / 256 threads / radix-8 steps
//  -> 2 waves/SIMD, 7.4 G elem/s, against 9.6 for this geometry at the time)
constexpr int NTT_MAX_TILE_LOG = 11;

struct NttPassParams {
    const uint4 *in;       // element e = in[2e], in[2e+1] (reference memory layout)
    uint4 *out;
    const uint4 *tw;       // bit-reversed twiddle table, internal layout (8 x u32, LS limb first)
    uint64_t in_batch_stride;   // elements between consecutive transforms of a batch
    uint64_t out_batch_stride;
    uint32_t L;            // log2 N
    uint32_t s0;           // first stage of this pass
    uint32_t r;            // stages in this pass
    uint32_t logC;         // log2 columns per tile
    uint32_t nsteps;
    uint32_t k[8];         // stages per register step, sum = r
    const uint4 *cos_lo, *cos_hi;   // coset powers h^e = cos_lo[e & mask] * cos_hi[e >> cos_hbits] (internal layout), or null
    uint32_t cos_hbits;
    uint32_t cos_in;       // first pass: multiply element e by h^e on load (Polynomial::scale, polynomial/mod.rs:259-271)
    uint32_t cos_out;      // last pass of an inverse transform: multiply natural output i by h^-i * N^-1 (folded into cos_hi)
    uint64_t in_mask;      // first pass of a low-degree extension: element g is read from in[g & in_mask] (see ntt256.hip)
    uint32_t lazy_in;      // input of this pass may be non-canonical (< 24p): a previous lazy pass wrote it
    uint32_t dbg;          // ablation builds only (-DLW_HIP_ABLATION, see LW_DBG): bit0 skip butterflies, bit1 skip global loads, bit2 skip global stores
    uint32_t wave_sync;    // WL kernels: bit s set = the exchange before register step s stays inside a wavefront (no workgroup barrier)
    uint32_t scale;        // multiply outputs by sc (last pass of an inverse transform)
    uint32_t sc[8];
};

__device__ __forceinline__ uint32_t bitrev_bits(uint32_t x, uint32_t bits) {
    return bits ? (__brev(x) >> (32 - bits)) : 0u;
}

template <class F>
__device__ __forceinline__ Fe<F> tw_load(const uint4 *tw, uint64_t g) {
    uint4 a = tw[2 * g], b = tw[2 * g + 1];
    Fe<F> r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
template <class F>
__device__ __forceinline__ void tw_store(uint4 *tw, uint64_t g, const Fe<F> &x) {
    tw[2 * g] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    tw[2 * g + 1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

// element in reference memory layout (two 16-byte halves) <-> limbs; pure register renaming
template <class F>
__device__ __forceinline__ Fe<F> unpack_mem(uint4 q0, uint4 q1) {
    Fe<F> r;
    r.v[6] = q0.x; r.v[7] = q0.y; r.v[4] = q0.z; r.v[5] = q0.w;
    r.v[2] = q1.x; r.v[3] = q1.y; r.v[0] = q1.z; r.v[1] = q1.w;
    return r;
}
template <class F>
__device__ __forceinline__ void pack_mem(const Fe<F> &a, uint4 &q0, uint4 &q1) {
    q0 = make_uint4(a.v[6], a.v[7], a.v[4], a.v[5]);
    q1 = make_uint4(a.v[2], a.v[3], a.v[0], a.v[1]);
}

// LDS slot (16-byte units within a plane) of tile element (row m, column c).
//   plain layout  (WL = false): rows of C columns, (m << logC) | c — work-items walk columns fastest.
//   column layout (WL = true):  every column's 2^r rows are contiguous and work-items walk ROWS fastest, so that a
//       column's butterflies of all stages belong to one wavefront (64 work-items x 4 elements = 256 rows) and the
//       exchanges between register steps need no workgroup barrier.  The low four row bits are XOR-ed with row bits 4-5
//       (into both bit pairs) and with the column, which makes every access pattern of the pass conflict-free over the
//       64 banks, 16 lanes at a time: the steps' row sets {0-3}, {0,1,4,5}, {2-5} and the (column, row&1) sets of the
//       coalesced global phases all map bijectively onto the four low slot bits.
template <bool WL>
__device__ __forceinline__ uint32_t lds_slot(uint32_t m, uint32_t c, uint32_t r, uint32_t logC) {
    if (!WL) return (m << logC) | c;
    const uint32_t sw = r >= 4 ? ((5u * ((m >> 4) & 3u)) ^ ((c & 7u) << 1)) : 0u;
    return (c << r) | (m ^ sw);
}
// exchange through LDS between two register steps when producer and consumer lanes share a wavefront: LDS operations
// of one wave execute in order, so only the compiler has to be kept from reordering them
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One work-item: 2^K elements, K stages in registers.
// EXTRA: the pass carries a coset scaling or the N^-1 factor (kept out of the plain transform's code: the last-pass
// kernel is ~60 KiB of straight-line MAC chains and shares a 64 KiB instruction cache with its neighbour CU)
// FX = r: a full-size tile of the 2048-element configuration (r = 8 stages x 8 columns or 6 x 32, r / 2 radix-4 steps, one
// item per work-item and step) with its shape as compile-time constants, so that the shifts, masks, bit reversals and swizzles
// of the index arithmetic fold (every pass of a 2^24 transform, the last pass from 2^16 on).
template <class F, int K, bool LAST, int TILE, bool EXTRA, bool WL, int FX = 0>
__device__ __forceinline__ void ntt_item(const NttPassParams &p, uint4 (*lds)[TILE], uint4 (*ltw)[256], const uint4 *gin,
                                         uint32_t w, uint32_t step, uint32_t t0, uint64_t base, uint32_t lgS,
                                         uint32_t hi_uniform, uint32_t hi_low, bool last_step, bool stage_tw) {
    constexpr int E = 1 << K;
    const uint32_t r = FX ? (uint32_t)FX : p.r, logC = FX ? (uint32_t)(11 - FX) : p.logC, L = p.L;
    const uint32_t sh = r - t0 - K;
    uint32_t c, mr;
    if ((LAST && step == 0) || (WL && (LAST || step > 0))) {   // rows fastest: contiguous global rows / one column per wave
        mr = w & ((1u << (r - K)) - 1);
        c = w >> (r - K);
    } else {                           // columns fastest
        c = w & ((1u << logC) - 1);
        mr = w >> logC;
    }
    const uint32_t m_low = mr & ((1u << sh) - 1);
    const uint32_t m_high = mr >> sh;
    const uint32_t mbase = (m_high << (sh + K)) | m_low;
    uint32_t hi_c = hi_uniform;
    uint64_t gbase = base;
    if (LAST) {
        hi_c = (bitrev_bits(c, logC) << (L - r - logC)) | hi_low;
        gbase = (uint64_t)hi_c << r;
    }

    // Non-last passes stage the tile's 2^r - 1 twiddles in LDS (stage t, group x -> slot 2^t - 1 + x holds
    // T[(hi << t) | x], shared by every column).  Their loads are issued first and the data loads right behind
    // them, so the workgroup pays one memory latency, not two, before its first butterfly.
    uint4 tq0, tq1;
    const uint32_t ti = threadIdx.x;
    if (!LAST && !WL && stage_tw && ti + 1 < (1u << r)) {
        const uint32_t t = 31 - __clz(ti + 1), xg = ti + 1 - (1u << t);
        const uint64_t g = ((uint64_t)hi_uniform << t) | xg;
        tq0 = p.tw[2 * g];
        tq1 = p.tw[2 * g + 1];
    }

    Fe<F> x[E];
    if (step == 0) {   // branch hoisted out of the element loop: the 2E global loads are issued back to back
#pragma unroll
        for (int j = 0; j < E; j++) {
            const uint32_t m = mbase | ((uint32_t)j << sh);
            const uint64_t g = (LAST ? (gbase + m) : (gbase + ((uint64_t)m << lgS) + c)) & p.in_mask;
            uint4 q0, q1;
            if (LW_DBG(p) & 2) {
                q0 = make_uint4(m, c, 1, 2);
                q1 = make_uint4(3, 4, 5, 6);
            } else {
                q0 = gin[2 * g];
                q1 = gin[2 * g + 1];
            }
            x[j] = unpack_mem<F>(q0, q1);
        }
        if (EXTRA && p.cos_in) {   // evaluate_offset_fft: c_e * h^e, fused into the first pass's load
#pragma unroll
            for (int j = 0; j < E; j++) {
                const uint32_t m = mbase | ((uint32_t)j << sh);
                const uint64_t e = (LAST ? (gbase + m) : (gbase + ((uint64_t)m << lgS) + c)) & p.in_mask;
                Fe<F> pw = fe_mul<F>(tw_load<F>(p.cos_lo, e & ((1ull << p.cos_hbits) - 1)), tw_load<F>(p.cos_hi, e >> p.cos_hbits));
                x[j] = fe_mul<F>(x[j], pw);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; j++) {
            const uint32_t idx = lds_slot<WL>(mbase | ((uint32_t)j << sh), c, r, logC);
            x[j] = unpack_mem<F>(lds[0][idx], lds[1][idx ^ (WL ? 1u : 0u)]);
        }
    }

    if (!LAST && !WL && stage_tw) {   // uniform across the workgroup (see the kernel)
        if (ti + 1 < (1u << r)) {
            ltw[0][ti] = tq0;
            ltw[1][ti] = tq1;
        }
        __syncthreads();
    }

    // Lazy reduction (fields with 4+ spare bits, F::LAZY): values ride in [0, 17p) — a butterfly is
    //   t = w*y in [0,2p) (no final subtraction), x' = x + t, y' = x + 2p - t, so the bound grows by 2p per stage;
    // a pass has <= 8 stages and fully reduces its input on load (< p, so < 17p at its end); the last pass canonicalises on exit.
    // Every result is still the unique canonical residue when it leaves the transform, so parity is unaffected.
    if (F::LAZY && step == 0 && p.lazy_in) {
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = fe_reduce_full(x[j]);
    }

    // stage u of this step == stage s0 + t0 + u of the transform.  The E-1 twiddle groups of the step are walked in
    // order q = 2^u - 1 + jt; group q+1's twiddle is fetched (LDS, or the table in the last pass) before group q's
    // butterflies run, so its latency hides behind a Montgomery product (the sched_barrier below pins it there).
    // Fetching all of a step's twiddles up front measured slower (more live registers, no fewer stalls).
    auto fetch_tw = [&](int q) -> Fe<F> {
        const int u = 31 - __builtin_clz(q + 1), jt = q + 1 - (1 << u);
        Fe<F> tw;
        if (LAST || WL) {   // from the table (L1/L2): per-lane twiddles would cost LDS bandwidth the exchanges need
            const uint64_t gt = ((uint64_t)hi_c << (t0 + u)) | ((uint64_t)m_high << u);
            tw = tw_load<F>(p.tw, (LW_DBG(p) & 32) ? ((gt | (uint32_t)jt) & 0xff) : (gt | (uint32_t)jt));
        } else {   // slot 2^t - 1 + x of the staged table
            const uint32_t li = (1u << (t0 + u)) - 1 + ((m_high << u) | (uint32_t)jt);
            uint4 a = ltw[0][li], b = ltw[1][li];
            tw.v[0] = a.x; tw.v[1] = a.y; tw.v[2] = a.z; tw.v[3] = a.w;
            tw.v[4] = b.x; tw.v[5] = b.y; tw.v[6] = b.z; tw.v[7] = b.w;
        }
        return tw;
    };
    if (!(LW_DBG(p) & 1)) {
        Fe<F> tw_next = fetch_tw(0);
#pragma unroll
        for (int q = 0; q < E - 1; q++) {
            const int u = 31 - __builtin_clz(q + 1), jt = q + 1 - (1 << u);
            const int half = 1 << (K - 1 - u);
            const Fe<F> tw = tw_next;
            if (q + 1 < E - 1) tw_next = fetch_tw(q + 1);
            // T[0] = 1: the first group of every stage multiplies by one (2^-t of stage t's butterflies, i.e. a
            // quarter of the first pass's products).  The reference multiplies anyway (fft.rs:40-43); the product
            // by the Montgomery one is the identity on canonical residues, so skipping it changes no byte.
            const bool unit = (jt == 0) && (hi_c == 0) && (m_high == 0);
#pragma unroll
            for (int jl = 0; jl < half; jl++) {
                const int j = (jt << (K - u)) | jl;
                if (F::LAZY) {
                    Fe<F> wb = unit ? fe_reduce_full(x[j + half]) : fe_mul_lazy<F>(tw, x[j + half]);
                    Fe<F> a = x[j];
                    x[j] = fe_add_raw<F>(a, wb);
                    x[j + half] = fe_add2p_sub_raw<F>(a, wb);
                } else {
                    Fe<F> wb = unit ? x[j + half] : fe_mul<F>(tw, x[j + half]);
                    Fe<F> a = x[j];
                    x[j] = fe_add<F>(a, wb);
                    x[j + half] = fe_sub<F>(a, wb);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep butterflies serial: interleaved products cost too many VGPRs
            }
        }
    }

    if (EXTRA && LAST && last_step && p.cos_out) {   // interpolate_offset_fft: N^-1 and h^-i in one product
#pragma unroll
        for (int j = 0; j < E; j++) {
            const uint32_t m = mbase | ((uint32_t)j << sh);
            const uint64_t i_nat = ((uint64_t)bitrev_bits(m, r) << (L - r)) + ((uint64_t)blockIdx.x << logC) + c;
            Fe<F> pw = fe_mul<F>(tw_load<F>(p.cos_lo, i_nat & ((1ull << p.cos_hbits) - 1)), tw_load<F>(p.cos_hi, i_nat >> p.cos_hbits));
            if (F::LAZY) x[j] = fe_cond_sub_kp<F, 0>(fe_mul_lazy<F>(pw, x[j]));
            else x[j] = fe_mul<F>(x[j], pw);
        }
    } else if (EXTRA && last_step && p.scale) {
        Fe<F> sc;
#pragma unroll
        for (int i = 0; i < 8; i++) sc.v[i] = p.sc[i];
#pragma unroll
        for (int j = 0; j < E; j++) {
            if (F::LAZY) x[j] = fe_cond_sub_kp<F, 0>(fe_mul_lazy<F>(sc, x[j]));   // N^-1 < p: product in [0,2p)
            else x[j] = fe_mul<F>(x[j], sc);
        }
    } else if (F::LAZY && LAST && last_step) {
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = fe_reduce_full(x[j]);
    }

#pragma unroll
    for (int j = 0; j < E; j++) {
        const uint32_t idx = lds_slot<WL>(mbase | ((uint32_t)j << sh), c, r, logC);
        uint4 q0, q1;
        pack_mem<F>(x[j], q0, q1);
        lds[0][idx] = q0;
        lds[1][idx ^ (WL ? 1u : 0u)] = q1;   // the planes are offset by one slot so that plane-interleaved reads spread too
    }
}

template <class F, bool LAST, class CFG, bool EXTRA, bool WL, int FX = 0>
__global__ __launch_bounds__(CFG::THREADS, CFG::WAVES_PER_SIMD) void ntt_pass_kernel(NttPassParams p) {
    constexpr int NTT_THREADS = CFG::THREADS;
    constexpr int NTT_KMAX = CFG::KMAX;
    constexpr int NTT_TILE = CFG::TILE;
    __shared__ uint4 lds[2][NTT_TILE];
    __shared__ uint4 ltw[LAST ? 1 : 2][LAST ? 1 : 256];   // non-last passes: the tile's twiddles (<= 255 x 32 B)
    const uint32_t tid = threadIdx.x;
    static_assert(!FX || (NTT_TILE == 2048 && NTT_THREADS == 512 && (FX == 8 || FX == 7 || FX == 6)), "FX: 2^8 x 8, 2^7 x 16 or 2^6 x 32 rows x columns");
    const uint32_t r = FX ? (uint32_t)FX : p.r, logC = FX ? (uint32_t)(11 - FX) : p.logC, L = p.L;
    const uint32_t tile_log = r + logC;
    const uint4 *gin = p.in + 2 * (uint64_t)blockIdx.y * p.in_batch_stride;
    uint4 *gout = p.out + 2 * (uint64_t)blockIdx.y * p.out_batch_stride;
    const uint32_t b = blockIdx.x;

    uint64_t base = 0;
    uint32_t lgS = 0, hi_uniform = 0, hi_low = 0;
    if (!LAST) {
        lgS = L - p.s0 - r;                       // log2 of the row stride
        const uint32_t lo_bits = lgS - logC;      // column blocks per `hi`
        const uint32_t lo_blk = b & ((1u << lo_bits) - 1);
        hi_uniform = b >> lo_bits;
        base = ((uint64_t)hi_uniform << (L - p.s0)) + ((uint64_t)lo_blk << logC);
    } else {
        hi_low = bitrev_bits(b, L - r - logC);
    }

    // twiddle staging happens inside the first register step when every thread runs it (the usual case);
    // tiles with fewer items than threads (small transforms) stage up front
    const bool stage_inside = FX ? (!LAST && !WL)
                                 : (!LAST && !WL && (1u << (tile_log - p.k[0])) >= (uint32_t)NTT_THREADS && (1u << r) <= (uint32_t)NTT_THREADS);
    if (!LAST && !WL && !stage_inside) {
        // stage t of the pass uses T[(hi << t) | x], x < 2^t, shared by every column of the tile
        for (uint32_t i = tid; i + 1 < (1u << r); i += NTT_THREADS) {
            const uint32_t t = 31 - __clz(i + 1), x = i + 1 - (1u << t);
            const uint64_t g = ((uint64_t)hi_uniform << t) | x;
            ltw[0][i] = p.tw[2 * g];
            ltw[1][i] = p.tw[2 * g + 1];
        }
        __syncthreads();
    }
    if constexpr (FX) {
#define LW_FX_STEP(S)                                                                                                                  \
    do {                                                                                                                                \
        if (S) {                                                                                                                        \
            if (WL && ((p.wave_sync >> (S)) & 1u)) lds_wave_sync();                                                                     \
            else __syncthreads();                                                                                                       \
        }                                                                                                                               \
        ntt_item<F, 2, LAST, NTT_TILE, EXTRA, WL, FX>(p, lds, (uint4 (*)[256])ltw, gin, tid, (S), 2u * (S), base, lgS, hi_uniform,      \
                                                      hi_low, 2 * ((S) + 1) == FX, stage_inside && (S) == 0);                           \
    } while (0)
        LW_FX_STEP(0u);
        LW_FX_STEP(1u);
        LW_FX_STEP(2u);
        if constexpr (FX == 8) LW_FX_STEP(3u);
        if constexpr (FX == 7) {   // the odd stage: a radix-2 step, two items per work-item
            __syncthreads();
            ntt_item<F, 1, LAST, NTT_TILE, EXTRA, WL, FX>(p, lds, (uint4 (*)[256])ltw, gin, tid, 3u, 6u, base, lgS, hi_uniform, hi_low, true, false);
            ntt_item<F, 1, LAST, NTT_TILE, EXTRA, WL, FX>(p, lds, (uint4 (*)[256])ltw, gin, tid + NTT_THREADS, 3u, 6u, base, lgS, hi_uniform, hi_low, true,
                                                          false);
        }
#undef LW_FX_STEP
    } else {
    uint32_t t0 = 0;
    for (uint32_t step = 0; step < p.nsteps; step++) {
        const uint32_t k = p.k[step];
        const uint32_t nitems = 1u << (tile_log - k);
        const bool last_step = (step + 1 == p.nsteps);
        if (step) {
            if (WL && ((p.wave_sync >> step) & 1u)) lds_wave_sync();
            else __syncthreads();
        }
        for (uint32_t w = tid; w < nitems; w += NTT_THREADS) {
            if (NTT_KMAX >= 3 && k == 3) ntt_item<F, (NTT_KMAX >= 3 ? 3 : 1), LAST, NTT_TILE, EXTRA, WL>(p, lds, (uint4 (*)[256])ltw, gin, w, step, t0, base, lgS, hi_uniform, hi_low, last_step, stage_inside && step == 0 && w == tid);
            else if (k == 2) ntt_item<F, 2, LAST, NTT_TILE, EXTRA, WL>(p, lds, (uint4 (*)[256])ltw, gin, w, step, t0, base, lgS, hi_uniform, hi_low, last_step, stage_inside && step == 0 && w == tid);
            else ntt_item<F, 1, LAST, NTT_TILE, EXTRA, WL>(p, lds, (uint4 (*)[256])ltw, gin, w, step, t0, base, lgS, hi_uniform, hi_low, last_step, stage_inside && step == 0 && w == tid);
        }
        t0 += k;
    }
    }
    __syncthreads();

    // coalesced write-out: two lanes per element, 16 B each
    auto write_one = [&](uint32_t f) {
        const uint32_t e = f >> 1, plane = f & 1;
        const uint32_t c = e & ((1u << logC) - 1);
        const uint32_t m = e >> logC;
        uint64_t g;
        if (!LAST) g = base + ((uint64_t)m << lgS) + c;
        else g = ((uint64_t)bitrev_bits(m, r) << (L - r)) + ((uint64_t)b << logC) + c;
        if (!(LW_DBG(p) & 4)) gout[2 * g + plane] = lds[plane][lds_slot<WL>(m, c, r, logC) ^ (WL ? plane : 0u)];
    };
    if constexpr (FX) {
#pragma unroll
        for (int q = 0; q < 2 * NTT_TILE / NTT_THREADS; q++) write_one(tid + (uint32_t)q * NTT_THREADS);
    } else {
        const uint32_t total = 2u << tile_log;
        for (uint32_t f = tid; f < total; f += NTT_THREADS) write_one(f);
    }
}

// T[g] = w^bitrev_{bits}(g), from two small power tables: w^e = lo[e & mask] * hi[e >> hbits]
template <class F>
__global__ void twiddle_fill_kernel(uint4 *tw, const uint4 *lo, const uint4 *hi, uint32_t bits, uint32_t hbits,
                                    uint64_t count) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count) return;
    uint32_t e = bitrev_bits((uint32_t)g, bits);
    Fe<F> a = tw_load<F>(lo, e & ((1u << hbits) - 1));
    Fe<F> b = tw_load<F>(hi, e >> hbits);
    tw_store<F>(tw, g, fe_mul<F>(a, b));
}

// The two power tables of a coset offset on the device: lo[i] = base^i (i < 2^hbits), hi[j] = scale * base^(j * 2^hbits)
// (j < hi_count), one square-and-multiply per entry — a few thousand work-items, nothing uploaded, nothing to wait for.
// (Built on the host and uploaded, a new offset cost a stream synchronisation, ~3000 serial products and two copies:
// ~110 us per call, every layer of a FRI commit phase.)
struct FeWords8 {
    uint32_t w[8];
};
template <class F>
__global__ void power_tables_kernel(uint4 *lo, uint4 *hi, FeWords8 base, uint32_t hbits, uint64_t hi_count, FeWords8 scale, int has_scale) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo_count = 1ull << hbits;
    if (t >= lo_count + hi_count) return;
    const bool is_lo = t < lo_count;
    const uint64_t e = is_lo ? t : ((t - lo_count) << hbits);
    Fe<F> b, r = Fe<F>::one();
#pragma unroll
    for (int k = 0; k < 8; k++) b.v[k] = base.w[k];
    for (int bit = 63 - (e ? __clzll((long long)e) : 63); bit >= 0; bit--) {
        r = fe_sqr<F>(r);
        if ((e >> bit) & 1) r = fe_mul<F>(r, b);
    }
    if (!is_lo && has_scale) {
        Fe<F> sc;
#pragma unroll
        for (int k = 0; k < 8; k++) sc.v[k] = scale.w[k];
        r = fe_mul<F>(r, sc);
    }
    if (is_lo) tw_store<F>(lo, t, r);
    else tw_store<F>(hi, t - lo_count, r);
}

}  // namespace lw
