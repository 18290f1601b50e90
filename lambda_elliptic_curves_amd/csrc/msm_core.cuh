// Curve-generic part of the Pippenger MSM (see msm.hip for the schedule); instantiated once per curve in its own
// translation unit so the four groups compile in parallel.
#pragma once
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "context.h"
#include "ec.cuh"

namespace lw {

uint32_t msm_ch(uint64_t items);       // max points per accumulate work-item (a bucket is cut into equal pieces <= CH)
int msm_piece_order_enabled();         // LW_HIP_MSM_ORDER=0: work-items take their pieces in key order (A/B)
uint64_t msm_quad_max_lanes();          // LW_HIP_MSM_QUAD: levels of the bucket reduce with at most this many lanes (8 per group) spread each addition over a quad; 0 = never, ~0 = not set
uint64_t msm_accumulate_quad_max_lanes();   // LW_HIP_MSM_ACCQ: accumulate launches of projective rows with at most this many lanes (4 per piece) use the quad kernel
uint32_t msm_g_log();                  // log2 buckets per running-sum group: 3 (8 buckets; 16 -> 8 saved 1 ms of dependent-add latency per MSM, 4 is no better)
constexpr int MSM_THREADS = 128;

// host launchers for the curve-independent kernels (defined in msm.hip)
uint32_t msm_sort_coarse_bins(uint32_t c, uint32_t W, uint64_t n);
uint32_t msm_max_window_bits();
uint64_t msm_sort_padded_points(uint64_t n);
void msm_launch_digits(Context &c, const uint32_t *scalars, uint64_t n, uint32_t cb, uint32_t W, uint32_t *dig, hipStream_t s);
void msm_launch_sort(Context &c, const uint32_t *dig, uint64_t n, uint32_t cb, uint32_t W, uint32_t *coarse_cnt,
                     uint32_t *coarse_off, uint32_t *coarse_cursor, uint64_t *items, uint32_t *sorted, uint32_t *off, uint32_t K,
                     uint32_t *maxlen, uint32_t *scan_tmp, uint32_t *sub_off, uint32_t *key_cnt, uint32_t *key_cursor, uint64_t fold_stride,
                     uint64_t win0, hipStream_t s);
int ensure_aux_stream(Context &c);   // msm.hip: the context's side stream
void msm_launch_scan(const uint32_t *in, uint32_t *out, uint32_t K, int mode, uint32_t *maxlen, uint32_t *scratch, hipStream_t s);
size_t msm_scan_scratch_bytes(uint32_t K);
void msm_launch_piece_order(Context &c, const uint32_t *seg_off, const uint32_t *out_off, uint32_t K, uint32_t P, uint32_t *order_tmp,
                            uint32_t *perm_t, uint32_t *perm_key, hipStream_t s);
size_t msm_order_tmp_bytes();
int msm_waves_per_simd();   // LW_HIP_MSM_WAVES (2 or 3): register budget of the accumulate kernel

// ---------------------------------------------------------------- accumulate
// All device point arrays (caller's points, partial sums, buckets, running-sum temporaries) use the reference
// memory layout, so one load path serves every round.
template <class C>
__device__ __forceinline__ Point<C> pt_ld(const void *base, size_t i) { return pt_load<C>((const char *)base + i * (3 * C::B::BYTES)); }
template <class C>
__device__ __forceinline__ void pt_st(void *base, size_t i, const Point<C> &p) { pt_store<C>((char *)base + i * (3 * C::B::BYTES), p); }

// Work-item t sums <= CH items of ONE key.
//   out_off != nullptr: t is a (key, piece) pair (key from perm_key or by binary search in out_off).  A key with ONE piece
//                       is finished by this work-item: its sum goes to buckets[key]; the pieces of a longer key go to
//                       pout[t] and are summed by the next round.
//   out_off == nullptr: every key has <= CH items; t is the key; result (identity when the key is empty) -> buckets[key],
//                       the dense bucket array.
//   later rounds (index == nullptr) skip keys whose segment is a single partial: an earlier round finished them.
// index != nullptr: item i is the point +-pts[index[i] & 0x7fffffff], negated when bit 31 is set (first round: the caller's
// points through the sorted list of signed digits); otherwise item i is pts[i] (partial sums of the previous round).
// AFFINE: the rows of `pts` are affine pairs of a pre-normalised SRS (lw_hip_srs_*): 2/3 of the bytes per gather and
// the mixed addition (19 N^2 MACs instead of 21 N^2).  Only first-round launches (index != nullptr) use it.
template <class C, int WAVES, bool AFFINE>
__global__ __launch_bounds__(MSM_THREADS, WAVES) void msm_accumulate_kernel(const void *pts, const uint32_t *index,
                                                                             const uint32_t *seg_off, const uint32_t *out_off,
                                                                             const uint32_t *perm_t, const uint32_t *perm_key,
                                                                             uint32_t K, uint32_t total_items, void *pout, void *buckets) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total_items) return;
    // perm_t: pieces in descending order of length (msm_piece_order_kernel), so the lanes of a wave run equally long
    const uint32_t t = perm_t ? perm_t[r] : r;
    uint32_t b, e;
    void *dst = buckets;
    size_t slot = t;
    if (out_off) {
        uint32_t lo = 0, hi = K;   // largest key with out_off[key] <= t
        if (perm_key) {
            lo = perm_key[r];
        } else {
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (out_off[mid] <= t) lo = mid; else hi = mid;
            }
        }
        // the key's items are cut into np = ceil(len / CH) pieces of equal length (+-1), so the lanes of a wave run
        // the same number of additions instead of full pieces next to a short remainder
        const uint32_t s0 = seg_off[lo], len = seg_off[lo + 1] - s0, np = out_off[lo + 1] - out_off[lo], j = t - out_off[lo];
        b = s0 + (uint32_t)(((uint64_t)len * j) / np);
        e = s0 + (uint32_t)(((uint64_t)len * (j + 1)) / np);
        if (!index && len <= 1) return;
        if (np == 1) slot = lo; else dst = pout;
    } else {
        b = seg_off[t];
        e = seg_off[t + 1];
        if (!index && e - b <= 1) return;
    }
    // The gather of a point (96-288 B from a random row) is a dependent chain index -> row.  The next index is
    // fetched one addition ahead, and the next row's cache lines are touched (one dword per 128 B, discarded) before
    // the current addition starts, so the real load at the top of the next iteration hits L2.
    constexpr int PW = (AFFINE ? 2 : 3) * C::B::BYTES / 4;   // row payload in dwords
    constexpr size_t ROW = AFFINE ? aff_stride<C>() : (size_t)PW * 4;
    auto row_ptr = [&](uint32_t i) { return (const char *)pts + (size_t)i * ROW; };
    using B = typename C::B;
    constexpr uint32_t IDX = 0x7fffffffu;
    Point<C> acc = pt_identity<C>();
    uint32_t idx_next = b;
    if (b < e) {
        const uint32_t i0 = index ? index[b] : b;
        if constexpr (AFFINE) acc = aff_to_point<C>(aff_load<C>(row_ptr(i0 & IDX)));
        else acc = pt_load<C>(row_ptr(i0 & IDX));
        if (index && (i0 >> 31)) acc.y = B::neg(acc.y);
        if (b + 1 < e) idx_next = index ? index[b + 1] : b + 1;
    }
#pragma nounroll
    for (uint32_t i = b + 1; i < e; i++) {
        const char *cur = row_ptr(idx_next & IDX);
        const bool neg = index && (idx_next >> 31);
        uint32_t touch0 = 0, touch1 = 0, touch2 = 0;
        if constexpr (AFFINE) {
            AffPoint<C> q = aff_load<C>(cur);
            q.y = B::select(neg, B::neg(q.y), q.y);   // -0 = 0: the identity row (0, 0) stays the identity
            if (i + 1 < e) {
                idx_next = index ? index[i + 1] : i + 1;
                const uint32_t *row = reinterpret_cast<const uint32_t *>(row_ptr(idx_next & IDX));
                touch0 = row[0];
                if constexpr (!(ROW % 128 == 0 && PW * 4 <= 128)) {   // a padded 128-byte row is one line: one touch
                    touch1 = row[32 < PW ? 32 : 0];
                    touch2 = row[PW - 1];
                }
            }
            if (!aff_is_identity<C>(q)) acc = pt_add_mixed<C>(acc, q);
        } else {
            Point<C> q = pt_load<C>(cur);
            q.y = B::select(neg, B::neg(q.y), q.y);
            if (i + 1 < e) {
                idx_next = index ? index[i + 1] : i + 1;
                const uint32_t *row = reinterpret_cast<const uint32_t *>(row_ptr(idx_next & IDX));
                touch0 = row[0];
                touch1 = row[32 < PW ? 32 : 0];
                touch2 = row[PW - 1];
            }
            acc = pt_add<C>(acc, q);
        }
        asm volatile("" ::"v"(touch0), "v"(touch1), "v"(touch2));   // consume the touches after the MACs
    }
    pt_st<C>(dst, slot, acc);
}

// The same work-items on four lanes each: when the pieces do not fill the machine a work-item's
// chain of dependent additions IS the kernel time, so every addition is spread over a quad (pt_add_quad, ec.cuh): lane s
// gathers, keeps and stores coordinate s only; the next row is fetched one addition ahead.  Same (key, piece) logic, same
// results up to the projective representative.
template <class C, bool AFFINE>
__global__ __launch_bounds__(MSM_THREADS) void msm_accumulate_quad_kernel(const void *pts, const uint32_t *index, const uint32_t *seg_off,
                                                                           const uint32_t *out_off, const uint32_t *perm_t,
                                                                           const uint32_t *perm_key, uint32_t K, uint32_t total_items, void *pout,
                                                                           void *buckets) {
    using B = typename C::B;
    using T = typename B::T;
    constexpr size_t ROW = AFFINE ? aff_stride<C>() : 3 * B::BYTES;   // AFFINE: rows of a pre-normalised SRS (first round only)
    constexpr uint32_t IDX = 0x7fffffffu;
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = gt >> 2;
    if (r >= total_items) return;                // quads are never split
    const uint32_t s = (gt & 3) == 3 ? 0u : (gt & 3);
    const uint32_t t = perm_t ? perm_t[r] : r;
    uint32_t b, e;
    void *dst = buckets;
    size_t slot = t;
    if (out_off) {
        uint32_t lo = 0, hi = K;
        if (perm_key) {
            lo = perm_key[r];
        } else {
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (out_off[mid] <= t) lo = mid; else hi = mid;
            }
        }
        const uint32_t s0 = seg_off[lo], len = seg_off[lo + 1] - s0, np = out_off[lo + 1] - out_off[lo], j = t - out_off[lo];
        b = s0 + (uint32_t)(((uint64_t)len * j) / np);
        e = s0 + (uint32_t)(((uint64_t)len * (j + 1)) / np);
        if (!index && len <= 1) return;
        if (np == 1) slot = lo; else dst = pout;
    } else {
        b = seg_off[t];
        e = seg_off[t + 1];
        if (!index && e - b <= 1) return;
    }
    const bool ylane = s == 1;
    auto fetch = [&](uint32_t i, bool &neg) {
        const uint32_t ix = index ? index[i] : i;
        neg = index && (ix >> 31) && ylane;
        const char *row = (const char *)pts + (size_t)(ix & IDX) * ROW;
        if constexpr (AFFINE) {   // (x, y) with z = 1 implied; the identity row (0, 0) becomes (0 : 1 : 0)
            const T v = s < 2 ? B::load(row + s * B::BYTES) : B::one();
            const int zero = s < 2 && B::is_zero(v);
            const bool ident = __builtin_amdgcn_mov_dpp(zero, 0x00, 0xF, 0xF, true) & __builtin_amdgcn_mov_dpp(zero, 0x55, 0xF, 0xF, true);
            return B::select(ident, ylane ? B::one() : B::zero(), v);
        } else {
            return B::load(row + s * B::BYTES);
        }
    };
    T acc = ylane ? B::one() : B::zero();        // (0 : 1 : 0)
    if (b < e) {
        bool neg;
        acc = fetch(b, neg);
        acc = B::select(neg, B::neg(acc), acc);
        T qn = acc;
        bool negn = false;
        if (b + 1 < e) qn = fetch(b + 1, negn);
#pragma nounroll
        for (uint32_t i = b + 1; i < e; i++) {
            const T q = B::select(negn, B::neg(qn), qn);
            if (i + 1 < e) qn = fetch(i + 1, negn);
            acc = pt_add_quad<C>(acc, q, s);
        }
    }
    if ((gt & 3) != 3) B::store((char *)dst + slot * (3 * B::BYTES) + s * B::BYTES, acc);   // partial sums and buckets are projective rows
}

// SRS preparation: projective rows -> affine pairs with Montgomery's batch inversion (the reference's
// FieldElement::inplace_batch_inverse, field/element.rs:47-65): a work-item walks `chk` points (a strided set, see the kernel), stores the running
// products of their z in a scratch array, inverts the last product once (Fermat), and walks
// back peeling one inverse per point — 3 products per point plus 1/chk of an inversion instead of a full inversion each.
// Identity rows (z = 0) are left out of the product and written as (0, 0).
template <class C>
__global__ __launch_bounds__(MSM_THREADS) void msm_to_affine_kernel(const void *in, uint64_t n, uint32_t chk, void *out, void *prefix) {
    using B = typename C::B;
    using T = typename B::T;
    constexpr size_t PBY = 3 * B::BYTES, ABY = aff_stride<C>();
    // A workgroup owns a CONTIGUOUS block of blockDim * chk points and work-item t of it the points base + t, base + t + S,
    // base + t + 2S, ... with S = blockDim: at every step of the walk the lanes of a wave touch consecutive rows (coalesced),
    // and the whole workgroup stays inside a few MB (round 1 strode over the whole array, S = all work-items of the grid;
    // same speed, worse locality).
    const uint64_t S = blockDim.x;
    const uint64_t base = (uint64_t)blockIdx.x * blockDim.x * chk;
    const uint64_t t = base + threadIdx.x;
    if (t >= n) return;
    uint32_t cnt = 0;
    while (cnt < chk && t + (uint64_t)cnt * S < n) cnt++;
    const char *pin = (const char *)in;
    char *pout = (char *)out;
    char *ppre = (char *)prefix;   // running products, B::BYTES per point: a compact array (whole lines written and read back)
    T acc = B::one();
#pragma nounroll
    for (uint32_t j = 0; j < cnt; j++) {
        const uint64_t i = t + (uint64_t)j * S;
        const T z = B::load(pin + i * PBY + 2 * B::BYTES);
        if (!B::is_zero(z)) acc = B::mul(acc, z);
        B::store(ppre + i * B::BYTES, acc);       // running product through this work-item's j-th point
    }
    T inv = B::inv(acc);                          // acc != 0: a product of non-zero field elements (or one)
    // rows land on the cheaper isomorphic model (L^2 x, L^3 y), see ec.cuh: L^2 rides on the inverse chain (every 1/z_i
    // peeled off below is then L^2 / z_i, one product per work-item), which leaves x * w and (y * w) * L: three products
    // per point instead of four
    T isoL = B::one();
    if constexpr (IsoOf<C>::has) {
        using I = typename IsoOf<C>::type;
        inv = B::mul(inv, I::konst(0));
        isoL = B::mul(I::konst(1), I::konst(2));   // L = L^3 * L^-2
    }
#pragma nounroll
    for (uint32_t j = cnt; j-- > 0;) {
        const uint64_t i = t + (uint64_t)j * S;
        const T z = B::load(pin + i * PBY + 2 * B::BYTES);
        if (B::is_zero(z)) {
            aff_store<C>(pout + i * ABY, AffPoint<C>{B::zero(), B::zero()});
            continue;
        }
        const T prev = j == 0 ? B::one() : B::load(ppre + (i - S) * B::BYTES);
        const T zinv = B::mul(inv, prev);         // 1 / z_i
        inv = B::mul(inv, z);                     // 1 / (running product through the previous point)
        const T x = B::load(pin + i * PBY), y = B::load(pin + i * PBY + B::BYTES);
        if constexpr (IsoOf<C>::has) {   // zinv = L^2 / z_i here
            aff_store<C>(pout + i * ABY, AffPoint<C>{B::mul(x, zinv), B::mul(B::mul(y, zinv), isoL)});
        } else {
            aff_store<C>(pout + i * ABY, AffPoint<C>{B::mul(x, zinv), B::mul(y, zinv)});
        }
    }
}

// Window-shifted copy of an affine point set (folded SRS): out[i] = 2^c * in[i], projective (normalised by the caller).
template <class C>
__global__ __launch_bounds__(MSM_THREADS) void msm_shift_kernel(const void *aff_in, uint64_t n, uint32_t cbits, void *proj_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Point<C> p = aff_to_point<C>(aff_load<C>((const char *)aff_in + i * aff_stride<C>()));
#pragma nounroll
    for (uint32_t k = 0; k < cbits; k++) p = pt_dbl<C>(p);   // complete doubling: the identity row stays the identity
    pt_st<C>(proj_out, i, p);
}

// Batched group law (IsGroup::operate_with, short_weierstrass/point.rs:171-207): out[j*m + i] = rows[i] + cols[j].
// Used to synthesise large sets of distinct points from two short runs (bench inputs: P = [s0 + i*d]G + [j*m*d]G).
template <class C>
__global__ __launch_bounds__(MSM_THREADS) void ec_add_outer_kernel(const void *rows, uint32_t m, const void *cols, uint32_t k, void *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)m * k) return;
    pt_st<C>(out, t, pt_add<C>(pt_ld<C>(rows, t % m), pt_ld<C>(cols, t / m)));
}

// ---------------------------------------------------------------- bucket reduce
// in: nwin arrays of n points.  Group j of array w covers d in [j*g, (j+1)*g):
//   A[w][j] = sum in[d],  Q[w][j] = sum (d - j*g) * in[d]     (running sum from the top, pippenger.rs:85-98)
// out is laid out [2*nwin][ngroups]: rows 0..nwin-1 hold A, rows nwin..2*nwin-1 hold Q, so the next level
// reduces both families in one launch.
template <class C>
__global__ __launch_bounds__(MSM_THREADS) void msm_group_sum_kernel(const void *in, uint32_t n, uint32_t g,
                                                                     uint32_t ngroups, uint32_t nwin, void *out) {
    // A level costs the latency of its chain of dependent additions (the kernel runs at about one wave per SIMD).  The
    // two additions of a step, q += running and running += in[d], both read the old `running`, so a pair of lanes does
    // them side by side: the even lane keeps `running`, the odd lane keeps `q` and receives the even lane's value by
    // DPP before each step.  The chain is g additions long instead of 2g.
    using B = typename C::B;
    // lanes are packed densely over (array, group, role): the small levels have hundreds of arrays with a handful of groups
    // each, and one workgroup per array left most of every wave idle while the launch took several rounds of workgroups
    // (n = 8 with 1088 arrays: 0.26 ms for a chain of 4 additions; packed: one round)
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pair = t >> 1;
    const bool is_q = t & 1;
    if (pair >= ngroups * nwin) return;          // pairs are never split: 2*pair and 2*pair+1 leave together
    const uint32_t w = pair / ngroups, j = pair - w * ngroups;
    const size_t base = (size_t)w * n;
    const uint32_t d0 = j * g, d1 = min(n, d0 + g);
    // the first step would add the top bucket to the identity (and the identity to itself): start from it instead — a chain
    // of g - 1 additions, 13 real ones per group of 8 where the plain loop computes 16
    const Point<C> top = pt_ld<C>(in, base + d1 - 1), id = pt_identity<C>();
    Point<C> acc = Point<C>{B::select(is_q, id.x, top.x), B::select(is_q, id.y, top.y), B::select(is_q, id.z, top.z)};
#pragma nounroll
    for (uint32_t d = d1 - 1; d-- > d0;) {
        const Point<C> x = pt_ld<C>(in, base + d);
        Point<C> operand;
        operand.x = B::select(is_q, B::lane_swap(acc.x), x.x);
        operand.y = B::select(is_q, B::lane_swap(acc.y), x.y);
        operand.z = B::select(is_q, B::lane_swap(acc.z), x.z);
        acc = pt_add<C>(acc, operand);
    }
    pt_st<C>(out, (size_t)(is_q ? nwin + w : w) * ngroups + j, acc);
}

// Level results (one point per array, rows as above) -> S[w] = sumQ[w] + 2^k * S(A)[w],  A[w] = sumA[w]
template <class C>
__global__ void msm_combine_kernel(const void *S2, const void *A2, uint32_t k, uint32_t nwin, void *S, void *A) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    Point<C> x = pt_ld<C>(S2, w);
#pragma nounroll
    for (uint32_t i = 0; i < k; i++) x = pt_dbl<C>(x);
    pt_st<C>(S, w, pt_add<C>(pt_ld<C>(A2, nwin + w), x));
    pt_st<C>(A, w, pt_ld<C>(A2, w));
}

// The same two kernels for the levels that are nothing but a chain of dependent additions (a few thousand groups and
// less): eight lanes per group — a quad for `running`, a quad for `q` — with every addition spread over the three lanes
// of its quad (pt_add_quad, ec.cuh: 7 N^2 MACs per lane and link instead of 21).  Lane s of a quad loads, keeps and stores
// coordinate s only; the q quad reads the running quad's coordinates by DPP (row_shr:4).  Same group elements as
// msm_group_sum_kernel / msm_combine_kernel (another projective representative; the MSM result is normalised at the end).
template <class C>
__global__ __launch_bounds__(MSM_THREADS) void msm_group_sum_quad_kernel(const void *in, uint32_t n, uint32_t g, uint32_t ngroups,
                                                                          uint32_t nwin, void *out) {
    using B = typename C::B;
    using T = typename B::T;
    constexpr size_t PBY = 3 * B::BYTES;
    static_assert(MSM_THREADS % 16 == 0, "groups of eight lanes must not straddle a DPP row");
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = t >> 3;
    if (grp >= ngroups * nwin) return;           // groups are never split: their eight lanes leave together
    const bool is_q = (t >> 2) & 1;
    const uint32_t s = (t & 3) == 3 ? 0u : (t & 3);
    const uint32_t w = grp / ngroups, j = grp - w * ngroups;
    const char *src = (const char *)in + (size_t)w * n * PBY + s * B::BYTES;
    const uint32_t d0 = j * g, d1 = min(n, d0 + g);
    // running starts from the top bucket, q from the identity (0 : 1 : 0): g - 1 links, as in msm_group_sum_kernel
    T acc = B::select(is_q, s == 1 ? B::one() : B::zero(), B::load(src + (size_t)(d1 - 1) * PBY));
#pragma nounroll
    for (uint32_t d = d1 - 1; d-- > d0;) {
        const T x = B::load(src + (size_t)d * PBY);
        const T operand = B::select(is_q, B::template dpp<0x114>(acc), x);
        acc = pt_add_quad<C>(acc, operand, s);
    }
    if ((t & 3) != 3) B::store((char *)out + ((size_t)(is_q ? nwin + w : w) * ngroups + j) * PBY + s * B::BYTES, acc);
}
template <class C>
__global__ __launch_bounds__(64) void msm_combine_quad_kernel(const void *S2, const void *A2, uint32_t k, uint32_t nwin, void *S, void *A) {
    using B = typename C::B;
    using T = typename B::T;
    constexpr size_t PBY = 3 * B::BYTES;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t w = t >> 2;
    if (w >= nwin) return;
    const uint32_t s = (t & 3) == 3 ? 0u : (t & 3);
    const size_t co = s * B::BYTES;
    T x = B::load((const char *)S2 + (size_t)w * PBY + co);
#pragma nounroll
    for (uint32_t i = 0; i < k; i++) x = pt_add_quad<C>(x, x, s);
    x = pt_add_quad<C>(B::load((const char *)A2 + (size_t)(nwin + w) * PBY + co), x, s);
    if ((t & 3) != 3) {
        B::store((char *)S + (size_t)w * PBY + co, x);
        B::store((char *)A + (size_t)w * PBY + co, B::load((const char *)A2 + (size_t)w * PBY + co));
    }
}

// ---------------------------------------------------------------- sharded MSM (comm.hip): bucket-slice exchange
// buckets[i] = identity (a rank without pairs still owns a slice of everybody's buckets)
template <class C>
__global__ void msm_fill_identity_kernel(void *buckets, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) pt_st<C>(buckets, i, pt_identity<C>());
}
// recv[w][g][j]: bucket j of MY bucket range in window w as rank g accumulated it.  out[w][j] = sum over g.
template <class C>
__global__ __launch_bounds__(MSM_THREADS) void msm_slice_sum_kernel(const void *recv, uint32_t G, uint32_t Bs, uint32_t nwin, void *out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)nwin * Bs) return;
    const uint64_t w = t / Bs, j = t - w * Bs;
    Point<C> acc = pt_ld<C>(recv, (w * G) * Bs + j);
#pragma nounroll
    for (uint32_t g = 1; g < G; g++) acc = pt_add<C>(acc, pt_ld<C>(recv, (w * G + g) * Bs + j));
    pt_st<C>(out, t, acc);
}

// ---------------------------------------------------------------- host orchestration
struct Carver {   // bump allocator over the context workspace
    char *base;
    size_t cap, used = 0;
    void *take(size_t bytes) {
        used = (used + 255) & ~(size_t)255;
        void *p = base ? base + used : nullptr;
        used += bytes;
        return p;
    }
    // true once a real (non-dry) carve-out has run past the workspace the dry run sized: checked before every launch
    // that would touch the new pointers
    bool overrun() const { return base && used > cap; }
};
#define LW_MSM_WS_CHECK(cv)                                                                                          \
    do {                                                                                                             \
        if ((cv).overrun()) {                                                                                        \
            set_error("internal: MSM workspace of %zu bytes is too small (%zu needed so far)", (cv).cap, (cv).used); \
            return LW_ERR_ALLOC;                                                                                     \
        }                                                                                                            \
    } while (0)

// Window width c (signed digits, W = ceil(257 / c) windows of 2^(c-1) buckets): the bucket additions N * W fall with c, the
// running sums over W * 2^(c-1) buckets (about three additions per bucket, latency-bound levels) grow with it.  Only widths
// whose top window is either empty or well filled for 254/255-bit scalars are candidates — c = 8 and 16 (the window at
// bit 256 holds a carry only for scalars >= 2^255) and c = 20 (15 bits left for the top window); with c = 15, 17 or 18 a
// top window of 0-3 bits puts N items into a handful of buckets and costs extra rounds (2^22: c = 16 15.6 ms, c = 17
// 18.9, c = 18 18.3, c = 19 17.1, c = 20 16.5).  Measured with LW_HIP_MSM_C (tools/ab_msm_csweep.sh): 2^14 c = 8 1.60 ms
// against 1.92 at c = 16; 2^16 2.02 (c = 16) against 2.10; 2^23 c = 20 27.3 against 28.6; 2^24 c = 20 49.1 against 54.8.
static uint32_t pick_window(size_t n) {
    const char *e = tuning_env("LW_HIP_MSM_C");   // tuning and tests only; read per call so a test can sweep it
    const int c_env = e ? atoi(e) : 0;
    if (c_env >= 3 && c_env <= (int)msm_max_window_bits()) return (uint32_t)c_env;
    if (n < ((size_t)1 << 15)) return 8u;
    if (n < ((size_t)1 << 23)) return 16u;
    return 20u;
}

template <class C>
struct MsmRunner {
    Context &c;
    hipStream_t stream;
    uint32_t W;
    bool affine = false;   // d_points are affine pairs (pre-normalised SRS)
    uint32_t fold_c = 0;        // folded SRS (lw_hip_srs_*): window width the shifted copies were built for, and
    uint64_t fold_stride = 0;   // rows per copy: d_points[w * fold_stride + i] = 2^(c w) * P_i
    hipEvent_t points_ready = nullptr;   // recorded on a side stream once d_points is complete; joined before the first accumulation

    // SRS preparation (lw_hip_srs_create*): n projective rows -> n affine pairs
    int normalize(const void *d_in, size_t n, void *d_out) {
        if (!n) return LW_OK;
        hipEvent_t pe = nullptr;
        // run length: long runs amortise the inversion (2^24 points: 5.8 ms at 128 against 8.3 ms at 32), short ones
        // keep enough work-items in flight for small sets (2^20: 1.2 ms at 32 against 2.0 ms at 128)
        static const uint32_t chk_env = [] { const char *e = tuning_env("LW_HIP_MSM_CHK"); return e ? (uint32_t)atoi(e) : 0u; }();   // tuning only
        // (2^22 and 2^23 sit between the two: inside an MSM, where this kernel runs beside the sort and is the longer of the two
        // below 2^24, 2^22 takes 13.7 ms at 32 against 14.0 at 128 and 2^23 24.1 at 64 against 24.5, profiles/r03_ab_msm_chk_mid.txt)
        const uint32_t chk = chk_env ? std::min(std::max(chk_env, 1u), 1024u) : (n >= ((size_t)1 << 24) ? 128 : n >= ((size_t)1 << 23) ? 64 : 32);
        const uint64_t items = (n + chk - 1) / chk;
        const uint32_t blocks = (uint32_t)((items + MSM_THREADS - 1) / MSM_THREADS);
        if (c.msm_prefix.ensure(n * C::B::BYTES)) return LW_ERR_ALLOC;   // running products, one element per point
        pe = c.prof_begin(stream);
        // (Measured and dropped, profiles/r03_ab_msm_norm.txt: the two sweeps and the inversions as three kernels — standalone
        // 2.83 ms either way at 2^24, inside the MSM 2^22 14.7 -> 16.3-17.0 ms, because the inversions alone are a short grid of
        // long dependent chains (85 product-equivalents each) that this kernel hides behind the other waves' memory phases;
        // and the same in phases of 2^20 points so that the back sweep re-reads rows from the Infinity Cache: 21.6 ms, one
        // exposed inversion latency per phase.)
        hipLaunchKernelGGL((msm_to_affine_kernel<C>), dim3(blocks), dim3(MSM_THREADS), 0, stream, d_in, (uint64_t)n, chk, d_out, c.msm_prefix.p);
        c.prof_end("msm_to_affine_kernel", pe, stream);
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
        return LW_OK;
    }

    static size_t affine_bytes(size_t n) { return n * aff_stride<C>(); }

    // Folded SRS (lw_hip_srs_*): rows[w * n + i] = 2^(c w) * P_i for w = 1 .. W-1, built from the affine rows 0 .. n-1.
    // With the 2^(c w) factors in the points, the items of ALL windows can share one set of 2^(c-1) buckets: the running
    // sums run over 1 window instead of W, which is what lets c = 20 pay from 2^19 points on (HBM is the price: W copies).
    int build_fold(void *d_rows, size_t n, uint32_t cbits) {
        const uint32_t Wf = (256 + cbits) / cbits;
        if (!n) return LW_OK;
        if (c.msm_affine.ensure(n * PB)) return LW_ERR_ALLOC;   // projective scratch of one copy
        for (uint32_t w = 1; w < Wf; w++) {
            const char *prev = (const char *)d_rows + (size_t)(w - 1) * n * aff_stride<C>();
            hipEvent_t pe = c.prof_begin(stream);
            hipLaunchKernelGGL((msm_shift_kernel<C>), dim3((uint32_t)((n + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0, stream,
                               (const void *)prev, (uint64_t)n, cbits, c.msm_affine.p);
            c.prof_end("msm_shift_kernel", pe, stream);
            LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
            int rc = normalize(c.msm_affine.p, n, (char *)d_rows + (size_t)w * n * aff_stride<C>());
            if (rc) return rc;
        }
        return LW_OK;
    }
    int add_outer(const void *d_rows, uint32_t m, const void *d_cols, uint32_t k, void *d_out) {
        const uint64_t total = (uint64_t)m * k;
        if (!total) return LW_OK;
        hipLaunchKernelGGL((ec_add_outer_kernel<C>), dim3((uint32_t)((total + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0, stream,
                           d_rows, m, d_cols, k, d_out);
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
        return LW_OK;
    }

    // One level of the bucket reduce.  Wide levels (more groups than the chip has lanes for) are bound by the additions'
    // throughput: two lanes per group.  The others are bound by the chain of g dependent additions: eight lanes per group,
    // each addition spread over a quad (msm_group_sum_quad_kernel).
    void launch_group_sum(const char *in, uint32_t n, uint32_t g, uint32_t ng, uint32_t nwin, char *out, hipStream_t stream) {
        const uint64_t groups = (uint64_t)ng * nwin;
        // widest level on the quad kernel: 2^18 lanes; 2^20 for BLS12-381 G2, whose pair kernel needs 256 VGPRs (one wave per
        // SIMD) where the quad kernel needs 192 (tools/ab_msm_quad_g2.py: reduce 2.30 -> 2.15 ms; the other groups lose 5 %)
        constexpr uint64_t QUAD_DEFAULT = (uint64_t)1 << (std::is_same<typename C::B, Fp2Ops<Fp381>>::value ? 20 : 18);
        const uint64_t quad_env = msm_quad_max_lanes();
        const uint64_t quad_max = quad_env == ~(uint64_t)0 ? QUAD_DEFAULT : quad_env;
        hipEvent_t pe = c.prof_begin(stream);
        if (8 * groups <= quad_max)
            hipLaunchKernelGGL((msm_group_sum_quad_kernel<C>), dim3((uint32_t)((8 * groups + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0,
                               stream, (const void *)in, n, g, ng, nwin, (void *)out);
        else
            hipLaunchKernelGGL((msm_group_sum_kernel<C>), dim3((uint32_t)((2 * groups + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0,
                               stream, (const void *)in, n, g, ng, nwin, (void *)out);
        c.prof_end("msm_group_sum_kernel", pe, stream);
    }

    // in: nwin arrays of n points.  Returns device arrays S[nwin] (sum d*in[d]) and A[nwin] (sum in[d]).
    static constexpr size_t PB = 3 * C::B::BYTES;
    int reduce(const char *in, uint32_t n, uint32_t nwin, Carver &cv, char **S_out, char **A_out, hipStream_t stream) {
        const uint32_t MSM_G_LOG = msm_g_log();
        const uint32_t g = 1u << MSM_G_LOG;
        if (n <= g) {   // one work-item per array: S = Q (d0 = 0), A = running sum
            char *out = (char *)cv.take(PB * 2 * (size_t)nwin);
            LW_MSM_WS_CHECK(cv);
            if (cv.base) launch_group_sum(in, n, n, 1u, nwin, out, stream);
            *A_out = out;
            *S_out = cv.base ? out + PB * nwin : nullptr;
            return LW_OK;
        }
        const uint32_t ng = (n + g - 1) / g;
        char *lvl = (char *)cv.take(PB * 2 * (size_t)nwin * ng);
        LW_MSM_WS_CHECK(cv);
        if (cv.base) launch_group_sum(in, n, g, ng, nwin, lvl, stream);
        char *S2, *A2;
        int rc = reduce(lvl, ng, 2 * nwin, cv, &S2, &A2, stream);
        if (rc) return rc;
        char *S = (char *)cv.take(PB * nwin), *A = (char *)cv.take(PB * nwin);
        LW_MSM_WS_CHECK(cv);
        if (cv.base) {
            hipEvent_t pe = c.prof_begin(stream);
            if (msm_quad_max_lanes() != 0)
                hipLaunchKernelGGL((msm_combine_quad_kernel<C>), dim3((4 * nwin + 63) / 64), dim3(64), 0, stream, (const void *)S2, (const void *)A2,
                                   MSM_G_LOG, nwin, (void *)S, (void *)A);
            else
                hipLaunchKernelGGL((msm_combine_kernel<C>), dim3((nwin + 63) / 64), dim3(64), 0, stream, (const void *)S2, (const void *)A2, MSM_G_LOG,
                                   nwin, (void *)S, (void *)A);
            c.prof_end("msm_combine_kernel", pe, stream);
        }
        *S_out = S;
        *A_out = A;
        return LW_OK;
    }

    // The windows of one MSM are processed as one or two independent SLICES [w0, w0 + Wh): each has its own sort arrays,
    // bucket array and running sums, exactly as if it were an MSM with Wh windows; the digit matrix is shared.  Two
    // slices let the memory-bound sort of the second and the latency-bound running sums of the first run on the side
    // stream UNDER the compute-bound accumulation of the other slice (run() below).
    struct Slice {
        uint32_t w0 = 0, Wh = 0, NW = 0, K = 0, CB = 0;
        uint32_t *coarse_cnt = nullptr, *coarse_cursor = nullptr, *maxlen_d = nullptr, *key_cnt = nullptr, *key_cursor = nullptr;
        uint32_t *coarse_off = nullptr, *sub_off = nullptr, *off = nullptr, *scan_tmp = nullptr, *sorted = nullptr, *order_tmp = nullptr;
        uint64_t *items = nullptr;
        char *buckets = nullptr, *S = nullptr, *A = nullptr;
        volatile uint32_t *maxlen_h = nullptr;   // pinned host word the sort's longest bucket is copied to
    };

    // carve-outs of a slice's sort; with cv.base == nullptr only the sizes are added up
    int carve_sort(Slice &sl, size_t n, uint32_t cbits, Carver &cv) {
        sl.NW = fold_stride ? 1u : sl.Wh;       // bucket sets: one per window, or one for all (folded SRS)
        sl.K = sl.NW << (cbits - 1);            // signed digits: 2^(c-1) buckets per window, bucket j = multiplier j + 1
        sl.CB = msm_sort_coarse_bins(cbits, sl.NW, fold_stride ? (uint64_t)W * fold_stride : n);
        sl.coarse_cnt = (uint32_t *)cv.take(4 * (size_t)(sl.CB + 1));
        sl.coarse_cursor = (uint32_t *)cv.take(4 * (size_t)(sl.CB + 1));
        sl.maxlen_d = (uint32_t *)cv.take(256);
        sl.key_cnt = (uint32_t *)cv.take(4 * (size_t)sl.K);        // zeroed with the counters above (contiguous)
        sl.key_cursor = (uint32_t *)cv.take(4 * (size_t)sl.K);
        sl.coarse_off = (uint32_t *)cv.take(4 * (size_t)(sl.CB + 1));
        sl.sub_off = (uint32_t *)cv.take(4 * (size_t)(sl.CB + 1));
        sl.off = (uint32_t *)cv.take(4 * (size_t)(sl.K + 1));
        sl.scan_tmp = (uint32_t *)cv.take(msm_scan_scratch_bytes(sl.K));
        sl.sorted = (uint32_t *)cv.take(4 * n * sl.Wh);
        sl.items = (uint64_t *)cv.take(8 * n * sl.Wh);
        sl.order_tmp = (uint32_t *)cv.take(msm_order_tmp_bytes());
        sl.buckets = (char *)cv.take(PB * (size_t)sl.K);
        LW_MSM_WS_CHECK(cv);
        return LW_OK;
    }

    // sort of the slice's windows (rows w0 .. w0 + Wh - 1 of the digit matrix) on stream `s`; the longest bucket lands in
    // *sl.maxlen_h once `s` gets there
    int launch_sort(Slice &sl, const uint32_t *dig, size_t n, uint32_t cbits, hipStream_t s) {
        // coarse_cnt, coarse_cursor, maxlen, key_cnt and key_cursor are adjacent carve-outs: one memset clears them all
        LW_HIP_CHECK(hipMemsetAsync(sl.coarse_cnt, 0, (size_t)((char *)sl.coarse_off - (char *)sl.coarse_cnt), s), LW_ERR_LAUNCH);
        msm_launch_sort(c, dig + (size_t)sl.w0 * msm_sort_padded_points(n), (uint64_t)n, cbits, sl.Wh, sl.coarse_cnt, sl.coarse_off,
                        sl.coarse_cursor, sl.items, sl.sorted, sl.off, sl.K, sl.maxlen_d, sl.scan_tmp, sl.sub_off, sl.key_cnt,
                        sl.key_cursor, fold_stride, (uint64_t)sl.w0, s);
        LW_HIP_CHECK(hipMemcpyAsync((void *)sl.maxlen_h, sl.maxlen_d, 4, hipMemcpyDeviceToHost, s), LW_ERR_LAUNCH);
        return LW_OK;
    }

    // accumulate rounds of a slice on stream `s`: while some bucket is longer than CH, cut every bucket into CH-sized
    // pieces.  `maxlen`: the longest bucket (real run: from the sort; dry run: the worst case).  before_first_launch() is
    // called once, right before the first accumulate kernel is enqueued.
    template <class Hook>
    int accumulate(Slice &sl, const void *d_points, size_t n, uint32_t maxlen, Carver &cv, hipStream_t s, Hook before_first_launch) {
        const bool dry = cv.base == nullptr;
        const uint32_t K = sl.K;
        const uint32_t CH = msm_ch((uint64_t)n * W);
        const bool ordered = msm_piece_order_enabled() != 0;   // first round only: later rounds sum equal numbers of partials
        const uint32_t *seg = sl.off;
        const void *pts = d_points;         // first round: the caller's points through the sorted index list
        const uint32_t *index = sl.sorted;
        uint64_t len = maxlen;             // longest segment
        uint64_t items_bound = (uint64_t)n * sl.Wh;   // upper bound on items in this round
        bool first = true;                 // (the dry run has no pointers to tell the rounds apart)
        bool hooked = false;
        char *buckets = sl.buckets;
        auto launch = [&](const uint32_t *out_off, const uint32_t *perm_t, const uint32_t *perm_key, uint32_t total, void *pout,
                          const char *name) {
            if (!hooked) { before_first_launch(); hooked = true; }
            const uint32_t blocks = (total + MSM_THREADS - 1) / MSM_THREADS;
            hipEvent_t pe = c.prof_begin(s);
            if (4 * (uint64_t)total <= msm_accumulate_quad_max_lanes()) {   // few pieces: their chains are the kernel time
                const dim3 qgrid((uint32_t)((4 * (uint64_t)total + MSM_THREADS - 1) / MSM_THREADS));
                if (index && affine)
                    hipLaunchKernelGGL((msm_accumulate_quad_kernel<C, true>), qgrid, dim3(MSM_THREADS), 0, s, pts, index, seg, out_off, perm_t,
                                       perm_key, K, total, pout, (void *)buckets);
                else
                    hipLaunchKernelGGL((msm_accumulate_quad_kernel<C, false>), qgrid, dim3(MSM_THREADS), 0, s, pts, index, seg, out_off, perm_t,
                                       perm_key, K, total, pout, (void *)buckets);
            } else if (index && affine)
                hipLaunchKernelGGL((msm_accumulate_kernel<C, C::ACC_WAVES, true>), dim3(blocks), dim3(MSM_THREADS), 0, s, pts, index, seg,
                                   out_off, perm_t, perm_key, K, total, pout, (void *)buckets);
            else if (out_off && C::ACC_WAVES == 2 && msm_waves_per_simd() == 3)
                hipLaunchKernelGGL((msm_accumulate_kernel<C, (C::ACC_WAVES == 2 ? 3 : C::ACC_WAVES), false>), dim3(blocks), dim3(MSM_THREADS), 0,
                                   s, pts, index, seg, out_off, perm_t, perm_key, K, total, pout, (void *)buckets);
            else
                hipLaunchKernelGGL((msm_accumulate_kernel<C, C::ACC_WAVES, false>), dim3(blocks), dim3(MSM_THREADS), 0, s, pts, index, seg,
                                   out_off, perm_t, perm_key, K, total, pout, (void *)buckets);
            c.prof_end(name, pe, s);
        };
        while (len > CH) {
            uint32_t *out_off = (uint32_t *)cv.take(4 * (size_t)(K + 1));
            uint64_t out_bound = items_bound / CH + K;
            char *pout = (char *)cv.take(PB * out_bound);
            const bool ord = ordered && first;
            uint32_t *perm_t = ord ? (uint32_t *)cv.take(4 * out_bound) : nullptr;
            uint32_t *perm_key = ord ? (uint32_t *)cv.take(4 * out_bound) : nullptr;
            LW_MSM_WS_CHECK(cv);
            if (!dry) {
                msm_launch_scan(seg, out_off, K, (int)CH, sl.maxlen_d, sl.scan_tmp, s);
                uint32_t total = 0;
                LW_HIP_CHECK(hipMemcpyAsync(&total, out_off + K, 4, hipMemcpyDeviceToHost, s), LW_ERR_LAUNCH);
                LW_HIP_CHECK(hipStreamSynchronize(s), LW_ERR_LAUNCH);
                if (total > out_bound) {
                    set_error("internal: MSM partial count %u exceeds bound %llu", total, (unsigned long long)out_bound);
                    return LW_ERR_LAUNCH;
                }
                if (ord) msm_launch_piece_order(c, seg, out_off, K, total, sl.order_tmp, perm_t, perm_key, s);
                if (total) launch(out_off, perm_t, perm_key, total, (void *)pout, index ? "msm_accumulate_kernel" : "msm_accumulate_kernel<partials>");
            }
            seg = out_off;
            pts = pout;
            index = nullptr;
            first = false;
            len = (len + CH - 1) / CH;
            items_bound = out_bound;
        }
        {
            const bool ord = ordered && first;
            uint32_t *perm_t = ord ? (uint32_t *)cv.take(4 * (size_t)K) : nullptr;
            LW_MSM_WS_CHECK(cv);
            if (!dry) {
                if (ord) msm_launch_piece_order(c, seg, nullptr, K, K, sl.order_tmp, perm_t, nullptr, s);
                launch(nullptr, perm_t, nullptr, K, nullptr, index ? "msm_accumulate_kernel" : "msm_accumulate_kernel<final>");
            }
        }
        return LW_OK;
    }

    // `to` waits for everything enqueued on `from` so far (untimed event from the context pool)
    int chain(hipStream_t from, hipStream_t to, std::vector<hipEvent_t> &taken, hipEvent_t *out = nullptr) {
        hipEvent_t e = nullptr;
        if (!c.sync_pool.empty()) { e = c.sync_pool.back(); c.sync_pool.pop_back(); }
        else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return LW_ERR_LAUNCH; }
        taken.push_back(e);
        LW_HIP_CHECK(hipEventRecord(e, from), LW_ERR_LAUNCH);
        if (to) LW_HIP_CHECK(hipStreamWaitEvent(to, e, 0), LW_ERR_LAUNCH);
        if (out) *out = e;
        return LW_OK;
    }

    // window sums -> the MSM: fold most-significant first, acc <- 2^c * acc + sum_w (pippenger.rs:101).  a_top: the plain sum of
    // the top slot's buckets, needed when c divides 256 (see below).
    static Point<C> fold_windows(const std::vector<Point<C>> &wsum, const Point<C> &a_top, uint32_t cbits, uint32_t W, bool folded) {
        const uint32_t NW = folded ? 1u : W;
        uint32_t top = NW - 1;
        Point<C> result = wsum[top];   // folded: the copies already carry the 2^(c w) factors, one sum is the result
        if (!folded && (W - 1) * cbits == 256) {
            // the top window's values above 2^(c-1) live in slot W-1 with 2^(c-1) taken off (msm_digits_kernel): both slots
            // weigh 2^(256-c); slot W-1 owes 2^(c-1) times its plain sum
            Point<C> extra = a_top;
            for (uint32_t i = 0; i + 1 < cbits; i++) extra = pt_dbl<C>(extra);
            top--;
            result = pt_add<C>(pt_add<C>(result, extra), wsum[top]);
        }
        for (uint32_t w = top; w-- > 0;) {
            for (uint32_t i = 0; i < cbits; i++) result = pt_dbl<C>(result);
            result = pt_add<C>(result, wsum[w]);
        }
        return result;
    }

    // ---- sharded MSM (comm.hip msm_sharded_run): three phases around the bucket-slice exchange --------------------------
    // Phase 1: digits + sort + accumulation of the local pairs with the window width all ranks agreed on; leaves the dense
    // bucket array [W][2^(c-1)] in the context workspace (*buckets_out, valid until the next MSM on this context).
    int shard_accumulate(const uint64_t *d_scalars, const void *d_points, size_t n, uint32_t cbits, char **buckets_out) {
        if ((n >> 31) || (((uint64_t)n * ((256 + cbits) / cbits)) >> 32)) { set_error("MSM shard of %zu points: index width", n); return LW_ERR_BAD_ARG; }
        W = (256 + cbits) / cbits;
        Slice sl;
        sl.w0 = 0;
        sl.Wh = W;
        if (n == 0) {   // no pairs here: identity buckets (this rank still owns a slice of everybody's)
            const uint64_t K = (uint64_t)W << (cbits - 1);
            if (c.msm_ws.ensure(PB * K + 4096)) return LW_ERR_ALLOC;
            hipLaunchKernelGGL((msm_fill_identity_kernel<C>), dim3((uint32_t)((K + 255) / 256)), dim3(256), 0, stream, c.msm_ws.p, K);
            LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
            *buckets_out = (char *)c.msm_ws.p;
            return LW_OK;
        }
        auto nothing = [] {};
        Carver dry{nullptr, 0};
        uint32_t *dig = (uint32_t *)dry.take(4 * (size_t)W * msm_sort_padded_points(n));
        int rc = carve_sort(sl, n, cbits, dry);
        if (!rc) rc = accumulate(sl, nullptr, n, (uint32_t)std::min<size_t>(n, 0xffffffffu), dry, stream, nothing);
        if (rc) return rc;
        if (c.msm_ws.ensure(dry.used + 4096)) return LW_ERR_ALLOC;
        if (!c.pinned_words) LW_HIP_CHECK(hipHostMalloc((void **)&c.pinned_words, 256, hipHostMallocDefault), LW_ERR_ALLOC);
        Carver cv{(char *)c.msm_ws.p, c.msm_ws.bytes};
        dig = (uint32_t *)cv.take(4 * (size_t)W * msm_sort_padded_points(n));
        rc = carve_sort(sl, n, cbits, cv);
        if (rc) return rc;
        sl.maxlen_h = c.pinned_words;
        *sl.maxlen_h = 0;
        msm_launch_digits(c, (const uint32_t *)d_scalars, (uint64_t)n, cbits, W, dig, stream);
        rc = launch_sort(sl, dig, n, cbits, stream);
        if (rc) return rc;
        LW_HIP_CHECK(hipStreamSynchronize(stream), LW_ERR_LAUNCH);
        if (points_ready) LW_HIP_CHECK(hipStreamWaitEvent(stream, points_ready, 0), LW_ERR_LAUNCH);
        rc = accumulate(sl, d_points, n, *sl.maxlen_h, cv, stream, nothing);
        if (rc) return rc;
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
        *buckets_out = sl.buckets;
        return LW_OK;
    }

    // Phase 2: recv[w][g][j] = bucket j of my bucket range [rank*Bs, (rank+1)*Bs) of window w as rank g accumulated it
    // (Bs = 2^(c-1) / G).  Adds the G contributions and runs the running sums over the slice:
    // d_sa[w] = S_w = sum_d d * B[d], d_sa[W + w] = A_w = sum_d B[d], d counted from 0 INSIDE the slice — the slice's
    // offset is applied in phase 3.  The bucket reduce of an MSM thus costs 1/G of the single-GPU one per rank.
    int shard_reduce(const char *recv, uint32_t G, uint32_t cbits, char *d_sa) {
        W = (256 + cbits) / cbits;
        const uint32_t Bs = (1u << (cbits - 1)) / G;
        if (Bs == 0 || Bs * G != (1u << (cbits - 1))) { set_error("2^%u buckets cannot be cut into %u slices", cbits - 1, G); return LW_ERR_BAD_ARG; }
        char *S_d = nullptr, *A_d = nullptr;
        Carver dry{nullptr, 0};
        (void)dry.take(PB * (size_t)W * Bs);
        int rc = reduce(nullptr, Bs, W, dry, &S_d, &A_d, stream);
        if (rc) return rc;
        if (c.msm_ws.ensure(dry.used + 4096)) return LW_ERR_ALLOC;   // (the local buckets have left through the exchange)
        Carver cv{(char *)c.msm_ws.p, c.msm_ws.bytes};
        char *summed = (char *)cv.take(PB * (size_t)W * Bs);
        hipEvent_t pe = c.prof_begin(stream);
        hipLaunchKernelGGL((msm_slice_sum_kernel<C>), dim3((uint32_t)(((uint64_t)W * Bs + MSM_THREADS - 1) / MSM_THREADS)), dim3(MSM_THREADS), 0,
                           stream, (const void *)recv, G, Bs, W, (void *)summed);
        c.prof_end("msm_slice_sum_kernel", pe, stream);
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
        rc = reduce(summed, Bs, W, cv, &S_d, &A_d, stream);
        if (rc) return rc;
        LW_HIP_CHECK(hipMemcpyAsync(d_sa, S_d, PB * W, hipMemcpyDeviceToDevice, stream), LW_ERR_LAUNCH);
        LW_HIP_CHECK(hipMemcpyAsync(d_sa + PB * W, A_d, PB * W, hipMemcpyDeviceToDevice, stream), LW_ERR_LAUNCH);
        return LW_OK;
    }

    // Phase 3 (host, every rank the same): sa_all[g] = rank g's (S[W], A[W]).  Bucket d of slice g is bucket g*Bs + d of the
    // window, multiplier g*Bs + d + 1, so   window sum = sum_g (S_g + A_g) + Bs * sum_g g * A_g.
    static void shard_combine_host(const char *sa_all, uint32_t G, uint32_t cbits, void *out_host) {
        const uint32_t W = (256 + cbits) / cbits, Bs = (1u << (cbits - 1)) / G;
        uint32_t lgBs = 0;
        while ((1u << lgBs) < Bs) lgBs++;
        std::vector<Point<C>> wsum(W);
        Point<C> a_top = pt_identity<C>();
        for (uint32_t w = 0; w < W; w++) {
            Point<C> x = pt_identity<C>(), run = pt_identity<C>(), t = pt_identity<C>(), a_all = pt_identity<C>();
            for (uint32_t g = G; g-- > 0;) {   // sum_g g * A_g by a running sum from the top slice (pippenger.rs:85-98, over ranks)
                const char *sg = sa_all + (size_t)g * 2 * W * PB;
                const Point<C> S = pt_load<C>(sg + PB * w), A = pt_load<C>(sg + PB * (W + w));
                x = pt_add<C>(x, pt_add<C>(S, A));
                a_all = pt_add<C>(a_all, A);
                if (g > 0) { run = pt_add<C>(run, A); t = pt_add<C>(t, run); }
            }
            for (uint32_t i = 0; i < lgBs; i++) t = pt_dbl<C>(t);
            wsum[w] = pt_add<C>(x, t);
            if (w + 1 == W) a_top = a_all;
        }
        Point<C> result = fold_windows(wsum, a_top, cbits, W, false);
        result = pt_unmap_result<C>(pt_to_affine<C>(result));
        pt_store<C>(out_host, result);
    }

    int run(const uint64_t *d_scalars, const void *d_points, size_t n, void *out_host) {
        Point<C> result = pt_identity<C>();
        if (n > 0) {
            if (n >> 32) {
                set_error("MSM of %zu points: index width is 32 bits", n);
                return LW_ERR_BAD_ARG;
            }
            if (n >> 31) {
                set_error("MSM of %zu points: the sorted list keeps the digit's sign in bit 31 of the index", n);
                return LW_ERR_BAD_ARG;
            }
            const uint32_t cbits = fold_stride ? fold_c : pick_window(n);
            W = (256 + cbits) / cbits;   // ceil(257 / c): the signed recoding of a 256-bit scalar never carries out of the top window
            if (fold_stride && (((uint64_t)W * fold_stride) >> 31)) {
                set_error("folded SRS of %llu x %u rows overflows the 31-bit point index", (unsigned long long)fold_stride, W);
                return LW_ERR_BAD_ARG;
            }
            if (((uint64_t)n * W) >> 32) {
                set_error("MSM of %zu points x %u windows overflows 32-bit item offsets; shard the input", n, W);
                return LW_ERR_BAD_ARG;
            }
            // Two slices (LW_HIP_MSM_SLICES=2, off by default): MEASURED AND DROPPED as the default — with the second slice's
            // sort and the first slice's running sums on a high-priority side stream under the other slice's accumulation,
            // 2^24 BLS12-381 G1 took 45.6 ms against 45.2-45.6 ms in one slice, and 2^22 15.6 against 14.8
            // (profiles/r03_ab_msm_slices.txt): the accumulate kernel is bound by VALU issue with 3 waves per SIMD hiding its
            // gathers, so every wave slot, LDS allocation and issue cycle the side kernels take comes out of it one for one
            // (it ran 1.3 ms longer; the side kernels, starved, took 3-5x their standalone time).
            static const bool split_env = [] { const char *e = tuning_env("LW_HIP_MSM_SLICES"); return e && atoi(e) == 2; }();   // A/B only
            const bool two = split_env && !fold_stride && W >= 4 && n >= ((size_t)1 << 22);
            Slice sl[2];
            const int ns = two ? 2 : 1;
            sl[0].w0 = 0;
            sl[0].Wh = two ? (W + 1) / 2 : W;
            sl[1].w0 = sl[0].Wh;
            sl[1].Wh = W - sl[0].Wh;
            // size the workspace for the worst case: one bucket holding every item.  A folded SRS sorts the items of all W
            // windows into one bucket set, so its longest bucket can hold n * W items (every scalar with the same digit in
            // every window), which takes more rounds of partial sums than n items do.
            const uint64_t worst_len = fold_stride ? (uint64_t)n * W : (uint64_t)n;
            const uint32_t worst = (uint32_t)std::min<uint64_t>(worst_len, 0xffffffffu);
            auto nothing = [] {};
            auto plan = [&](Carver &cv, uint32_t **dig_out) -> int {   // every carve-out that does not depend on the data
                *dig_out = (uint32_t *)cv.take(4 * (size_t)W * msm_sort_padded_points(n));
                for (int k = 0; k < ns; k++) {
                    int rc = carve_sort(sl[k], n, cbits, cv);
                    if (rc) return rc;
                }
                return LW_OK;
            };
            Carver dry{nullptr, 0};
            uint32_t *dig = nullptr;
            int rc = plan(dry, &dig);
            for (int k = 0; k < ns && !rc; k++) {
                rc = accumulate(sl[k], nullptr, n, worst, dry, stream, nothing);
                if (!rc) rc = reduce(sl[k].buckets, 1u << (cbits - 1), sl[k].NW, dry, &sl[k].S, &sl[k].A, stream);
            }
            if (rc) return rc;
            if (c.msm_ws.ensure(dry.used + 4096)) return LW_ERR_ALLOC;
            if (!c.pinned_words) LW_HIP_CHECK(hipHostMalloc((void **)&c.pinned_words, 256, hipHostMallocDefault), LW_ERR_ALLOC);
            Carver cv{(char *)c.msm_ws.p, c.msm_ws.bytes};
            rc = plan(cv, &dig);
            if (rc) return rc;
            std::vector<hipEvent_t> taken;
            struct Giveback { Context &c; std::vector<hipEvent_t> &t; ~Giveback() { for (hipEvent_t e : t) c.sync_pool.push_back(e); } } giveback{c, taken};
            for (int k = 0; k < ns; k++) { sl[k].maxlen_h = c.pinned_words + k; *sl[k].maxlen_h = 0; }
            hipStream_t side = nullptr;
            if (two) {
                rc = ensure_aux_stream(c);
                if (rc) return rc;
                side = c.aux_hi;   // high priority: its short kernels take CU slots as the long accumulate kernel's workgroups retire
            }
            // digits of all windows, then the sort of the first slice, on the caller's stream
            msm_launch_digits(c, (const uint32_t *)d_scalars, (uint64_t)n, cbits, W, dig, stream);
            rc = launch_sort(sl[0], dig, n, cbits, stream);
            if (rc) return rc;
            if (c.msm_after_sort) {   // host-buffer call: the points are uploaded (and normalised) while the sort above runs
                auto hook = std::move(c.msm_after_sort);
                c.msm_after_sort = nullptr;
                rc = hook();
                if (rc) return rc;
            }
            LW_HIP_CHECK(hipStreamSynchronize(stream), LW_ERR_LAUNCH);
            if (points_ready) LW_HIP_CHECK(hipStreamWaitEvent(stream, points_ready, 0), LW_ERR_LAUNCH);   // normalised points
            hipEvent_t sorted1 = nullptr;
            int hook_rc = LW_OK;
            auto start_side_sort = [&] {   // the second slice's sort starts when the first slice's accumulation does
                if (!two) return;
                hook_rc = chain(stream, side, taken);
                if (!hook_rc) hook_rc = launch_sort(sl[1], dig, n, cbits, side);
                if (!hook_rc) hook_rc = chain(side, nullptr, taken, &sorted1);
            };
            rc = accumulate(sl[0], d_points, n, *sl[0].maxlen_h, cv, stream, start_side_sort);
            if (rc || hook_rc) return rc ? rc : hook_rc;
            if (two) {
                // running sums of the first slice on the side stream, under the second slice's accumulation
                rc = chain(stream, side, taken);
                if (!rc) rc = reduce(sl[0].buckets, 1u << (cbits - 1), sl[0].NW, cv, &sl[0].S, &sl[0].A, side);
                if (rc) return rc;
                LW_HIP_CHECK(hipEventSynchronize(sorted1), LW_ERR_LAUNCH);           // the host needs the longest bucket
                LW_HIP_CHECK(hipStreamWaitEvent(stream, sorted1, 0), LW_ERR_LAUNCH);
                rc = accumulate(sl[1], d_points, n, *sl[1].maxlen_h, cv, stream, nothing);
                if (!rc) rc = reduce(sl[1].buckets, 1u << (cbits - 1), sl[1].NW, cv, &sl[1].S, &sl[1].A, stream);
                if (!rc) rc = chain(side, stream, taken);
            } else {
                rc = reduce(sl[0].buckets, 1u << (cbits - 1), sl[0].NW, cv, &sl[0].S, &sl[0].A, stream);
            }
            if (rc) return rc;
            LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
            const uint32_t NW = fold_stride ? 1u : W;
            std::vector<char> S(PB * NW), A(PB * NW);
            for (int k = 0; k < ns; k++) {
                LW_HIP_CHECK(hipMemcpyAsync(S.data() + PB * sl[k].w0, sl[k].S, PB * sl[k].NW, hipMemcpyDeviceToHost, stream), LW_ERR_LAUNCH);
                LW_HIP_CHECK(hipMemcpyAsync(A.data() + PB * sl[k].w0, sl[k].A, PB * sl[k].NW, hipMemcpyDeviceToHost, stream), LW_ERR_LAUNCH);
            }
            LW_HIP_CHECK(hipStreamSynchronize(stream), LW_ERR_LAUNCH);
            // window sum = sum (j + 1) * bucket[j] = S_w + A_w
            std::vector<Point<C>> wsum(NW);
            for (uint32_t w = 0; w < NW; w++) wsum[w] = pt_add<C>(pt_load<C>(S.data() + PB * w), pt_load<C>(A.data() + PB * w));
            result = fold_windows(wsum, pt_load<C>(A.data() + PB * (NW - 1)), cbits, W, fold_stride != 0);
        }
        result = pt_unmap_result<C>(pt_to_affine<C>(result));
        pt_store<C>(out_host, result);
        return LW_OK;
    }
};

// one translation unit per curve (they compile in parallel): the three entry points msm.hip dispatches to
#define LW_MSM_INSTANTIATE(CURVE, SUFFIX)                                                                                        \
    int msm_run_##SUFFIX(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_points, size_t n, void *out, int affine, \
                         hipEvent_t points_ready) {                                                                                \
        if (affine && IsoOf<CURVE>::has) {   /* normalised rows live on the isomorphic model (msm_to_affine_kernel) */           \
            MsmRunner<typename IsoOf<CURVE>::type> ri{c, s, 0};                                                                    \
            ri.affine = true;                                                                                                      \
            ri.points_ready = points_ready;                                                                                        \
            ri.fold_c = c.msm_fold_c;                                                                                              \
            ri.fold_stride = c.msm_fold_stride;                                                                                    \
            return ri.run(d_scalars, d_points, n, out);                                                                            \
        }                                                                                                                          \
        MsmRunner<CURVE> r{c, s, 0};                                                                                               \
        r.affine = affine != 0;                                                                                                    \
        r.points_ready = points_ready;                                                                                             \
        if (affine) {                                                                                                              \
            r.fold_c = c.msm_fold_c;                                                                                               \
            r.fold_stride = c.msm_fold_stride;                                                                                     \
        }                                                                                                                          \
        return r.run(d_scalars, d_points, n, out);                                                                                 \
    }                                                                                                                              \
    int msm_normalize_##SUFFIX(Context &c, hipStream_t s, const void *d_in, size_t n, void *d_out) {                               \
        MsmRunner<CURVE> r{c, s, 0};                                                                                               \
        return r.normalize(d_in, n, d_out);                                                                                        \
    }                                                                                                                              \
    size_t msm_affine_bytes_##SUFFIX(size_t n) { return MsmRunner<CURVE>::affine_bytes(n); }                                      \
    int msm_fold_build_##SUFFIX(Context &c, hipStream_t s, void *d_rows, size_t n, uint32_t cbits) {   /* rows live on IsoOf<CURVE> */ \
        MsmRunner<typename IsoOf<CURVE>::type> r{c, s, 0};                                                                         \
        return r.build_fold(d_rows, n, cbits);                                                                                     \
    }                                      \
    /* sharded MSM phases: always on the normalised (affine) rows, i.e. on IsoOf<CURVE> where the curve has a cheaper model */  \
    int msm_shard_accumulate_##SUFFIX(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_aff, size_t n, uint32_t cbits, \
                                      char **buckets) {                                                                             \
        MsmRunner<typename IsoOf<CURVE>::type> r{c, s, 0};                                                                         \
        r.affine = true;                                                                                                           \
        return r.shard_accumulate(d_scalars, d_aff, n, cbits, buckets);                                                            \
    }                                                                                                                              \
    int msm_shard_reduce_##SUFFIX(Context &c, hipStream_t s, const char *recv, uint32_t G, uint32_t cbits, char *d_sa) {           \
        MsmRunner<typename IsoOf<CURVE>::type> r{c, s, 0};                                                                         \
        return r.shard_reduce(recv, G, cbits, d_sa);                                                                               \
    }                                                                                                                              \
    void msm_shard_combine_##SUFFIX(const char *sa_all, uint32_t G, uint32_t cbits, void *out) {                                   \
        MsmRunner<typename IsoOf<CURVE>::type>::shard_combine_host(sa_all, G, cbits, out);                                         \
    }                                                                                                                              \
    int ec_add_outer_##SUFFIX(Context &c, hipStream_t s, const void *d_rows, uint32_t m, const void *d_cols, uint32_t k, void *d_out) { \
        MsmRunner<CURVE> r{c, s, 0};                                                                                               \
        return r.add_outer(d_rows, m, d_cols, k, d_out);                                                                           \
    }

}  // namespace lw
