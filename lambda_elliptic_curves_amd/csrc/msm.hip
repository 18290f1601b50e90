// Pippenger multi-scalar multiplication on gfx950: sum_i k_i * P_i  (math/src/msm/pippenger.rs:18-103).
//
// The reference walks windows sequentially, adding each point into one of 2^c - 1 buckets and folding the
// buckets with a running sum (pippenger.rs:69-101).  The group element computed here is the same; the
// schedule is rebuilt for a GPU:
//   0. normalise   large projective inputs -> affine rows on a side stream (batch inversion; BLS12-381 G1 and BN254 G2
//                  land on a cheaper isomorphic curve, ec.cuh), so that step 3 uses the mixed addition.
//   1. digits      every scalar is recoded into W = ceil(257/c) SIGNED c-bit digits (the reference uses unsigned ones,
//                  pippenger.rs:76-77); (window, |digit| - 1) is a bucket key, the sign negates the point; digit 0
//                  contributes nothing (:78).  c = 8 / 16 / 20 by size (msm_core.cuh pick_window).
//   2. scatter     two-level counting sort of (point index, sign) by key through LDS (coarse bins per window, then the
//                  keys of every coarse bin); the order inside a bucket is irrelevant to the group sum.
//   3. accumulate  segmented reduction over each bucket's list: every work-item sums one piece (<= CH points) of ONE
//                  bucket with the complete addition law (ec.cuh), pieces handed out in descending order of length;
//                  buckets longer than CH are finished in further rounds, so a skewed scalar distribution (all scalars
//                  equal) costs extra rounds, not one serial thread.
//   4. bucket reduce  sum_j (j+1)*B[j] per window by a hierarchical running sum: groups of g buckets give
//                  (A_j, Q_j) = (sum B[d], sum (d-d0) B[d]); then sum_d d*B[d] = sum_j Q_j + g * sum_j j*A_j,
//                  the second term being the same problem on n/g points; the window sum is S + A.
//   5. combine     the <= 33 window sums are folded most-significant first, acc <- 2^c * acc + S_w
//                  (pippenger.rs:101), on the host with the same limb code, and normalised to (x/z : y/z : 1).
// lw_hip_srs_* handles of >= 2^19 points keep window-shifted copies of the points, which lets all windows share one
// bucket set (msm_core.cuh build_fold).
#include <stdlib.h>
#include "msm_core.cuh"

namespace lw {

// ---- scalar preparation: FrElement (Montgomery form) -> canonical integer, i.e. `.representative()` --------------
// Every caller of msm() first maps its witness / coefficients through representative() on the CPU
// (provers/groth16/src/prover.rs:69-78, crypto/src/commitments/kzg.rs:159-163), one Montgomery product per scalar.
template <class F>
__global__ void msm_scalars_from_mont_kernel(const void *in, void *out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store<F>((char *)out + i * 32, fe_from_mont<F>(fe_load<F>((const char *)in + i * 32)));
}
void msm_launch_scalar_prep(int bn254, const void *in, void *out, uint64_t n, hipStream_t s) {
    dim3 grid((uint32_t)((n + 255) / 256));
    if (bn254) hipLaunchKernelGGL((msm_scalars_from_mont_kernel<Fr254>), grid, dim3(256), 0, s, in, out, n);
    else hipLaunchKernelGGL((msm_scalars_from_mont_kernel<Fr381>), grid, dim3(256), 0, s, in, out, n);
}

// ---- bucket scatter: two-level counting sort of (point index, sign) by key = (window, |digit| - 1) -------------
// Digits are SIGNED: with u = (c raw bits) + carry, u > 2^(c-1) becomes the digit u - 2^c and carries one into the next
// window, so |digit| <= 2^(c-1) and a window has 2^(c-1) buckets — half the running-sum work of the reference's unsigned
// digits (pippenger.rs:76-81) for the same c, which is what makes c = 20 affordable at 2^24 points (13 windows instead
// of 16: 19 % fewer bucket additions).  A negative digit adds -P (one field negation of y in the accumulate kernel).
// W = ceil(257 / c) windows, so the top window never carries out (scalars are below 2^256).  Same group element as the
// reference's sum; bucket j of a window stands for the multiplier j + 1.
// Level 0 cuts every scalar into its W digits once (msm_digits_kernel: 32 B read, W x 4 B written per point), so the
// per-window passes below read 4 bytes per point instead of the whole scalar.
// Level A partitions the N items of ONE window (grid.y = window) into coarse bins (high key bits): a workgroup takes
// 16384 points, histograms them in LDS, reserves one contiguous run per bin with a single global atomic and writes its
// items of that bin into the run (ranks from LDS atomics).  The first version of this level did all W windows in one
// workgroup of 1024 points: 4096 bins per workgroup, 4-item runs and 67 M contended global atomics (16384 workgroups x
// 4096 counters); that was 6.8 of the sort's 9.5 ms at 2^24.
// Level B gives every coarse bin to one or more workgroups, which counting-sort it by the low key bits through LDS and
// emit the per-key offsets the accumulation needs.
constexpr uint32_t SORT_PPB = 16384;            // points per level-A workgroup (one window)
constexpr uint32_t SORT_THREADS = 512;
constexpr uint32_t SORT_MAX_COARSE = 512;       // coarse bins per window
constexpr uint32_t SORT_MAX_FINE = 1024;        // keys per coarse bin
constexpr uint32_t MSM_MAX_C = 20;              // key bits c - 1 <= 9 + 10

// split of the c - 1 key bits of a window into (coarse, fine) and the item width.  Narrow items (32 bits: fine digit << 25
// | sign << 24 | index) halve the traffic of the intermediate list; they fit up to 2^24 points and 7 fine bits.
struct SortSplit { uint32_t kb, fine, hb; bool wide; };
static SortSplit sort_split(uint32_t c, uint64_t n) {   // n: one more than the largest point index an item can carry
    SortSplit sp;
    sp.kb = c - 1;
    sp.wide = n > (1ull << 24) || sp.kb > 15;
    const uint32_t fmax = sp.wide ? 10u : 7u;
    sp.fine = sp.kb < fmax ? sp.kb : fmax;
    sp.hb = sp.kb - sp.fine;
    if (sp.hb > 9) { sp.hb = 9; sp.fine = sp.kb - 9; }   // wide only: kb <= 19 keeps fine <= 10
    return sp;
}

__device__ __forceinline__ void load_scalar_words(const uint32_t *scalars, uint64_t i, uint32_t *s) {
    const uint4 *q = reinterpret_cast<const uint4 *>(scalars + i * 8);
    uint4 a = q[0], b = q[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
}

// dig[w * n_pad + i] = (|d| << 1) | (d < 0) for the signed digit d of window w of scalar i; 0 when d = 0; rows padded with
// zeros to n_pad.  The scalar's words are walked with compile-time register indices (a runtime word index would send the
// eight words through scratch memory: 3.2 ms instead of 0.4 at 2^24).
constexpr uint32_t DIGITS_PER_THREAD = 8;   // points per work-item: 8192 workgroups at 2^24 instead of 65536 tiny ones
// When c divides 256 (c = 8, 16) the window at bit 256 would hold nothing but the carry of the one below it: zero for every
// scalar below 2^255 (all the reference's callers pass representatives below r), but ONE bucket with half of all points
// for uniform 256-bit scalars.  So the top c-bit window is not recoded: its value u = raw + carry <= 2^c is taken
// unsigned and split over the last two window slots, u <= 2^(c-1) into slot W-2 (bucket u - 1) and larger values into slot
// W-1 (bucket u - 2^(c-1) - 1); both slots sit at bit 256 - c and the host fold adds 2^(c-1) times slot W-1's plain sum
// (msm_core.cuh run()).
__global__ __launch_bounds__(256) void msm_digits_kernel(const uint32_t *scalars, uint64_t n, uint64_t n_pad, uint32_t c, uint32_t W,
                                                         uint32_t *dig) {
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
    const bool split_top = (W - 1) * c == 256;
    const uint64_t i0 = (uint64_t)blockIdx.x * (256 * DIGITS_PER_THREAD) + threadIdx.x;
#pragma nounroll
    for (uint32_t q = 0; q < DIGITS_PER_THREAD; q++) {
        const uint64_t i = i0 + (uint64_t)q * 256;
        if (i >= n_pad) return;
        uint32_t s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (i < n) load_scalar_words(scalars, i, s);
        const uint32_t t[9] = {s[6], s[7], s[4], s[5], s[2], s[3], s[0], s[1], 0u};   // 32-bit words, least significant first
        uint32_t o = 0, w = 0, carry = 0;
        uint32_t *out = dig + i;
        auto emit = [&](uint32_t raw) {
            const uint32_t u = raw + carry;                 // 0 .. 2^c
            if (split_top && w + 2 == W) {
                const bool hi = u > half;
                out[(uint64_t)w * n_pad] = (!hi && u) ? (u << 1) : 0u;
                out[(uint64_t)(w + 1) * n_pad] = hi ? ((u - half) << 1) : 0u;
                carry = 0;
                w += 2;
                return;
            }
            const uint32_t neg = u > half;
            const uint32_t m = neg ? (mask + 1 - u) : u;    // |digit| <= 2^(c-1)
            carry = neg;
            out[(uint64_t)w * n_pad] = m ? ((m << 1) | neg) : 0u;
            w++;
        };
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint64_t v = (uint64_t)t[j] | ((uint64_t)t[j + 1] << 32);
            while (w < W && o < 32u * (j + 1)) {
                emit((uint32_t)(v >> (o - 32u * j)) & mask);
                o += c;
            }
        }
        while (w < W) emit(0u);   // a window that starts at bit 256 (c divides 256) holds only the carry
    }
}

// An item is (fine key bits, sign, point index).
template <class ITEM> struct ItemPack;
template <> struct ItemPack<uint32_t> {
    static __device__ __forceinline__ uint32_t make(uint32_t fine, uint32_t neg, uint64_t idx) { return (fine << 25) | (neg << 24) | (uint32_t)idx; }
    static __device__ __forceinline__ uint32_t fine(uint32_t it) { return it >> 25; }
    static __device__ __forceinline__ uint32_t entry(uint32_t it) { return (it & 0xffffffu) | ((it << 7) & 0x80000000u); }
};
template <> struct ItemPack<uint64_t> {
    static __device__ __forceinline__ uint64_t make(uint32_t fine, uint32_t neg, uint64_t idx) { return ((uint64_t)fine << 32) | (neg << 31) | (uint32_t)idx; }
    static __device__ __forceinline__ uint32_t fine(uint64_t it) { return (uint32_t)(it >> 32); }
    static __device__ __forceinline__ uint32_t entry(uint64_t it) { return (uint32_t)it; }
};
// entry of the sorted list handed to the accumulate kernel: sign << 31 | point index
// The intermediate list in memory: narrow items as they are; wide items split into the 32-bit entry and the 16-bit fine
// key (6 bytes written, 2 read by the key count, 6 read by the scatter: 14 bytes per item instead of 24).
template <class ITEM> struct ItemMem;
template <> struct ItemMem<uint32_t> {
    uint32_t *p;
    __device__ __forceinline__ uint32_t load(uint32_t i) const { return p[i]; }
    __device__ __forceinline__ uint32_t fine(uint32_t i) const { return p[i] >> 25; }
    __device__ __forceinline__ void store(uint32_t i, uint32_t it) const { p[i] = it; }
};
template <> struct ItemMem<uint64_t> {
    uint32_t *lo;
    uint16_t *hi;
    __device__ __forceinline__ uint64_t load(uint32_t i) const { return ((uint64_t)hi[i] << 32) | lo[i]; }
    __device__ __forceinline__ uint32_t fine(uint32_t i) const { return hi[i]; }
    __device__ __forceinline__ void store(uint32_t i, uint64_t it) const { lo[i] = (uint32_t)it; hi[i] = (uint16_t)(it >> 32); }
};

// exclusive scan of a[0 .. NMAX) in LDS by the whole workgroup (tmp: SORT_THREADS words); returns the total.
// Round 3: the scan across the 512 work-items is a wave-level shuffle scan (6 steps, no barrier) plus the 8 wave totals —
// three barriers per call where the Hillis-Steele loop over LDS took eighteen; every 8192-item chunk of the coarse and
// fine scatters calls it once, and a workgroup's time was mostly those barriers.
template <uint32_t NMAX>
__device__ __forceinline__ uint32_t block_scan_exclusive(uint32_t *a, uint32_t *tmp) {
    constexpr uint32_t PER = NMAX / SORT_THREADS;
    static_assert(PER * SORT_THREADS == NMAX, "scan size");
    static_assert(SORT_THREADS / 64 <= 64, "one wave scans the wave totals");
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) {
        v[i] = a[tid * PER + i];
        sum += v[i];
    }
    uint32_t inc = sum;   // inclusive scan inside the wave
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(inc, d);
        if (lane >= d) inc += x;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < SORT_THREADS / 64; w++) {
        const uint32_t t = tmp[w];
        if (w < wave) wbase += t;
        total += t;
    }
    uint32_t run = wbase + inc - sum;
#pragma unroll
    for (uint32_t i = 0; i < PER; i++) {
        a[tid * PER + i] = run;
        run += v[i];
    }
    __syncthreads();
    return total;
}

// pass A0 (msm_coarse_count_kernel): coarse histogram.  pass A1 (msm_coarse_kernel): scatter packed items into the runs reserved per coarse bin.
// grid = (ceil(n_pad / SORT_PPB), W); four digits (one uint4) per load.  The scatter stages every COARSE_CHUNK points in
// LDS in bin order and writes them out linearly, so a bin's share of the chunk leaves as one contiguous run; storing each
// item straight from its lane (4-byte stores into up to 64 runs per instruction) took 2.3 ms.  A staged item is 32 bits
// whatever the item width: position in the chunk (13 bits) | sign << 13 | fine key bits << 14.
constexpr uint32_t COARSE_CHUNK = 8192;
constexpr int COARSE_LOADS = COARSE_CHUNK / 4 / SORT_THREADS;   // uint4 loads per work-item and chunk
__device__ __forceinline__ uint32_t digit_of(const uint4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
// coarse bin of an encoded digit (SORT_MAX_COARSE = the dummy slot of zero digits, which contribute nothing, pippenger.rs:78)
__device__ __forceinline__ uint32_t coarse_bin(uint32_t enc, uint32_t fine_bits) { return enc ? (((enc >> 1) - 1) >> fine_bits) : SORT_MAX_COARSE; }

// A/B only (LW_HIP_MSM_BALLOT=1): rank of an item among the items of its wave with the same bin by wave-wide
// ballots — a match-any over the 10 bin bits, lane rank = popcount of the matching lanes below, one LDS atomic per
// distinct bin and wave instead of one per item.  Returns the rank the plain atomicAdd(&h[bin], 1) would have returned
// (up to the order inside the bin, which is irrelevant).
__device__ __forceinline__ uint32_t ballot_rank_add(uint32_t *h, uint32_t bin) {
    uint64_t m = __ballot(1);
#pragma unroll
    for (int b = 0; b < 10; b++) {
        const uint32_t bit = (bin >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    const uint32_t lane = __lane_id();
    const uint32_t below = __popcll(m & ((1ull << lane) - 1));
    uint32_t base = 0;
    if (below == 0) base = atomicAdd(&h[bin], (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    return base + below;
}

template <bool BALLOT = false>
__global__ __launch_bounds__(SORT_THREADS) void msm_coarse_count_kernel(const uint32_t *dig, uint64_t n_pad, uint32_t hb, uint32_t fine_bits,
                                                                       uint32_t folded, uint32_t *coarse_cnt) {
    __shared__ uint32_t h[SORT_MAX_COARSE + 1];
    const uint32_t NB = 1u << hb, w = blockIdx.y, tid = threadIdx.x;
    for (uint32_t b = tid; b <= SORT_MAX_COARSE; b += SORT_THREADS) h[b] = 0;
    __syncthreads();
    const uint64_t q0 = (uint64_t)blockIdx.x * (SORT_PPB / 4);
    const uint64_t q1 = min(n_pad / 4, q0 + SORT_PPB / 4);
    const uint4 *d4 = reinterpret_cast<const uint4 *>(dig + (uint64_t)w * n_pad);
    if constexpr (BALLOT) {
        for (uint64_t q = q0; q < q1; q += SORT_THREADS) {   // whole waves stay in the loop: ballots need every lane
            const uint4 v = q + tid < q1 ? d4[q + tid] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; e++) (void)ballot_rank_add(h, coarse_bin(digit_of(v, e), fine_bits));
        }
    } else {
    for (uint64_t q = q0 + tid; q < q1; q += SORT_THREADS) {
        const uint4 v = d4[q];
#pragma unroll
        for (int e = 0; e < 4; e++) atomicAdd(&h[coarse_bin(digit_of(v, e), fine_bits)], 1u);
    }
    }
    __syncthreads();
    for (uint32_t b = tid; b < NB; b += SORT_THREADS)
        if (h[b]) atomicAdd(&coarse_cnt[(folded ? 0u : (w << hb)) + b], h[b]);
}
template <class ITEM, bool BALLOT = false>
__global__ __launch_bounds__(SORT_THREADS) void msm_coarse_kernel(const uint32_t *dig, uint64_t n_pad, uint32_t hb, uint32_t fine_bits,
                                                                 uint64_t idx_stride, uint32_t folded, uint32_t win0, const uint32_t *coarse_off,
                                                                 uint32_t *coarse_cursor, ItemMem<ITEM> items) {
    __shared__ uint32_t h[SORT_MAX_COARSE + 1];   // + dummy slot for zero digits
    __shared__ uint32_t base[SORT_MAX_COARSE];    // next free slot of this workgroup's run per bin
    __shared__ uint32_t pre[SORT_MAX_COARSE];     // chunk-local exclusive offsets
    __shared__ uint32_t tmp[SORT_THREADS];
    __shared__ uint32_t buf[COARSE_CHUNK];
    __shared__ uint16_t bbin[COARSE_CHUNK];       // bin of every staged item
    const uint32_t NB = 1u << hb, w = blockIdx.y, tid = threadIdx.x;
    const uint32_t bin0 = folded ? 0u : (w << hb);   // folded: every window sorts into the one shared bucket set
    const uint64_t idx0 = (uint64_t)(win0 + w) * idx_stride;   // and its items point at the window's own copy of the points
    for (uint32_t b = tid; b <= SORT_MAX_COARSE; b += SORT_THREADS) h[b] = 0;
    __syncthreads();
    const uint64_t q0 = (uint64_t)blockIdx.x * (SORT_PPB / 4);
    const uint64_t q1 = min(n_pad / 4, q0 + SORT_PPB / 4);
    const uint4 *d4 = reinterpret_cast<const uint4 *>(dig + (uint64_t)w * n_pad);
    if constexpr (BALLOT) {
        for (uint64_t q = q0; q < q1; q += SORT_THREADS) {
            const uint4 v = q + tid < q1 ? d4[q + tid] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; e++) (void)ballot_rank_add(h, coarse_bin(digit_of(v, e), fine_bits));
        }
    } else {
    for (uint64_t q = q0 + tid; q < q1; q += SORT_THREADS) {
        const uint4 v = d4[q];
#pragma unroll
        for (int e = 0; e < 4; e++) atomicAdd(&h[coarse_bin(digit_of(v, e), fine_bits)], 1u);
    }
    }
    __syncthreads();
    for (uint32_t b = tid; b < SORT_MAX_COARSE; b += SORT_THREADS) {   // one global atomic per bin reserves this workgroup's run
        const uint32_t cnt = b < NB ? h[b] : 0;
        base[b] = cnt ? coarse_off[bin0 + b] + atomicAdd(&coarse_cursor[bin0 + b], cnt) : 0;
    }
    __syncthreads();
    const uint32_t fmask = (1u << fine_bits) - 1;
    for (uint64_t qc = q0; qc < q1; qc += COARSE_CHUNK / 4) {
        for (uint32_t b = tid; b <= SORT_MAX_COARSE; b += SORT_THREADS) h[b] = 0;
        __syncthreads();
        uint4 v[COARSE_LOADS];
        uint32_t rk[COARSE_LOADS][4];
#pragma unroll
        for (int l = 0; l < COARSE_LOADS; l++) {
            const uint64_t q = qc + l * SORT_THREADS + tid;
            v[l] = q < q1 ? d4[q] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int l = 0; l < COARSE_LOADS; l++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                rk[l][e] = BALLOT ? ballot_rank_add(h, coarse_bin(digit_of(v[l], e), fine_bits))
                                  : atomicAdd(&h[coarse_bin(digit_of(v[l], e), fine_bits)], 1u);
        __syncthreads();
        for (uint32_t b = tid; b < SORT_MAX_COARSE; b += SORT_THREADS) pre[b] = h[b];   // zero digits stay in the dummy slot
        __syncthreads();
        const uint32_t total = block_scan_exclusive<SORT_MAX_COARSE>(pre, tmp);
#pragma unroll
        for (int l = 0; l < COARSE_LOADS; l++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t enc = digit_of(v[l], e);
                if (enc) {
                    const uint32_t key = (enc >> 1) - 1, bin = key >> fine_bits;
                    const uint32_t pos = pre[bin] + rk[l][e];
                    buf[pos] = ((uint32_t)(l * SORT_THREADS + tid) * 4 + e) | ((enc & 1) << 13) | ((key & fmask) << 14);
                    bbin[pos] = (uint16_t)bin;
                }
            }
        __syncthreads();
        for (uint32_t e = tid; e < total; e += SORT_THREADS) {
            const uint32_t bin = bbin[e], st = buf[e];
            items.store(base[bin] + (e - pre[bin]), ItemPack<ITEM>::make(st >> 14, (st >> 13) & 1, idx0 + qc * 4 + (st & 0x1fffu)));
        }
        __syncthreads();
        for (uint32_t b = tid; b < SORT_MAX_COARSE; b += SORT_THREADS) base[b] += h[b];
        __syncthreads();
    }
}

// pass B: counting sort of every coarse bin by the fine key bits; writes the sorted (sign, index) entries; the per-key
// offsets come from a scan of the key counts.  A coarse bin is cut into sub-blocks of FINE_SUB items, one workgroup each,
// so a skewed scalar distribution (a prover's witness is mostly 0 / 1 / small values: one key of window 0 then holds a
// large share of all items, and a short top window puts all its items into a fraction of the keys) costs more
// workgroups, not one serial workgroup walking millions of items — the round-1 kernel (one workgroup per coarse bin) took
// 8 ms for a 4 M-item bin.  B0 counts the keys (LDS histogram per sub-block, one global atomic per non-empty key); B1
// takes its sub-block in chunks of FINE_CHUNK items, counting-sorts a chunk inside LDS, reserves a run per key with one
// global atomic and writes the chunk out linearly, so every key receives its share of the chunk as one contiguous run
// instead of single 4-byte stores scattered over the cursors.
constexpr uint32_t FINE_SUB = 131072;
constexpr uint32_t FINE_CHUNK = 8192;
constexpr int FINE_PER = FINE_CHUNK / SORT_THREADS;

// sub_off[bin] = number of sub-blocks before `bin` (exclusive scan of ceil(len / FINE_SUB)); returns false past the end
__device__ __forceinline__ bool fine_locate(const uint32_t *coarse_off, const uint32_t *sub_off, uint32_t CB, uint32_t blk,
                                            uint32_t &bin, uint32_t &i0, uint32_t &i1) {
    if (blk >= sub_off[CB]) return false;
    uint32_t lo = 0, hi = CB;   // largest bin with sub_off[bin] <= blk (sub-blocks are counted with at least one per bin)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sub_off[mid] <= blk) lo = mid; else hi = mid;
    }
    bin = lo;
    const uint32_t b0 = coarse_off[lo], b1 = coarse_off[lo + 1];
    i0 = b0 + (blk - sub_off[lo]) * FINE_SUB;
    i1 = min(b1, i0 + FINE_SUB);
    return true;
}

template <class ITEM>
__global__ __launch_bounds__(SORT_THREADS) void msm_fine_count_kernel(ItemMem<ITEM> items, const uint32_t *coarse_off, const uint32_t *sub_off,
                                                                     uint32_t CB, uint32_t fine_bits, uint32_t *key_cnt) {
    __shared__ uint32_t h[SORT_MAX_FINE];
    const uint32_t tid = threadIdx.x, NF = 1u << fine_bits;
    uint32_t bin, i0, i1;
    if (!fine_locate(coarse_off, sub_off, CB, blockIdx.x, bin, i0, i1)) return;
    for (uint32_t k = tid; k < NF; k += SORT_THREADS) h[k] = 0;
    __syncthreads();
    for (uint32_t i = i0 + tid; i < i1; i += 8 * SORT_THREADS) {
        uint32_t f[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t k = i + j * SORT_THREADS;
            f[j] = k < i1 ? items.fine(k) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (i + j * SORT_THREADS < i1) atomicAdd(&h[f[j]], 1u);
    }
    __syncthreads();
    for (uint32_t k = tid; k < NF; k += SORT_THREADS)
        if (h[k]) atomicAdd(&key_cnt[(bin << fine_bits) | k], h[k]);
}

template <class ITEM>
__global__ __launch_bounds__(SORT_THREADS) void msm_fine_kernel(ItemMem<ITEM> items, const uint32_t *coarse_off, const uint32_t *sub_off,
                                                               uint32_t CB, uint32_t fine_bits, const uint32_t *off, uint32_t *key_cursor,
                                                               uint32_t *sorted) {
    __shared__ uint32_t h[SORT_MAX_FINE];     // chunk-local counts
    __shared__ uint32_t pre[SORT_MAX_FINE];   // chunk-local exclusive offsets
    __shared__ uint32_t run[SORT_MAX_FINE];   // start of this chunk's run in `sorted`, per fine key
    __shared__ uint32_t tmp[SORT_THREADS];
    __shared__ ITEM buf[FINE_CHUNK];
    const uint32_t tid = threadIdx.x, NF = 1u << fine_bits;
    uint32_t bin, i0, i1;
    if (!fine_locate(coarse_off, sub_off, CB, blockIdx.x, bin, i0, i1)) return;
    const uint32_t key0 = bin << fine_bits;
    for (uint32_t c0 = i0; c0 < i1; c0 += FINE_CHUNK) {
        const uint32_t cn = min(FINE_CHUNK, i1 - c0);
        for (uint32_t k = tid; k < SORT_MAX_FINE; k += SORT_THREADS) h[k] = 0;
        __syncthreads();
        ITEM it[FINE_PER];
        uint32_t rk[FINE_PER];
#pragma unroll
        for (int j = 0; j < FINE_PER; j++) {
            const uint32_t e = j * SORT_THREADS + tid;
            it[j] = e < cn ? items.load(c0 + e) : (ITEM)0;
        }
#pragma unroll
        for (int j = 0; j < FINE_PER; j++)
            if (j * SORT_THREADS + tid < cn) rk[j] = atomicAdd(&h[ItemPack<ITEM>::fine(it[j])], 1u);
        __syncthreads();
        for (uint32_t k = tid; k < SORT_MAX_FINE; k += SORT_THREADS) {
            const uint32_t lcnt = h[k];   // > 0 only below NF
            run[k] = lcnt ? off[key0 + k] + atomicAdd(&key_cursor[key0 + k], lcnt) : 0;
            pre[k] = lcnt;
        }
        __syncthreads();
        block_scan_exclusive<SORT_MAX_FINE>(pre, tmp);
#pragma unroll
        for (int j = 0; j < FINE_PER; j++)
            if (j * SORT_THREADS + tid < cn) buf[pre[ItemPack<ITEM>::fine(it[j])] + rk[j]] = it[j];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < FINE_PER; j++) {
            const uint32_t e = j * SORT_THREADS + tid;
            if (e < cn) {
                const ITEM x = buf[e];
                const uint32_t f = ItemPack<ITEM>::fine(x);
                sorted[run[f] + (e - pre[f])] = ItemPack<ITEM>::entry(x);
            }
        }
        __syncthreads();
    }
    (void)NF;
}

// Exclusive scan over K keys in three launches (block partials -> top scan -> final).
//   mode 0:         in = counts[K]            -> out[K+1] = exclusive scan(counts)
//   mode = chunk>0: in = segment offsets[K+1] -> out[K+1] = exclusive scan(ceil(len/chunk))
// *maxlen receives the largest count / segment length seen.
constexpr uint32_t SCAN_BLOCK = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ uint32_t scan_len(const uint32_t *in, uint32_t k, int mode) { return mode ? (in[k + 1] - in[k]) : in[k]; }
// chunk 0 = mode 0.  In chunk mode an empty segment still gets one (empty) piece, so that every key owns an output slot
__device__ __forceinline__ uint32_t scan_val(uint32_t len, uint32_t chunk) { return chunk ? max(1u, (len + chunk - 1) / chunk) : len; }

__global__ __launch_bounds__(SCAN_BLOCK) void msm_scan_partial_kernel(const uint32_t *in, uint32_t K, int mode, uint32_t *bsum,
                                                                      uint32_t *bmax) {
    __shared__ uint32_t ssum[SCAN_BLOCK], smax[SCAN_BLOCK];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * SCAN_TILE + tid * SCAN_ITEMS;
    uint32_t sum = 0, mx = 0;
    for (uint32_t i = 0; i < SCAN_ITEMS; i++) {
        uint32_t k = base + i;
        if (k < K) {
            uint32_t len = scan_len(in, k, mode);
            mx = max(mx, len);
            sum += scan_val(len, mode);
        }
    }
    ssum[tid] = sum;
    smax[tid] = mx;
    __syncthreads();
    for (uint32_t d = SCAN_BLOCK / 2; d > 0; d >>= 1) {
        if (tid < d) {
            ssum[tid] += ssum[tid + d];
            smax[tid] = max(smax[tid], smax[tid + d]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        bsum[blockIdx.x] = ssum[0];
        bmax[blockIdx.x] = smax[0];
    }
}

// one workgroup: exclusive scan of the block sums in place, total -> out[K], max -> *maxlen
__global__ __launch_bounds__(1024) void msm_scan_top_kernel(uint32_t *bsum, const uint32_t *bmax, uint32_t nblocks, uint32_t *out,
                                                            uint32_t K, uint32_t *maxlen) {
    __shared__ uint32_t part[1024], pmax[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t b = tid * per, e = min(nblocks, b + per);
    uint32_t sum = 0, mx = 0;
    for (uint32_t k = b; k < e; k++) {
        sum += bsum[k];
        mx = max(mx, bmax[k]);
    }
    part[tid] = sum;
    pmax[tid] = mx;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0;
        uint32_t m = tid >= d ? pmax[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        pmax[tid] = max(pmax[tid], m);
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint32_t k = b; k < e; k++) {
        uint32_t v = bsum[k];
        bsum[k] = run;
        run += v;
    }
    if (tid == 1023) {
        out[K] = part[1023];
        *maxlen = pmax[1023];
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void msm_scan_final_kernel(const uint32_t *in, uint32_t *out, uint32_t K, int mode,
                                                                    const uint32_t *bsum) {
    __shared__ uint32_t ssum[SCAN_BLOCK];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * SCAN_TILE + tid * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
    for (uint32_t i = 0; i < SCAN_ITEMS; i++) {
        uint32_t k = base + i;
        v[i] = k < K ? scan_val(scan_len(in, k, mode), mode) : 0;
        sum += v[i];
    }
    ssum[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < SCAN_BLOCK; d <<= 1) {
        uint32_t x = tid >= d ? ssum[tid - d] : 0;
        __syncthreads();
        ssum[tid] += x;
        __syncthreads();
    }
    uint32_t run = bsum[blockIdx.x] + ssum[tid] - sum;
    for (uint32_t i = 0; i < SCAN_ITEMS; i++) {
        uint32_t k = base + i;
        if (k < K) out[k] = run;
        run += v[i];
    }
}

// ---- piece order: the accumulate work-items sorted by their number of additions --------------------------------
// A wave runs as long as its longest piece.  Bucket lengths of uniform scalars are Poisson (mean 16 at 2^20 points with
// c = 16: the longest of 64 neighbours is ~25), so work-items are handed their pieces in descending order of length: a
// counting sort of the piece ids by length (<= ORDER_BINS - 1) — every wave then holds pieces of one length.
//   out_off == nullptr: piece t is the key t, length seg_off[t+1] - seg_off[t]
//   else:               key k owns the pieces out_off[k] .. out_off[k+1]-1 (at least one), equal shares of its segment
// pass 0 counts the lengths, pass 1 (same walk) scatters: perm_t[rank] = piece id, perm_key[rank] = its key.
constexpr uint32_t ORDER_BINS = 130;    // lengths 0 .. 128 (MSM_CH <= 128) + slack
constexpr uint32_t ORDER_PER = 8;       // consecutive pieces per work-item
__device__ __forceinline__ uint32_t order_first_key(const uint32_t *out_off, uint32_t K, uint32_t t) {
    uint32_t lo = 0, hi = K;   // largest key with out_off[key] <= t
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (out_off[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}
template <bool SCATTER>
__global__ __launch_bounds__(SORT_THREADS) void msm_piece_order_kernel(const uint32_t *seg_off, const uint32_t *out_off, uint32_t K,
                                                                      uint32_t P, uint32_t *bin_cnt, uint32_t *bin_cursor,
                                                                      uint32_t *perm_t, uint32_t *perm_key) {
    __shared__ uint32_t h[ORDER_BINS], base[ORDER_BINS];
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < ORDER_BINS; b += SORT_THREADS) h[b] = 0;
    __syncthreads();
    const uint32_t t0 = (blockIdx.x * SORT_THREADS + tid) * ORDER_PER;
    uint32_t key = 0, o0 = 0, o1 = 0, s0 = 0, len = 0;
    uint32_t bin[ORDER_PER], rk[ORDER_PER], kk[ORDER_PER];
    if (t0 < P && out_off) {
        key = order_first_key(out_off, K, t0);
        o0 = out_off[key]; o1 = out_off[key + 1]; s0 = seg_off[key]; len = seg_off[key + 1] - s0;
    }
#pragma unroll
    for (uint32_t q = 0; q < ORDER_PER; q++) {
        const uint32_t t = t0 + q;
        bin[q] = ORDER_BINS;
        if (t >= P) continue;
        uint32_t L;
        if (out_off) {
            while (t >= o1) {   // every key owns at least one piece: at most one step per piece
                key++;
                o0 = o1; o1 = out_off[key + 1]; s0 = seg_off[key]; len = seg_off[key + 1] - s0;
            }
            const uint32_t np = o1 - o0, j = t - o0;
            L = (uint32_t)(((uint64_t)len * (j + 1)) / np) - (uint32_t)(((uint64_t)len * j) / np);
            kk[q] = key;
        } else {
            L = seg_off[t + 1] - seg_off[t];
            kk[q] = t;
        }
        bin[q] = ORDER_BINS - 1 - min(L, ORDER_BINS - 1);   // long pieces first
        rk[q] = atomicAdd(&h[bin[q]], 1u);
    }
    __syncthreads();
    if constexpr (!SCATTER) {
        for (uint32_t b = tid; b < ORDER_BINS; b += SORT_THREADS)
            if (h[b]) atomicAdd(&bin_cnt[b], h[b]);
    } else {
        if (tid == 0) {   // exclusive scan of the global bin counts (130 values)
            uint32_t run = 0;
            for (uint32_t b = 0; b < ORDER_BINS; b++) { base[b] = run; run += bin_cnt[b]; }
        }
        __syncthreads();
        for (uint32_t b = tid; b < ORDER_BINS; b += SORT_THREADS)
            base[b] += h[b] ? atomicAdd(&bin_cursor[b], h[b]) : 0;
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < ORDER_PER; q++)
            if (bin[q] < ORDER_BINS) {
                const uint32_t r = base[bin[q]] + rk[q];
                perm_t[r] = t0 + q;
                if (perm_key) perm_key[r] = kk[q];
            }
    }
}
// order_tmp: 2 * ORDER_BINS u32
void msm_launch_piece_order(Context &c, const uint32_t *seg_off, const uint32_t *out_off, uint32_t K, uint32_t P, uint32_t *order_tmp,
                            uint32_t *perm_t, uint32_t *perm_key, hipStream_t s) {
    (void)hipMemsetAsync(order_tmp, 0, 8 * ORDER_BINS, s);
    const uint32_t blocks = (P + SORT_THREADS * ORDER_PER - 1) / (SORT_THREADS * ORDER_PER);
    hipEvent_t pe = c.prof_begin(s);
    hipLaunchKernelGGL((msm_piece_order_kernel<false>), dim3(blocks), dim3(SORT_THREADS), 0, s, seg_off, out_off, K, P, order_tmp,
                       order_tmp + ORDER_BINS, perm_t, perm_key);
    hipLaunchKernelGGL((msm_piece_order_kernel<true>), dim3(blocks), dim3(SORT_THREADS), 0, s, seg_off, out_off, K, P, order_tmp,
                       order_tmp + ORDER_BINS, perm_t, perm_key);
    c.prof_end("msm_piece_order_kernel", pe, s);
}
size_t msm_order_tmp_bytes() { return 8 * ORDER_BINS; }
// Max points per accumulate work-item.  With pieces handed out in order of length the cut only has to bound the longest
// dependent chain: long pieces (64) save partial sums when there is plenty of work (2^24, c = 20: 49.1 ms against 50.0 at
// 32); when the items do not fill the machine a work-item's chain IS the kernel time, so short pieces and one more
// round win (2^14, c = 8: 1.60 ms at 16, 2.07 at 64).  LW_HIP_MSM_CH: tuning only.
uint32_t msm_ch(uint64_t items) {
    const char *e = tuning_env("LW_HIP_MSM_CH");   // read per call so that a test can force several rounds of partial sums
    const int env = e ? atoi(e) : 0;
    if (env) return (uint32_t)(env < 4 ? 4 : (env > 128 ? 128 : env));
    // below 2^20 items (c = 8: fewer than 2^15 points) a second round of short chains beats one of long ones: 2^10 0.95 ->
    // 0.82 ms, 2^12 0.98 -> 0.90, 2^14 1.35 -> 1.15 at 8 (4 is no better); 2^16 (c = 16, 2^20.1 items) keeps 16: 1.47 ms
    // against 1.53 at 8 (tools/ab_msm_ch_small.sh, profiles/r03_ab_msm_ch_small.txt)
    return items < (1ull << 20) ? 8u : items < (1ull << 22) ? 16u : items < (1ull << 27) ? 32u : 64u;
}
int msm_piece_order_enabled() {
    static int v = [] { const char *e = tuning_env("LW_HIP_MSM_ORDER"); return e ? atoi(e) : 1; }();
    return v;
}

uint32_t msm_sort_coarse_bins(uint32_t c, uint32_t W, uint64_t n) { return W << sort_split(c, n).hb; }   // folded: W = 1, n = W * stride
uint32_t msm_max_window_bits() { return MSM_MAX_C; }
// row length of the digit matrix: a multiple of 8 (uint4 loads) that is not a power of two, so that the W rows a wave
// writes do not all fall on the same memory channel
uint64_t msm_sort_padded_points(uint64_t n) { return ((n + 7) & ~(uint64_t)7) + 1032; }
// level 0: the W x n_pad digit matrix of all windows (shared by the window slices of msm_core.cuh run())
void msm_launch_digits(Context &c, const uint32_t *scalars, uint64_t n, uint32_t cb, uint32_t W, uint32_t *dig, hipStream_t s) {
    const uint64_t n_pad = msm_sort_padded_points(n);
    hipEvent_t pe = c.prof_begin(s);
    hipLaunchKernelGGL(msm_digits_kernel, dim3((uint32_t)((n_pad + 256 * DIGITS_PER_THREAD - 1) / (256 * DIGITS_PER_THREAD))), dim3(256), 0, s,
                       scalars, n, n_pad, cb, W, dig);
    c.prof_end("msm_digits_kernel", pe, s);
}
// `dig`: row 0 = the first of the W windows sorted here (a slice of the matrix); win0: that window's number in the MSM
template <class ITEM>
static void launch_sort_t(Context &c, uint64_t n, uint32_t cb, uint32_t W, const SortSplit &sp, uint32_t CB,
                          const uint32_t *dig, uint32_t *coarse_cnt, uint32_t *coarse_off, uint32_t *coarse_cursor, ItemMem<ITEM> items,
                          uint32_t *sorted, uint32_t *off, uint32_t K, uint32_t *maxlen, uint32_t *scan_tmp, uint32_t *sub_off,
                          uint32_t *key_cnt, uint32_t *key_cursor, uint64_t fold_stride, uint32_t win0, hipStream_t s) {
    const uint64_t n_pad = msm_sort_padded_points(n);
    const uint32_t folded = fold_stride != 0;
    const uint32_t fine = sp.fine, hb = sp.hb;
    const dim3 grid((uint32_t)((n_pad + SORT_PPB - 1) / SORT_PPB), W);
    hipEvent_t pe = c.prof_begin(s);
    static const bool ballot = [] { const char *e = tuning_env("LW_HIP_MSM_BALLOT"); return e && atoi(e) == 1; }();   // A/B only
    if (ballot) hipLaunchKernelGGL((msm_coarse_count_kernel<true>), grid, dim3(SORT_THREADS), 0, s, (const uint32_t *)dig, n_pad, hb, fine, folded, coarse_cnt);
    else hipLaunchKernelGGL((msm_coarse_count_kernel<false>), grid, dim3(SORT_THREADS), 0, s, (const uint32_t *)dig, n_pad, hb, fine, folded, coarse_cnt);
    c.prof_end("msm_coarse_kernel<count>", pe, s);
    msm_launch_scan(coarse_cnt, coarse_off, CB, 0, maxlen + 1, scan_tmp, s);   // maxlen[1]: coarse max (unused)
    pe = c.prof_begin(s);
    if (ballot) hipLaunchKernelGGL((msm_coarse_kernel<ITEM, true>), grid, dim3(SORT_THREADS), 0, s, (const uint32_t *)dig, n_pad, hb, fine, fold_stride, folded,
                                   win0, (const uint32_t *)coarse_off, coarse_cursor, items);
    else hipLaunchKernelGGL((msm_coarse_kernel<ITEM, false>), grid, dim3(SORT_THREADS), 0, s, (const uint32_t *)dig, n_pad, hb, fine, fold_stride, folded,
                            win0, (const uint32_t *)coarse_off, coarse_cursor, items);
    c.prof_end("msm_coarse_kernel<scatter>", pe, s);
    // level B: sub-blocks of the coarse bins -> key counts -> key offsets (+ the longest bucket) -> sorted index list
    msm_launch_scan(coarse_off, sub_off, CB, (int)FINE_SUB, maxlen + 1, scan_tmp, s);
    const uint32_t UB = CB + (uint32_t)(((uint64_t)n * W + FINE_SUB - 1) / FINE_SUB);   // >= sub_off[CB]; surplus workgroups exit
    pe = c.prof_begin(s);
    hipLaunchKernelGGL((msm_fine_count_kernel<ITEM>), dim3(UB), dim3(SORT_THREADS), 0, s, items, (const uint32_t *)coarse_off,
                       (const uint32_t *)sub_off, CB, fine, key_cnt);
    c.prof_end("msm_fine_count_kernel", pe, s);
    msm_launch_scan(key_cnt, off, K, 0, maxlen, scan_tmp, s);
    pe = c.prof_begin(s);
    hipLaunchKernelGGL((msm_fine_kernel<ITEM>), dim3(UB), dim3(SORT_THREADS), 0, s, items, (const uint32_t *)coarse_off,
                       (const uint32_t *)sub_off, CB, fine, (const uint32_t *)off, key_cursor, sorted);
    c.prof_end("msm_fine_kernel", pe, s);
}
// dig: W rows of padded(n) u32; coarse_cnt / coarse_cursor: CB + 1 zeroed u32 each; coarse_off, sub_off: CB + 1; items: n*W u64;
// key_cnt / key_cursor: K zeroed u32 each; off: K + 1; maxlen: zeroed.  K = W << (cb - 1).
void msm_launch_sort(Context &c, const uint32_t *dig, uint64_t n, uint32_t cb, uint32_t W, uint32_t *coarse_cnt,
                     uint32_t *coarse_off, uint32_t *coarse_cursor, uint64_t *items, uint32_t *sorted, uint32_t *off, uint32_t K,
                     uint32_t *maxlen, uint32_t *scan_tmp, uint32_t *sub_off, uint32_t *key_cnt, uint32_t *key_cursor, uint64_t fold_stride,
                     uint64_t win0, hipStream_t s) {
    // fold_stride != 0 (lw_hip_srs_* with window-shifted copies of the points): one bucket set of 2^(cb-1) keys for all W
    // windows; the item of window w and scalar i points at row w * fold_stride + i
    const SortSplit sp = sort_split(cb, fold_stride ? (uint64_t)W * fold_stride : n);
    const uint32_t CB = (fold_stride ? 1u : W) << sp.hb;
    if (!sp.wide)
        launch_sort_t<uint32_t>(c, n, cb, W, sp, CB, dig, coarse_cnt, coarse_off, coarse_cursor, ItemMem<uint32_t>{(uint32_t *)items},
                                sorted, off, K, maxlen, scan_tmp, sub_off, key_cnt, key_cursor, fold_stride, (uint32_t)win0, s);
    else   // the 8 * n * W bytes of `items` hold n * W entries followed by n * W fine keys
        launch_sort_t<uint64_t>(c, n, cb, W, sp, CB, dig, coarse_cnt, coarse_off, coarse_cursor,
                                ItemMem<uint64_t>{(uint32_t *)items, (uint16_t *)((uint32_t *)items + n * W)}, sorted, off, K, maxlen,
                                scan_tmp, sub_off, key_cnt, key_cursor, fold_stride, (uint32_t)win0, s);
}
// scratch: 2 * ceil(K / SCAN_TILE) u32 (block sums, block maxima)
void msm_launch_scan(const uint32_t *in, uint32_t *out, uint32_t K, int mode, uint32_t *maxlen, uint32_t *scratch, hipStream_t s) {
    const uint32_t nblocks = (K + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t *bsum = scratch, *bmax = scratch + nblocks;
    hipLaunchKernelGGL(msm_scan_partial_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, in, K, mode, bsum, bmax);
    hipLaunchKernelGGL(msm_scan_top_kernel, dim3(1), dim3(1024), 0, s, bsum, bmax, nblocks, out, K, maxlen);
    hipLaunchKernelGGL(msm_scan_final_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, in, out, K, mode, bsum);
}
size_t msm_scan_scratch_bytes(uint32_t K) { return 8 * (size_t)((K + SCAN_TILE - 1) / SCAN_TILE) + 256; }

uint32_t msm_g_log() {
    static uint32_t g = [] { const char *e = tuning_env("LW_HIP_MSM_GLOG"); int v = e ? atoi(e) : 3; return (uint32_t)(v < 1 ? 1 : (v > 6 ? 6 : v)); }();
    return g;
}
uint64_t msm_quad_max_lanes() {   // tuning: LW_HIP_MSM_QUAD = log2 of the widest level (in lanes) that takes the quad kernels, 0 = none
    const char *e = tuning_env("LW_HIP_MSM_QUAD");   // read per call so that a test can sweep it
    if (!e) return ~(uint64_t)0;                     // not set: the group's own default (msm_core.cuh launch_group_sum)
    const int v = atoi(e);
    return v <= 0 ? (uint64_t)0 : (uint64_t)1 << (v > 30 ? 30 : v);
}
uint64_t msm_accumulate_quad_max_lanes() {   // tuning: LW_HIP_MSM_ACCQ = log2 of the widest accumulate launch (in lanes) on the quad kernel, 0 = none; read per call
    const char *e = tuning_env("LW_HIP_MSM_ACCQ");
    const int v = e ? atoi(e) : 19;
    return v <= 0 ? (uint64_t)0 : (uint64_t)1 << (v > 30 ? 30 : v);
}
int msm_waves_per_simd() {
    static int w = [] { const char *e = tuning_env("LW_HIP_MSM_WAVES"); int v = e ? atoi(e) : 2; return v == 3 ? 3 : 2; }();
    return w;
}

int msm_run_bls12381_g1(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_points, size_t n, void *out, int affine,
                      hipEvent_t points_ready);
int msm_run_bn254_g1(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_points, size_t n, void *out, int affine,
                      hipEvent_t points_ready);
int msm_run_bn254_g2(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_points, size_t n, void *out, int affine,
                      hipEvent_t points_ready);
int msm_run_bls12381_g2(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_points, size_t n, void *out, int affine,
                      hipEvent_t points_ready);
int msm_normalize_bls12381_g1(Context &c, hipStream_t s, const void *d_in, size_t n, void *d_out);
int msm_normalize_bn254_g1(Context &c, hipStream_t s, const void *d_in, size_t n, void *d_out);
int msm_normalize_bn254_g2(Context &c, hipStream_t s, const void *d_in, size_t n, void *d_out);
int msm_normalize_bls12381_g2(Context &c, hipStream_t s, const void *d_in, size_t n, void *d_out);

int msm_normalize_device(Context &c, lw_curve_t curve, const void *d_in, size_t n, void *d_out, hipStream_t stream);
size_t msm_affine_bytes_bls12381_g1(size_t n);
size_t msm_affine_bytes_bn254_g1(size_t n);
size_t msm_affine_bytes_bn254_g2(size_t n);
size_t msm_affine_bytes_bls12381_g2(size_t n);
int msm_fold_build_bls12381_g1(Context &c, hipStream_t s, void *d_rows, size_t n, uint32_t cbits);
int msm_fold_build_bn254_g1(Context &c, hipStream_t s, void *d_rows, size_t n, uint32_t cbits);
int msm_fold_build_bn254_g2(Context &c, hipStream_t s, void *d_rows, size_t n, uint32_t cbits);
int msm_fold_build_bls12381_g2(Context &c, hipStream_t s, void *d_rows, size_t n, uint32_t cbits);
int msm_fold_build(Context &c, lw_curve_t curve, void *d_rows, size_t n, uint32_t cbits, hipStream_t stream) {
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: return msm_fold_build_bls12381_g1(c, stream, d_rows, n, cbits);
        case LW_CURVE_BN254_G1: return msm_fold_build_bn254_g1(c, stream, d_rows, n, cbits);
        case LW_CURVE_BN254_G2: return msm_fold_build_bn254_g2(c, stream, d_rows, n, cbits);
        case LW_CURVE_BLS12_381_G2: return msm_fold_build_bls12381_g2(c, stream, d_rows, n, cbits);
        default: return LW_ERR_BAD_ARG;
    }
}
// bytes of the device-resident affine form of n points (rows may be padded, ec.cuh aff_stride)
size_t msm_affine_bytes(lw_curve_t curve, size_t n) {
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: return msm_affine_bytes_bls12381_g1(n);
        case LW_CURVE_BN254_G1: return msm_affine_bytes_bn254_g1(n);
        case LW_CURVE_BN254_G2: return msm_affine_bytes_bn254_g2(n);
        case LW_CURVE_BLS12_381_G2: return msm_affine_bytes_bls12381_g2(n);
        default: return 0;
    }
}
int ec_add_outer_bls12381_g1(Context &c, hipStream_t s, const void *d_rows, uint32_t m, const void *d_cols, uint32_t k, void *d_out);
int ec_add_outer_bn254_g1(Context &c, hipStream_t s, const void *d_rows, uint32_t m, const void *d_cols, uint32_t k, void *d_out);
int ec_add_outer_bn254_g2(Context &c, hipStream_t s, const void *d_rows, uint32_t m, const void *d_cols, uint32_t k, void *d_out);
int ec_add_outer_bls12381_g2(Context &c, hipStream_t s, const void *d_rows, uint32_t m, const void *d_cols, uint32_t k, void *d_out);

int ec_add_outer_device(Context &c, lw_curve_t curve, const void *d_rows, uint32_t m, const void *d_cols, uint32_t k, void *d_out,
                        hipStream_t stream) {
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: return ec_add_outer_bls12381_g1(c, stream, d_rows, m, d_cols, k, d_out);
        case LW_CURVE_BN254_G1: return ec_add_outer_bn254_g1(c, stream, d_rows, m, d_cols, k, d_out);
        case LW_CURVE_BN254_G2: return ec_add_outer_bn254_g2(c, stream, d_rows, m, d_cols, k, d_out);
        case LW_CURVE_BLS12_381_G2: return ec_add_outer_bls12381_g2(c, stream, d_rows, m, d_cols, k, d_out);
        default: set_error("bad curve %d", (int)curve); return LW_ERR_BAD_ARG;
    }
}

// smallest input (log2 points) for which normalising first pays, per group (tools/ab_msm_curve.py, LW_HIP_MSM_NORM_MIN)
static int msm_normalize_min_log2(lw_curve_t curve) {
    switch (curve) {   // measured break-even (normalise + mixed additions against projective additions)
        case LW_CURVE_BN254_G2: return 18;        // 2^20: 12.3 -> 11.2 ms, 2^21: 18.7 -> 16.6 (mixed addition on the isomorphic curve)
        case LW_CURVE_BLS12_381_G2: return 19;    // 2^20: 24.2 -> 22.6 ms, 2^21: 37.2 -> 34.6; 2^18: 12.7 against 13.4
        case LW_CURVE_BN254_G1: return 19;        // 2^18: 1.61 against 1.59 ms, 2^19: 2.24 -> 2.15, 2^20: 3.27 -> 3.11
        default: return 19;                       // BLS12-381 G1: 2^18 2.87 against 2.92 ms, 2^19 4.02 -> 3.90, 2^20 6.10 -> 5.67, 2^21 10.0 -> 9.14
    }
}

// affine_points: d_points are affine pairs produced by msm_normalize_device (2 field elements per row)
// The context's side stream (MSM: normalisation beside the sort; sharded NTT: exchanges beside the kernels).
int ensure_aux_stream(Context &c) {
    if (c.aux_stream) return LW_OK;
    // lowest priority: the sort on the caller's stream (2.7 ms alone) keeps its pace and the normalisation fills the
    // issue slots it leaves; with equal priorities the sort kernels queued behind the normalisation's workgroups
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // (hipDeviceGetStreamPriorityRange: numerically lower = higher priority)
    // (Measured, profiles/r03_ab_msm_normcus.txt: a side stream restricted to 64-192 CUs, so that the normalisation takes a smaller
    // share of HBM from the sort on the critical path — the sort kernels then run at their standalone times but the
    // normalisation takes 3.5-7.9 ms and the MSM 47.2-52.4 ms against 45.4.)
    if (hipStreamCreateWithPriority(&c.aux_stream, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority(&c.aux_hi, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&c.aux_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c.aux_join, hipEventDisableTiming) != hipSuccess) {
        set_error("cannot create the side stream");
        return LW_ERR_LAUNCH;
    }
    return LW_OK;
}

int msm_device(Context &c, lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n, void *out_host,
               hipStream_t stream, int scalars_montgomery, int affine_points, const void *h_points) {
    // h_points (host-buffer entry points): the points are still in host memory and d_points is the device buffer they go to.
    // The sort needs the scalars only, so it is enqueued first and the upload runs under it (msm_after_sort below);
    // without a normalisation the points are needed by the first kernel after the sort and go up front.
    const size_t pbytes = lw_hip_curve_point_bytes(curve);
    bool upload_pending = h_points != nullptr && n != 0;
    if (scalars_montgomery && n) {
        if (c.msm_scalars.ensure(n * 32)) return LW_ERR_ALLOC;
        const int bn = (curve == LW_CURVE_BN254_G1 || curve == LW_CURVE_BN254_G2);
        hipEvent_t pe = c.prof_begin(stream);
        msm_launch_scalar_prep(bn, d_scalars, c.msm_scalars.p, n, stream);
        c.prof_end("msm_scalars_from_mont_kernel", pe, stream);
        LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
        d_scalars = (const uint64_t *)c.msm_scalars.p;
    }
    // Large projective inputs are normalised first (batch inversion, ~3 ms at 2^24) so that the accumulation can use
    // the mixed addition, one-line gathers and — BLS12-381 G1, BN254 G2 — the cheaper isomorphic curve (~10 ms less at
    // 2^24); below the per-group threshold (msm_normalize_min_log2) the conversion costs more than it saves.
    // LW_HIP_MSM_NORMALIZE=0 keeps the projective path.
    static const bool auto_norm = [] { const char *e = tuning_env("LW_HIP_MSM_NORMALIZE"); return !e || atoi(e) != 0; }();
    hipEvent_t join = nullptr;
    static const int norm_min_env = [] { const char *e = tuning_env("LW_HIP_MSM_NORM_MIN"); return e ? atoi(e) : -1; }();   // tuning only
    const int norm_min_log2 = norm_min_env >= 0 ? norm_min_env : msm_normalize_min_log2(curve);
    if (!affine_points && auto_norm && n >= ((size_t)1 << norm_min_log2)) {
        const size_t aff_bytes = msm_affine_bytes(curve, n);
        if (c.msm_affine.ensure(aff_bytes)) return LW_ERR_ALLOC;
        // The normalisation reads only the points and the bucket sort only the scalars, so the normalisation runs on a
        // side stream beside the sort and the main stream joins it just before the first accumulation launch (both are
        // memory-bound: side by side they take about the sum of their standalone times less 0.5 ms, LW_HIP_MSM_SIDE).
        static const bool side = [] { const char *e = tuning_env("LW_HIP_MSM_SIDE"); return !e || atoi(e) != 0; }();   // A/B only
        if (!side) {
            if (upload_pending) {
                LW_HIP_CHECK(hipMemcpyAsync((void *)d_points, h_points, n * pbytes, hipMemcpyHostToDevice, stream), LW_ERR_LAUNCH);
                upload_pending = false;
            }
            int rc = msm_normalize_device(c, curve, d_points, n, c.msm_affine.p, stream);
            if (rc) return rc;
        } else if (upload_pending) {
            int rc = ensure_aux_stream(c);
            if (rc) return rc;
            LW_HIP_CHECK(hipEventRecord(c.aux_fork, stream), LW_ERR_LAUNCH);
            LW_HIP_CHECK(hipStreamWaitEvent(c.aux_stream, c.aux_fork, 0), LW_ERR_LAUNCH);
            const void *src = d_points;
            // in chunks of 2^20 points: chunk k is normalised (second side stream) while chunk k + 1 is on the bus, so that what
            // is left after the last byte arrives is one chunk's normalisation, not the whole set's
            c.msm_after_sort = [&c, curve, src, h_points, n, pbytes]() -> int {
                const size_t CHUNK = (size_t)1 << 20, astride = msm_affine_bytes(curve, 1);
                hipEvent_t ev = nullptr;
                if (!c.sync_pool.empty()) { ev = c.sync_pool.back(); c.sync_pool.pop_back(); }
                else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return LW_ERR_LAUNCH; }
                int r = LW_OK;
                for (size_t off = 0; off < n && !r; off += CHUNK) {
                    const size_t m = n - off < CHUNK ? n - off : CHUNK;
                    const char *d = (const char *)src + off * pbytes;
                    if (hipMemcpyAsync((void *)d, (const char *)h_points + off * pbytes, m * pbytes, hipMemcpyHostToDevice, c.aux_stream) != hipSuccess ||
                        hipEventRecord(ev, c.aux_stream) != hipSuccess || hipStreamWaitEvent(c.aux_hi, ev, 0) != hipSuccess) {
                        set_error("upload of the points failed");
                        r = LW_ERR_LAUNCH;
                        break;
                    }
                    r = msm_normalize_device(c, curve, d, m, (char *)c.msm_affine.p + off * astride, c.aux_hi);
                }
                c.sync_pool.push_back(ev);
                if (r) return r;
                LW_HIP_CHECK(hipEventRecord(c.aux_join, c.aux_hi), LW_ERR_LAUNCH);
                return LW_OK;
            };
            upload_pending = false;
            join = c.aux_join;
        } else {
        {
            int rc = ensure_aux_stream(c);
            if (rc) return rc;
        }
        LW_HIP_CHECK(hipEventRecord(c.aux_fork, stream), LW_ERR_LAUNCH);
        LW_HIP_CHECK(hipStreamWaitEvent(c.aux_stream, c.aux_fork, 0), LW_ERR_LAUNCH);
        // (Enqueueing the normalisation behind the sort, or behind the digit kernel only, measured worse: whichever sort
        // kernel first overlaps the normalisation takes ~2-3 ms longer, and a later start only moves that cost: 56.9 ms with
        // the normalisation first, 58.2 / 59.5 ms with the sort / the digit kernel first.)
        {
            int rc = msm_normalize_device(c, curve, d_points, n, c.msm_affine.p, c.aux_stream);
            if (rc) return rc;
            LW_HIP_CHECK(hipEventRecord(c.aux_join, c.aux_stream), LW_ERR_LAUNCH);
        }
        join = c.aux_join;
        }
        d_points = c.msm_affine.p;
        affine_points = 1;
    }
    if (upload_pending)   // no normalisation: the accumulation reads the rows as they are
        LW_HIP_CHECK(hipMemcpyAsync((void *)d_points, h_points, n * pbytes, hipMemcpyHostToDevice, stream), LW_ERR_LAUNCH);
    int rc;
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: rc = msm_run_bls12381_g1(c, stream, d_scalars, d_points, n, out_host, affine_points, join); break;
        case LW_CURVE_BN254_G1: rc = msm_run_bn254_g1(c, stream, d_scalars, d_points, n, out_host, affine_points, join); break;
        case LW_CURVE_BN254_G2: rc = msm_run_bn254_g2(c, stream, d_scalars, d_points, n, out_host, affine_points, join); break;
        case LW_CURVE_BLS12_381_G2: rc = msm_run_bls12381_g2(c, stream, d_scalars, d_points, n, out_host, affine_points, join); break;
        default: set_error("bad curve %d", (int)curve); return LW_ERR_BAD_ARG;
    }
    c.msm_after_sort = nullptr;                                  // (a run that failed before its sort never took it)
    if (rc && join) {   // do not leave the side streams running into a failed call's buffers
        (void)hipStreamSynchronize(c.aux_stream);
        if (c.aux_hi) (void)hipStreamSynchronize(c.aux_hi);
    }
    return rc;
}

// ---- sharded MSM (comm.hip): per-curve dispatch of the three phases around the bucket-slice exchange ----
#define LW_SHARD_DECL(SUFFIX)                                                                                                                 \
    int msm_shard_accumulate_##SUFFIX(Context &c, hipStream_t s, const uint64_t *d_scalars, const void *d_aff, size_t n, uint32_t cbits, char **buckets); \
    int msm_shard_reduce_##SUFFIX(Context &c, hipStream_t s, const char *recv, uint32_t G, uint32_t cbits, char *d_sa);                        \
    void msm_shard_combine_##SUFFIX(const char *sa_all, uint32_t G, uint32_t cbits, void *out);
LW_SHARD_DECL(bls12381_g1) LW_SHARD_DECL(bn254_g1) LW_SHARD_DECL(bn254_g2) LW_SHARD_DECL(bls12381_g2)
#undef LW_SHARD_DECL
uint32_t msm_window_bits_for(size_t n) { return pick_window(n); }
// local pairs -> dense bucket array in the context workspace.  The points are normalised first whatever n is, so that every
// rank's buckets live on the same curve model (the isomorphic one where the curve has it) and can be added across ranks.
int msm_shard_accumulate(Context &c, lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n, uint32_t cbits, hipStream_t s,
                         char **buckets) {
    if (n) {
        if (c.msm_affine.ensure(msm_affine_bytes(curve, n))) return LW_ERR_ALLOC;
        int rc = msm_normalize_device(c, curve, d_points, n, c.msm_affine.p, s);
        if (rc) return rc;
    }
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: return msm_shard_accumulate_bls12381_g1(c, s, d_scalars, c.msm_affine.p, n, cbits, buckets);
        case LW_CURVE_BN254_G1: return msm_shard_accumulate_bn254_g1(c, s, d_scalars, c.msm_affine.p, n, cbits, buckets);
        case LW_CURVE_BN254_G2: return msm_shard_accumulate_bn254_g2(c, s, d_scalars, c.msm_affine.p, n, cbits, buckets);
        case LW_CURVE_BLS12_381_G2: return msm_shard_accumulate_bls12381_g2(c, s, d_scalars, c.msm_affine.p, n, cbits, buckets);
        default: set_error("bad curve %d", (int)curve); return LW_ERR_BAD_ARG;
    }
}
int msm_shard_reduce(Context &c, lw_curve_t curve, const char *recv, uint32_t G, uint32_t cbits, char *d_sa, hipStream_t s) {
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: return msm_shard_reduce_bls12381_g1(c, s, recv, G, cbits, d_sa);
        case LW_CURVE_BN254_G1: return msm_shard_reduce_bn254_g1(c, s, recv, G, cbits, d_sa);
        case LW_CURVE_BN254_G2: return msm_shard_reduce_bn254_g2(c, s, recv, G, cbits, d_sa);
        case LW_CURVE_BLS12_381_G2: return msm_shard_reduce_bls12381_g2(c, s, recv, G, cbits, d_sa);
        default: set_error("bad curve %d", (int)curve); return LW_ERR_BAD_ARG;
    }
}
int msm_shard_combine(lw_curve_t curve, const char *sa_all, uint32_t G, uint32_t cbits, void *out) {
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: msm_shard_combine_bls12381_g1(sa_all, G, cbits, out); return LW_OK;
        case LW_CURVE_BN254_G1: msm_shard_combine_bn254_g1(sa_all, G, cbits, out); return LW_OK;
        case LW_CURVE_BN254_G2: msm_shard_combine_bn254_g2(sa_all, G, cbits, out); return LW_OK;
        case LW_CURVE_BLS12_381_G2: msm_shard_combine_bls12381_g2(sa_all, G, cbits, out); return LW_OK;
        default: set_error("bad curve %d", (int)curve); return LW_ERR_BAD_ARG;
    }
}

// Sum of a few projective points on the host, normalised like every MSM result (the combine step of the sharded MSM).
template <class C>
static void sum_points_host_t(const void *pts, size_t n, void *out) {
    Point<C> acc = pt_identity<C>();
    for (size_t i = 0; i < n; i++) acc = pt_add<C>(acc, pt_load<C>((const char *)pts + i * 3 * C::B::BYTES));
    pt_store<C>(out, pt_to_affine<C>(acc));
}
int msm_sum_points_host(lw_curve_t curve, const void *pts, size_t n, void *out) {
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: sum_points_host_t<Bls12381G1>(pts, n, out); return LW_OK;
        case LW_CURVE_BN254_G1: sum_points_host_t<Bn254G1>(pts, n, out); return LW_OK;
        case LW_CURVE_BN254_G2: sum_points_host_t<Bn254G2>(pts, n, out); return LW_OK;
        case LW_CURVE_BLS12_381_G2: sum_points_host_t<Bls12381G2>(pts, n, out); return LW_OK;
        default: set_error("bad curve %d", (int)curve); return LW_ERR_BAD_ARG;
    }
}

int msm_normalize_device(Context &c, lw_curve_t curve, const void *d_in, size_t n, void *d_out, hipStream_t stream) {
    switch (curve) {
        case LW_CURVE_BLS12_381_G1: return msm_normalize_bls12381_g1(c, stream, d_in, n, d_out);
        case LW_CURVE_BN254_G1: return msm_normalize_bn254_g1(c, stream, d_in, n, d_out);
        case LW_CURVE_BN254_G2: return msm_normalize_bn254_g2(c, stream, d_in, n, d_out);
        case LW_CURVE_BLS12_381_G2: return msm_normalize_bls12381_g2(c, stream, d_in, n, d_out);
        default: set_error("bad curve %d", (int)curve); return LW_ERR_BAD_ARG;
    }
}

}  // namespace lw
