// Keccak-256 Merkle commitment of LDE columns on the device (SURVEY 8f "next" #1).
//
// What the STARK prover does right after every LDE (provers/stark/src/prover.rs:229-244): bit-reverse-permute each
// column (:232-234), turn columns into rows (columns2rows) and build BatchedMerkleTree<BatchKeccak256Backend> over the
// rows (crypto/src/merkle_tree/merkle.rs:31-56):
//     leaf_i   = Keccak256( as_bytes(col_0[bitrev(i)]) || as_bytes(col_1[bitrev(i)]) || ... )      field_element_vector.rs:41-49
//     parent   = Keccak256( left || right )                                                       field_element_vector.rs:51-58
//     nodes    = [inner nodes, root first | leaves]                                               utils.rs:44-72
// as_bytes is the raw Montgomery value big-endian (math/src/field/fields/montgomery_backed_prime_fields.rs:367-373).
// Here the permutation and the transposition are folded into the leaf kernel's gather, so the LDE output never
// leaves HBM and no permuted / row-major copy is materialised.
// Keccak-256 comes from the `sha3` crate (0.10) in the reference; this is the published Keccak-f[1600] with rate 136
// and the original 0x01..0x80 padding.
#include <algorithm>
#include <stdlib.h>
#include "context.h"

namespace lw {

__constant__ uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// Keccak-f[1600] on explicit 32-bit halves: a 64-bit rotation by a constant is two v_alignbit_b32, chi's a ^ (~b & c) one
// v_bitop3_b32 per half and theta's column parities three-input XORs (written on uint64_t the compiler emitted 64-bit
// shift pairs and v_bfi + v_xor: ~300 instructions per round against ~190 here).
struct U64H { uint32_t lo, hi; };
__device__ __forceinline__ U64H rotl64h(U64H x, int n) {   // n: compile-time constant, 1 .. 63
    if (n == 32) return U64H{x.hi, x.lo};
    if (n < 32) return U64H{__builtin_amdgcn_alignbit(x.lo, x.hi, 32 - n), __builtin_amdgcn_alignbit(x.hi, x.lo, 32 - n)};
    return U64H{__builtin_amdgcn_alignbit(x.hi, x.lo, 64 - n), __builtin_amdgcn_alignbit(x.lo, x.hi, 64 - n)};
}
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }   // a ^ b ^ c

__device__ __forceinline__ void keccak_f1600(uint64_t (&st)[25]) {
    constexpr int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    constexpr int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    U64H a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = U64H{(uint32_t)st[i], (uint32_t)(st[i] >> 32)};
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        U64H bc[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            bc[i].lo = xor3(xor3(a[i].lo, a[i + 5].lo, a[i + 10].lo), a[i + 15].lo, a[i + 20].lo);
            bc[i].hi = xor3(xor3(a[i].hi, a[i + 5].hi, a[i + 10].hi), a[i + 15].hi, a[i + 20].hi);
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const U64H u = bc[(i + 4) % 5], v = rotl64h(bc[(i + 1) % 5], 1);
#pragma unroll
            for (int j = 0; j < 25; j += 5) a[j + i] = U64H{xor3(a[j + i].lo, u.lo, v.lo), xor3(a[j + i].hi, u.hi, v.hi)};
        }
        U64H t = a[1];
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const int j = PIL[i];
            const U64H b = a[j];
            a[j] = rotl64h(t, ROT[i]);
            t = b;
        }
#pragma unroll
        for (int j = 0; j < 25; j += 5) {
#pragma unroll
            for (int i = 0; i < 5; i++) bc[i] = a[j + i];
#pragma unroll
            for (int i = 0; i < 5; i++) {
                a[j + i].lo = bc[i].lo ^ (~bc[(i + 1) % 5].lo & bc[(i + 2) % 5].lo);
                a[j + i].hi = bc[i].hi ^ (~bc[(i + 1) % 5].hi & bc[(i + 2) % 5].hi);
            }
        }
        const uint64_t rc = KECCAK_RC[round];
        a[0].lo ^= (uint32_t)rc;
        a[0].hi ^= (uint32_t)(rc >> 32);
    }
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = ((uint64_t)a[i].hi << 32) | a[i].lo;
}

// parent k of the level starting at new_begin from its two children in the level starting at level_begin
__device__ __forceinline__ void merkle_parent(uint64_t *nodes, uint64_t level_begin, uint64_t new_begin, uint64_t k) {
    const uint64_t *ch = nodes + (level_begin + 2 * k) * 4;
    uint64_t st[25];
#pragma unroll
    for (int j = 0; j < 8; j++) st[j] = ch[j];
    st[8] = 0x01ull;
#pragma unroll
    for (int j = 9; j < 25; j++) st[j] = 0;
    st[16] = 0x8000000000000000ull;
    keccak_f1600(st);
    uint64_t *out = nodes + (new_begin + k) * 4;
    out[0] = st[0]; out[1] = st[1]; out[2] = st[2]; out[3] = st[3];
}

// `levels` levels of the subtree a workgroup owns: its 2 * cnt children in the level starting at level_begin (children
// [2 * cnt * blockIdx.x, ...) of that level) -> cnt parents, cnt / 2 grandparents, ... with a barrier between levels.  The
// children of every step but the first were written by this workgroup a moment ago and are read back through L2
// (a launch per level costs ~10 us of dependent-launch latency on top of a kernel that is a few us long near the top of a
// tree; the FRI commit phase builds twenty trees per proof).
__device__ __forceinline__ void merkle_subtree(uint64_t *nodes, uint64_t level_begin, uint32_t cnt, uint32_t levels) {
    for (uint32_t lv = 0; lv < levels; lv++) {
        const uint64_t new_begin = level_begin / 2;
        if (threadIdx.x < cnt) merkle_parent(nodes, level_begin, new_begin, (uint64_t)blockIdx.x * cnt + threadIdx.x);
        __threadfence_block();   // workgroup scope: an agent-scope release writes back the whole L2 of the XCD (measured: leaf kernel 0.8 -> 13 ms)
        __syncthreads();
        level_begin = new_begin;
        cnt >>= 1;
    }
}

// one work-item per leaf; columns[c] starts at cols + c * col_stride elements.  EB = bytes per element as the reference keeps
// it in memory; as_bytes of an element is its raw (Montgomery) value big-endian:
//   32: MontgomeryBackendPrimeField<_, 4>, four u64 limbs most significant first (montgomery_backed_prime_fields.rs:367-373)
//    8: MontgomeryBackendPrimeField<_, 1> (BabyBear on a u64 limb), the limb big-endian (same impl)
//    4: U32MontgomeryBackendPrimeField (BabyBear u32), value().to_be_bytes() (u32_montgomery_backend_prime_field.rs:258-262)
template <int EB>
__global__ __launch_bounds__(256) void merkle_leaves_kernel(const void *cols_v, uint32_t n_cols, uint64_t col_stride, uint32_t log2n,
                                                            int bit_reverse, uint64_t *nodes, uint32_t fused_levels) {
    const uint64_t n = 1ull << log2n;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;   // (whole workgroups only when fused_levels != 0: n is then a multiple of 256)
    // Bit-reversed gather of a large tree: work-item i reading row bitrev(i) makes every 32-byte read of a wave land 2^(n-8)
    // rows from the next.  Leaves are handed out in 16 x 16 tiles instead — i = u * 2^(n-4) + M * 16 + v for the
    // workgroup M and tid = 16 u + v — so that the rows of a fixed v are 16 consecutive ones (bitrev(i) ends in bitrev4(u))
    // and the leaves of a fixed u are 16 consecutive ones: reads and writes both in 512-byte runs.  (Trees with fused
    // levels keep consecutive leaves per workgroup.)  The host asks for it with bit_reverse = 2.
    // Narrow elements take wider tiles on the read side: 2^UB rows x 2^(8-UB) leaves with UB = 4 / 5 / 6 for 32- / 8- /
    // 4-byte elements (512 / 256 / 256-byte runs of rows per column, 512 / 256 / 128-byte runs of leaf hashes).
    constexpr int UB = EB == 32 ? 4 : (EB == 8 ? 5 : 6), VB = 8 - UB;
    if (bit_reverse == 2)
        i = ((uint64_t)(threadIdx.x >> VB) << (log2n - UB)) | ((uint64_t)blockIdx.x << VB) | (threadIdx.x & ((1u << VB) - 1));
    const uint64_t src = bit_reverse ? (log2n ? (uint64_t)(__brevll(i) >> (64 - log2n)) : 0) : i;
    const uint32_t total_bytes = (uint32_t)EB * n_cols;
    const uint32_t total = total_bytes / 8;         // whole 8-byte lanes of leaf data
    const uint32_t tail = total_bytes % 8;          // 4 for an odd number of u32 columns, else 0
    const uint32_t nblocks = total_bytes / 136 + 1; // rate = 17 lanes = 136 bytes; the padding always adds at least one byte
    // a Keccak lane is 8 stream bytes little-endian; lane s of the leaf's byte stream
    auto lane = [&](uint32_t s) -> uint64_t {
        if constexpr (EB == 32) {
            const uint64_t *cols = (const uint64_t *)cols_v;
            return __builtin_bswap64(cols[((uint64_t)(s >> 2) * col_stride + src) * 4 + (s & 3)]);
        } else if constexpr (EB == 8) {
            const uint64_t *cols = (const uint64_t *)cols_v;
            return __builtin_bswap64(cols[(uint64_t)s * col_stride + src]);
        } else {
            const uint32_t *cols = (const uint32_t *)cols_v;
            const uint32_t e0 = cols[(uint64_t)(2 * s) * col_stride + src];
            const uint32_t e1 = 2 * s + 1 < n_cols ? cols[(uint64_t)(2 * s + 1) * col_stride + src] : 0u;
            return (uint64_t)__builtin_bswap32(e0) | ((uint64_t)__builtin_bswap32(e1) << 32);
        }
    };
    uint64_t st[25];
#pragma unroll
    for (int k = 0; k < 25; k++) st[k] = 0;
    for (uint32_t b = 0; b < nblocks; b++) {
#pragma unroll
        for (int j = 0; j < 17; j++) {
            const uint32_t s = 17 * b + j;
            if (s < total) {
                st[j] ^= lane(s);
            } else if (s == total) {
                if (tail) st[j] ^= lane(s);              // the last u32 of an odd row (upper half zero)
                st[j] ^= 0x01ull << (8 * tail);          // first padding byte
            }
        }
        if (b + 1 == nblocks) st[16] ^= 0x8000000000000000ull;   // last padding byte of the rate
        keccak_f1600(st);
    }
    uint64_t *out = nodes + (n - 1 + i) * 4;
    out[0] = st[0]; out[1] = st[1]; out[2] = st[2]; out[3] = st[3];
    if (fused_levels) {   // the first levels above this workgroup's 256 leaves
        __threadfence_block();   // workgroup scope: an agent-scope release writes back the whole L2 of the XCD (measured: leaf kernel 0.8 -> 13 ms)
        __syncthreads();
        merkle_subtree(nodes, n - 1, 128, fused_levels);
    }
}

// `levels` levels per launch: a workgroup turns 512 children of the level starting at level_begin into 256 parents, 128
// grandparents, ...; the level must hold a multiple of 512 nodes
__global__ __launch_bounds__(256) void merkle_levels_kernel(uint64_t *nodes, uint64_t level_begin, uint32_t levels) {
    merkle_subtree(nodes, level_begin, 256, levels);
}

// The top of the tree in one launch: all levels from `count` <= 256 parents down to the root, one workgroup, a barrier
// between levels (a launch per level spends ~18 us on a kernel that hashes a handful of nodes: 9 launches saved per tree).
__global__ __launch_bounds__(256) void merkle_top_kernel(uint64_t *nodes, uint64_t level_begin, uint64_t level_end) {
    while (level_begin != level_end) {
        const uint64_t new_begin = level_begin / 2, count = level_begin - new_begin;
        const uint64_t k = threadIdx.x;
        if (k < count) merkle_parent(nodes, level_begin, new_begin, k);
        __threadfence_block();   // workgroup scope: an agent-scope release writes back the whole L2 of the XCD (measured: leaf kernel 0.8 -> 13 ms)
        __syncthreads();
        level_end = level_begin - 1;
        level_begin = new_begin;
    }
}

// d_nodes: (2 * 2^log2n - 1) * 32 bytes, root first
int merkle_commit_device(Context &c, const void *d_cols, uint32_t n_cols, uint64_t col_stride, uint32_t log2n, int bit_reverse,
                         void *d_nodes, hipStream_t stream, uint32_t elem_bytes) {
    const uint64_t n = 1ull << log2n;
    // Levels per launch (LW_HIP_MERKLE_FUSE, tuning: 1 = a launch per level as before): the tree above 2^m nodes is built
    // by the top kernel once m <= 9 (256 parents), by launches of up to `fuse` levels before that, the first of them inside
    // the leaf kernel.
    static const uint32_t fuse = [] { const char *e = tuning_env("LW_HIP_MERKLE_FUSE"); int v = e ? atoi(e) : 4; return (uint32_t)(v < 1 ? 1 : (v > 7 ? 7 : v)); }();
    // Wide levels are bound by the hashes' throughput and take a launch each (in a fused launch the upper levels of a
    // subtree run on 128, 64, 32 ... of the workgroup's 256 work-items while its other waves hold their slots: 1 x 2^24
    // 4.19 -> 4.35 ms fused everywhere); from 2^16 nodes down a level is a few microseconds long and the launches count.
    constexpr uint32_t FUSE_BELOW = 16;
    uint32_t m = log2n;   // the level whose parents are built next holds 2^m nodes
    const uint32_t leaf_fused = (fuse > 1 && m > 9 && m <= FUSE_BELOW) ? std::min(fuse, m - 9) : 0u;
    static const bool tile_env = [] { const char *e = tuning_env("LW_HIP_MERKLE_TILE"); return !e || atoi(e) != 0; }();   // A/B only
    if (bit_reverse) bit_reverse = (tile_env && leaf_fused == 0 && log2n >= 12) ? 2 : 1;
    hipEvent_t pe = c.prof_begin(stream);
    const dim3 grid((uint32_t)((n + 255) / 256));
    if (elem_bytes == 4)
        hipLaunchKernelGGL((merkle_leaves_kernel<4>), grid, dim3(256), 0, stream, d_cols, n_cols, col_stride, log2n, bit_reverse, (uint64_t *)d_nodes, leaf_fused);
    else if (elem_bytes == 8)
        hipLaunchKernelGGL((merkle_leaves_kernel<8>), grid, dim3(256), 0, stream, d_cols, n_cols, col_stride, log2n, bit_reverse, (uint64_t *)d_nodes, leaf_fused);
    else
        hipLaunchKernelGGL((merkle_leaves_kernel<32>), grid, dim3(256), 0, stream, d_cols, n_cols, col_stride, log2n, bit_reverse, (uint64_t *)d_nodes, leaf_fused);
    c.prof_end("merkle_leaves_kernel", pe, stream);
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    m -= leaf_fused;
    // crypto/src/merkle_tree/utils.rs:44-72: the level of 2^m nodes starts at node 2^m - 1
    while (m > 9) {
        const uint32_t lv = m <= FUSE_BELOW ? std::min(fuse, m - 9) : 1u;
        pe = c.prof_begin(stream);
        hipLaunchKernelGGL(merkle_levels_kernel, dim3((uint32_t)(((uint64_t)1 << m) / 512)), dim3(256), 0, stream, (uint64_t *)d_nodes,
                           ((uint64_t)1 << m) - 1, lv);
        c.prof_end("merkle_level_kernel", pe, stream);
        m -= lv;
    }
    if (m > 0) {   // the rest of the tree in one launch: at most 256 parents
        const uint64_t level_begin = ((uint64_t)1 << m) - 1;
        pe = c.prof_begin(stream);
        hipLaunchKernelGGL(merkle_top_kernel, dim3(1), dim3(256), 0, stream, (uint64_t *)d_nodes, level_begin, 2 * level_begin);
        c.prof_end("merkle_top_kernel", pe, stream);
    }
    LW_HIP_CHECK(hipGetLastError(), LW_ERR_LAUNCH);
    return LW_OK;
}

}  // namespace lw
