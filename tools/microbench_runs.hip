// HBM efficiency of a strided tile pass as a function of the run length: the memory side of an NTT pass without its
// butterflies.  A 2^26-word vector (256 MiB, the BabyBear 4 x 2^24 batch) is read and written back in place by
// workgroups that each own a tile of R rows x C adjacent words, rows 2^26 / R words apart — exactly the access pattern of a
// non-last pass with r = log2 R stages and C columns (csrc/ntt_bb.hip).  The 3-pass plan has R = 256, C = 32 (128-byte
// runs, 32 KiB tile); a 2-pass plan at 2^24 needs R = 4096 and therefore C <= 8 (32-byte runs, 128 KiB tile): this prints
// what each run length costs before any arithmetic.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_runs.hip -o tools/microbench_runs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void tile_pass(uint32_t *v, uint32_t logN, uint32_t logR, uint32_t logC) {
    extern __shared__ uint32_t lds[];
    const uint32_t R = 1u << logR, C = 1u << logC, S = 1u << (logN - logR);   // row stride in words
    const uint32_t tiles_per_row = S >> logC;
    const uint32_t b = blockIdx.x % tiles_per_row;
    uint32_t *base = v + ((size_t)(blockIdx.x / tiles_per_row) << logN) + ((size_t)b << logC);
    for (uint32_t e = threadIdx.x; e < R * C; e += THREADS) lds[e] = base[(size_t)(e >> logC) * S + (e & (C - 1))];
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < R * C; e += THREADS) base[(size_t)(e >> logC) * S + (e & (C - 1))] = lds[e ^ 1] + 1u;
}

int main() {
    const uint32_t logN = 24, batch = 4;
    const size_t words = (size_t)batch << logN;
    uint32_t *v;
    CK(hipMalloc(&v, words * 4));
    CK(hipMemset(v, 1, words * 4));
    CK(hipFuncSetAttribute((const void *)tile_pass<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)tile_pass<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Shape { uint32_t logR, logC; int threads; };
    const Shape shapes[] = {{8, 5, 256}, {8, 4, 256}, {8, 3, 256}, {10, 5, 1024}, {11, 4, 1024}, {12, 3, 1024}, {12, 3, 256}, {11, 3, 1024}, {10, 3, 1024}};
    printf("vector %u x 2^%u u32 words (%zu MiB), read + write in place\n", batch, logN, words * 4 >> 20);
    for (const Shape &sh : shapes) {
        const size_t lds = ((size_t)4 << (sh.logR + sh.logC));
        const uint32_t blocks = (uint32_t)(words >> (sh.logR + sh.logC));
        float best = 1e30f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipEventRecord(e0));
            if (sh.threads == 256) hipLaunchKernelGGL((tile_pass<256>), dim3(blocks), dim3(256), lds, 0, v, logN, sh.logR, sh.logC);
            else hipLaunchKernelGGL((tile_pass<1024>), dim3(blocks), dim3(1024), lds, 0, v, logN, sh.logR, sh.logC);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("RUNS rows 2^%-2u x %3u-byte runs, tile %3zu KiB, %4d threads: %7.3f ms  %7.1f GB/s (read + write)\n", sh.logR, 4u << sh.logC, lds >> 10,
               sh.threads, best, 2.0 * words * 4 / best / 1e6);
    }
    return 0;
}
