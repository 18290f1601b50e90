// Micro-benchmarks for the integer pipe of gfx950: v_mad_u64_u32 / v_mul_lo_u32 / 32-bit add rates and the
// Montgomery product of each field, plus an HBM copy.  Feeds DESIGN.md's VALU roofline (SURVEY §8d asks for a
// measured, not assumed, mul32 peak).  Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../lambda_elliptic_curves_amd/csrc/field.cuh"
using namespace lw;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ILP>
__global__ void k_mad64(uint32_t *out, int iters, uint32_t seed) {
    uint64_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 77u;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int ILP>
__global__ void k_mullo(uint32_t *out, int iters, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t b = blockIdx.x * 40503u + 77u + seed;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_add(uint32_t *out, int iters, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t b = blockIdx.x * 40503u + 77u + seed;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[i]) : "v"(b) : "vcc");
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mac96(uint32_t *out, int iters, uint32_t seed) {
    uint64_t lo[ILP]; uint32_t hi[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 77u;
#pragma unroll
    for (int i = 0; i < ILP; i++) { lo[i] = i + a; hi[i] = 0; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= lo[i] + hi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <class F, int CH>
__global__ void k_femul(uint32_t *io, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fe<F> a[CH], b;
#pragma unroll
    for (int c = 0; c < CH; c++)
        for (int i = 0; i < F::N; i++) a[c].v[i] = io[((size_t)tid * CH + c) % 4096 * F::N + i];
    for (int i = 0; i < F::N; i++) b.v[i] = io[(size_t)((tid + 17) % 4096) * F::N + i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) a[c] = fe_mul<F>(a[c], b);
    }
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++)
        for (int i = 0; i < F::N; i++) s ^= a[c].v[i];
    io[4096 * 12 + tid] = s;
}
// correctness: device fe_mul / add / sub vs host portable code
template <class F>
__global__ void k_check(const uint32_t *in, uint32_t *out, int n) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    Fe<F> a, b;
    for (int i = 0; i < F::N; i++) { a.v[i] = in[(2 * tid) * F::N + i]; b.v[i] = in[(2 * tid + 1) * F::N + i]; }
    Fe<F> m = fe_mul<F>(a, b), s = fe_add<F>(a, b), d = fe_sub<F>(a, b);
    for (int i = 0; i < F::N; i++) { out[(3 * tid) * F::N + i] = m.v[i]; out[(3 * tid + 1) * F::N + i] = s.v[i]; out[(3 * tid + 2) * F::N + i] = d.v[i]; }
}
__global__ void k_copy(const uint4 *in, uint4 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <class L>
static float time_ms(L launch, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

template <class F>
static void check_field(const char *name) {
    const int n = 4096;
    std::vector<uint32_t> in(2 * n * F::N), out(3 * n * F::N);
    uint64_t s = 88172645463325252ULL;
    for (int k = 0; k < 2 * n; k++) {
        Fe<F> x;
        for (int i = 0; i < F::N; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x.v[i] = (uint32_t)(s >> 16); }
        x.v[F::N - 1] &= (F::p(F::N - 1) >> 1);   // < p
        if (k == 0) x = Fe<F>::zero();
        if (k == 1) { x = Fe<F>::zero(); for (int i = 0; i < F::N; i++) x.v[i] = F::p(i); x.v[0] -= 1; }   // p-1
        if (k == 2) { for (int i = 0; i < F::N; i++) x.v[i] = F::p(i); x.v[0] -= 1; }
        if (k == 3) { for (int i = 0; i < F::N; i++) x.v[i] = F::p(i); x.v[0] -= 1; }
        for (int i = 0; i < F::N; i++) in[k * F::N + i] = x.v[i];
    }
    uint32_t *din, *dout;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, out.size() * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_check<F>), dim3(n / 256), dim3(256), 0, 0, din, dout, n);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int k = 0; k < n; k++) {
        Fe<F> a, b;
        for (int i = 0; i < F::N; i++) { a.v[i] = in[(2 * k) * F::N + i]; b.v[i] = in[(2 * k + 1) * F::N + i]; }
        Fe<F> m = fe_mul_portable<F>(a, b), sm = fe_add<F>(a, b), d = fe_sub<F>(a, b);
        for (int i = 0; i < F::N; i++)
            if (out[(3 * k) * F::N + i] != m.v[i] || out[(3 * k + 1) * F::N + i] != sm.v[i] || out[(3 * k + 2) * F::N + i] != d.v[i]) { bad++; break; }
    }
    printf("CHECK %-9s device fe_mul/add/sub vs host: %d mismatches of %d\n", name, bad, n);
    CK(hipFree(din)); CK(hipFree(dout));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    check_field<Stark252>("Stark252"); check_field<Fr381>("Fr381"); check_field<Fp381>("Fp381"); check_field<Fp254>("Fp254");

    uint32_t *buf;
    CK(hipMalloc(&buf, (size_t)64 << 20));
    CK(hipMemset(buf, 1, (size_t)64 << 20));
    const int blocks = cus * 8, threads = 256, iters = 4096;
    const double lanes = (double)blocks * threads;
    {
        float ms = time_ms([&] { hipLaunchKernelGGL((k_mad64<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1u); });
        printf("RATE v_mad_u64_u32      %8.2f Gop/s (lane-ops)  [ILP8, %d waves/CU]\n", lanes * iters * 8 / ms / 1e6, blocks * threads / 64 / cus);
        ms = time_ms([&] { hipLaunchKernelGGL((k_mullo<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1u); });
        printf("RATE v_mul_lo_u32       %8.2f Gop/s\n", lanes * iters * 8 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_add<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1u); });
        printf("RATE v_add_co+v_addc    %8.2f Gop/s (per 32-bit add instr)\n", lanes * iters * 8 * 2 / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_mac96<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1u); });
        printf("RATE mac96 (mad+addc)   %8.2f Gmac/s\n", lanes * iters * 8 / ms / 1e6);
    }
    {
        // dependent-chain issue: one MAC chain per lane (the shape of one FIPS column), by waves per SIMD
        for (int wps : {1, 2, 3, 4, 6, 8}) {
            const int blk = cus * wps;   // 256-thread blocks: 4 waves -> one per SIMD
            const double ln = (double)blk * 256;
            float ms = time_ms([&] { hipLaunchKernelGGL((k_mac96<1>), dim3(blk), dim3(256), 0, 0, buf, iters * 4, 1u); });
            float ms2 = time_ms([&] { hipLaunchKernelGGL((k_mac96<2>), dim3(blk), dim3(256), 0, 0, buf, iters * 2, 1u); });
            printf("RATE mac96 dependent chain, %d waves/SIMD: ILP1 %8.2f Gmac/s   ILP2 %8.2f Gmac/s\n", wps, ln * iters * 4 / ms / 1e6,
                   ln * iters * 2 * 2 / ms2 / 1e6);
        }
    }
    {
        const int it2 = 512;
        float ms;
#define FEMUL(F, NAME)                                                                                         \
        ms = time_ms([&] { hipLaunchKernelGGL((k_femul<F, 2>), dim3(blocks), dim3(threads), 0, 0, buf, it2); }); \
        printf("RATE fe_mul %-9s   %8.2f Gmul/s   (%.1f mad-equivalents/mul at the measured mad rate)\n", NAME, lanes * it2 * 2 / ms / 1e6, 0.0);
        FEMUL(Stark252, "Stark252") FEMUL(Fr381, "Fr381") FEMUL(Fp254, "Fp254") FEMUL(Fp381, "Fp381")
    }
    {
        const size_t bytes = (size_t)2 << 30;
        uint4 *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 3, bytes));
        float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, a, b, bytes / 16); });
        printf("RATE HBM copy 2GiB      %8.2f GB/s (read+write)\n", 2.0 * bytes / ms / 1e6);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
