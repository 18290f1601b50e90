// Go/no-go micro-benchmark for an FP64-FMA ("DPF") Montgomery product on gfx950.
//
// Question: MI355X's FP64 vector pipe is full rate; would a Montgomery product on L-bit limbs held in doubles
// (52x52-bit products split into high and low halves by the FMA rounding trick) beat the integer product the kernels
// use (v_mad_u64_u32 + v_addc_co_u32 per 32x32 partial product)?
//
// Measured here, on the same launch geometry as tools/microbench.hip:
//   * raw issue rates: v_fma_f64, v_add_f64, v_mul_f64, v_lshl_add_u64, v_cvt (the instructions a DPF product is made of)
//   * a complete DPF Montgomery product for BLS12-381 Fp (8 limbs x 48 bits = 384 bits, so R = 2^384 exactly as in the
//     reference, math/src/unsigned_integer/montgomery.rs:86-141 — results are the same canonical residues) and for
//     Stark252 (6 limbs x 43 bits, R' = 2^258 = 4R: dpf(4a, b) = a*b/R), checked BIT FOR BIT against fe_mul on 2^20
//     random pairs, then timed like k_femul in microbench.hip.
//
// How the DPF product works (round-toward-zero mode for doubles, set once per wave with s_setreg):
//   t  = fma(a, b, H)        H is a running sum that starts at 2^(L+52): its ulp is 2^L, so the FMA adds
//                            floor(a*b / 2^L) * 2^L exactly — the high halves of a column accumulate INSIDE the addend
//   s  = H_old - t           = -(high half) * 2^L, exact
//   lo = fma(a, b, s)        = a*b mod 2^L, exact
//   Lo += lo                 low halves of the column
// i.e. FOUR full-rate FP64 instructions per LxL partial product (the high halves ride in the FMA addend; the low half
// cannot: it needs the individual high half cancelled first, and Lo + s is not representable).  Column sums stay exact
// while 2N terms of L bits fit 52 bits (L = 48, N = 8: 2^52).
//
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/microbench_dpf.hip -o tools/microbench_dpf
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../lambda_elliptic_curves_amd/csrc/field.cuh"
using namespace lw;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- raw rates
template <int ILP>
__global__ void k_fma64(double *out, int iters, double seed) {
    double acc[ILP];
    double a = 1.0 + 1e-9 * threadIdx.x + seed, b = 0.999999 + 1e-10 * blockIdx.x;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_add64(double *out, int iters, double seed) {
    double acc[ILP];
    double a = 1e-9 * threadIdx.x + seed;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mul64(double *out, int iters, double seed) {
    double acc[ILP];
    double a = 1.0 + 1e-12 * threadIdx.x + seed * 1e-12;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_lshladd64(double *out, int iters, double seed) {
    uint64_t acc[ILP];
    uint64_t a = threadIdx.x * 2654435761ull + (uint64_t)seed;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc[i] = i + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(a));
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;
}
template <int ILP>
__global__ void k_cvt(double *out, int iters, double seed) {
    uint32_t acc[ILP];
    double d[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) { acc[i] = i + threadIdx.x + (uint32_t)seed; d[i] = 0; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(acc[i]));
#pragma unroll
        for (int i = 0; i < ILP; i++) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(acc[i]) : "v"(d[i]));
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)s;
}

// ---------------------------------------------------------------- DPF Montgomery product
template <int NL>
struct DpfMod {
    double p[NL];      // modulus limbs (L bits each)
    double pinv;       // -p^-1 mod 2^L
};

__device__ __forceinline__ void set_round_toward_zero_f64() {
    // MODE.FP_ROUND[3:2] (double/half precision) = 3.  Inline asm on purpose: with __builtin_amdgcn_s_setreg the
    // compiler's mode-register pass sees the change and puts MODE back to round-to-nearest in front of the first f64
    // instruction (it guarantees default rounding to every non-strict FP op, and strict FP is "unsupported on this
    // target"); it does not look inside asm.  First statement of the kernel, "memory" so that no load — and with the
    // loads every FP op — moves above it.
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory");
}

// r = a * b / 2^(NL*L) mod p; operands and result: NL doubles holding L-bit non-negative integers, result canonical
template <int NL, int L>
__device__ __forceinline__ void dpf_mont_mul(const double (&a)[NL], const double (&b)[NL], const DpfMod<NL> &M, double (&r)[NL]) {
    const double C = __builtin_ldexp(1.0, L + 52);        // ulp 2^L: fma(x, y, C) keeps floor(x*y / 2^L) * 2^L
    const double TWO_L = __builtin_ldexp(1.0, L), INV_L = __builtin_ldexp(1.0, -L);
    double H[2 * NL], Lo[2 * NL];                          // per column: C + sum(high halves)*2^L, sum(low halves)
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) { H[k] = C; Lo[k] = 0.0; }
    // a * b
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const double t = __builtin_fma(a[i], b[j], H[i + j]);
            const double s = H[i + j] - t;
            Lo[i + j] += __builtin_fma(a[i], b[j], s);
            H[i + j] = t;
        }
    // Montgomery reduction, one limb of m per column; carry = value of column k-1 above 2^L (in units of 2^L)
    double carry = 0.0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        // x = Lo[k] + (high halves of column k-1) + carry
        double x = Lo[k] + carry;
        if (k > 0) x = __builtin_fma(H[k - 1] - C, INV_L, x);
        // low L bits of x, and what is above them
        const double xt = (x + C) - C;                     // floor(x / 2^L) * 2^L (x < 2^53 < C's range)
        const double xl = x - xt;
        // m = xl * pinv mod 2^L
        const double mt = __builtin_fma(xl, M.pinv, C);
        const double m = __builtin_fma(xl, M.pinv, C - mt);
        // column k += m * p[0] (makes its low L bits zero), columns k+j += m * p[j]
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const double t = __builtin_fma(m, M.p[j], H[k + j]);
            const double s = H[k + j] - t;
            Lo[k + j] += __builtin_fma(m, M.p[j], s);
            H[k + j] = t;
        }
        // column k is now a multiple of 2^L: its value / 2^L carries into column k+1
        double y = Lo[k] + carry;
        if (k > 0) y = __builtin_fma(H[k - 1] - C, INV_L, y);
        carry = y * INV_L;                                  // exact: y is a multiple of 2^L
        (void)xt;
    }
    // upper half: propagate carries, L bits per limb
#pragma unroll
    for (int k = NL; k < 2 * NL; k++) {
        double x = Lo[k] + carry;
        x = __builtin_fma(H[k - 1] - C, INV_L, x);
        const double xt = (x + C) - C;
        r[k - NL] = x - xt;
        carry = xt * INV_L;
    }
    // the top column's high halves and the last carry are zero for a, b < p (the result is < 2p < 2^(NL*L))
    // conditional subtraction of p: d = r - p with borrows
    double d[NL], borrow = 0.0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        double v = r[k] - M.p[k] - borrow;
        borrow = v < 0.0 ? 1.0 : 0.0;
        d[k] = v < 0.0 ? v + TWO_L : v;
    }
#pragma unroll
    for (int k = 0; k < NL; k++) r[k] = borrow != 0.0 ? r[k] : d[k];
}

// u64 <-> double for integers below 2^53, built from 32-bit conversions only.  (The compiler's own u64 -> f64 expansion
// ends in an addition that must round to nearest, so its mode-register pass resets MODE.FP_ROUND to 0 right after our
// s_setreg and every later FMA would round to nearest: hi halves come out rounded instead of truncated.)
__device__ __forceinline__ double u64_to_f64(uint64_t v) {
    return __builtin_fma((double)(uint32_t)(v >> 32), 4294967296.0, (double)(uint32_t)v);   // exact: no rounding happens
}
__device__ __forceinline__ uint64_t f64_to_u64(double x) {
    const uint32_t hi = (uint32_t)(x * (1.0 / 4294967296.0));          // v_cvt_u32_f64 truncates whatever the mode
    const uint32_t lo = (uint32_t)__builtin_fma((double)hi, -4294967296.0, x);
    return ((uint64_t)hi << 32) | lo;
}

// 32-bit words (least significant first) <-> L-bit limbs in doubles
template <int NW, int NL, int L>
__device__ __forceinline__ void words_to_limbs(const uint32_t (&w)[NW], double (&x)[NL]) {
#pragma unroll
    for (int k = 0; k < NL; k++) {
        uint64_t v = 0;
        const int bit = k * L, w0 = bit / 32, sh = bit % 32;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (w0 + q < NW) {
                const int pos = 32 * q - sh;
                if (pos < 64) v |= pos >= 0 ? ((uint64_t)w[w0 + q] << pos) : ((uint64_t)w[w0 + q] >> -pos);
            }
        }
        v &= (1ull << L) - 1;
        x[k] = u64_to_f64(v);
    }
}
template <int NW, int NL, int L>
__device__ __forceinline__ void limbs_to_words(const double (&x)[NL], uint32_t (&w)[NW]) {
    uint64_t v[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) v[k] = f64_to_u64(x[k]);
#pragma unroll
    for (int q = 0; q < NW; q++) {
        const int bit = 32 * q, k0 = bit / L, sh = bit % L;
        uint64_t lo = v[k0] >> sh;
        if (L - sh < 32 && k0 + 1 < NL) lo |= v[k0 + 1] << (L - sh);
        w[q] = (uint32_t)lo;
    }
}

template <class F, int NL, int L>
__global__ void k_dpf_check(const uint32_t *in, uint32_t *out, int n, DpfMod<NL> M) {
    set_round_toward_zero_f64();
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    uint32_t aw[F::N], bw[F::N], rw[F::N];
    for (int i = 0; i < F::N; i++) { aw[i] = in[(size_t)(2 * tid) * F::N + i]; bw[i] = in[(size_t)(2 * tid + 1) * F::N + i]; }
    double a[NL], b[NL], r[NL];
    words_to_limbs<F::N, NL, L>(aw, a);
    words_to_limbs<F::N, NL, L>(bw, b);
    dpf_mont_mul<NL, L>(a, b, M, r);
    limbs_to_words<F::N, NL, L>(r, rw);
    for (int i = 0; i < F::N; i++) out[(size_t)tid * F::N + i] = rw[i];
}
template <class F>
__global__ void k_int_ref(const uint32_t *in, uint32_t *out, int n, int pre_scale_log2) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    Fe<F> a, b;
    for (int i = 0; i < F::N; i++) { a.v[i] = in[(size_t)(2 * tid) * F::N + i]; b.v[i] = in[(size_t)(2 * tid + 1) * F::N + i]; }
    Fe<F> m = fe_mul<F>(a, b);
    for (int i = 0; i < F::N; i++) out[(size_t)tid * F::N + i] = m.v[i];
}
// a <- 2^k * a mod p (the R' = 2^k R correction of the first operand), in place
template <class F>
__global__ void k_scale(uint32_t *in, int n, int k) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    Fe<F> a;
    for (int i = 0; i < F::N; i++) a.v[i] = in[(size_t)(2 * tid) * F::N + i];
    for (int q = 0; q < k; q++) a = fe_add<F>(a, a);
    for (int i = 0; i < F::N; i++) in[(size_t)(2 * tid) * F::N + i] = a.v[i];
}

template <int NL, int L, int CH>
__global__ void k_dpf_rate(double *io, int iters, DpfMod<NL> M) {
    set_round_toward_zero_f64();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double a[CH][NL], b[NL];
#pragma unroll
    for (int c = 0; c < CH; c++)
        for (int i = 0; i < NL; i++) a[c][i] = u64_to_f64(((uint64_t)(tid * 7 + c * 13 + i * 3 + 1) * 0x9E3779B97F4A7C15ull) >> (64 - L + 1));
    for (int i = 0; i < NL; i++) b[i] = u64_to_f64(((uint64_t)(tid * 11 + i * 5 + 2) * 0xC2B2AE3D27D4EB4Full) >> (64 - L + 1));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            double r[NL];
            dpf_mont_mul<NL, L>(a[c], b, M, r);
#pragma unroll
            for (int i = 0; i < NL; i++) a[c][i] = r[i];
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++)
        for (int i = 0; i < NL; i++) s += a[c][i];
    io[tid] = s;
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

// host: modulus limbs and -p^-1 mod 2^L from the field's 32-bit words
template <class F, int NL, int L>
static DpfMod<NL> make_mod() {
    DpfMod<NL> M;
    unsigned __int128 acc = 0;
    uint64_t limbs[NL];
    for (int k = 0; k < NL; k++) {
        uint64_t v = 0;
        for (int bit = 0; bit < L; bit++) {
            const int g = k * L + bit;
            if (g / 32 < F::N && ((F::p(g / 32) >> (g % 32)) & 1)) v |= 1ull << bit;
        }
        limbs[k] = v;
        M.p[k] = (double)v;
    }
    (void)acc;
    uint64_t inv = 1;                       // Newton: inv = p0^-1 mod 2^64
    for (int i = 0; i < 6; i++) inv *= 2 - limbs[0] * inv;
    M.pinv = (double)((0 - inv) & ((1ull << L) - 1));
    return M;
}

template <class F, int NL, int L>
static void run_field(const char *name, int pre_scale_log2, double int_rate_gmul, int cus) {
    const int n = 1 << 20;
    std::vector<uint32_t> in((size_t)2 * n * F::N), out_d((size_t)n * F::N), out_i((size_t)n * F::N);
    uint64_t s = 88172645463325252ULL;
    for (int k = 0; k < 2 * n; k++) {
        Fe<F> x;
        for (int i = 0; i < F::N; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x.v[i] = (uint32_t)(s >> 16); }
        x.v[F::N - 1] &= (F::p(F::N - 1) >> 1);   // < p
        if (k == 0 || k == 5) x = Fe<F>::zero();
        if (k == 1 || k == 2 || k == 3) { for (int i = 0; i < F::N; i++) x.v[i] = F::p(i); x.v[0] -= 1; }   // p - 1
        for (int i = 0; i < F::N; i++) in[(size_t)k * F::N + i] = x.v[i];
    }
    uint32_t *din, *dout;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, out_d.size() * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_int_ref<F>), dim3(n / 256), dim3(256), 0, 0, din, dout, n, 0);
    CK(hipMemcpy(out_i.data(), dout, out_i.size() * 4, hipMemcpyDeviceToHost));
    if (pre_scale_log2) hipLaunchKernelGGL((k_scale<F>), dim3(n / 256), dim3(256), 0, 0, din, n, pre_scale_log2);
    const DpfMod<NL> M = make_mod<F, NL, L>();
    hipLaunchKernelGGL((k_dpf_check<F, NL, L>), dim3(n / 256), dim3(256), 0, 0, din, dout, n, M);
    CK(hipMemcpy(out_d.data(), dout, out_d.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < out_d.size(); i += F::N)
        for (int q = 0; q < F::N; q++)
            if (out_d[i + q] != out_i[i + q]) { bad++; break; }
    printf("CHECK dpf_mont_mul %-9s (%d limbs x %d bits) vs integer fe_mul: %zu mismatches of %d\n", name, NL, L, bad, n);
    double *buf;
    CK(hipMalloc(&buf, (size_t)cus * 8 * 256 * 8));
    const int blocks = cus * 8, threads = 256, it2 = 256;
    float ms = time_ms([&] { hipLaunchKernelGGL((k_dpf_rate<NL, L, 2>), dim3(blocks), dim3(threads), 0, 0, buf, it2, M); });
    const double rate = (double)blocks * threads * it2 * 2 / ms / 1e6;
    const int ops = 4 * 2 * NL * NL + 5 * 2 * NL + 6 * NL + 5 * NL;
    printf("RATE dpf_mont_mul %-9s %8.2f Gmul/s   integer fe_mul %8.2f Gmul/s   ratio %.2f   (~%d FP64 instructions per product)\n",
           name, rate, int_rate_gmul, rate / int_rate_gmul, ops);
    CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(buf));
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

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    double *buf;
    CK(hipMalloc(&buf, (size_t)64 << 20));
    CK(hipMemset(buf, 0, (size_t)64 << 20));
    const int blocks = cus * 8, threads = 256, iters = 4096;
    const double lanes = (double)blocks * threads;
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL((k_fma64<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1.0); });
    printf("RATE v_fma_f64          %8.2f Gop/s (lane-ops)  [ILP8, %d waves/CU]\n", lanes * iters * 8 / ms / 1e6, blocks * threads / 64 / cus);
    ms = time_ms([&] { hipLaunchKernelGGL((k_add64<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1.0); });
    printf("RATE v_add_f64          %8.2f Gop/s\n", lanes * iters * 8 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_mul64<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1.0); });
    printf("RATE v_mul_f64          %8.2f Gop/s\n", lanes * iters * 8 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_lshladd64<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1.0); });
    printf("RATE v_lshl_add_u64     %8.2f Gop/s\n", lanes * iters * 8 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_cvt<8>), dim3(blocks), dim3(threads), 0, 0, buf, iters, 1.0); });
    printf("RATE v_cvt_f64_u32 + v_cvt_u32_f64   %8.2f Gop/s (per instruction)\n", lanes * iters * 8 * 2 / ms / 1e6);
    // the integer products, same box, same geometry (microbench.hip's k_femul)
    uint32_t *ibuf = (uint32_t *)buf;
    CK(hipMemset(ibuf, 1, (size_t)64 << 20));
    const int it2 = 512;
    ms = time_ms([&] { hipLaunchKernelGGL((k_femul<Fp381, 2>), dim3(blocks), dim3(threads), 0, 0, ibuf, it2); });
    const double fp381 = lanes * it2 * 2 / ms / 1e6;
    ms = time_ms([&] { hipLaunchKernelGGL((k_femul<Stark252, 2>), dim3(blocks), dim3(threads), 0, 0, ibuf, it2); });
    const double stark = lanes * it2 * 2 / ms / 1e6;
    printf("RATE fe_mul Fp381 (integer, this box)     %8.2f Gmul/s\n", fp381);
    printf("RATE fe_mul Stark252 (integer, this box)  %8.2f Gmul/s\n", stark);
    run_field<Fp381, 8, 48>("Fp381", 0, fp381, cus);
    run_field<Stark252, 6, 43>("Stark252", 2, stark, cus);
    return 0;
}
