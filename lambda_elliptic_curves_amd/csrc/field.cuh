// Montgomery prime-field arithmetic for gfx950 (and the host side of the library).
//
// Values are the same canonical Montgomery residues the reference stores (R = 2^(64*NUM_LIMBS),
// math/src/field/fields/montgomery_backed_prime_fields.rs:34-51), so any correct implementation is
// bit-identical to the reference's `cios_optimized_for_moduli_with_one_spare_bit`
// (math/src/unsigned_integer/montgomery.rs:86-141).  Here a value lives in registers as N 32-bit limbs,
// least-significant first: gfx950 has no 64x64 multiply, `v_mad_u64_u32` (32x32+64 -> 64) is the widest
// integer MAC, so 8x32 (256-bit) / 12x32 (384-bit) limbs are the native shape.
//
// Memory layout at the boundary is the reference's: u64 limbs, MOST significant limb first
// (math/src/unsigned_integer/element.rs:29-37); conversion happens only in load/store.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LW_HD __host__ __device__ __forceinline__
#else
#define LW_HD inline
#endif

namespace lw {

// ---------------------------------------------------------------- field parameter packs
// p(i), one(i), r2(i): 32-bit limb i (LS first) of the modulus, R mod p, R^2 mod p.  INV = -p^-1 mod 2^32.
struct Stark252 {   // math/src/field/fields/fft_friendly/stark_252_prime_field.rs:13-24
    static constexpr int N = 8;
    // p < 2^252: 2^256 / p ~ 32, so NTT butterflies can carry values in [0, 24p) between reductions
    static constexpr bool LAZY = true;
    static constexpr uint32_t INV = 0xffffffffu;
    static constexpr uint32_t TWO_ADICITY = 192;
    LW_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t t[N] = {0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000011u, 0x08000000u};
        return t[i];
    }
    LW_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t t[N] = {0xffffffe1u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xfffffdf0u, 0x07ffffffu};
        return t[i];
    }
    LW_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t t[N] = {0x7e000401u, 0xfffffd73u, 0x330fffffu, 0x00000001u, 0xff6f8000u, 0xffffffffu, 0x5e008810u, 0x07ffd4abu};
        return t[i];
    }
    // TWO_ADIC_PRIMITVE_ROOT_OF_UNITY (canonical), 32-bit limbs LS first
    LW_HD static constexpr uint32_t root(int i) {
        constexpr uint32_t t[N] = {0x42f8ef94u, 0x6070024fu, 0xe11a6161u, 0xad187148u, 0x9c8b0fa5u, 0x3f046451u, 0x87529cfau, 0x005282dbu};
        return t[i];
    }
};
struct Fr381 {      // math/src/elliptic_curve/short_weierstrass/curves/bls12_381/default_types.rs:15-30
    static constexpr int N = 8;
    static constexpr bool LAZY = false;   // 255-bit modulus: no headroom above 2p
    static constexpr uint32_t INV = 0xffffffffu;
    static constexpr uint32_t TWO_ADICITY = 32;
    LW_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t t[N] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return t[i];
    }
    LW_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t t[N] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return t[i];
    }
    LW_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t t[N] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return t[i];
    }
    LW_HD static constexpr uint32_t root(int i) {
        constexpr uint32_t t[N] = {0xbb30bbb7u, 0x54cc39d9u, 0xfa78eb2bu, 0xc5e433d6u, 0x349d9b3cu, 0x84dd396cu, 0xa08a499du, 0x2ab00961u};
        return t[i];
    }
};
struct Fp381 {      // math/src/elliptic_curve/short_weierstrass/curves/bls12_381/field_extension.rs:13-22
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;
    LW_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t t[N] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return t[i];
    }
    LW_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t t[N] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return t[i];
    }
    LW_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t t[N] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return t[i];
    }
};
struct Fp254 {      // math/src/elliptic_curve/short_weierstrass/curves/bn_254/field_extension.rs:15-25
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xe4866389u;
    LW_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t t[N] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return t[i];
    }
    LW_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t t[N] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return t[i];
    }
    LW_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t t[N] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return t[i];
    }
};

struct Fr254 {      // BN254 scalar field (bn_254/default_types.rs:13-20): MSM scalar preparation only
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xefffffffu;
    static constexpr bool LAZY = false;
    LW_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t t[N] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return t[i];
    }
    LW_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t t[N] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return t[i];
    }
    LW_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t t[N] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return t[i];
    }
};

// ---------------------------------------------------------------- multi-limb element
template <class F>
struct Fe {
    static constexpr int N = F::N;
    uint32_t v[F::N];

    LW_HD static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = 0;
        return r;
    }
    LW_HD static Fe one() {   // R mod p
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = F::one(i);
        return r;
    }
    LW_HD static Fe r2() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = F::r2(i);
        return r;
    }
    LW_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= v[i];
        return o == 0;
    }
    LW_HD bool operator==(const Fe &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    LW_HD bool operator!=(const Fe &b) const { return !(*this == b); }
};

// ---- N-limb carry chains ----
// Written with __builtin_addc / __builtin_subc so that hipcc emits one v_add_co/v_addc_co (v_sub_co/v_subb_co) per
// limb; the same sums written with 64-bit temporaries compile to a v_lshl_add_u64 plus one or two v_mov per limb,
// which tripled the cost of every field addition.  Modulus limbs enter the chains through lw_k(): an SGPR the
// optimiser cannot see through, otherwise zero limbs break the chain into sub + cndmask pairs.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t lw_k(uint32_t x) {
    asm("" : "+s"(x));
    return x;
}
#else
inline uint32_t lw_k(uint32_t x) { return x; }
#endif
#if defined(__clang__)
#define LW_ADDC(a, b, c) __builtin_addc((a), (b), (c), &(c))
#define LW_SUBC(a, b, c) __builtin_subc((a), (b), (c), &(c))
#else   // the host-only sanitizer build of this header uses g++
static inline uint32_t lw_addc_portable(uint32_t a, uint32_t b, unsigned &c) {
    const uint64_t t = (uint64_t)a + b + c;
    c = (unsigned)(t >> 32);
    return (uint32_t)t;
}
static inline uint32_t lw_subc_portable(uint32_t a, uint32_t b, unsigned &c) {
    const uint64_t t = (uint64_t)a - b - c;
    c = (unsigned)((t >> 32) & 1);
    return (uint32_t)t;
}
#define LW_ADDC(a, b, c) lw_addc_portable((a), (b), (c))
#define LW_SUBC(a, b, c) lw_subc_portable((a), (b), (c))
#endif
template <int N>
LW_HD uint32_t limbs_add(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = LW_ADDC(a[i], b[i], c);
    return c;
}
template <int N>
LW_HD uint32_t limbs_sub(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = LW_SUBC(a[i], b[i], c);
    return c;
}
// limb i of K*p for K = 2^LOGK
template <class F, int LOGK>
LW_HD constexpr uint32_t kp_limb(int i) {
    if (LOGK == 0) return F::p(i);
    return (F::p(i) << LOGK) | (i > 0 ? (F::p(i - 1) >> (32 - LOGK)) : 0u);
}
// limbs of K*p as chain operands
template <class F, int LOGK>
LW_HD void limbs_kp(uint32_t (&k)[F::N]) {
#pragma unroll
    for (int i = 0; i < F::N; i++) k[i] = lw_k(kp_limb<F, LOGK>(i));
}
// limbs of p AND mask; zero limbs stay literal zeros behind lw_k
template <class F>
LW_HD void limbs_p_masked(uint32_t (&k)[F::N], uint32_t mask) {
#pragma unroll
    for (int i = 0; i < F::N; i++) k[i] = F::p(i) == 0 ? lw_k(0u) : (lw_k(F::p(i)) & mask);
}

// r = a - p if a >= p else a       (a < 2p, all moduli here have a spare top bit)
template <class F>
LW_HD Fe<F> reduce_once(const Fe<F> &a) {
    constexpr int N = F::N;
    uint32_t pk[N];
    limbs_kp<F, 0>(pk);
    Fe<F> d;
    const uint32_t borrow = limbs_sub<N>(d.v, a.v, pk);
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}

// IsField::add (montgomery_backed_prime_fields.rs:121-135, spare-bit branch)
template <class F>
LW_HD Fe<F> fe_add(const Fe<F> &a, const Fe<F> &b) {
    Fe<F> s;
    limbs_add<F::N>(s.v, a.v, b.v);
    return reduce_once<F>(s);
}

// IsField::sub (:157-163): a - b, + p on borrow
template <class F>
LW_HD Fe<F> fe_sub(const Fe<F> &a, const Fe<F> &b) {
    constexpr int N = F::N;
    Fe<F> d;
    const uint32_t borrow = limbs_sub<N>(d.v, a.v, b.v);
    uint32_t pm[N];
    limbs_p_masked<F>(pm, 0u - borrow);
    Fe<F> r;
    limbs_add<N>(r.v, d.v, pm);
    return r;
}

// IsField::neg (:166-172)
template <class F>
LW_HD Fe<F> fe_neg(const Fe<F> &a) {
    constexpr int N = F::N;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= a.v[i];
    uint32_t pk[N];
    limbs_kp<F, 0>(pk);
    Fe<F> r;
    limbs_sub<N>(r.v, pk, a.v);
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = nz ? r.v[i] : 0u;
    return r;
}

template <class F>
LW_HD Fe<F> fe_dbl(const Fe<F> &a) { return fe_add<F>(a, a); }

// Montgomery product a*b*R^-1 mod p, canonical in [0,p).
//
// Host build: operand-scanning CIOS over 32-bit limbs in plain C++ (used for twiddle seeds, N^-1, the
// final MSM window combine).
//
// Device build (gfx950): product-scanning (FIPS) Montgomery.  Each column sum lives in a 96-bit
// accumulator (one VGPR pair + one VGPR); every partial product is exactly one `v_mad_u64_u32`
// (32x32+64 -> 64, carry-out to VCC) followed by one `v_addc_co_u32` into the top word.  Plain C++ makes
// hipcc materialise zero-extended pairs (3-4 v_mov + a 64-bit add per product), so the MAC chains are
// inline asm, several MACs per statement (hipcc pads one wait state after every asm statement).
// Multiplications by the modulus limbs are resolved at compile time: zero limbs vanish, small limbs
// become inline constants, the rest are SGPR literals (Stark252: p = 2^251 + 17*2^192 + 1, INV = -1).
template <class F>
LW_HD Fe<F> fe_mul_portable(const Fe<F> &a, const Fe<F> &b) {
    constexpr int N = F::N;
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            c += (uint64_t)a.v[j] * b.v[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        uint32_t tN = (uint32_t)c;          // t < 2p < 2^(32N) on row entry (spare top bit), so word N starts at 0
        uint32_t m = t[0] * F::INV;
        c = ((uint64_t)m * F::p(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c += (uint64_t)m * F::p(j) + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += tN;
        t[N - 1] = (uint32_t)c;             // (t + a*b_i + m*p) / 2^32 < 2p: the high word is 0
    }
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return reduce_once<F>(r);
}

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
// Host build on a 64-bit machine: the same CIOS over 64-bit limbs with 128-bit products (4-5x the 32-bit loop above).
// The host only combines <= 64 window sums per MSM and derives table seeds, but for small MSMs that fold was half the
// call (2^10 points: 0.5 of 1.3 ms).
#define LW_HOST_MUL64 1
template <class F>
inline Fe<F> fe_mul_host64(const Fe<F> &a, const Fe<F> &b) {
    constexpr int N = F::N, M = N / 2;
    static_assert(N % 2 == 0, "even number of 32-bit limbs");
    typedef unsigned __int128 u128;
    uint64_t A[M], B[M], P[M], t[M + 2];
    for (int i = 0; i < M; i++) {
        A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
        B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
        P[i] = (uint64_t)F::p(2 * i) | ((uint64_t)F::p(2 * i + 1) << 32);
    }
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    uint64_t z = (uint64_t)(0u - F::INV);        // p^-1 mod 2^32 ...
    z *= 2 - P[0] * z;                           // ... one Newton step: p^-1 mod 2^64
    const uint64_t inv = 0 - z;                  // -p^-1 mod 2^64
    for (int i = 0; i < M; i++) {
        u128 c = 0;
        for (int j = 0; j < M; j++) {
            c += (u128)A[j] * B[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[M];
        t[M] = (uint64_t)c;
        t[M + 1] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * inv;
        c = ((u128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < M; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[M];
        t[M - 1] = (uint64_t)c;
        t[M] = t[M + 1] + (uint64_t)(c >> 64);
    }
    Fe<F> r;
    for (int i = 0; i < M; i++) {
        r.v[2 * i] = (uint32_t)t[i];
        r.v[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    return reduce_once<F>(r);                    // t < 2p: every modulus here has a spare top bit
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define LW_MAC_V(A, B) "v_mad_u64_u32 %0, vcc, " A ", " B ", %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"

// acc(96) += a0*b0 [+ a1*b1 ...]; all operands VGPRs
__device__ __forceinline__ void mac96_x1(uint64_t &lo, uint32_t &hi, uint32_t a0, uint32_t b0) {
    asm(LW_MAC_V("%2", "%3") : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0) : "vcc");
}
__device__ __forceinline__ void mac96_x2(uint64_t &lo, uint32_t &hi, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
    asm(LW_MAC_V("%2", "%3") LW_MAC_V("%4", "%5") : "+v"(lo), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
__device__ __forceinline__ void mac96_x4(uint64_t &lo, uint32_t &hi, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                         uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
    asm(LW_MAC_V("%2", "%3") LW_MAC_V("%4", "%5") LW_MAC_V("%6", "%7") LW_MAC_V("%8", "%9")
        : "+v"(lo), "+v"(hi)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
        : "vcc");
}
// Column start: acc(96) = init(64) + a0*b0 [+ ...].  The first MAC takes the previous column's carry as its
// addend and the first add-with-carry materialises the top word (0 + 0 + carry), so a column change costs one
// register copy instead of three.
#define LW_MAC_FIRST(A, B, INIT) "v_mad_u64_u32 %0, vcc, " A ", " B ", " INIT "\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
__device__ __forceinline__ void mac96_first_x1(uint64_t &lo, uint32_t &hi, uint64_t init, uint32_t a0, uint32_t b0) {
    asm(LW_MAC_FIRST("%3", "%4", "%2") : "=v"(lo), "=v"(hi) : "v"(init), "v"(a0), "v"(b0) : "vcc");
}
__device__ __forceinline__ void mac96_first_x2(uint64_t &lo, uint32_t &hi, uint64_t init, uint32_t a0, uint32_t b0, uint32_t a1,
                                               uint32_t b1) {
    asm(LW_MAC_FIRST("%3", "%4", "%2") LW_MAC_V("%5", "%6")
        : "=&v"(lo), "=&v"(hi) : "v"(init), "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
__device__ __forceinline__ void mac96_first_x4(uint64_t &lo, uint32_t &hi, uint64_t init, uint32_t a0, uint32_t b0, uint32_t a1,
                                               uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
    asm(LW_MAC_FIRST("%3", "%4", "%2") LW_MAC_V("%5", "%6") LW_MAC_V("%7", "%8") LW_MAC_V("%9", "%10")
        : "=&v"(lo), "=&v"(hi)
        : "v"(init), "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
        : "vcc");
}
// acc(96) += m * C for a compile-time constant C
template <uint32_t C>
__device__ __forceinline__ void mac96_c1(uint64_t &lo, uint32_t &hi, uint32_t m0) {
    if constexpr (C == 0) {
    } else if constexpr (C <= 64) {
        asm(LW_MAC_V("%2", "%3") : "+v"(lo), "+v"(hi) : "v"(m0), "n"(C) : "vcc");
    } else {
        asm(LW_MAC_V("%2", "%3") : "+v"(lo), "+v"(hi) : "v"(m0), "s"(C) : "vcc");
    }
}
template <uint32_t C0, uint32_t C1, uint32_t C2, uint32_t C3>
__device__ __forceinline__ void mac96_s4(uint64_t &lo, uint32_t &hi, uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3) {
    asm(LW_MAC_V("%2", "%3") LW_MAC_V("%4", "%5") LW_MAC_V("%6", "%7") LW_MAC_V("%8", "%9")
        : "+v"(lo), "+v"(hi)
        : "v"(m0), "s"(C0), "v"(m1), "s"(C1), "v"(m2), "s"(C2), "v"(m3), "s"(C3)
        : "vcc");
}

// column K: acc += sum_{i=I}^{IEND-1} a[i]*b[K-i]
template <class F, int K, int I, int IEND>
__device__ __forceinline__ void col_ab(uint64_t &lo, uint32_t &hi, const Fe<F> &a, const Fe<F> &b) {
    if constexpr (I + 4 <= IEND) {
        mac96_x4(lo, hi, a.v[I], b.v[K - I], a.v[I + 1], b.v[K - I - 1], a.v[I + 2], b.v[K - I - 2], a.v[I + 3], b.v[K - I - 3]);
        col_ab<F, K, I + 4, IEND>(lo, hi, a, b);
    } else if constexpr (I + 2 <= IEND) {
        mac96_x2(lo, hi, a.v[I], b.v[K - I], a.v[I + 1], b.v[K - I - 1]);
        col_ab<F, K, I + 2, IEND>(lo, hi, a, b);
    } else if constexpr (I + 1 <= IEND) {
        mac96_x1(lo, hi, a.v[I], b.v[K - I]);
    }
}
template <uint32_t C>
constexpr bool lw_is_literal() { return C > 64; }
// column K: acc += sum_{i=I}^{IEND-1} m[i]*p[K-i]
template <class F, int K, int I, int IEND>
__device__ __forceinline__ void col_mp(uint64_t &lo, uint32_t &hi, const uint32_t (&m)[F::N]) {
    if constexpr (I + 4 <= IEND) {
        if constexpr (lw_is_literal<F::p(K - I)>() && lw_is_literal<F::p(K - I - 1)>() && lw_is_literal<F::p(K - I - 2)>() &&
                      lw_is_literal<F::p(K - I - 3)>()) {
            mac96_s4<F::p(K - I), F::p(K - I - 1), F::p(K - I - 2), F::p(K - I - 3)>(lo, hi, m[I], m[I + 1], m[I + 2], m[I + 3]);
            col_mp<F, K, I + 4, IEND>(lo, hi, m);
        } else {
            mac96_c1<F::p(K - I)>(lo, hi, m[I]);
            col_mp<F, K, I + 1, IEND>(lo, hi, m);
        }
    } else if constexpr (I + 1 <= IEND) {
        mac96_c1<F::p(K - I)>(lo, hi, m[I]);
        col_mp<F, K, I + 1, IEND>(lo, hi, m);
    }
}
#include "mac_chains.inc"

// every limb of p in [LO, HI) is a literal (> 64, so not an inline constant and not zero): the whole m*p part of a
// column can be one statement with SGPR operands
template <class F, int LO, int HI>
constexpr bool lw_all_literal() {
    for (int i = LO; i < HI; i++)
        if (F::p(i) <= 64) return false;
    return true;
}

// column K, first chunk: acc = init + a[I]*b[K-I] + ...; returns via lo/hi, continues with col_ab
template <class F, int K, int I, int IEND>
__device__ __forceinline__ void col_ab_first(uint64_t &lo, uint32_t &hi, uint64_t init, const Fe<F> &a, const Fe<F> &b) {
    if constexpr (I + 4 <= IEND) {
        mac96_first_x4(lo, hi, init, a.v[I], b.v[K - I], a.v[I + 1], b.v[K - I - 1], a.v[I + 2], b.v[K - I - 2], a.v[I + 3], b.v[K - I - 3]);
        col_ab<F, K, I + 4, IEND>(lo, hi, a, b);
    } else if constexpr (I + 2 <= IEND) {
        mac96_first_x2(lo, hi, init, a.v[I], b.v[K - I], a.v[I + 1], b.v[K - I - 1]);
        col_ab<F, K, I + 2, IEND>(lo, hi, a, b);
    } else {
        static_assert(I + 1 <= IEND, "column without products");
        mac96_first_x1(lo, hi, init, a.v[I], b.v[K - I]);
    }
}
// Column K of sum_{q<P} a[q]*b[q] + m*p.  P > 1 accumulates several products before the single Montgomery
// reduction (fe_dot below): the a*b MACs of every product land in the same 96-bit column accumulator and the m*p
// MACs are paid once, so a sum of P products costs (P+1)*N^2 MACs instead of 2*P*N^2.
template <class F, int P, int K>
__device__ __forceinline__ void fips_col(uint64_t init, const Fe<F> *const (&a)[P], const Fe<F> *const (&b)[P], uint32_t (&m)[F::N],
                                         uint32_t (&t)[F::N]) {
    constexpr int N = F::N;
    if constexpr (K == 2 * N - 1) {
        t[N - 1] = (uint32_t)init;   // no products left: the last carry is the top limb
    } else {
        uint64_t lo;
        uint32_t hi;
        if constexpr (K < N) {
#if defined(LW_NO_COLUMN_CHAINS)
            col_ab_first<F, K, 0, K + 1>(lo, hi, init, *a[0], *b[0]);
#else
            col_ab_first_dispatch<F, K, 0, K + 1>(lo, hi, init, *a[0], *b[0]);
#endif
            if constexpr (P > 1) col_ab<F, K, 0, K + 1>(lo, hi, *a[1], *b[1]);
            if constexpr (P > 2) col_ab<F, K, 0, K + 1>(lo, hi, *a[2], *b[2]);
            if constexpr (P > 3) col_ab<F, K, 0, K + 1>(lo, hi, *a[3], *b[3]);
#if defined(LW_NO_COLUMN_CHAINS)
            col_mp<F, K, 0, K>(lo, hi, m);
#else
            if constexpr (K >= 1 && lw_all_literal<F, 1, K + 1>()) col_mp_dispatch<F, K, 0, K>(lo, hi, m);
            else col_mp<F, K, 0, K>(lo, hi, m);
#endif
            uint32_t mk;
            if constexpr (F::INV == 0xffffffffu) mk = 0u - (uint32_t)lo;
            else mk = (uint32_t)lo * F::INV;
            m[K] = mk;
            mac96_c1<F::p(0)>(lo, hi, mk);
        } else {
#if defined(LW_NO_COLUMN_CHAINS)
            col_ab_first<F, K, K - N + 1, N>(lo, hi, init, *a[0], *b[0]);
#else
            col_ab_first_dispatch<F, K, K - N + 1, 2 * N - 1 - K>(lo, hi, init, *a[0], *b[0]);
#endif
            if constexpr (P > 1) col_ab<F, K, K - N + 1, N>(lo, hi, *a[1], *b[1]);
            if constexpr (P > 2) col_ab<F, K, K - N + 1, N>(lo, hi, *a[2], *b[2]);
            if constexpr (P > 3) col_ab<F, K, K - N + 1, N>(lo, hi, *a[3], *b[3]);
#if defined(LW_NO_COLUMN_CHAINS)
            col_mp<F, K, K - N + 1, N>(lo, hi, m);
#else
            if constexpr (lw_all_literal<F, K - N + 1, N>()) col_mp_dispatch<F, K, K - N + 1, 2 * N - 1 - K>(lo, hi, m);
            else col_mp<F, K, K - N + 1, N>(lo, hi, m);
#endif
            t[K - N] = (uint32_t)lo;
        }
        fips_col<F, P, K + 1>((lo >> 32) | ((uint64_t)hi << 32), a, b, m, t);
    }
}
template <class F>
__device__ __forceinline__ Fe<F> fe_mul_gfx9(const Fe<F> &a, const Fe<F> &b) {
    constexpr int N = F::N;
    uint32_t m[N], t[N];
    const Fe<F> *const pa[1] = {&a}, *const pb[1] = {&b};
    fips_col<F, 1, 0>(0ull, pa, pb, m, t);
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return reduce_once<F>(r);
}
#endif

template <class F>
LW_HD Fe<F> fe_mul(const Fe<F> &a, const Fe<F> &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_gfx9<F>(a, b);
#elif defined(LW_HOST_MUL64)
    return fe_mul_host64<F>(a, b);
#else
    return fe_mul_portable<F>(a, b);
#endif
}

template <class F>
LW_HD Fe<F> fe_sqr(const Fe<F> &a) { return fe_mul<F>(a, a); }

// p - a without the zero check of fe_neg: in (0, p], congruent to -a.  Only ever fed to fe_dot.
template <class F>
LW_HD Fe<F> fe_neg_raw(const Fe<F> &a) {
    uint32_t pk[F::N];
    limbs_kp<F, 0>(pk);
    Fe<F> r;
    limbs_sub<F::N>(r.v, pk, a.v);
    return r;
}
// 2p - a for a < 2p: in (0, 2p], congruent to -a (an unreduced operand of fe_dot, counted as 2 in KSUM)
template <class F>
LW_HD Fe<F> fe_neg_raw_2p(const Fe<F> &a) {
    uint32_t pk[F::N];
    limbs_kp<F, 1>(pk);
    Fe<F> r;
    limbs_sub<F::N>(r.v, pk, a.v);
    return r;
}
// sum_{q<P} a[q]*b[q] * R^-1 mod p, canonical, operands <= p.  One reduction for P products; the result before the
// final subtraction is < (P*p/R + 1)*p, which must stay below 2p: P*(top limb + 1) <= 2^32 (Fp381: P <= 9, Fp254: P <= 5).
// KSUM: bound of sum a[q]*b[q] in units of p^2 — P for reduced operands; more when some are unreduced sums (e.g. an
// operand below 2p times one below 3p counts 6), as long as KSUM * p / R stays below 1 (Fp381: KSUM <= 9).  Device only.
template <class F, int P, int KSUM = P>
LW_HD Fe<F> fe_dot(const Fe<F> *const (&a)[P], const Fe<F> *const (&b)[P]) {
    static_assert(P >= 1 && P <= 4, "fe_dot: 1..4 products");
    static_assert(KSUM >= P, "fe_dot: KSUM counts every product at least once");
    static_assert((uint64_t)KSUM * ((uint64_t)F::p(F::N - 1) + 1) <= (1ull << 32), "fe_dot: sum of products would exceed 2p after reduction");
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int N = F::N;
    uint32_t m[N], t[N];
    fips_col<F, P, 0>(0ull, a, b, m, t);
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return reduce_once<F>(r);
#else
    static_assert(KSUM == P, "fe_dot: unreduced operands are a device-code feature");
    Fe<F> acc = fe_mul<F>(reduce_once<F>(*a[0]), *b[0]);
    for (int q = 1; q < P; q++) acc = fe_add<F>(acc, fe_mul<F>(reduce_once<F>(*a[q]), *b[q]));
    return acc;
#endif
}

// ---- lazy-reduction helpers (fields with F::LAZY; used by the NTT butterflies only) ----
// a - K*p if a >= K*p else a
template <class F, int LOGK>
LW_HD Fe<F> fe_cond_sub_kp(const Fe<F> &a) {
    constexpr int N = F::N;
    uint32_t kp[N];
    limbs_kp<F, LOGK>(kp);
    Fe<F> d;
    const uint32_t borrow = limbs_sub<N>(d.v, a.v, kp);
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}
// Full reduction of any 256-bit value for p = 2^251 + 17*2^192 + 1 (Stark252): q = x >> 251 is floor(x/p) or one
// more (x*(p - 2^251)/(p*2^251) < 2^-48), and q*p = q + 17q*2^192 + q*2^251 needs no multiplication chain:
// one 8-limb subtraction plus a conditional add of p.
LW_HD Fe<Stark252> fe_reduce_full(const Fe<Stark252> &x) {
    const uint32_t q = x.v[7] >> 27;
    const uint32_t z = lw_k(0u);
    const uint32_t sub[8] = {q, z, z, z, z, z, 17u * q, q << 27};
    Fe<Stark252> d;
    const uint32_t borrow = limbs_sub<8>(d.v, x.v, sub);
    uint32_t pm[8];
    limbs_p_masked<Stark252>(pm, 0u - borrow);
    Fe<Stark252> r;
    limbs_add<8>(r.v, d.v, pm);
    return r;
}
template <class F>
LW_HD Fe<F> fe_reduce_full(const Fe<F> &x) { return reduce_once<F>(x); }   // non-lazy fields never leave [0, 2p)

// plain N-limb add / (a + 2p - b); the caller guarantees the range fits N limbs
template <class F>
LW_HD Fe<F> fe_add_raw(const Fe<F> &a, const Fe<F> &b) {
    Fe<F> s;
    limbs_add<F::N>(s.v, a.v, b.v);
    return s;
}
template <class F>
LW_HD Fe<F> fe_add2p_sub_raw(const Fe<F> &a, const Fe<F> &b) {   // a + 2p - b, b <= a + 2p
    constexpr int N = F::N;
    uint32_t kp[N];
    limbs_kp<F, 1>(kp);
    Fe<F> d, r;
    limbs_sub<N>(d.v, a.v, b.v);      // wraps mod 2^(32N) when b > a; adding 2p brings it back into range
    limbs_add<N>(r.v, d.v, kp);
    return r;
}
// Montgomery product without the final conditional subtraction: result in [0, 2p) whenever a < p
// (b may be any N-limb value): (a*b + m*p) / R < a*b/R + p < 2p.
template <class F>
LW_HD Fe<F> fe_mul_lazy(const Fe<F> &a, const Fe<F> &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int N = F::N;
    uint32_t m[N], t[N];
    const Fe<F> *const pa[1] = {&a}, *const pb[1] = {&b};
    fips_col<F, 1, 0>(0ull, pa, pb, m, t);
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = t[i];
    return r;
#else
    return fe_mul_portable<F>(a, b);
#endif
}

// x^e for a small exponent, Montgomery domain
template <class F>
LW_HD Fe<F> fe_pow_u64(Fe<F> base, uint64_t e) {
    Fe<F> r = Fe<F>::one();
    while (e) {
        if (e & 1) r = fe_mul<F>(r, base);
        base = fe_sqr<F>(base);
        e >>= 1;
    }
    return r;
}

// Inverse by Fermat (a^(p-2)); result is the canonical residue, hence bit-identical to the reference's
// binary-EEA inv (montgomery_backed_prime_fields.rs:175-247).  Off the hot path (a handful per call).
template <class F>
LW_HD Fe<F> fe_inv(const Fe<F> &a) {
    constexpr int N = F::N;
    uint32_t e[N];
    // e = p - 2
    uint64_t borrow = 2;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)F::p(i) - borrow;
        e[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    Fe<F> r = Fe<F>::one();
    for (int i = N - 1; i >= 0; i--) {
        for (int bit = 31; bit >= 0; bit--) {
            r = fe_sqr<F>(r);
            if ((e[i] >> bit) & 1) r = fe_mul<F>(r, a);
        }
    }
    return r;
}

// ---- inversion by a bounded binary GCD ---------------------------------------------------------------------------------
// FieldElement::inv in the reference is a binary extended Euclid on the Montgomery representation
// (montgomery_backed_prime_fields.rs:175-247); fe_inv above is Fermat (p - 2 is a 380-bit exponent: ~570 products).  This
// is the binary GCD in the form Pornin gave it ("Optimized binary GCD for modular inversion", 2020; BearSSL's
// br_i31_moddiv), re-cut for 32-bit limbs: a = y, b = p, u = 1, v = 0 with a = y*u, b = y*v (mod p) throughout.  The 30
// steps of a round — halve an even a or b, else subtract the smaller from the larger and halve — are decided on the top 64
// and low 32 bits of a and b only and recorded as a 2x2 matrix of factors |f| <= 2^30; then a, b and u, v are updated
// with it in one sweep each (exact division by 2^30 for a, b; Montgomery-style division mod p for u, v).  Every round
// shortens len(a) + len(b) by at least 29 bits, so 2*bits/29 + 2 rounds end with {a, b} = {1, 0} and the inverse in u | v.
// Branch-free: every lane of a wave runs the same instructions.  Cost ~ 55 K VALU instructions for Fp381 (~85 products)
// against ~570 products; same results (tests/test_host_sanitizers.py checks it against fe_inv on the host, every GPU MSM
// test runs through it).
template <class F>
LW_HD constexpr int fe_modulus_bits() {
    int top = 0;
    for (int b = 0; b < 32; b++)
        if ((F::p(F::N - 1) >> b) & 1u) top = b + 1;
    return 32 * (F::N - 1) + top;
}
// x = (x*f + y*g) >> 30 and y = (x*h + y*k) >> 30 over N-limb magnitudes (exact divisions); a negative result is negated
// and reported (bit 0: x, bit 1: y)
template <int N>
LW_HD uint32_t bingcd_update_ab(uint32_t (&x)[N], uint32_t (&y)[N], int32_t f, int32_t g, int32_t h, int32_t k) {
    int64_t cx = 0, cy = 0;
    uint32_t wx[N + 1], wy[N + 1];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int64_t zx = (int64_t)x[i] * f + (int64_t)y[i] * g + cx;
        const int64_t zy = (int64_t)x[i] * h + (int64_t)y[i] * k + cy;
        wx[i] = (uint32_t)zx;
        wy[i] = (uint32_t)zy;
        cx = zx >> 32;
        cy = zy >> 32;
    }
    wx[N] = (uint32_t)cx;
    wy[N] = (uint32_t)cy;
    const uint32_t nx = (uint32_t)((uint64_t)cx >> 63), ny = (uint32_t)((uint64_t)cy >> 63);
    // shift right by 30 and negate when negative: -r = ~r + 1
    uint32_t carx = nx, cary = ny;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t rx = (wx[i] >> 30) | (wx[i + 1] << 2), ry = (wy[i] >> 30) | (wy[i + 1] << 2);
        rx ^= 0u - nx;
        ry ^= 0u - ny;
        const uint32_t sx = rx + carx, sy = ry + cary;
        carx = sx < rx ? 1u : 0u;
        cary = sy < ry ? 1u : 0u;
        x[i] = sx;
        y[i] = sy;
    }
    return nx | (ny << 1);
}
// u = (u*f + v*g) / 2^30 mod p, v = (u*h + v*k) / 2^30 mod p for u, v in [0, p)
template <class F>
LW_HD void bingcd_update_uv(uint32_t (&u)[F::N], uint32_t (&v)[F::N], int32_t f, int32_t g, int32_t h, int32_t k) {
    constexpr int N = F::N;
    // multiple of p that makes the sum divisible by 2^30: F::INV = -p^-1 mod 2^32
    const uint32_t fu = ((uint32_t)(u[0] * (uint32_t)f + v[0] * (uint32_t)g) * F::INV) & 0x3fffffffu;
    const uint32_t fv = ((uint32_t)(u[0] * (uint32_t)h + v[0] * (uint32_t)k) * F::INV) & 0x3fffffffu;
    int64_t cu = 0, cv = 0;
    uint32_t wu[N + 1], wv[N + 1];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int64_t zu = (int64_t)u[i] * f + (int64_t)v[i] * g + (int64_t)((uint64_t)F::p(i) * fu) + cu;
        const int64_t zv = (int64_t)u[i] * h + (int64_t)v[i] * k + (int64_t)((uint64_t)F::p(i) * fv) + cv;
        wu[i] = (uint32_t)zu;
        wv[i] = (uint32_t)zv;
        cu = zu >> 32;
        cv = zv >> 32;
    }
    wu[N] = (uint32_t)cu;
    wv[N] = (uint32_t)cv;
    const uint32_t nu = (uint32_t)((uint64_t)cu >> 63), nv = (uint32_t)((uint64_t)cv >> 63);
    uint32_t ru[N], rv[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        ru[i] = (wu[i] >> 30) | (wu[i + 1] << 2);
        rv[i] = (wv[i] >> 30) | (wv[i + 1] << 2);
    }
    // the quotient lies in (-p, 2p): add p when negative, else subtract p when >= p
    uint32_t pk[N], du[N], dv[N], su[N], sv[N];
#pragma unroll
    for (int i = 0; i < N; i++) pk[i] = F::p(i);
    const uint32_t bu = limbs_sub<N>(du, ru, pk), bv = limbs_sub<N>(dv, rv, pk);
    limbs_add<N>(su, ru, pk);
    limbs_add<N>(sv, rv, pk);
#pragma unroll
    for (int i = 0; i < N; i++) {
        u[i] = nu ? su[i] : (bu ? ru[i] : du[i]);
        v[i] = nv ? sv[i] : (bv ? rv[i] : dv[i]);
    }
}
// the integer inverse of y modulo p, 0 < y < p (0 -> 0)
template <class F>
LW_HD Fe<F> fe_inv_int_bingcd(const Fe<F> &y) {
    constexpr int N = F::N;
    constexpr int ROUNDS = (2 * fe_modulus_bits<F>()) / 29 + 2;
    uint32_t a[N], b[N], u[N], v[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        a[i] = y.v[i];
        b[i] = F::p(i);
        u[i] = i == 0 ? 1u : 0u;
        v[i] = 0u;
    }
#pragma nounroll
    for (int round = 0; round < ROUNDS; round++) {
        // the two most significant limbs at the highest position where a or b is non-zero (one limb when that is limb 0)
        uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = 0xffffffffu, c1 = 0xffffffffu;
#pragma unroll
        for (int j = N - 1; j >= 0; j--) {
            const uint32_t aw = a[j], bw = b[j];
            a0 ^= (a0 ^ aw) & c0;
            a1 ^= (a1 ^ aw) & c1;
            b0 ^= (b0 ^ bw) & c0;
            b1 ^= (b1 ^ bw) & c1;
            c1 = c0;
            c0 &= ((aw | bw) == 0u) ? 0xffffffffu : 0u;
        }
        a1 |= a0 & c1;
        a0 &= ~c1;
        b1 |= b0 & c1;
        b0 &= ~c1;
        uint64_t a_hi = ((uint64_t)a0 << 32) | a1, b_hi = ((uint64_t)b0 << 32) | b1;
        uint32_t a_lo = a[0], b_lo = b[0];
        int32_t pa = 1, pb = 0, qa = 0, qb = 1;
#pragma unroll 2
        for (int i = 0; i < 30; i++) {
            const uint32_t r = a_hi > b_hi ? 1u : 0u;
            const uint32_t oa = (a_lo >> i) & 1u, ob = (b_lo >> i) & 1u;
            const uint32_t cAB = oa & ob & r, cBA = oa & ob & (r ^ 1u), cA = cAB | (oa ^ 1u);
            const uint32_t mAB = 0u - cAB, mBA = 0u - cBA, mA = 0u - cA;
            a_lo -= b_lo & mAB;
            a_hi -= b_hi & (uint64_t)(int64_t)(int32_t)mAB;
            pa -= qa & (int32_t)mAB;
            pb -= qb & (int32_t)mAB;
            b_lo -= a_lo & mBA;
            b_hi -= a_hi & (uint64_t)(int64_t)(int32_t)mBA;
            qa -= pa & (int32_t)mBA;
            qb -= pb & (int32_t)mBA;
            // the halved side keeps its low word and factors, the other side doubles them: the common 2^30 goes at the end
            a_lo += a_lo & ~mA;
            pa += pa & (int32_t)~mA;
            pb += pb & (int32_t)~mA;
            a_hi = cA ? (a_hi >> 1) : a_hi;
            b_lo += b_lo & mA;
            qa += qa & (int32_t)mA;
            qb += qb & (int32_t)mA;
            b_hi = cA ? b_hi : (b_hi >> 1);
        }
        const uint32_t neg = bingcd_update_ab<N>(a, b, pa, pb, qa, qb);
        if (neg & 1u) { pa = -pa; pb = -pb; }
        if (neg & 2u) { qa = -qa; qb = -qb; }
        bingcd_update_uv<F>(u, v, pa, pb, qa, qb);
    }
    Fe<F> r;
    const bool zero = y.is_zero();
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = zero ? 0u : (u[i] | v[i]);   // {a, b} = {1, 0}: the side that reached 0 has factor 0
    return r;
}
// Montgomery-form inverse: (a R)^-1 = a^-1 R^-1 as an integer; times R^3 / R / R ... one product with R^3 gives a^-1 R
template <class F>
LW_HD Fe<F> fe_inv_fast(const Fe<F> &a) {
    const Fe<F> r3 = fe_mul<F>(Fe<F>::r2(), Fe<F>::r2());   // R^2 * R^2 / R = R^3
    return fe_mul<F>(fe_inv_int_bingcd<F>(a), r3);           // a^-1 R^-1 * R^3 / R = a^-1 R
}

// to / from Montgomery form (from_base_type :280-282, representative :291-293)
template <class F>
LW_HD Fe<F> fe_to_mont(const Fe<F> &a) { return fe_mul<F>(a, Fe<F>::r2()); }
template <class F>
LW_HD Fe<F> fe_from_mont(const Fe<F> &a) {
    Fe<F> o = Fe<F>::zero();
    o.v[0] = 1;
    return fe_mul<F>(a, o);
}
template <class F>
LW_HD Fe<F> fe_from_u64(uint64_t x) {
    Fe<F> o = Fe<F>::zero();
    o.v[0] = (uint32_t)x;
    o.v[1] = (uint32_t)(x >> 32);
    return fe_to_mont<F>(o);
}

// ---------------------------------------------------------------- boundary layout
// Reference memory: N/2 u64 limbs, limbs[0] most significant, each u64 native little-endian.
// As a u32 array m[]: m[2i] = lo32(limb i), m[2i+1] = hi32(limb i).  Internal limb k (LS first) sits at
// m[2*(N/2-1-k/2) + (k&1)].
template <class F>
LW_HD constexpr int mem_index(int k) { return 2 * (F::N / 2 - 1 - k / 2) + (k & 1); }

template <class F>
LW_HD Fe<F> fe_load(const void *ptr) {
    constexpr int N = F::N;
    const uint4 *q = reinterpret_cast<const uint4 *>(ptr);
    uint32_t m[N];
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
        uint4 x = q[i];
        m[4 * i + 0] = x.x; m[4 * i + 1] = x.y; m[4 * i + 2] = x.z; m[4 * i + 3] = x.w;
    }
    Fe<F> r;
#pragma unroll
    for (int k = 0; k < N; k++) r.v[k] = m[mem_index<F>(k)];
    return r;
}
template <class F>
LW_HD void fe_store(void *ptr, const Fe<F> &a) {
    constexpr int N = F::N;
    uint32_t m[N];
#pragma unroll
    for (int k = 0; k < N; k++) m[mem_index<F>(k)] = a.v[k];
    uint4 *q = reinterpret_cast<uint4 *>(ptr);
#pragma unroll
    for (int i = 0; i < N / 4; i++) q[i] = make_uint4(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
}

// ---------------------------------------------------------------- BabyBear (31-bit), R = 2^32 in registers
// math/src/field/fields/u32_montgomery_backend_prime_field.rs:84-118,273-303
struct BabyBear {
    static constexpr uint32_t P = 2013265921u;   // 0x78000001
    static constexpr uint32_t MU = 0x88000001u;  // +p^-1 mod 2^32 (NOT negated, :33-49)
    static constexpr uint32_t R2 = 0x45dddde3u;  // 2^64 mod p
    static constexpr uint32_t ONE = 0x0ffffffeu; // 2^32 mod p
    static constexpr uint32_t TWO_ADICITY = 24;  // babybear.rs:29 / babybear_u32.rs:17
    static constexpr uint32_t ROOT = 21;
};

LW_HD uint32_t bb_reduce(uint64_t x) {   // x * 2^-32 mod p, x < p * 2^32
    // m = -lo(x) * p^-1, so x + m*p is a multiple of 2^32 and (x + m*p) / 2^32 < 2p is congruent to x * 2^-32: the same
    // canonical residue as the reference's montgomery_reduction (u32_montgomery_backend_prime_field.rs:278-292), which
    // subtracts with the un-negated inverse instead.  On gfx950 this is v_mul_lo + v_mad_u64_u32 + v_sub + v_min.
    const uint32_t m = (uint32_t)x * (0u - BabyBear::MU);
    const uint32_t r = (uint32_t)((x + (uint64_t)m * BabyBear::P) >> 32);
    const uint32_t d = r - BabyBear::P;      // wraps above r exactly when r < p
    return d < r ? d : r;
}
LW_HD uint32_t bb_mul(uint32_t a, uint32_t b) { return bb_reduce((uint64_t)a * b); }
LW_HD uint32_t bb_add(uint32_t a, uint32_t b) {
    const uint32_t s = a + b, d = s - BabyBear::P;
    return d < s ? d : s;
}
LW_HD uint32_t bb_sub(uint32_t a, uint32_t b) {
    const uint32_t d = a - b, e = d + BabyBear::P;   // a < b: d wraps to 2^32 - (b - a) and e = p - (b - a)
    return e < d ? e : d;
}
LW_HD uint32_t bb_pow(uint32_t a, uint64_t e) {
    uint32_t r = BabyBear::ONE;
    while (e) {
        if (e & 1) r = bb_mul(r, a);
        a = bb_mul(a, a);
        e >>= 1;
    }
    return r;
}
LW_HD uint32_t bb_inv(uint32_t a) { return bb_pow(a, (uint64_t)BabyBear::P - 2); }
// The reference also keeps BabyBear as MontgomeryBackendPrimeField<_,1> with R = 2^64 (babybear.rs:19-20):
// memory word = a*2^64 mod p in a u64.  Convert at the boundary: r64 -> r32 is one reduction (x*2^-32),
// r32 -> r64 is a Montgomery product with 2^64 mod p (x*2^64*2^-32 = x*2^32).
LW_HD uint32_t bb_from_r64(uint64_t w) { return bb_reduce(w); }            // w < p
LW_HD uint64_t bb_to_r64(uint32_t v) { return (uint64_t)bb_mul(v, BabyBear::R2); }

}  // namespace lw
