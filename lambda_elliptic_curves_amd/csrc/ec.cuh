// Short-Weierstrass (a = 0) group law in homogeneous projective coordinates for the four groups in scope.
//
// The reference's operate_with (math/src/elliptic_curve/short_weierstrass/point.rs:171-207) branches on the
// exceptional cases (P+P, P+(-P), identity).  A bucket accumulation of a structured SRS can hit all of them, and
// divergent branches cost a whole wavefront, so the device uses the COMPLETE addition law of Renes-Costello-
// Batina (Eurocrypt 2016, Alg. 7/9 for a = 0): one straight-line formula, valid for every pair of inputs
// including the identity (0:1:0), on curves without rational 2-torsion (all four groups have odd order).
// The group element computed is the same; only the projective representative differs, and the reference
// itself defines point equality up to scaling (math/src/elliptic_curve/point.rs:57-63).
#pragma once
#include "field.cuh"

namespace lw {

// ---------------------------------------------------------------- base-field op packs
template <class F>
struct FpOps {
    using T = Fe<F>;
    using Field = F;
    static constexpr int WORDS32 = F::N;
    LW_HD static T add(const T &a, const T &b) { return fe_add<F>(a, b); }
    LW_HD static T sub(const T &a, const T &b) { return fe_sub<F>(a, b); }
    // A point addition holds ~10 field elements live; letting hipcc interleave its independent products
    // (for ILP) pushes a 384-bit kernel past 256 VGPRs.  Pinning the products in program order keeps the
    // live set near its minimum so 2-3 waves fit per SIMD, which is what hides the MAC-chain latency
    // (profiles/r01_microbench.txt: dependent chains reach 72 % of the MAC rate at 3 waves/SIMD).
    LW_HD static T mul(const T &a, const T &b) {
        T r = fe_mul<F>(a, b);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        return r;
    }
    // a*b + c*d and a*b - c*d with one Montgomery reduction (fe_dot)
    LW_HD static T dot2(const T &a, const T &b, const T &c, const T &d) {
        const T *const pa[2] = {&a, &c}, *const pb[2] = {&b, &d};
        T r = fe_dot<F, 2>(pa, pb);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        return r;
    }
    LW_HD static T dot2s(const T &a, const T &b, const T &c, const T &d) { return dot2(a, b, fe_neg_raw<F>(c), d); }
#if defined(__HIP_DEVICE_COMPILE__)
    // Unreduced sums as product operands (fields with headroom only, see fe_dot's KSUM): a + b below 2p for reduced a, b
    // saves the conditional subtraction (24 of 36 instructions at 12 limbs).  K* = bound of the products' sum in p^2.
    static constexpr bool HEADROOM = (uint64_t)9 * ((uint64_t)F::p(F::N - 1) + 1) <= (1ull << 32);    // Fp381: 2^384 / p = 9.8
    static constexpr bool HEADROOM5 = (uint64_t)5 * ((uint64_t)F::p(F::N - 1) + 1) <= (1ull << 32);   // Fp254: 2^256 / p = 5.3
    __device__ static T add_nr(const T &a, const T &b) {
        T s;
        limbs_add<F::N>(s.v, a.v, b.v);
        return s;
    }
    template <int K>
    __device__ static T mul_k(const T &a, const T &b) {
        const T *const pa[1] = {&a}, *const pb[1] = {&b};
        T r = fe_dot<F, 1, K>(pa, pb);
        __builtin_amdgcn_sched_barrier(0);
        return r;
    }
    template <int K>
    __device__ static T dot2_k(const T &a, const T &b, const T &c, const T &d) {
        const T *const pa[2] = {&a, &c}, *const pb[2] = {&b, &d};
        T r = fe_dot<F, 2, K>(pa, pb);
        __builtin_amdgcn_sched_barrier(0);
        return r;
    }
#endif
    LW_HD static T sqr(const T &a) { return mul(a, a); }
    LW_HD static T neg(const T &a) { return fe_neg<F>(a); }
    LW_HD static T dbl(const T &a) { return fe_add<F>(a, a); }
    // value held by the neighbouring lane (lanes 2i <-> 2i+1), one DPP move per limb; device code only
    LW_HD static T lane_swap(const T &a) {
#if defined(__HIP_DEVICE_COMPILE__)
        T r;
#pragma unroll
        for (int i = 0; i < F::N; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1, 0xF, 0xF, true);
        return r;
#else
        return a;
#endif
    }
    LW_HD static T select(bool c, const T &a, const T &b) {
        T r;
#pragma unroll
        for (int i = 0; i < F::N; i++) r.v[i] = c ? a.v[i] : b.v[i];
        return r;
    }
    // value held by another lane, one DPP move per limb; device code only.  CTRL: 0x00-0xFF quad_perm (lane i of a quad
    // reads lane (CTRL >> 2i) & 3), 0x110 + n row_shr:n (lane i reads lane i - n of its row of 16)
    template <int CTRL>
    LW_HD static T dpp(const T &a) {
#if defined(__HIP_DEVICE_COMPILE__)
        T r;
#pragma unroll
        for (int i = 0; i < F::N; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], CTRL, 0xF, 0xF, true);
        return r;
#else
        return a;
#endif
    }
    LW_HD static T zero() { return T::zero(); }
    LW_HD static T one() { return T::one(); }
    LW_HD static bool is_zero(const T &a) { return a.is_zero(); }
    LW_HD static bool eq(const T &a, const T &b) { return a == b; }
    LW_HD static T inv(const T &a) { return fe_inv_fast<F>(a); }   // bounded binary GCD (field.cuh), ~7x cheaper than Fermat
    // reference memory (N/2 u64, MS limb first) <-> limbs
    LW_HD static T load(const void *p) { return fe_load<F>(p); }
    LW_HD static void store(void *p, const T &a) { fe_store<F>(p, a); }
    static constexpr int BYTES = F::N * 4;
};

// Fp2 = Fp[u]/(u^2+1); element stored as [c0, c1] (bls12_381/field_extension.rs:29, bn_254/field_extension.rs:36).
// Karatsuba product for both towers: the result is the same field element as the reference's schoolbook
// BN254 product (bn_254/field_extension.rs:47-49).
template <class F>
struct Fp2 {
    Fe<F> c0, c1;
};
template <class F>
struct Fp2Ops {
    using T = Fp2<F>;
    static constexpr int WORDS32 = 2 * F::N;
    LW_HD static T add(const T &a, const T &b) { return T{fe_add<F>(a.c0, b.c0), fe_add<F>(a.c1, b.c1)}; }
    LW_HD static T sub(const T &a, const T &b) { return T{fe_sub<F>(a.c0, b.c0), fe_sub<F>(a.c1, b.c1)}; }
    LW_HD static void fence() {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    // (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u: two 2-term dot products (6 N^2 MACs like Karatsuba,
    // but two reductions instead of three and none of its five additions)
    LW_HD static T mul(const T &a, const T &b) {
        const Fe<F> na1 = fe_neg_raw<F>(a.c1);
        const Fe<F> *const ra[2] = {&a.c0, &na1}, *const rb[2] = {&b.c0, &b.c1};
        const Fe<F> *const ia[2] = {&a.c0, &a.c1}, *const ib[2] = {&b.c1, &b.c0};
        T r;
        r.c0 = fe_dot<F, 2>(ra, rb);
        fence();
        r.c1 = fe_dot<F, 2>(ia, ib);
        fence();
        return r;
    }
    // a*b + c*d over Fp2: each component is a 4-term dot product
    LW_HD static T dot2(const T &a, const T &b, const T &c, const T &d) {
        const Fe<F> na1 = fe_neg_raw<F>(a.c1), nc1 = fe_neg_raw<F>(c.c1);
        const Fe<F> *const ra[4] = {&a.c0, &na1, &c.c0, &nc1}, *const rb[4] = {&b.c0, &b.c1, &d.c0, &d.c1};
        const Fe<F> *const ia[4] = {&a.c0, &a.c1, &c.c0, &c.c1}, *const ib[4] = {&b.c1, &b.c0, &d.c1, &d.c0};
        T r;
        r.c0 = fe_dot<F, 4>(ra, rb);
        fence();
        r.c1 = fe_dot<F, 4>(ia, ib);
        fence();
        return r;
    }
    LW_HD static T dot2s(const T &a, const T &b, const T &c, const T &d) {
        return dot2(a, b, T{fe_neg_raw<F>(c.c0), fe_neg_raw<F>(c.c1)}, d);
    }
    LW_HD static T sqr(const T &a) {
        Fe<F> v0 = fe_mul<F>(a.c0, a.c1);
        Fe<F> c0 = fe_mul<F>(fe_add<F>(a.c0, a.c1), fe_sub<F>(a.c0, a.c1));
        return T{c0, fe_add<F>(v0, v0)};
    }
    LW_HD static T neg(const T &a) { return T{fe_neg<F>(a.c0), fe_neg<F>(a.c1)}; }
    LW_HD static T dbl(const T &a) { return add(a, a); }
    LW_HD static T lane_swap(const T &a) { return T{FpOps<F>::lane_swap(a.c0), FpOps<F>::lane_swap(a.c1)}; }
    LW_HD static T select(bool c, const T &a, const T &b) { return T{FpOps<F>::select(c, a.c0, b.c0), FpOps<F>::select(c, a.c1, b.c1)}; }
    template <int CTRL>
    LW_HD static T dpp(const T &a) { return T{FpOps<F>::template dpp<CTRL>(a.c0), FpOps<F>::template dpp<CTRL>(a.c1)}; }
    LW_HD static T zero() { return T{Fe<F>::zero(), Fe<F>::zero()}; }
    LW_HD static T one() { return T{Fe<F>::one(), Fe<F>::zero()}; }
    LW_HD static bool is_zero(const T &a) { return a.c0.is_zero() && a.c1.is_zero(); }
    LW_HD static bool eq(const T &a, const T &b) { return a.c0 == b.c0 && a.c1 == b.c1; }
    LW_HD static T inv(const T &a) {
        Fe<F> n = fe_inv_fast<F>(fe_add<F>(fe_sqr<F>(a.c0), fe_sqr<F>(a.c1)));
        return T{fe_mul<F>(a.c0, n), fe_mul<F>(fe_neg<F>(a.c1), n)};
    }
    LW_HD static T load(const void *p) {
        return T{fe_load<F>(p), fe_load<F>((const char *)p + F::N * 4)};
    }
    LW_HD static void store(void *p, const T &a) {
        fe_store<F>(p, a.c0);
        fe_store<F>((char *)p + F::N * 4, a.c1);
    }
    static constexpr int BYTES = 2 * F::N * 4;
};

// ---------------------------------------------------------------- curves (b3 = 3*b)
// ACC_WAVES: waves per SIMD the MSM accumulate kernel is register-budgeted for
struct Bls12381G1 {   // y^2 = x^3 + 4   (bls12_381/curve.rs:38-46)
    using B = FpOps<Fp381>;
    static constexpr int ACC_WAVES = 2;
    LW_HD static B::T mul_b3(const B::T &x) {   // 12x
        B::T x4 = B::dbl(B::dbl(x));
        return B::add(B::dbl(x4), x4);
    }
};
struct Bn254G1 {      // y^2 = x^3 + 3   (bn_254/curve.rs:32-40)
    using B = FpOps<Fp254>;
    static constexpr int ACC_WAVES = 2;
    LW_HD static B::T mul_b3(const B::T &x) {   // 9x
        return B::add(B::dbl(B::dbl(B::dbl(x))), x);
    }
};
struct Bls12381G2 {   // y^2 = x^3 + 4(1+u)   (bls12_381/twist.rs:40-48)
    using B = Fp2Ops<Fp381>;
    static constexpr int ACC_WAVES = 2;   // 256 VGPRs + 77 spilled; a 1-wave budget (322 VGPRs, no spills) measured 15 % slower
    LW_HD static B::T mul_b3(const B::T &x) {   // 12(1+u) * (x0 + x1 u) = 12(x0 - x1) + 12(x0 + x1) u
        Fe<Fp381> d = fe_sub<Fp381>(x.c0, x.c1), s = fe_add<Fp381>(x.c0, x.c1);
        Fe<Fp381> d4 = fe_dbl<Fp381>(fe_dbl<Fp381>(d)), s4 = fe_dbl<Fp381>(fe_dbl<Fp381>(s));
        return B::T{fe_add<Fp381>(fe_dbl<Fp381>(d4), d4), fe_add<Fp381>(fe_dbl<Fp381>(s4), s4)};
    }
};
struct Bn254G2 {      // y^2 = x^3 + 3/(9+u)   (bn_254/twist.rs:46-61); 3b' precomputed, Montgomery form
    using B = Fp2Ops<Fp254>;
    static constexpr int ACC_WAVES = 2;
    LW_HD static constexpr uint32_t b3c0(int i) {
        constexpr uint32_t t[8] = {0xb62e0d6au, 0x3baa927cu, 0xd1b664fdu, 0xd71e7c52u, 0xd95d4664u, 0x03873e63u, 0x082ab8f4u, 0x0e75b5b1u};
        return t[i];
    }
    LW_HD static constexpr uint32_t b3c1(int i) {
        constexpr uint32_t t[8] = {0x7596fe35u, 0xaab7c666u, 0xbb6a27bau, 0x31d21a78u, 0x680401ffu, 0x85dd7297u, 0xdf39a7e9u, 0x03c52d6au};
        return t[i];
    }
    LW_HD static B::T mul_b3(const B::T &x) {
        B::T k;
#pragma unroll
        for (int i = 0; i < 8; i++) { k.c0.v[i] = b3c0(i); k.c1.v[i] = b3c1(i); }
        return B::mul(x, k);
    }
};

// The BN254 twist has a generic constant (b' = 3/(9+u)), so the two multiplications by b3 = 3b' of every complete addition
// are full Fp2 products: 12 N^2 of the 72 N^2 MACs of a mixed addition.  The curve y^2 = x^3 + (9 + u) is isomorphic to it
// under (x, y) -> (L^2 x, L^3 y) with L^6 = (9 + u)/b' = (9 + u)^2 / 3 (a sixth power in Fp2; L computed with plain big-integer
// arithmetic, checked on the generator), and there b3 = 3(9 + u) costs 14 field additions.  Point sets that the library
// normalises anyway (msm_to_affine_kernel: every MSM from 2^22 points, every lw_hip_srs_*) are mapped while they are
// normalised (two more products per point, once), the whole Pippenger then runs on the isomorphic curve
// (MsmRunner<Bn254G2Iso>) and the single result is mapped back.  Same group element, bit-identical output.
struct Bn254G2Iso {
    using B = Fp2Ops<Fp254>;
    static constexpr int ACC_WAVES = 2;
    LW_HD static Fe<Fp254> times9(const Fe<Fp254> &a) {
        Fe<Fp254> a8 = fe_dbl<Fp254>(fe_dbl<Fp254>(fe_dbl<Fp254>(a)));
        return fe_add<Fp254>(a8, a);
    }
    LW_HD static B::T mul_b3(const B::T &x) {   // 3 (9 + u)(x0 + x1 u) = 3 [(9 x0 - x1) + (x0 + 9 x1) u]
        const Fe<Fp254> r0 = fe_sub<Fp254>(times9(x.c0), x.c1), r1 = fe_add<Fp254>(x.c0, times9(x.c1));
        return B::T{fe_add<Fp254>(fe_dbl<Fp254>(r0), r0), fe_add<Fp254>(fe_dbl<Fp254>(r1), r1)};
    }
    // L^2, L^3 and their inverses, Montgomery form, least significant limb first
    LW_HD static constexpr uint32_t k(int which, int comp, int i) {
        constexpr uint32_t t[4][2][8] = {
            {{0xf4ea760eu, 0x863358b0u, 0x8866346au, 0x5542f3eau, 0x320755ebu, 0xc1a1b0feu, 0x8db66114u, 0x26bed9dfu},
             {0xcf5aeac5u, 0x64175e24u, 0xaab12f2eu, 0x917d0fa8u, 0x20ce7296u, 0x251213edu, 0x507803a1u, 0x0bff36bcu}},
            {{0x26dc456du, 0x922bc93bu, 0xfc6bd081u, 0xc7fab160u, 0x6075cf91u, 0x939a58a8u, 0xa23e25e7u, 0x222afc9eu},
             {0x8dac7c1cu, 0x8159c18bu, 0xfb52354eu, 0x1cb9ad90u, 0x26651e6fu, 0xd9c4ca12u, 0x742c0bffu, 0x1f3b4390u}},
            {{0x628ccb76u, 0xaa1e5596u, 0x7632787du, 0xc4a592f3u, 0xd5fba75bu, 0x2ca71636u, 0x469ac37fu, 0x251b9825u},
             {0xab22864cu, 0xc7a4b634u, 0xed37081fu, 0xe87a1ac9u, 0x83c7e14fu, 0x05a828d3u, 0xd881bdb9u, 0x0fc3a027u}},
            {{0x521e31d6u, 0x86edc1fau, 0x9e0038bbu, 0xa29b17ecu, 0xc638e992u, 0x4f43fbe0u, 0x4677b41eu, 0x0b249a75u},
             {0x3215c5f8u, 0x461ab99fu, 0xbd1e697du, 0x8871022eu, 0xf4fd856cu, 0x58c34f79u, 0xb7d214bdu, 0x0380d139u}}};
        return t[which][comp][i];
    }
    LW_HD static B::T konst(int which) {   // 0: L^2, 1: L^3, 2: L^-2, 3: L^-3
        B::T r;
#pragma unroll
        for (int i = 0; i < 8; i++) { r.c0.v[i] = k(which, 0, i); r.c1.v[i] = k(which, 1, i); }
        return r;
    }
};
// BLS12-381 G1 has b = 4, so every complete addition multiplies twice by b3 = 12: four reduced additions each, 288 of a
// mixed addition's ~6150 instructions.  1/6 is a sixth power in Fp (L^6 = 1/6, L computed with plain big-integer
// arithmetic and checked on the generator), so (x, y) -> (L^2 x, L^3 y) lands on y^2 = x^3 + 2/3, where b3 = 2 is ONE
// addition.  Used exactly like Bn254G2Iso: normalised point sets are mapped while they are normalised, the Pippenger runs
// on the isomorphic curve, the one result is mapped back.  Same group element, bit-identical output.
struct Bls12381G1Iso {
    using B = FpOps<Fp381>;
    static constexpr int ACC_WAVES = 2;
    LW_HD static B::T mul_b3(const B::T &x) { return B::dbl(x); }   // 3 * (2/3) = 2
    // L^2, L^3 and their inverses, Montgomery form, least significant limb first
    LW_HD static constexpr uint32_t k(int which, int i) {
        constexpr uint32_t t[4][12] = {
            {0x15377704u, 0x6ed8f73eu, 0x1f3b7cebu, 0x8921d5c2u, 0xe0844ffeu, 0xac49fd07u, 0xb61bbf57u, 0xc8f90ce5u, 0x7a43682bu, 0x2ea1d8ebu, 0xe3242f6au, 0x0d4c1fabu},   // L^2
            {0xeaf23d8cu, 0xd8a04617u, 0xdc37d6e0u, 0x22f8df4cu, 0x047f015bu, 0x1f9dad26u, 0x6ab9b82cu, 0xb2d5d4b4u, 0x70b6c83du, 0x77cdd5beu, 0x09fc1c8du, 0x0aafebafu},   // L^3
            {0x8ff0085au, 0xca5a8e1eu, 0x2ebd0db3u, 0xf0a3ac83u, 0x71706d4au, 0x3ed76996u, 0x75036192u, 0x49ba450du, 0xbcc4dbaau, 0xfb1e6033u, 0x8112ddf4u, 0x102025cdu},   // L^-2
            {0x81ae1bf2u, 0x9fc3a48fu, 0xc6a70945u, 0x947d3bcfu, 0x2d981bdau, 0xef5069a2u, 0x99502b89u, 0x68146530u, 0x1db157c3u, 0x389bb30au, 0xc8e8de1cu, 0x0c1d6245u}};  // L^-3
        return t[which][i];
    }
    LW_HD static B::T konst(int which) {   // 0: L^2, 1: L^3, 2: L^-2, 3: L^-3
        B::T r;
#pragma unroll
        for (int i = 0; i < 12; i++) r.v[i] = k(which, i);
        return r;
    }
};
// IsoOf<C>: the curve the accumulation of a NORMALISED point set runs on (C itself unless a cheaper model exists)
template <class C> struct IsoOf { using type = C; static constexpr bool has = false; };
template <> struct IsoOf<Bn254G2> { using type = Bn254G2Iso; static constexpr bool has = true; };
template <> struct IsoOf<Bls12381G1> { using type = Bls12381G1Iso; static constexpr bool has = true; };
template <class C> struct IsIso { static constexpr bool value = false; };
template <> struct IsIso<Bn254G2Iso> { static constexpr bool value = true; };
template <> struct IsIso<Bls12381G1Iso> { static constexpr bool value = true; };

// ---------------------------------------------------------------- points
template <class C>
struct Point {
    typename C::B::T x, y, z;
};

template <class C>
LW_HD Point<C> pt_identity() {   // (0 : 1 : 0)  (short_weierstrass/point.rs:156-162)
    using B = typename C::B;
    return Point<C>{B::zero(), B::one(), B::zero()};
}
template <class C>
LW_HD bool pt_is_identity(const Point<C> &p) { return C::B::is_zero(p.z); }

// Reference memory layout: X, Y, Z consecutive (elliptic_curve/point.rs:8-10)
template <class C>
LW_HD Point<C> pt_load(const void *p) {
    using B = typename C::B;
    const char *c = (const char *)p;
    return Point<C>{B::load(c), B::load(c + B::BYTES), B::load(c + 2 * B::BYTES)};
}
template <class C>
LW_HD void pt_store(void *p, const Point<C> &a) {
    using B = typename C::B;
    char *c = (char *)p;
    B::store(c, a.x);
    B::store(c + B::BYTES, a.y);
    B::store(c + 2 * B::BYTES, a.z);
}

// Complete addition, RCB16 Algorithm 7 (a = 0): 12M + 2 m_3b + 19a, evaluated as 6M + 3 two-term dot products
template <class C>
LW_HD Point<C> pt_add(const Point<C> &p, const Point<C> &q) {
    using B = typename C::B;
    using T = typename B::T;
    T t0 = B::mul(p.x, q.x);
    T t1 = B::mul(p.y, q.y);
    T t2 = B::mul(p.z, q.z);
    T t3 = B::mul(B::add(p.x, p.y), B::add(q.x, q.y));
    T t4 = B::add(t0, t1);
    t3 = B::sub(t3, t4);
    t4 = B::mul(B::add(p.y, p.z), B::add(q.y, q.z));
    T x3 = B::add(t1, t2);
    t4 = B::sub(t4, x3);
    x3 = B::mul(B::add(p.x, p.z), B::add(q.x, q.z));
    T y3 = B::add(t0, t2);
    y3 = B::sub(x3, y3);
    x3 = B::add(t0, t0);
    t0 = B::add(x3, t0);
    t2 = C::mul_b3(t2);
    T z3 = B::add(t1, t2);
    t1 = B::sub(t1, t2);
    y3 = C::mul_b3(y3);
    // the last six products of the algorithm pair up into three sums, each reduced once:
    //   X3 = t3*t1 - t4*y3,  Y3 = t1*z3 + y3*t0,  Z3 = z3*t4 + t0*t3
    x3 = B::dot2s(t3, t1, t4, y3);
    T yo = B::dot2(t1, z3, y3, t0);
    T zo = B::dot2(z3, t4, t0, t3);
    return Point<C>{x3, yo, zo};
}

// Complete doubling, RCB16 Algorithm 9 (a = 0): 6M + 2S + 1 m_3b
template <class C>
LW_HD Point<C> pt_dbl(const Point<C> &p) {
    using B = typename C::B;
    using T = typename B::T;
    T t0 = B::sqr(p.y);
    T z3 = B::dbl(B::dbl(B::dbl(t0)));
    T t1 = B::mul(p.y, p.z);
    T t2 = B::sqr(p.z);
    t2 = C::mul_b3(t2);
    T x3 = B::mul(t2, z3);
    T y3 = B::add(t0, t2);
    z3 = B::mul(t1, z3);
    t1 = B::dbl(t2);
    t2 = B::add(t1, t2);
    t0 = B::sub(t0, t2);
    y3 = B::mul(t0, y3);
    y3 = B::add(x3, y3);
    t1 = B::mul(p.x, p.y);
    x3 = B::mul(t0, t1);
    x3 = B::dbl(x3);
    return Point<C>{x3, y3, z3};
}

// The same complete addition spread over the lanes s = 0, 1, 2 of a quad (lane 3 mirrors lane 0): lane s holds coordinate
// s (X, Y, Z) of both operands and returns coordinate s of the sum.  A chain of dependent additions run by ONE lane costs
// 6 products + 3 dots = 21 N^2 MACs per link; here a link is 2 products + 1 dot = 7 N^2 per lane:
//   lane s:  P_s = a_s b_s,  C_s = (a_s + a_s+1)(b_s + b_s+1) - P_s - P_s+1      (indices mod 3)
//   i.e.     t0, t1, t2 = P_0, P_1, P_2   and   t3, t4, y3 = C_0, C_1, C_2   of the formula above, broadcast by quad_perm,
//   lane 0:  X3 = t3 t1' - t4 (b3 y3),  lane 1:  Y3 = t1' z3 + (b3 y3)(3 t0),  lane 2:  Z3 = z3 t4 + (3 t0) t3
// with t1' = t1 - b3 t2, z3 = t1 + b3 t2 computed by every lane.  For the latency-bound levels of the bucket reduce
// (msm_core.cuh), where lanes are free and the chain is everything.  p + p is a valid input (complete formula), so the
// doublings of the reduce use it too.
template <class C>
LW_HD typename C::B::T pt_add_quad(const typename C::B::T &a, const typename C::B::T &b, uint32_t s) {
    using B = typename C::B;
    using T = typename B::T;
    constexpr int ROT1 = 0x49;   // quad_perm [1, 2, 0, 1]: lane s reads lane (s + 1) mod 3
    const T a1 = B::template dpp<ROT1>(a), b1 = B::template dpp<ROT1>(b);
    const T P = B::mul(a, b);
    T Cs = B::mul(B::add(a, a1), B::add(b, b1));
    Cs = B::sub(Cs, B::add(P, B::template dpp<ROT1>(P)));
    const T t0 = B::template dpp<0x00>(P), t1 = B::template dpp<0x55>(P), t2 = C::mul_b3(B::template dpp<0xAA>(P));
    const T t3 = B::template dpp<0x00>(Cs), t4 = B::template dpp<0x55>(Cs), y3 = C::mul_b3(B::template dpp<0xAA>(Cs));
    const T z3 = B::add(t1, t2), t1m = B::sub(t1, t2);
    const T t0x3 = B::add(B::add(t0, t0), t0);
    const bool l0 = s == 0, l1 = s == 1;
    const T oa = B::select(l0, t3, B::select(l1, t1m, z3));
    const T ob = B::select(l0, t1m, B::select(l1, z3, t4));
    const T oc = B::select(l0, B::neg(t4), B::select(l1, y3, t0x3));
    const T od = B::select(l0, y3, B::select(l1, t0x3, t3));
    return B::dot2(oa, ob, oc, od);
}

template <class C>
LW_HD Point<C> pt_neg(const Point<C> &p) { return Point<C>{p.x, C::B::neg(p.y), p.z}; }

// ---------------------------------------------------------------- affine points (pre-normalised SRS, lw_hip_srs_*)
// (x, y) with z = 1 implied; (0, 0) encodes the identity — it is on none of these curves (b != 0).
template <class C>
struct AffPoint {
    typename C::B::T x, y;
};
// Row stride of a device-resident affine point set: the 96-byte rows of BLS12-381 G1 are padded to 128 B so that every
// gather of the accumulation touches exactly one 128-byte line instead of 1.75 on average (the affine copy is
// library-owned, so its layout is free); the other groups' rows (64, 128, 192 B) are already line-friendly.
template <class C>
constexpr size_t aff_stride() { return 2 * C::B::BYTES == 96 ? 128 : 2 * C::B::BYTES; }
template <class C>
LW_HD bool aff_is_identity(const AffPoint<C> &p) { return C::B::is_zero(p.x) && C::B::is_zero(p.y); }
template <class C>
LW_HD AffPoint<C> aff_load(const void *p) {
    using B = typename C::B;
    return AffPoint<C>{B::load(p), B::load((const char *)p + B::BYTES)};
}
template <class C>
LW_HD void aff_store(void *p, const AffPoint<C> &a) {
    using B = typename C::B;
    B::store(p, a.x);
    B::store((char *)p + B::BYTES, a.y);
}
template <class C>
LW_HD Point<C> aff_to_point(const AffPoint<C> &a) {
    if (aff_is_identity<C>(a)) return pt_identity<C>();
    return Point<C>{a.x, a.y, C::B::one()};
}

// Complete mixed addition, RCB16 Algorithm 8 (a = 0, Z2 = 1): 11M + 2 m_3b + 13a, evaluated as 5M + 3 two-term dots
// (19 N^2 MACs against 21 N^2 for the projective formula).  q must not be the identity.
template <class C>
LW_HD Point<C> pt_add_mixed(const Point<C> &p, const AffPoint<C> &q) {
    using B = typename C::B;
    using T = typename B::T;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LW_NO_LAZY_SUMS)
    if constexpr (std::is_same<B, FpOps<Fp381>>::value || std::is_same<B, FpOps<Fp254>>::value) {
        // Same formula with some of its sums left unreduced where they only feed products: a product sum of up to K p^2
        // still reduces to below 2p while K p / R < 1 (fe_dot's KSUM) — Fp381: K <= 9 (2^384 / p = 9.8), six sums:
        // (x2 + y2), (x1 + y1), 2 t0, 3 t0, y2 z1 + y1, t1 + b3 z1 (144 of the addition's ~6150 instructions); Fp254:
        // K <= 5, five sums (t1 + b3 z1 stays reduced).  Every output is the canonical residue, as before.
        static_assert(B::HEADROOM5, "unreduced sums need headroom above 5p");
        constexpr bool WIDE = B::HEADROOM;
        using F = typename B::Field;
        T t0 = B::mul(p.x, q.x);
        T t1 = B::mul(p.y, q.y);
        T t3 = B::template mul_k<4>(B::add_nr(q.x, q.y), B::add_nr(p.x, p.y));        // (< 2p)(< 2p)
        t3 = B::sub(t3, B::add(t0, t1));                                                  // reduced
        const T t4 = B::add_nr(B::mul(q.y, p.z), p.y);                                    // < 2p
        T y3 = B::add(B::mul(q.x, p.z), p.x);
        const T t0x3 = B::add_nr(B::add_nr(t0, t0), t0);                                  // < 3p
        const T t2 = C::mul_b3(p.z);
        constexpr int KZ = WIDE ? 2 : 1;
        const T z3 = WIDE ? B::add_nr(t1, t2) : B::add(t1, t2);                           // < KZ p
        t1 = B::sub(t1, t2);
        constexpr bool B3_IS_2 = std::is_same<C, Bls12381G1Iso>::value;                   // b3 y3 = y3 + y3, left unreduced
        if constexpr (B3_IS_2) y3 = B::add_nr(y3, y3);                                    // < 2p
        else y3 = C::mul_b3(y3);
        constexpr int KY = B3_IS_2 ? 2 : 1;
        const T xo = B::template dot2_k<1 + 2 * KY>(t3, t1, fe_neg_raw_2p<F>(t4), y3);    // p*p + (2p - t4 <= 2p)*y3
        const T yo = B::template dot2_k<KZ + 3 * KY>(t1, z3, y3, t0x3);                   // p*z3 + y3*3p
        const T zo = B::template dot2_k<2 * KZ + 3>(z3, t4, t0x3, t3);                    // z3*2p + 3p*p
        return Point<C>{xo, yo, zo};
    }
#endif
    T t0 = B::mul(p.x, q.x);
    T t1 = B::mul(p.y, q.y);
    T t3 = B::mul(B::add(q.x, q.y), B::add(p.x, p.y));
    T t4 = B::add(t0, t1);
    t3 = B::sub(t3, t4);
    t4 = B::add(B::mul(q.y, p.z), p.y);
    T y3 = B::add(B::mul(q.x, p.z), p.x);
    T x3 = B::add(t0, t0);
    t0 = B::add(x3, t0);
    T t2 = C::mul_b3(p.z);
    T z3 = B::add(t1, t2);
    t1 = B::sub(t1, t2);
    y3 = C::mul_b3(y3);
    x3 = B::dot2s(t3, t1, t4, y3);
    T yo = B::dot2(t1, z3, y3, t0);
    T zo = B::dot2(z3, t4, t0, t3);
    return Point<C>{x3, yo, zo};
}

// projective -> affine pair (short_weierstrass/point.rs:91-129 `to_affine`), identity -> (0, 0)
template <class C>
LW_HD AffPoint<C> pt_to_aff(const Point<C> &p) {
    using B = typename C::B;
    if (B::is_zero(p.z)) return AffPoint<C>{B::zero(), B::zero()};
    typename B::T zi = B::inv(p.z);
    return AffPoint<C>{B::mul(p.x, zi), B::mul(p.y, zi)};
}

// (x/z : y/z : 1), identity -> (0:1:0)   (elliptic_curve/point.rs:41-54)
template <class C>
LW_HD Point<C> pt_to_affine(const Point<C> &p) {
    using B = typename C::B;
    if (B::is_zero(p.z)) return pt_identity<C>();
    typename B::T zi = B::inv(p.z);
    return Point<C>{B::mul(p.x, zi), B::mul(p.y, zi), B::one()};
}

// normalised result of an MSM that ran on an isomorphic model -> coordinates of the caller's curve
template <class C>
LW_HD Point<C> pt_unmap_result(const Point<C> &p) {
    if constexpr (IsIso<C>::value) {
        if (C::B::is_zero(p.z)) return p;
        return Point<C>{C::B::mul(p.x, C::konst(2)), C::B::mul(p.y, C::konst(3)), p.z};
    } else {
        return p;
    }
}

}  // namespace lw
