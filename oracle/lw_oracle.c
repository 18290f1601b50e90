/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * Plain-C CPU restatement of the lambdaworks (lambdaclass/lambda_elliptic_curves @ v0.11.0) NTT + MSM
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (lambda_elliptic_curves_amd/) never links, imports or calls it.
 *
 * Parity pinning: the reference is Rust and no Rust toolchain exists in the build container or on
 * the GPU box, so oracle/_ref cannot be built.  This restatement is pinned by
 *   (1) every fixed vector the reference's own tests hold for the path (tests/golden/reference_kats.json,
 *       each entry cites its reference file:line), and
 *   (2) independent Python big-integer definitions (oracle/bigint_def.py: NTT as sum c_j w^{ij} mod p,
 *       MSM as affine sum k_i P_i) checked in tests/test_oracle_*.py.
 *
 * Memory layout = the reference's: FieldElement = UnsignedInteger{limbs:[u64;N]}, limbs[0] MOST
 * significant, Montgomery form; projective point = X,Y,Z consecutive; Fp2 = [c0,c1].
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "orc_field.h"
#include "orc_keccak.h"
#include "lw_oracle.h"

/* ------------------------------------------------------------------ fields */
static orc_field F_STARK, F_FR381, F_BB64, F_FP381, F_FP254, F_FR254;
static orc_field32 F_BB32;
static u32 BB32_ROOT;     /* 21 in Montgomery form */
static int g_init_done = 0;

static void hex_to_limbs(const char *hex, u64 *out, int n) {
    size_t len = strlen(hex);
    for (int i = 0; i < n; i++) out[i] = 0;
    for (size_t k = 0; k < len; k++) {
        char c = hex[len - 1 - k];
        u64 v = (c >= '0' && c <= '9') ? (u64)(c - '0') : (c >= 'a' && c <= 'f') ? (u64)(c - 'a' + 10) : (u64)(c - 'A' + 10);
        size_t bit = 4 * k;
        out[n - 1 - bit / 64] |= v << (bit % 64);
    }
}

static void init_fft_field(orc_field *f, int n, const char *qhex, u64 two_adicity, const char *roothex) {
    u64 q[ORC_MAXL], r[ORC_MAXL];
    hex_to_limbs(qhex, q, n);
    orc_field_init(f, n, q);
    if (roothex) {
        hex_to_limbs(roothex, r, n);
        /* FieldElement::new(TWO_ADIC_PRIMITVE_ROOT_OF_UNITY) = from_base_type (traits.rs:82-84) */
        fp_from_base_type(f, n, f->root, r);
        f->two_adicity = two_adicity;
        f->is_fft = 1;
    }
}

static void orc_init(void) {
    if (g_init_done) return;
    /* stark_252_prime_field.rs:13-14,19-24 */
    init_fft_field(&F_STARK, 4, "800000000000011000000000000000000000000000000000000000000000001", 192,
                   "5282db87529cfa3f0464519c8b0fa5ad187148e11a61616070024f42f8ef94");
    /* bls12_381/default_types.rs:15-17,25-30 */
    init_fft_field(&F_FR381, 4, "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001", 32,
                   "2ab00961a08a499d84dd396c349d9b3cc5e433d6fa78eb2b54cc39d9bb30bbb7");
    /* babybear.rs:14-20,28-36 */
    init_fft_field(&F_BB64, 1, "78000001", 24, "15");
    /* bls12_381/field_extension.rs:13 */
    init_fft_field(&F_FP381, 6, "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab", 0, NULL);
    /* bn_254/field_extension.rs:15-16 */
    init_fft_field(&F_FP254, 4, "30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47", 0, NULL);
    /* bn_254/default_types.rs:14-16 */
    init_fft_field(&F_FR254, 4, "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", 0, NULL);
    /* babybear_u32.rs:6,16-24 */
    orc_field32_init(&F_BB32, 2013265921u);
    BB32_ROOT = m32_mul(&F_BB32, 21u, F_BB32.r2);
    g_init_done = 1;
}
__attribute__((constructor)) static void orc_ctor(void) { orc_init(); }

static const orc_field *field_by_id(int id) {
    switch (id) {
        case ORC_F_STARK252: return &F_STARK;
        case ORC_F_FR381: return &F_FR381;
        case ORC_F_BABYBEAR_U64: return &F_BB64;
        case ORC_F_BABYBEAR_EXT4: return &F_BB64;
        case ORC_F_FP381: return &F_FP381;
        case ORC_F_FP254: return &F_FP254;
        case ORC_F_FR254: return &F_FR254;
        default: return NULL;
    }
}

/* ------------------------------------------------------------------ NTT instances */
typedef struct { u64 l[4]; } fe4;
typedef struct { u64 l[1]; } fe1;
typedef struct { u64 l[6]; } fe6;
typedef struct { fe1 c[4]; } fe1x4;   /* quartic BabyBear extension element: [FieldElement<Babybear31>;4] (quartic_babybear.rs:16-19) */

/* 4-limb fields (Stark252, Fr381) */
#define NTT_NAME(x) ntt4_##x
#define EL_T fe4
#define TW_T fe4
#define CTX_T orc_field
#define TW_MUL(c, r, a, b) fp_mul(c, 4, (r)->l, (a)->l, (b)->l)
#define TW_ONE(c, r) ui_copy((r)->l, (c)->one, 4)
#define EL_MULTW(c, r, w, x) fp_mul(c, 4, (r)->l, (w)->l, (x)->l)
#define EL_ADD(c, r, a, b) fp_add(c, 4, (r)->l, (a)->l, (b)->l)
#define EL_SUB(c, r, a, b) fp_sub(c, 4, (r)->l, (a)->l, (b)->l)
#include "orc_ntt_tmpl.h"
#undef NTT_NAME
#undef EL_T
#undef TW_T
#undef CTX_T
#undef TW_MUL
#undef TW_ONE
#undef EL_MULTW
#undef EL_ADD
#undef EL_SUB

/* 1-limb field (BabyBear as MontgomeryBackendPrimeField<_,1>) */
#define NTT_NAME(x) ntt1_##x
#define EL_T fe1
#define TW_T fe1
#define CTX_T orc_field
#define TW_MUL(c, r, a, b) fp_mul(c, 1, (r)->l, (a)->l, (b)->l)
#define TW_ONE(c, r) ui_copy((r)->l, (c)->one, 1)
#define EL_MULTW(c, r, w, x) fp_mul(c, 1, (r)->l, (w)->l, (x)->l)
#define EL_ADD(c, r, a, b) fp_add(c, 1, (r)->l, (a)->l, (b)->l)
#define EL_SUB(c, r, a, b) fp_sub(c, 1, (r)->l, (a)->l, (b)->l)
#include "orc_ntt_tmpl.h"
#undef NTT_NAME
#undef EL_T
#undef TW_T
#undef CTX_T
#undef TW_MUL
#undef TW_ONE
#undef EL_MULTW
#undef EL_ADD
#undef EL_SUB

/* values in the quartic extension, domain in the base field (quartic_babybear.rs:155-166) */
ORC_INLINE void x4_multw(const orc_field *c, fe1x4 *r, const fe1 *w, const fe1x4 *x) {
    for (int k = 0; k < 4; k++) fp_mul(c, 1, r->c[k].l, w->l, x->c[k].l);
}
ORC_INLINE void x4_add(const orc_field *c, fe1x4 *r, const fe1x4 *a, const fe1x4 *b) {
    for (int k = 0; k < 4; k++) fp_add(c, 1, r->c[k].l, a->c[k].l, b->c[k].l);
}
ORC_INLINE void x4_sub(const orc_field *c, fe1x4 *r, const fe1x4 *a, const fe1x4 *b) {
    for (int k = 0; k < 4; k++) fp_sub(c, 1, r->c[k].l, a->c[k].l, b->c[k].l);
}
#define NTT_NAME(x) ntt1x4_##x
#define EL_T fe1x4
#define TW_T fe1
#define CTX_T orc_field
#define TW_MUL(c, r, a, b) fp_mul(c, 1, (r)->l, (a)->l, (b)->l)
#define TW_ONE(c, r) ui_copy((r)->l, (c)->one, 1)
#define EL_MULTW(c, r, w, x) x4_multw(c, r, w, x)
#define EL_ADD(c, r, a, b) x4_add(c, r, a, b)
#define EL_SUB(c, r, a, b) x4_sub(c, r, a, b)
#include "orc_ntt_tmpl.h"
#undef NTT_NAME
#undef EL_T
#undef TW_T
#undef CTX_T
#undef TW_MUL
#undef TW_ONE
#undef EL_MULTW
#undef EL_ADD
#undef EL_SUB

/* 32-bit BabyBear (U32MontgomeryBackendPrimeField) */
typedef struct { u32 v; } fe32;
#define NTT_NAME(x) ntt32_##x
#define EL_T fe32
#define TW_T fe32
#define CTX_T orc_field32
#define TW_MUL(c, r, a, b) ((r)->v = m32_mul(c, (a)->v, (b)->v))
#define TW_ONE(c, r) ((r)->v = (c)->one)
#define EL_MULTW(c, r, w, x) ((r)->v = m32_mul(c, (w)->v, (x)->v))
#define EL_ADD(c, r, a, b) ((r)->v = m32_add(c, (a)->v, (b)->v))
#define EL_SUB(c, r, a, b) ((r)->v = m32_sub(c, (a)->v, (b)->v))
#include "orc_ntt_tmpl.h"
#undef NTT_NAME
#undef EL_T
#undef TW_T
#undef CTX_T
#undef TW_MUL
#undef TW_ONE
#undef EL_MULTW
#undef EL_ADD
#undef EL_SUB

/* ------------------------------------------------------------------ generic field API */
size_t orc_field_elem_bytes(int field) {
    switch (field) {
        case ORC_F_STARK252: case ORC_F_FR381: case ORC_F_FP254: case ORC_F_FR254: return 32;
        case ORC_F_BABYBEAR_U64: return 8;
        case ORC_F_BABYBEAR_U32: return 4;
        case ORC_F_BABYBEAR_EXT4: return 32;
        case ORC_F_FP381: return 48;
        default: return 0;
    }
}
/* bytes of one twiddle (domain-field element) */
static size_t tw_bytes(int field) {
    return field == ORC_F_BABYBEAR_EXT4 ? 8 : orc_field_elem_bytes(field);
}

int orc_field_params(int field, u64 *q, u64 *mu, u64 *r2, u64 *one) {
    if (field == ORC_F_BABYBEAR_U32) {
        q[0] = F_BB32.q; *mu = F_BB32.mu; r2[0] = F_BB32.r2; one[0] = F_BB32.one;
        return 0;
    }
    const orc_field *f = field_by_id(field);
    if (!f) return ORC_ERR_BAD_ARG;
    ui_copy(q, f->q, f->n); *mu = f->mu; ui_copy(r2, f->r2, f->n); ui_copy(one, f->one, f->n);
    return 0;
}

/* Montgomery parameters for an arbitrary modulus (KATs: montgomery_backed_prime_fields.rs:446-480) */
int orc_derive_params(int n, const u64 *q, u64 *mu, u64 *r2, u64 *one, int *spare_bit) {
    if (n < 1 || n > ORC_MAXL) return ORC_ERR_BAD_ARG;
    orc_field f;
    orc_field_init(&f, n, q);
    *mu = f.mu; ui_copy(r2, f.r2, n); ui_copy(one, f.one, n); *spare_bit = f.spare_bit;
    return 0;
}

int orc_mont_cios(int n, const u64 *a, const u64 *b, const u64 *q, u64 mu, u64 *r) {
    if (n < 1 || n > ORC_MAXL) return ORC_ERR_BAD_ARG;
    mont_cios(r, a, b, q, mu, n);
    return 0;
}
int orc_mont_cios_spare(int n, const u64 *a, const u64 *b, const u64 *q, u64 mu, u64 *r) {
    if (n < 1 || n > ORC_MAXL) return ORC_ERR_BAD_ARG;
    mont_cios_spare(r, a, b, q, mu, n);
    return 0;
}

/* field ops with an arbitrary modulus (KATs :1012-1082 use ad-hoc 256-bit moduli) */
int orc_fe_op_mod(int n, const u64 *q, int op, const u64 *a, const u64 *b, u64 *r) {
    if (n < 1 || n > ORC_MAXL) return ORC_ERR_BAD_ARG;
    orc_field f;
    orc_field_init(&f, n, q);
    switch (op) {
        case ORC_OP_ADD: fp_add(&f, n, r, a, b); return 0;
        case ORC_OP_SUB: fp_sub(&f, n, r, a, b); return 0;
        case ORC_OP_MUL: fp_mul(&f, n, r, a, b); return 0;
        case ORC_OP_NEG: fp_neg(&f, n, r, a); return 0;
        case ORC_OP_INV: return fp_inv(&f, n, r, a) ? ORC_ERR_INV_ZERO : 0;
        case ORC_OP_TO_MONT: fp_from_base_type(&f, n, r, a); return 0;
        case ORC_OP_FROM_MONT: fp_representative(&f, n, r, a); return 0;
        default: return ORC_ERR_BAD_ARG;
    }
}

int orc_fe_op(int field, int op, const void *a, const void *b, void *r) {
    if (field == ORC_F_BABYBEAR_U32) {
        const orc_field32 *f = &F_BB32;
        u32 x = *(const u32 *)a, y = b ? *(const u32 *)b : 0, z;
        switch (op) {
            case ORC_OP_ADD: z = m32_add(f, x, y); break;
            case ORC_OP_SUB: z = m32_sub(f, x, y); break;
            case ORC_OP_MUL: z = m32_mul(f, x, y); break;
            case ORC_OP_NEG: z = m32_neg(f, x); break;
            case ORC_OP_INV: if (x == 0) return ORC_ERR_INV_ZERO; z = m32_pow(f, x, (u64)f->q - 2); break;
            case ORC_OP_TO_MONT: z = m32_mul(f, x, f->r2); break;
            case ORC_OP_FROM_MONT: z = m32_mul(f, x, 1); break;
            default: return ORC_ERR_BAD_ARG;
        }
        *(u32 *)r = z;
        return 0;
    }
    const orc_field *f = field_by_id(field);
    if (!f || field == ORC_F_BABYBEAR_EXT4) return ORC_ERR_BAD_ARG;
    return orc_fe_op_mod(f->n, f->q, op, (const u64 *)a, (const u64 *)b, (u64 *)r);
}

/* traits.rs:82-94 */
int orc_get_primitive_root_of_unity(int field, u64 order, void *out) {
    if (field == ORC_F_BABYBEAR_U32) {
        if (order == 0) { *(u32 *)out = F_BB32.one; return 0; }
        if (order > 24) return ORC_ERR_ROOT_OF_UNITY;
        u32 r = BB32_ROOT;
        for (u64 i = 0; i < 24 - order; i++) r = m32_mul(&F_BB32, r, r);
        *(u32 *)out = r;
        return 0;
    }
    const orc_field *f = field_by_id(field);
    if (!f || !f->is_fft) return ORC_ERR_BAD_ARG;
    int n = f->n;
    if (order == 0) { ui_copy((u64 *)out, f->one, n); return 0; }
    if (order > f->two_adicity) return ORC_ERR_ROOT_OF_UNITY;
    u64 r[ORC_MAXL];
    ui_copy(r, f->root, n);
    for (u64 i = 0; i < f->two_adicity - order; i++) fp_mul(f, n, r, r, r);
    ui_copy((u64 *)out, r, n);
    return 0;
}

/* roots_of_unity.rs:13-48.  config: 0 Natural, 1 NaturalInversed, 2 BitReverse, 3 BitReverseInversed */
int orc_get_powers_of_primitive_root(int field, u64 n, size_t count, int config, void *out) {
    if (count == 0) return 0;
    int inversed = (config == ORC_ROOTS_NATURAL_INV || config == ORC_ROOTS_BITREV_INV);
    int bitrev = (config == ORC_ROOTS_BITREV || config == ORC_ROOTS_BITREV_INV);
    if (field == ORC_F_BABYBEAR_U32) {
        fe32 root;
        int rc = orc_get_primitive_root_of_unity(field, n, &root.v);
        if (rc) return rc;
        if (inversed) root.v = m32_pow(&F_BB32, root.v, (u64)F_BB32.q - 2);
        ntt32_powers(&F_BB32, &root, count, bitrev, (fe32 *)out);
        return 0;
    }
    const orc_field *f = field_by_id(field);
    if (!f || !f->is_fft) return ORC_ERR_BAD_ARG;
    u64 root[ORC_MAXL];
    int rc = orc_get_primitive_root_of_unity(field, n, root);
    if (rc) return rc;
    if (inversed) fp_inv(f, f->n, root, root);
    if (f->n == 4) { fe4 r; ui_copy(r.l, root, 4); ntt4_powers(f, &r, count, bitrev, (fe4 *)out); }
    else { fe1 r; r.l[0] = root[0]; ntt1_powers(f, &r, count, bitrev, (fe1 *)out); }
    return 0;
}

/* roots_of_unity.rs:66-75 */
int orc_get_twiddles(int field, u64 order, int config, void *out) {
    if (order > 63) return ORC_ERR_ORDER;
    return orc_get_powers_of_primitive_root(field, order, ((size_t)1 << order) / 2, config, out);
}

int orc_bit_reverse_permute(int field, void *data, size_t n) {
    switch (orc_field_elem_bytes(field)) {
        case 32: ntt4_bitrev_el((fe4 *)data, n); return 0;
        case 8: ntt1_bitrev_el((fe1 *)data, n); return 0;
        case 4: ntt32_bitrev_el((fe32 *)data, n); return 0;
        default: return ORC_ERR_BAD_ARG;
    }
}

int orc_in_place_nr_2radix_fft(int field, void *data, size_t n, const void *twiddles) {
    switch (field) {
        case ORC_F_STARK252: case ORC_F_FR381:
            ntt4_nr_2radix(field_by_id(field), (fe4 *)data, n, (const fe4 *)twiddles); return 0;
        case ORC_F_BABYBEAR_U64: ntt1_nr_2radix(&F_BB64, (fe1 *)data, n, (const fe1 *)twiddles); return 0;
        case ORC_F_BABYBEAR_EXT4: ntt1x4_nr_2radix(&F_BB64, (fe1x4 *)data, n, (const fe1 *)twiddles); return 0;
        case ORC_F_BABYBEAR_U32: ntt32_nr_2radix(&F_BB32, (fe32 *)data, n, (const fe32 *)twiddles); return 0;
        default: return ORC_ERR_BAD_ARG;
    }
}

/* ops.rs:13-26 */
int orc_fft(int field, const void *in, size_t n, const void *twiddles, void *out) {
    if (n == 0 || (n & (n - 1))) return ORC_ERR_INPUT_NOT_POW2;
    switch (field) {
        case ORC_F_STARK252: case ORC_F_FR381:
            ntt4_fft(field_by_id(field), (const fe4 *)in, n, (const fe4 *)twiddles, (fe4 *)out); return 0;
        case ORC_F_BABYBEAR_U64: ntt1_fft(&F_BB64, (const fe1 *)in, n, (const fe1 *)twiddles, (fe1 *)out); return 0;
        case ORC_F_BABYBEAR_EXT4: ntt1x4_fft(&F_BB64, (const fe1x4 *)in, n, (const fe1 *)twiddles, (fe1x4 *)out); return 0;
        case ORC_F_BABYBEAR_U32: ntt32_fft(&F_BB32, (const fe32 *)in, n, (const fe32 *)twiddles, (fe32 *)out); return 0;
        default: return ORC_ERR_BAD_ARG;
    }
}

static int elem_is_zero(const void *p, size_t bytes) {
    const unsigned char *c = (const unsigned char *)p;
    for (size_t i = 0; i < bytes; i++) if (c[i]) return 0;
    return 1;
}

/* Polynomial::new — strip trailing zero coefficients (polynomial/mod.rs:19-31) */
static size_t poly_coeff_len(int field, const void *coeffs, size_t n) {
    size_t eb = orc_field_elem_bytes(field);
    while (n > 0 && elem_is_zero((const char *)coeffs + (n - 1) * eb, eb)) n--;
    return n;
}

static size_t next_pow2(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

/* fft/polynomial.rs:148-157 evaluate_fft_cpu */
static int evaluate_fft_cpu(int field, const void *coeffs, size_t len, void *out) {
    if (len == 0 || (len & (len - 1))) return ORC_ERR_INPUT_NOT_POW2;
    u64 order = (u64)__builtin_ctzll((u64)len);
    size_t tb = tw_bytes(field);
    void *tw = malloc((len / 2 + 1) * tb);
    if (!tw) return ORC_ERR_ALLOC;
    int tf = field == ORC_F_BABYBEAR_EXT4 ? ORC_F_BABYBEAR_U64 : field;
    int rc = orc_get_twiddles(tf, order, ORC_ROOTS_BITREV, tw);
    if (!rc) rc = orc_fft(field, coeffs, len, tw, out);
    free(tw);
    return rc;
}

/* multiply element (field E) by a domain-field element */
static void el_mul_tw(int field, void *r, const void *w, const void *x) {
    switch (field) {
        case ORC_F_STARK252: case ORC_F_FR381: fp_mul(field_by_id(field), 4, (u64 *)r, (const u64 *)w, (const u64 *)x); break;
        case ORC_F_BABYBEAR_U64: fp_mul(&F_BB64, 1, (u64 *)r, (const u64 *)w, (const u64 *)x); break;
        case ORC_F_BABYBEAR_EXT4: x4_multw(&F_BB64, (fe1x4 *)r, (const fe1 *)w, (const fe1x4 *)x); break;
        case ORC_F_BABYBEAR_U32: *(u32 *)r = m32_mul(&F_BB32, *(const u32 *)w, *(const u32 *)x); break;
    }
}
static void tw_one(int field, void *r) {
    if (field == ORC_F_BABYBEAR_U32) *(u32 *)r = F_BB32.one;
    else { const orc_field *f = field_by_id(field); ui_copy((u64 *)r, f->one, f->n); }
}
static void tw_mul(int field, void *r, const void *a, const void *b) {
    if (field == ORC_F_BABYBEAR_U32) *(u32 *)r = m32_mul(&F_BB32, *(const u32 *)a, *(const u32 *)b);
    else { const orc_field *f = field_by_id(field); fp_mul(f, f->n, (u64 *)r, (const u64 *)a, (const u64 *)b); }
}

/* Polynomial::scale (polynomial/mod.rs:259-271): c_i * factor^i by running power from one */
static void poly_scale(int field, void *coeffs, size_t n, const void *factor) {
    size_t eb = orc_field_elem_bytes(field), tb = tw_bytes(field);
    unsigned char power[48], nx[48], tmp[48];
    tw_one(field, power);
    for (size_t i = 0; i < n; i++) {
        el_mul_tw(field, tmp, power, (char *)coeffs + i * eb);
        memcpy((char *)coeffs + i * eb, tmp, eb);
        tw_mul(field, nx, power, factor);
        memcpy(power, nx, tb);
    }
}

/* Polynomial::evaluate_fft / evaluate_offset_fft (fft/polynomial.rs:25-68,74-82).
   offset == NULL → plain. out must hold *out_len elements; call with out == NULL to query the length. */
int orc_evaluate_fft(int field, const void *coeffs, size_t ncoeffs, size_t blowup, size_t domain_size,
                     const void *offset, void *out, size_t *out_len) {
    size_t eb = orc_field_elem_bytes(field);
    if (!eb || field == ORC_F_FP381 || field == ORC_F_FP254 || field == ORC_F_FR254) return ORC_ERR_BAD_ARG;
    size_t clen = poly_coeff_len(field, coeffs, ncoeffs);
    size_t m = clen > domain_size ? clen : domain_size;
    size_t len = next_pow2(m) * blowup;   /* usize::next_power_of_two(0) == 1 */
    *out_len = len;
    if (!out) return 0;
    if (clen == 0) { memset(out, 0, len * eb); return 0; }
    if (len == 0 || (len & (len - 1))) return ORC_ERR_INPUT_NOT_POW2;
    void *buf = calloc(len, eb);
    if (!buf) return ORC_ERR_ALLOC;
    memcpy(buf, coeffs, clen * eb);
    if (offset) poly_scale(field, buf, clen, offset);
    int rc = evaluate_fft_cpu(field, buf, len, out);
    free(buf);
    return rc;
}

/* Polynomial::interpolate_fft / interpolate_offset_fft (fft/polynomial.rs:87-127,159-174).
   Writes all n coefficients (the reference's Polynomial::new then strips trailing zeros; *coeff_len
   reports the stripped length). */
int orc_interpolate_fft(int field, const void *evals, size_t n, const void *offset, void *out, size_t *coeff_len) {
    size_t eb = orc_field_elem_bytes(field), tb = tw_bytes(field);
    if (!eb) return ORC_ERR_BAD_ARG;
    if (n == 0 || (n & (n - 1))) return ORC_ERR_INPUT_NOT_POW2;
    int tf = field == ORC_F_BABYBEAR_EXT4 ? ORC_F_BABYBEAR_U64 : field;
    u64 order = (u64)__builtin_ctzll((u64)n);
    void *tw = malloc((n / 2 + 1) * tb);
    if (!tw) return ORC_ERR_ALLOC;
    int rc = orc_get_twiddles(tf, order, ORC_ROOTS_BITREV_INV, tw);
    if (!rc) rc = orc_fft(field, evals, n, tw, out);
    free(tw);
    if (rc) return rc;
    /* scale_factor = FieldElement::from(n as u64).inv() */
    unsigned char sf[48];
    if (tf == ORC_F_BABYBEAR_U32) {
        u32 nn = m32_mul(&F_BB32, (u32)((u64)n % F_BB32.q), F_BB32.r2);
        *(u32 *)sf = m32_pow(&F_BB32, nn, (u64)F_BB32.q - 2);
    } else {
        const orc_field *f = field_by_id(tf);
        u64 t[ORC_MAXL];
        fp_from_u64(f, f->n, t, (u64)n);
        fp_inv(f, f->n, (u64 *)sf, t);
    }
    size_t clen = poly_coeff_len(field, out, n);
    unsigned char tmp[48];
    for (size_t i = 0; i < clen; i++) {
        el_mul_tw(field, tmp, sf, (char *)out + i * eb);
        memcpy((char *)out + i * eb, tmp, eb);
    }
    if (offset) {
        unsigned char oinv[48];
        if (tf == ORC_F_BABYBEAR_U32) *(u32 *)oinv = m32_pow(&F_BB32, *(const u32 *)offset, (u64)F_BB32.q - 2);
        else { const orc_field *f = field_by_id(tf); if (fp_inv(f, f->n, (u64 *)oinv, (const u64 *)offset)) return ORC_ERR_INV_ZERO; }
        poly_scale(field, out, clen, oinv);
    }
    if (coeff_len) *coeff_len = clen;
    return 0;
}

/* ------------------------------------------------------------------ curves */
/* --- Fp2 over Fp381: Karatsuba (bls12_381/field_extension.rs:42-47) ; over Fp254: schoolbook (bn_254/field_extension.rs:47-49) */
typedef struct { fe6 c[2]; } fe6x2;
typedef struct { fe4 c[2]; } fe4x2;

ORC_INLINE void fp2_381_mul(fe6x2 *r, const fe6x2 *a, const fe6x2 *b) {
    const orc_field *f = &F_FP381;
    fe6 a0b0, a1b1, s, t, z;
    fp_mul(f, 6, a0b0.l, a->c[0].l, b->c[0].l);
    fp_mul(f, 6, a1b1.l, a->c[1].l, b->c[1].l);
    fp_add(f, 6, s.l, a->c[0].l, a->c[1].l);
    fp_add(f, 6, t.l, b->c[0].l, b->c[1].l);
    fp_mul(f, 6, z.l, s.l, t.l);
    fp_sub(f, 6, r->c[0].l, a0b0.l, a1b1.l);
    fp_sub(f, 6, z.l, z.l, a0b0.l);
    fp_sub(f, 6, r->c[1].l, z.l, a1b1.l);
}
ORC_INLINE void fp2_254_mul(fe4x2 *r, const fe4x2 *a, const fe4x2 *b) {
    const orc_field *f = &F_FP254;
    fe4 t0, t1, t2, t3;
    fp_mul(f, 4, t0.l, a->c[0].l, b->c[0].l);
    fp_mul(f, 4, t1.l, a->c[1].l, b->c[1].l);
    fp_mul(f, 4, t2.l, a->c[0].l, b->c[1].l);
    fp_mul(f, 4, t3.l, a->c[1].l, b->c[0].l);
    fp_sub(f, 4, r->c[0].l, t0.l, t1.l);
    fp_add(f, 4, r->c[1].l, t2.l, t3.l);
}
/* Fp2 inverse: (a0 - a1 u) / (a0^2 + a1^2)  (u^2 = -1 in both towers) */
#define DEF_FP2_INV(NAME, T, E, F, N)                                        \
    static int NAME(T *r, const T *a) {                                      \
        E n0, n1, nrm, inv;                                                  \
        fp_mul(F, N, n0.l, a->c[0].l, a->c[0].l);                            \
        fp_mul(F, N, n1.l, a->c[1].l, a->c[1].l);                            \
        fp_add(F, N, nrm.l, n0.l, n1.l);                                     \
        if (fp_inv(F, N, inv.l, nrm.l)) return -1;                           \
        fp_mul(F, N, r->c[0].l, a->c[0].l, inv.l);                           \
        fp_neg(F, N, n0.l, a->c[1].l);                                       \
        fp_mul(F, N, r->c[1].l, n0.l, inv.l);                                \
        return 0;                                                            \
    }
DEF_FP2_INV(fp2_381_inv, fe6x2, fe6, &F_FP381, 6)
DEF_FP2_INV(fp2_254_inv, fe4x2, fe4, &F_FP254, 4)

/* G1 BLS12-381 */
#define EC_NAME(x) g1_381_##x
#define FE_T fe6
#define FE_MUL(r, a, b) fp_mul(&F_FP381, 6, (r)->l, (a)->l, (b)->l)
#define FE_ADD(r, a, b) fp_add(&F_FP381, 6, (r)->l, (a)->l, (b)->l)
#define FE_SUB(r, a, b) fp_sub(&F_FP381, 6, (r)->l, (a)->l, (b)->l)
#define FE_NEG(r, a) fp_neg(&F_FP381, 6, (r)->l, (a)->l)
#define FE_IS_ZERO(a) ui_is_zero((a)->l, 6)
#define FE_EQ(a, b) ui_eq((a)->l, (b)->l, 6)
#define FE_SET_ZERO(r) ui_set_u64((r)->l, 0, 6)
#define FE_SET_ONE(r) ui_copy((r)->l, F_FP381.one, 6)
#define FE_INV(r, a) fp_inv(&F_FP381, 6, (r)->l, (a)->l)
#include "orc_ec_tmpl.h"
#undef EC_NAME
#undef FE_T
#undef FE_MUL
#undef FE_ADD
#undef FE_SUB
#undef FE_NEG
#undef FE_IS_ZERO
#undef FE_EQ
#undef FE_SET_ZERO
#undef FE_SET_ONE
#undef FE_INV

/* G1 BN254 */
#define EC_NAME(x) g1_254_##x
#define FE_T fe4
#define FE_MUL(r, a, b) fp_mul(&F_FP254, 4, (r)->l, (a)->l, (b)->l)
#define FE_ADD(r, a, b) fp_add(&F_FP254, 4, (r)->l, (a)->l, (b)->l)
#define FE_SUB(r, a, b) fp_sub(&F_FP254, 4, (r)->l, (a)->l, (b)->l)
#define FE_NEG(r, a) fp_neg(&F_FP254, 4, (r)->l, (a)->l)
#define FE_IS_ZERO(a) ui_is_zero((a)->l, 4)
#define FE_EQ(a, b) ui_eq((a)->l, (b)->l, 4)
#define FE_SET_ZERO(r) ui_set_u64((r)->l, 0, 4)
#define FE_SET_ONE(r) ui_copy((r)->l, F_FP254.one, 4)
#define FE_INV(r, a) fp_inv(&F_FP254, 4, (r)->l, (a)->l)
#include "orc_ec_tmpl.h"
#undef EC_NAME
#undef FE_T
#undef FE_MUL
#undef FE_ADD
#undef FE_SUB
#undef FE_NEG
#undef FE_IS_ZERO
#undef FE_EQ
#undef FE_SET_ZERO
#undef FE_SET_ONE
#undef FE_INV

/* G2 BN254 (twist, Fp2 coordinates) */
#define EC_NAME(x) g2_254_##x
#define FE_T fe4x2
#define FE_MUL(r, a, b) do { fe4x2 _t; fp2_254_mul(&_t, a, b); *(r) = _t; } while (0)
#define FE_ADD(r, a, b) do { fp_add(&F_FP254, 4, (r)->c[0].l, (a)->c[0].l, (b)->c[0].l); fp_add(&F_FP254, 4, (r)->c[1].l, (a)->c[1].l, (b)->c[1].l); } while (0)
#define FE_SUB(r, a, b) do { fp_sub(&F_FP254, 4, (r)->c[0].l, (a)->c[0].l, (b)->c[0].l); fp_sub(&F_FP254, 4, (r)->c[1].l, (a)->c[1].l, (b)->c[1].l); } while (0)
#define FE_NEG(r, a) do { fp_neg(&F_FP254, 4, (r)->c[0].l, (a)->c[0].l); fp_neg(&F_FP254, 4, (r)->c[1].l, (a)->c[1].l); } while (0)
#define FE_IS_ZERO(a) (ui_is_zero((a)->c[0].l, 4) && ui_is_zero((a)->c[1].l, 4))
#define FE_EQ(a, b) (ui_eq((a)->c[0].l, (b)->c[0].l, 4) && ui_eq((a)->c[1].l, (b)->c[1].l, 4))
#define FE_SET_ZERO(r) do { ui_set_u64((r)->c[0].l, 0, 4); ui_set_u64((r)->c[1].l, 0, 4); } while (0)
#define FE_SET_ONE(r) do { ui_copy((r)->c[0].l, F_FP254.one, 4); ui_set_u64((r)->c[1].l, 0, 4); } while (0)
#define FE_INV(r, a) fp2_254_inv(r, a)
#include "orc_ec_tmpl.h"
#undef EC_NAME
#undef FE_T
#undef FE_MUL
#undef FE_ADD
#undef FE_SUB
#undef FE_NEG
#undef FE_IS_ZERO
#undef FE_EQ
#undef FE_SET_ZERO
#undef FE_SET_ONE
#undef FE_INV

/* G2 BLS12-381 (twist, Fp2 coordinates) */
#define EC_NAME(x) g2_381_##x
#define FE_T fe6x2
#define FE_MUL(r, a, b) do { fe6x2 _t; fp2_381_mul(&_t, a, b); *(r) = _t; } while (0)
#define FE_ADD(r, a, b) do { fp_add(&F_FP381, 6, (r)->c[0].l, (a)->c[0].l, (b)->c[0].l); fp_add(&F_FP381, 6, (r)->c[1].l, (a)->c[1].l, (b)->c[1].l); } while (0)
#define FE_SUB(r, a, b) do { fp_sub(&F_FP381, 6, (r)->c[0].l, (a)->c[0].l, (b)->c[0].l); fp_sub(&F_FP381, 6, (r)->c[1].l, (a)->c[1].l, (b)->c[1].l); } while (0)
#define FE_NEG(r, a) do { fp_neg(&F_FP381, 6, (r)->c[0].l, (a)->c[0].l); fp_neg(&F_FP381, 6, (r)->c[1].l, (a)->c[1].l); } while (0)
#define FE_IS_ZERO(a) (ui_is_zero((a)->c[0].l, 6) && ui_is_zero((a)->c[1].l, 6))
#define FE_EQ(a, b) (ui_eq((a)->c[0].l, (b)->c[0].l, 6) && ui_eq((a)->c[1].l, (b)->c[1].l, 6))
#define FE_SET_ZERO(r) do { ui_set_u64((r)->c[0].l, 0, 6); ui_set_u64((r)->c[1].l, 0, 6); } while (0)
#define FE_SET_ONE(r) do { ui_copy((r)->c[0].l, F_FP381.one, 6); ui_set_u64((r)->c[1].l, 0, 6); } while (0)
#define FE_INV(r, a) fp2_381_inv(r, a)
#include "orc_ec_tmpl.h"
#undef EC_NAME
#undef FE_T
#undef FE_MUL
#undef FE_ADD
#undef FE_SUB
#undef FE_NEG
#undef FE_IS_ZERO
#undef FE_EQ
#undef FE_SET_ZERO
#undef FE_SET_ONE
#undef FE_INV

size_t orc_curve_point_bytes(int curve) {
    switch (curve) {
        case ORC_C_BLS12_381_G1: return sizeof(g1_381_pt);
        case ORC_C_BN254_G1: return sizeof(g1_254_pt);
        case ORC_C_BN254_G2: return sizeof(g2_254_pt);
        case ORC_C_BLS12_381_G2: return sizeof(g2_381_pt);
        default: return 0;
    }
}

#define CURVE_DISPATCH(curve, CALL)                               \
    switch (curve) {                                              \
        case ORC_C_BLS12_381_G1: { typedef g1_381_pt PT; CALL(g1_381); } break; \
        case ORC_C_BN254_G1: { typedef g1_254_pt PT; CALL(g1_254); } break;     \
        case ORC_C_BN254_G2: { typedef g2_254_pt PT; CALL(g2_254); } break;     \
        case ORC_C_BLS12_381_G2: { typedef g2_381_pt PT; CALL(g2_381); } break; \
        default: return ORC_ERR_BAD_ARG;                          \
    }

int orc_ec_op(int curve, int op, const void *p, const void *q, void *r) {
#define CALL_OP(P)                                                                         \
    switch (op) {                                                                          \
        case ORC_EC_ADD: { PT t; P##_add(&t, (const PT *)p, (const PT *)q); *(PT *)r = t; return 0; } \
        case ORC_EC_DOUBLE: { PT t; P##_double(&t, (const PT *)p); *(PT *)r = t; return 0; }  \
        case ORC_EC_NEG: { PT t; P##_neg(&t, (const PT *)p); *(PT *)r = t; return 0; }        \
        case ORC_EC_TO_AFFINE: { PT t; P##_to_affine(&t, (const PT *)p); *(PT *)r = t; return 0; } \
        case ORC_EC_EQ: return P##_eq((const PT *)p, (const PT *)q) ? 1 : 0;                 \
        case ORC_EC_NEUTRAL: P##_neutral((PT *)r); return 0;                                 \
        default: return ORC_ERR_BAD_ARG;                                                   \
    }
    CURVE_DISPATCH(curve, CALL_OP)
#undef CALL_OP
    return ORC_ERR_BAD_ARG;
}

int orc_ec_mul(int curve, const void *p, const u64 *k, int k_limbs, void *r) {
    if (k_limbs < 1 || k_limbs > 8) return ORC_ERR_BAD_ARG;
#define CALL_MUL(P) { PT t; P##_mul(&t, (const PT *)p, k, k_limbs); *(PT *)r = t; return 0; }
    CURVE_DISPATCH(curve, CALL_MUL)
#undef CALL_MUL
    return ORC_ERR_BAD_ARG;
}

/* pippenger.rs:18-32.  n_scalars != n_points → LengthMismatch */
int orc_msm(int curve, const u64 *cs, size_t n_scalars, int k_limbs, const void *points, size_t n_points, void *out) {
    if (n_scalars != n_points) return ORC_ERR_LENGTH_MISMATCH;
    if (k_limbs < 1 || k_limbs > 8) return ORC_ERR_BAD_ARG;
#define CALL_MSM(P) return P##_msm(cs, k_limbs, (const PT *)points, n_points, (PT *)out) ? ORC_ERR_ALLOC : 0;
    CURVE_DISPATCH(curve, CALL_MSM)
#undef CALL_MSM
    return ORC_ERR_BAD_ARG;
}
int orc_msm_with(int curve, const u64 *cs, int k_limbs, const void *points, size_t n, size_t window, void *out) {
    if (k_limbs < 1 || k_limbs > 8) return ORC_ERR_BAD_ARG;
#define CALL_MSMW(P) return P##_msm_with(cs, k_limbs, (const PT *)points, n, window, (PT *)out) ? ORC_ERR_ALLOC : 0;
    CURVE_DISPATCH(curve, CALL_MSMW)
#undef CALL_MSMW
    return ORC_ERR_BAD_ARG;
}
int orc_parallel_msm_with(int curve, const u64 *cs, int k_limbs, const void *points, size_t n, size_t window, int threads, void *out) {
    if (k_limbs < 1 || k_limbs > 8 || window < 1 || window > 31) return ORC_ERR_BAD_ARG;
#define CALL_PMSM(P) return P##_parallel_msm_with(cs, k_limbs, (const PT *)points, n, window, threads, (PT *)out) ? ORC_ERR_ALLOC : 0;
    CURVE_DISPATCH(curve, CALL_PMSM)
#undef CALL_PMSM
    return ORC_ERR_BAD_ARG;
}
int orc_msm_naive(int curve, const u64 *cs, int k_limbs, const void *points, size_t n, void *out) {
    if (k_limbs < 1 || k_limbs > 8) return ORC_ERR_BAD_ARG;
#define CALL_NAIVE(P) { P##_msm_naive(cs, k_limbs, (const PT *)points, n, (PT *)out); return 0; }
    CURVE_DISPATCH(curve, CALL_NAIVE)
#undef CALL_NAIVE
    return ORC_ERR_BAD_ARG;
}
size_t orc_optimum_window_size(size_t n) { return g1_381_optimum_window_size(n); }

/* Synthetic SRS-like point set for benches: P_0 = [s0]G, P_i = P_{i-1} + [delta]G,
   built by repeated projective addition, so every triple arrives non-normalised (Z != 1) as real SRS
   points do (provers/groth16/src/setup.rs:127-135).  G = generator given by the caller (projective). */
int orc_gen_points(int curve, const void *gen, const u64 *s0, const u64 *delta, int k_limbs, size_t n,
                   void *out) {
#define CALL_GEN(P)                                                            \
    {                                                                          \
        PT cur, step;                                                          \
        P##_mul(&cur, (const PT *)gen, s0, k_limbs);                           \
        P##_mul(&step, (const PT *)gen, delta, k_limbs);                       \
        PT *o = (PT *)out;                                                     \
        for (size_t i = 0; i < n; i++) {                                       \
            o[i] = cur;                                                        \
            PT nx; P##_add(&nx, &cur, &step); cur = nx;                        \
        }                                                                      \
        return 0;                                                              \
    }
    CURVE_DISPATCH(curve, CALL_GEN)
#undef CALL_GEN
    return ORC_ERR_BAD_ARG;
}

/* ------------------------------------------------------------------ Merkle commitment */
int orc_keccak256_bytes(const uint8_t *data, size_t len, uint8_t *out32) {
    orc_keccak256(data, len, out32);
    return 0;
}

/* FieldElement::as_bytes for MontgomeryBackendPrimeField = value().to_bytes_be()
   (math/src/field/fields/montgomery_backed_prime_fields.rs:367-373): raw Montgomery limbs, big-endian */
static void elem_as_bytes_be(const u64 *limbs, int n, uint8_t *out) {
    for (int i = 0; i < n; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(limbs[i] >> (56 - 8 * b));
}

/* interpolate_and_commit_main's commitment (provers/stark/src/prover.rs:229-244): every LDE column is bit-reverse
   permuted (:232-234), columns2rows, then BatchedMerkleTree::<BatchKeccak256Backend>::build(rows)
   (crypto/src/merkle_tree/merkle.rs:31-56; leaf hash field_element_vector.rs:41-49; inner nodes utils.rs:44-72).
   columns: n_cols arrays of 2^log2n 4-limb elements (column-major).  nodes_out: (2*2^log2n - 1) * 32 bytes, root first. */
int orc_merkle_commit_columns(const u64 *columns, uint32_t n_cols, uint32_t log2n, int bit_reverse, uint8_t *nodes_out) {
    const size_t n = (size_t)1 << log2n;
    uint8_t *row = (uint8_t *)malloc((size_t)n_cols * 32 + 1);
    if (!row) return ORC_ERR_ALLOC;
    uint8_t *leaves = nodes_out + (n - 1) * 32;
    for (size_t i = 0; i < n; i++) {
        size_t src = bit_reverse ? ntt4_reverse_index(i, n) : i;
        for (uint32_t c = 0; c < n_cols; c++) elem_as_bytes_be(columns + ((size_t)c * n + src) * 4, 4, row + (size_t)c * 32);
        orc_keccak256(row, (size_t)n_cols * 32, leaves + i * 32);
    }
    free(row);
    /* utils.rs:44-72: parents of level [begin, end] are written to [begin/2, begin) */
    size_t level_begin = n - 1, level_end = 2 * level_begin;
    while (level_begin != level_end) {
        size_t new_begin = level_begin / 2;
        size_t new_len = level_begin - new_begin;
        for (size_t k = 0; k < new_len; k++) {
            uint8_t buf[64];
            memcpy(buf, nodes_out + (level_begin + 2 * k) * 32, 32);
            memcpy(buf + 32, nodes_out + (level_begin + 2 * k + 1) * 32, 32);
            orc_keccak256(buf, 64, nodes_out + (new_begin + k) * 32);
        }
        level_end = level_begin - 1;
        level_begin = new_begin;
    }
    return 0;
}

/* Leaves and every tree level are independent hashes; the checker at the sizes the GPU path is timed on (2^22-2^24
   leaves) splits exactly the loops of orc_merkle_commit_columns over `threads` workers.  Same nodes, same order. */
typedef struct { const u64 *columns; uint32_t n_cols; size_t n; int bit_reverse; uint8_t *nodes; size_t level_begin, new_begin, lo, hi; int leaves; uint32_t eb; } mk_job;
/* as_bytes of one element: eb = 32: four u64 limbs big-endian (montgomery_backed_prime_fields.rs:367-373); 8: the one limb of
   the u64-limb BabyBear, same impl; 4: U32MontgomeryBackendPrimeField's value().to_be_bytes()
   (u32_montgomery_backend_prime_field.rs:258-262) */
static void elem_as_bytes_eb(const void *columns, uint32_t eb, size_t index, uint8_t *out) {
    if (eb == 32) elem_as_bytes_be((const u64 *)columns + index * 4, 4, out);
    else if (eb == 8) elem_as_bytes_be((const u64 *)columns + index, 1, out);
    else { u32 v = ((const u32 *)columns)[index]; out[0] = (uint8_t)(v >> 24); out[1] = (uint8_t)(v >> 16); out[2] = (uint8_t)(v >> 8); out[3] = (uint8_t)v; }
}
static void *mk_worker(void *arg) {
    mk_job *j = (mk_job *)arg;
    if (j->leaves) {
        const uint32_t eb = j->eb ? j->eb : 32;
        uint8_t *row = (uint8_t *)malloc((size_t)j->n_cols * eb + 1);
        uint8_t *leaves = j->nodes + (j->n - 1) * 32;
        for (size_t i = j->lo; i < j->hi; i++) {
            size_t src = j->bit_reverse ? ntt4_reverse_index(i, j->n) : i;
            for (uint32_t c = 0; c < j->n_cols; c++) elem_as_bytes_eb(j->columns, eb, (size_t)c * j->n + src, row + (size_t)c * eb);
            orc_keccak256(row, (size_t)j->n_cols * eb, leaves + i * 32);
        }
        free(row);
    } else {
        for (size_t k = j->lo; k < j->hi; k++) {
            uint8_t buf[64];
            memcpy(buf, j->nodes + (j->level_begin + 2 * k) * 32, 32);
            memcpy(buf + 32, j->nodes + (j->level_begin + 2 * k + 1) * 32, 32);
            orc_keccak256(buf, 64, j->nodes + (j->new_begin + k) * 32);
        }
    }
    return NULL;
}
static void mk_run(mk_job *proto, size_t count, int threads) {
    pthread_t th[64];
    mk_job jobs[64];
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (count < 4096) threads = 1;
    size_t per = (count + threads - 1) / threads;
    int started = 0;
    for (int t = 0; t < threads; t++) {
        size_t lo = (size_t)t * per, hi = lo + per < count ? lo + per : count;
        if (lo >= hi) break;
        jobs[t] = *proto;
        jobs[t].lo = lo;
        jobs[t].hi = hi;
        pthread_create(&th[t], NULL, mk_worker, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}
/* the same tree over columns of elem_bytes-byte elements (32, 8 or 4), see elem_as_bytes_eb */
int orc_merkle_commit_columns_bytes(const void *columns, uint32_t elem_bytes, uint32_t n_cols, uint32_t log2n, int bit_reverse, uint8_t *nodes_out,
                                    int threads);
int orc_merkle_commit_columns_mt(const u64 *columns, uint32_t n_cols, uint32_t log2n, int bit_reverse, uint8_t *nodes_out, int threads) {
    return orc_merkle_commit_columns_bytes(columns, 32, n_cols, log2n, bit_reverse, nodes_out, threads);
}
int orc_merkle_commit_columns_bytes(const void *columns, uint32_t elem_bytes, uint32_t n_cols, uint32_t log2n, int bit_reverse, uint8_t *nodes_out,
                                    int threads) {
    if (elem_bytes != 32 && elem_bytes != 8 && elem_bytes != 4) return ORC_ERR_BAD_ARG;
    const size_t n = (size_t)1 << log2n;
    mk_job j = { (const u64 *)columns, n_cols, n, bit_reverse, nodes_out, 0, 0, 0, 0, 1, elem_bytes };
    mk_run(&j, n, threads);
    size_t level_begin = n - 1, level_end = 2 * level_begin;
    while (level_begin != level_end) {
        size_t new_begin = level_begin / 2;
        j.leaves = 0;
        j.level_begin = level_begin;
        j.new_begin = new_begin;
        mk_run(&j, level_begin - new_begin, threads);
        level_end = level_begin - 1;
        level_begin = new_begin;
    }
    return 0;
}

/* ------------------------------------------------------------------ FRI fold
   commit_phase's `FieldElement::<F>::from(2) * fold_polynomial(&current_poly, &zeta)` (provers/stark/src/fri/mod.rs:49):
   fold_polynomial (fri/fri_functions.rs:7-30) = even coefficients + zeta * odd coefficients, each side built with
   Polynomial::new (trailing zeros stripped) and padded to the longer one.  out: ceil(n/2) elements (all written);
   *out_len = length after Polynomial::new strips the product's trailing zeros.  4-limb fields. */
int orc_fri_fold_twice(int field, const u64 *coeffs, size_t n, const u64 *zeta, u64 *out, size_t *out_len) {
    const orc_field *f = field_by_id(field);
    if (!f || f->n != 4) return ORC_ERR_BAD_ARG;
    size_t clen = poly_coeff_len(field, coeffs, n);   /* current_poly is a Polynomial: already stripped */
    size_t n_out = (n + 1) / 2;
    u64 two[4];
    fp_from_u64(f, 4, two, 2);
    for (size_t i = 0; i < n_out; i++) {
        u64 even[4] = {0, 0, 0, 0}, odd[4] = {0, 0, 0, 0}, sum[4];
        if (2 * i < clen) ui_copy(even, coeffs + 8 * i, 4);
        if (2 * i + 1 < clen) fp_mul(f, 4, odd, coeffs + 8 * i + 4, zeta);
        fp_add(f, 4, sum, even, odd);
        fp_mul(f, 4, out + 4 * i, two, sum);
    }
    if (out_len) *out_len = poly_coeff_len(field, out, n_out);
    return 0;
}

/* ------------------------------------------------------------------ Groth16 quotient
   QuadraticArithmeticProgram::calculate_h_coefficients (provers/groth16/src/qap.rs:15-39) after
   scale_and_accumulate_variable_polynomials has summed the variable polynomials into L, R, O (ncoeffs coefficients each):
   l, r, o = evaluate_offset_fft(., 1, Some(2g), 7); t = evaluate_offset_fft(x^g - 1, 1, Some(2g), 7) then
   inplace_batch_inverse (field/element.rs:47-65); h_evaluated = (l*r - o)*t; interpolate_offset_fft(h_evaluated, 7).
   out: 2g elements; *coeff_len: length after Polynomial::new. */
int orc_groth16_h_coefficients(const u64 *l, const u64 *r, const u64 *o, size_t ncoeffs, size_t gates, u64 *out, size_t *coeff_len) {
    const int F = ORC_F_FR381;
    const orc_field *f = field_by_id(F);
    const size_t deg = 2 * gates;
    if (gates == 0 || (gates & (gates - 1))) return ORC_ERR_INPUT_NOT_POW2;
    u64 offset[4];
    fp_from_u64(f, 4, offset, 7);   /* ORDER_R_MINUS_1_ROOT_UNITY (provers/groth16/src/common.rs:26) */
    u64 *ev = (u64 *)malloc(4 * deg * 32), *tp = (u64 *)calloc(gates + 1, 32), *prefix = (u64 *)malloc(deg * 32);
    if (!ev || !tp || !prefix) { free(ev); free(tp); free(prefix); return ORC_ERR_ALLOC; }
    u64 *le = ev, *re = ev + 4 * deg, *oe = ev + 8 * deg, *t = ev + 12 * deg;
    size_t len = 0;
    int rc = orc_evaluate_fft(F, l, ncoeffs, 1, deg, offset, le, &len);
    if (!rc) rc = orc_evaluate_fft(F, r, ncoeffs, 1, deg, offset, re, &len);
    if (!rc) rc = orc_evaluate_fft(F, o, ncoeffs, 1, deg, offset, oe, &len);
    /* t_poly = new_monomial(1, g) - 1 */
    fp_neg(f, 4, tp, f->one);
    ui_copy(tp + 4 * gates, f->one, 4);
    if (gates == 0) rc = ORC_ERR_BAD_ARG;
    if (!rc) rc = orc_evaluate_fft(F, tp, gates + 1, 1, deg, offset, t, &len);
    if (!rc) {   /* inplace_batch_inverse */
        ui_copy(prefix, t, 4);
        for (size_t i = 1; i < deg; i++) fp_mul(f, 4, prefix + 4 * i, prefix + 4 * (i - 1), t + 4 * i);
        u64 bi[4], ai[4], nx[4];
        if (fp_inv(f, 4, bi, prefix + 4 * (deg - 1))) rc = ORC_ERR_INV_ZERO;
        for (size_t i = deg - 1; !rc && i >= 1; i--) {
            fp_mul(f, 4, ai, bi, prefix + 4 * (i - 1));
            fp_mul(f, 4, nx, bi, t + 4 * i);
            ui_copy(bi, nx, 4);
            ui_copy(t + 4 * i, ai, 4);
        }
        if (!rc) ui_copy(t, bi, 4);
    }
    if (!rc) {
        for (size_t i = 0; i < deg; i++) {
            u64 p[4], d[4];
            fp_mul(f, 4, p, le + 4 * i, re + 4 * i);
            fp_sub(f, 4, d, p, oe + 4 * i);
            fp_mul(f, 4, le + 4 * i, d, t + 4 * i);
        }
        rc = orc_interpolate_fft(F, le, deg, offset, out, coeff_len);
    }
    free(ev); free(tp); free(prefix);
    return rc;
}
