/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C CPU restatement of lambdaworks' limb + Montgomery field arithmetic.
 * Layout follows the reference: UnsignedInteger<N>{limbs:[u64;N]}, limbs[0] MOST significant
 * (math/src/unsigned_integer/element.rs:29-37).
 *
 * Follows:
 *   math/src/unsigned_integer/montgomery.rs:12-76    (cios)
 *   math/src/unsigned_integer/montgomery.rs:86-141   (cios_optimized_for_moduli_with_one_spare_bit)
 *   math/src/field/fields/montgomery_backed_prime_fields.rs:42-111 (R2, MU, ONE, spare-bit predicate)
 *   math/src/field/fields/montgomery_backed_prime_fields.rs:121-247 (add, sub, neg, inv)
 *   math/src/field/fields/montgomery_backed_prime_fields.rs:270-293 (from_u64, from_base_type, representative)
 *   math/src/field/fields/u32_montgomery_backend_prime_field.rs:21-118,273-303 (32-bit backend)
 */
#ifndef ORC_FIELD_H
#define ORC_FIELD_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

#define ORC_MAXL 6
#define ORC_INLINE static inline __attribute__((always_inline))

typedef struct {
    int n;               /* number of 64-bit limbs */
    int spare_bit;       /* MODULUS_HAS_ONE_SPARE_BIT (montgomery_backed_prime_fields.rs:109-111) */
    u64 q[ORC_MAXL];     /* modulus, MS limb first */
    u64 mu;              /* -q^{-1} mod 2^64 */
    u64 r2[ORC_MAXL];    /* R^2 mod q */
    u64 one[ORC_MAXL];   /* R mod q */
    u64 two_adicity;
    u64 root[ORC_MAXL];  /* TWO_ADIC_PRIMITVE_ROOT_OF_UNITY in Montgomery form */
    int is_fft;
} orc_field;

/* ---- UnsignedInteger helpers (MS limb first) ---- */
ORC_INLINE int ui_cmp(const u64 *a, const u64 *b, int n) {
    for (int i = 0; i < n; i++) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
ORC_INLINE int ui_is_zero(const u64 *a, int n) {
    u64 o = 0;
    for (int i = 0; i < n; i++) o |= a[i];
    return o == 0;
}
ORC_INLINE int ui_eq(const u64 *a, const u64 *b, int n) {
    u64 o = 0;
    for (int i = 0; i < n; i++) o |= a[i] ^ b[i];
    return o == 0;
}
/* r = a + b, returns carry (element.rs: UnsignedInteger::add) */
ORC_INLINE u64 ui_add(u64 *r, const u64 *a, const u64 *b, int n) {
    u128 c = 0;
    for (int i = n - 1; i >= 0; i--) {
        c += (u128)a[i] + b[i];
        r[i] = (u64)c;
        c >>= 64;
    }
    return (u64)c;
}
/* r = a - b, returns borrow */
ORC_INLINE u64 ui_sub(u64 *r, const u64 *a, const u64 *b, int n) {
    u64 borrow = 0;
    for (int i = n - 1; i >= 0; i--) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
ORC_INLINE void ui_shr1(u64 *a, int n) {
    for (int i = n - 1; i > 0; i--) a[i] = (a[i] >> 1) | (a[i - 1] << 63);
    a[0] >>= 1;
}
ORC_INLINE void ui_set_u64(u64 *a, u64 v, int n) {
    for (int i = 0; i < n; i++) a[i] = 0;
    a[n - 1] = v;
}
ORC_INLINE void ui_copy(u64 *r, const u64 *a, int n) {
    for (int i = 0; i < n; i++) r[i] = a[i];
}

/* ---- Montgomery multiplication ---- */
/* montgomery.rs:12-76 — generic CIOS with the two extra carry words */
ORC_INLINE void mont_cios(u64 *r, const u64 *a, const u64 *b, const u64 *q, u64 mu, int n) {
    u64 t[ORC_MAXL];
    u64 t_extra[2] = {0, 0};
    for (int i = 0; i < n; i++) t[i] = 0;
    for (int i = n - 1; i >= 0; i--) {
        u128 c = 0, cs;
        for (int j = n - 1; j >= 0; j--) {
            cs = (u128)t[j] + (u128)a[j] * (u128)b[i] + c;
            c = cs >> 64;
            t[j] = (u64)cs;
        }
        cs = (u128)t_extra[1] + c;
        t_extra[0] = (u64)(cs >> 64);
        t_extra[1] = (u64)cs;

        u64 m = t[n - 1] * mu;
        c = ((u128)t[n - 1] + (u128)m * (u128)q[n - 1]) >> 64;
        for (int j = n - 2; j >= 0; j--) {
            cs = (u128)t[j] + (u128)m * (u128)q[j] + c;
            c = cs >> 64;
            t[j + 1] = (u64)cs;
        }
        cs = (u128)t_extra[1] + c;
        c = cs >> 64;
        t[0] = (u64)cs;
        t_extra[1] = t_extra[0] + (u64)c;
    }
    int overflow = t_extra[1] > 0;
    if (overflow || ui_cmp(q, t, n) <= 0) ui_sub(t, t, q, n);
    ui_copy(r, t, n);
}

/* montgomery.rs:86-141 — EdMSM Algorithm 2, moduli with a spare top bit */
ORC_INLINE void mont_cios_spare(u64 *r, const u64 *a, const u64 *b, const u64 *q, u64 mu, int n) {
    u64 t[ORC_MAXL];
    for (int i = 0; i < n; i++) t[i] = 0;
    for (int i = n - 1; i >= 0; i--) {
        u128 c = 0, cs;
        for (int j = n - 1; j >= 0; j--) {
            cs = (u128)t[j] + (u128)a[j] * (u128)b[i] + c;
            c = cs >> 64;
            t[j] = (u64)cs;
        }
        u64 t_extra = (u64)c;
        u64 m = t[n - 1] * mu;
        c = ((u128)t[n - 1] + (u128)m * (u128)q[n - 1]) >> 64;
        for (int j = n - 2; j >= 0; j--) {
            cs = (u128)t[j] + (u128)m * (u128)q[j] + c;
            c = cs >> 64;
            t[j + 1] = (u64)cs;
        }
        cs = (u128)t_extra + c;
        t[0] = (u64)cs;
    }
    if (ui_cmp(q, t, n) <= 0) ui_sub(t, t, q, n);
    ui_copy(r, t, n);
}

/* IsField::mul (montgomery_backed_prime_fields.rs:139-149) */
ORC_INLINE void fp_mul(const orc_field *f, int n, u64 *r, const u64 *a, const u64 *b) {
    if (f->spare_bit) mont_cios_spare(r, a, b, f->q, f->mu, n);
    else mont_cios(r, a, b, f->q, f->mu, n);
}
/* IsField::add (:121-135) */
ORC_INLINE void fp_add(const orc_field *f, int n, u64 *r, const u64 *a, const u64 *b) {
    u64 s[ORC_MAXL];
    u64 overflow = ui_add(s, a, b, n);
    if (f->spare_bit) {
        if (ui_cmp(s, f->q, n) >= 0) ui_sub(s, s, f->q, n);
    } else if (overflow || ui_cmp(s, f->q, n) >= 0) {
        ui_sub(s, s, f->q, n);
    }
    ui_copy(r, s, n);
}
/* IsField::sub (:157-163) */
ORC_INLINE void fp_sub(const orc_field *f, int n, u64 *r, const u64 *a, const u64 *b) {
    u64 d[ORC_MAXL];
    if (ui_cmp(b, a, n) <= 0) {
        ui_sub(d, a, b, n);
    } else {
        ui_sub(d, b, a, n);
        ui_sub(d, f->q, d, n);
    }
    ui_copy(r, d, n);
}
/* IsField::neg (:166-172) */
ORC_INLINE void fp_neg(const orc_field *f, int n, u64 *r, const u64 *a) {
    if (ui_is_zero(a, n)) ui_copy(r, a, n);
    else ui_sub(r, f->q, a, n);
}
/* representative = cios(x, 1) (:291-293) */
ORC_INLINE void fp_representative(const orc_field *f, int n, u64 *r, const u64 *a) {
    u64 one[ORC_MAXL];
    ui_set_u64(one, 1, n);
    mont_cios(r, a, one, f->q, f->mu, n);
}
/* from_base_type = cios(x, R2) (:280-282) */
ORC_INLINE void fp_from_base_type(const orc_field *f, int n, u64 *r, const u64 *a) {
    mont_cios(r, a, f->r2, f->q, f->mu, n);
}
ORC_INLINE void fp_from_u64(const orc_field *f, int n, u64 *r, u64 v) {
    u64 t[ORC_MAXL];
    ui_set_u64(t, v, n);
    mont_cios(r, t, f->r2, f->q, f->mu, n);
}

/* IsField::inv — binary extended Euclid seeded with R2 (:175-247). Returns -1 on zero. */
static int fp_inv(const orc_field *f, int n, u64 *r, const u64 *a) {
    if (ui_is_zero(a, n)) return -1;
    u64 one[ORC_MAXL], u[ORC_MAXL], v[ORC_MAXL], b[ORC_MAXL], c[ORC_MAXL], t[ORC_MAXL];
    ui_set_u64(one, 1, n);
    int has_spare = (f->q[0] >> 63) == 0;
    ui_copy(u, a, n);
    ui_copy(v, f->q, n);
    ui_copy(b, f->r2, n);
    ui_set_u64(c, 0, n);
    while (!ui_eq(u, one, n) && !ui_eq(v, one, n)) {
        while ((u[n - 1] & 1) == 0) {
            ui_shr1(u, n);
            if ((b[n - 1] & 1) == 0) {
                ui_shr1(b, n);
            } else {
                u64 carry = ui_add(b, b, f->q, n);
                ui_shr1(b, n);
                if (!has_spare && carry) b[0] |= 1ull << 63;
            }
        }
        while ((v[n - 1] & 1) == 0) {
            ui_shr1(v, n);
            if ((c[n - 1] & 1) == 0) {
                ui_shr1(c, n);
            } else {
                u64 carry = ui_add(c, c, f->q, n);
                ui_shr1(c, n);
                if (!has_spare && carry) c[0] |= 1ull << 63;
            }
        }
        if (ui_cmp(v, u, n) <= 0) {
            ui_sub(u, u, v, n);
            if (ui_cmp(b, c, n) < 0) {
                ui_sub(t, f->q, c, n);
                ui_add(b, t, b, n);
            } else {
                ui_sub(b, b, c, n);
            }
        } else {
            ui_sub(v, v, u, n);
            if (ui_cmp(c, b, n) < 0) {
                ui_sub(t, f->q, b, n);
                ui_add(c, t, c, n);
            } else {
                ui_sub(c, c, b, n);
            }
        }
    }
    if (ui_eq(u, one, n)) ui_copy(r, b, n);
    else ui_copy(r, c, n);
    return 0;
}

/* ---- parameter derivation (montgomery_backed_prime_fields.rs:59-103) ---- */
static u64 orc_compute_mu(const u64 *q, int n) {
    /* Dusse-Kaliski on the least significant limb */
    u64 y = 1;
    for (int i = 2; i <= 64; i++) {
        u64 lo = q[n - 1] * y;
        u64 masked = (i == 64) ? lo : (lo & ((1ull << i) - 1));
        if (masked != 1) y += 1ull << (i - 1);
    }
    return (u64)(0 - y);
}
static void orc_compute_r2(const u64 *q, int n, u64 *r2) {
    /* l = number of leading-zero-free shift: smallest l with (q >> l) != 0 is 0, so as in the
       reference the loop breaks at l = 0 for any non-zero modulus; c = 1, then doubled 2*64*n times. */
    u64 c[ORC_MAXL], d[ORC_MAXL];
    ui_set_u64(c, 1, n);
    int total = 2 * n * 64;
    for (int i = 1; i <= total; i++) {
        u64 overflow = ui_add(d, c, c, n);
        if (ui_cmp(q, d, n) <= 0 || overflow) ui_sub(c, d, q, n);
        else ui_copy(c, d, n);
    }
    ui_copy(r2, c, n);
}
static void orc_field_init(orc_field *f, int n, const u64 *q) {
    memset(f, 0, sizeof(*f));
    f->n = n;
    ui_copy(f->q, q, n);
    f->spare_bit = q[0] < ((1ull << 63) - 1);
    f->mu = orc_compute_mu(q, n);
    orc_compute_r2(q, n, f->r2);
    u64 one[ORC_MAXL];
    ui_set_u64(one, 1, n);
    mont_cios(f->one, one, f->r2, f->q, f->mu, n);
}

/* ---- 32-bit Montgomery backend (u32_montgomery_backend_prime_field.rs) ---- */
typedef struct { u32 q, mu, r2, one; } orc_field32;

ORC_INLINE u32 m32_reduce(u64 x, u32 mu, u32 q) { /* :278-292 */
    u64 t = (x * (u64)mu) & 0xffffffffull;
    u64 u = t * (u64)q;
    u64 d = x - u;
    int over = x < u;
    u32 hi = (u32)(d >> 32);
    return hi + (over ? q : 0);
}
ORC_INLINE u32 m32_mul(const orc_field32 *f, u32 a, u32 b) { return m32_reduce((u64)a * (u64)b, f->mu, f->q); }
ORC_INLINE u32 m32_add(const orc_field32 *f, u32 a, u32 b) { /* :84-92 */
    u32 s = a + b;
    u32 c = s - f->q;
    return (s < f->q) ? s : c;
}
ORC_INLINE u32 m32_sub(const orc_field32 *f, u32 a, u32 b) { return (b <= a) ? a - b : f->q - (b - a); }
ORC_INLINE u32 m32_neg(const orc_field32 *f, u32 a) { return a == 0 ? 0 : f->q - a; }
static u32 m32_pow(const orc_field32 *f, u32 a, u64 e) {
    u32 r = f->one;
    while (e) {
        if (e & 1) r = m32_mul(f, r, a);
        a = m32_mul(f, a, a);
        e >>= 1;
    }
    return r;
}
static void orc_field32_init(orc_field32 *f, u32 q) {
    f->q = q;
    u32 y = 1; /* :33-49, NOT negated */
    for (int i = 2; i <= 32; i++) {
        u32 m = (u32)((u64)q * (u64)y);
        u32 masked = (i == 32) ? m : (m & ((1u << i) - 1));
        if (masked != 1) y += 1u << (i - 1);
    }
    f->mu = y;
    /* :51-81 — with (q >> 0) != 0 the scan stops at l = 0, c = 1, doubled 64 times */
    u32 c = 1;
    for (int i = 1; i <= 64; i++) {
        u32 d = c << 1;
        c = d >= q ? d - q : d;
    }
    f->r2 = c;
    f->one = m32_reduce((u64)1 * (u64)f->r2, f->mu, f->q);
}
#endif
