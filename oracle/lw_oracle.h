/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  See lw_oracle.c for the header statement and citations.
 */
#ifndef LW_ORACLE_H
#define LW_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC_F_STARK252 = 0,      /* Stark252PrimeField, 4xu64 */
    ORC_F_FR381 = 1,         /* BLS12-381 FrField, 4xu64 */
    ORC_F_BABYBEAR_U64 = 2,  /* Babybear31PrimeField as MontgomeryBackendPrimeField<_,1>, R = 2^64 */
    ORC_F_BABYBEAR_U32 = 3,  /* U32MontgomeryBackendPrimeField<2013265921>, R = 2^32 */
    ORC_F_BABYBEAR_EXT4 = 4, /* Degree4BabyBearExtensionField values, base-field domain */
    ORC_F_FP381 = 5,         /* BLS12-381 base field, 6xu64 */
    ORC_F_FP254 = 6,         /* BN254 base field */
    ORC_F_FR254 = 7          /* BN254 scalar field (not an FFT field in the reference) */
};
enum { ORC_C_BLS12_381_G1 = 0, ORC_C_BN254_G1 = 1, ORC_C_BN254_G2 = 2, ORC_C_BLS12_381_G2 = 3 };
enum { ORC_OP_ADD = 0, ORC_OP_SUB, ORC_OP_MUL, ORC_OP_NEG, ORC_OP_INV, ORC_OP_TO_MONT, ORC_OP_FROM_MONT };
enum { ORC_EC_ADD = 0, ORC_EC_DOUBLE, ORC_EC_NEG, ORC_EC_TO_AFFINE, ORC_EC_EQ, ORC_EC_NEUTRAL };
enum { ORC_ROOTS_NATURAL = 0, ORC_ROOTS_NATURAL_INV = 1, ORC_ROOTS_BITREV = 2, ORC_ROOTS_BITREV_INV = 3 };

/* error codes mirror FFTError / FieldError / MSMError variants */
enum {
    ORC_OK = 0,
    ORC_ERR_INPUT_NOT_POW2 = -1,  /* FFTError::InputError */
    ORC_ERR_ORDER = -2,           /* FFTError::OrderError */
    ORC_ERR_ROOT_OF_UNITY = -3,   /* FieldError::RootOfUnityError */
    ORC_ERR_LENGTH_MISMATCH = -4, /* MSMError::LengthMismatch */
    ORC_ERR_INV_ZERO = -5,        /* FieldError::InvZeroError */
    ORC_ERR_ALLOC = -6,
    ORC_ERR_BAD_ARG = -7
};

size_t orc_field_elem_bytes(int field);
int orc_field_params(int field, uint64_t *q, uint64_t *mu, uint64_t *r2, uint64_t *one);
int orc_derive_params(int n, const uint64_t *q, uint64_t *mu, uint64_t *r2, uint64_t *one, int *spare_bit);
int orc_mont_cios(int n, const uint64_t *a, const uint64_t *b, const uint64_t *q, uint64_t mu, uint64_t *r);
int orc_mont_cios_spare(int n, const uint64_t *a, const uint64_t *b, const uint64_t *q, uint64_t mu, uint64_t *r);
int orc_fe_op_mod(int n, const uint64_t *q, int op, const uint64_t *a, const uint64_t *b, uint64_t *r);
int orc_fe_op(int field, int op, const void *a, const void *b, void *r);

int orc_get_primitive_root_of_unity(int field, uint64_t order, void *out);
int orc_get_powers_of_primitive_root(int field, uint64_t n, size_t count, int config, void *out);
int orc_get_twiddles(int field, uint64_t order, int config, void *out);
int orc_bit_reverse_permute(int field, void *data, size_t n);
int orc_in_place_nr_2radix_fft(int field, void *data, size_t n, const void *twiddles);
int orc_fft(int field, const void *in, size_t n, const void *twiddles, void *out);
int orc_evaluate_fft(int field, const void *coeffs, size_t ncoeffs, size_t blowup, size_t domain_size,
                     const void *offset, void *out, size_t *out_len);
int orc_interpolate_fft(int field, const void *evals, size_t n, const void *offset, void *out, size_t *coeff_len);

size_t orc_curve_point_bytes(int curve);
int orc_ec_op(int curve, int op, const void *p, const void *q, void *r);
int orc_ec_mul(int curve, const void *p, const uint64_t *k, int k_limbs, void *r);
int orc_msm(int curve, const uint64_t *cs, size_t n_scalars, int k_limbs, const void *points, size_t n_points, void *out);
int orc_msm_with(int curve, const uint64_t *cs, int k_limbs, const void *points, size_t n, size_t window, void *out);
int orc_parallel_msm_with(int curve, const uint64_t *cs, int k_limbs, const void *points, size_t n, size_t window, int threads, void *out);
int orc_msm_naive(int curve, const uint64_t *cs, int k_limbs, const void *points, size_t n, void *out);
size_t orc_optimum_window_size(size_t n);
int orc_gen_points(int curve, const void *gen, const uint64_t *s0, const uint64_t *delta, int k_limbs, size_t n, void *out);

int orc_keccak256_bytes(const uint8_t *data, size_t len, uint8_t *out32);
int orc_merkle_commit_columns(const uint64_t *columns, uint32_t n_cols, uint32_t log2n, int bit_reverse, uint8_t *nodes_out);
int orc_merkle_commit_columns_bytes(const void *columns, uint32_t elem_bytes, uint32_t n_cols, uint32_t log2n, int bit_reverse, uint8_t *nodes_out,
                                    int threads);
int orc_merkle_commit_columns_mt(const uint64_t *columns, uint32_t n_cols, uint32_t log2n, int bit_reverse, uint8_t *nodes_out, int threads);
/* 2 * fold_polynomial(p, zeta) (provers/stark/src/fri/mod.rs:49, fri/fri_functions.rs:7-30) */
int orc_fri_fold_twice(int field, const uint64_t *coeffs, size_t n, const uint64_t *zeta, uint64_t *out, size_t *out_len);
/* calculate_h_coefficients (provers/groth16/src/qap.rs:15-39) on the accumulated L, R, O */
int orc_groth16_h_coefficients(const uint64_t *l, const uint64_t *r, const uint64_t *o, size_t ncoeffs, size_t gates, uint64_t *out,
                               size_t *coeff_len);

#ifdef __cplusplus
}
#endif
#endif
