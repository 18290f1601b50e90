/* Test driver (CPU hygiene): exercises the oracle under ASan/UBSan without the Python interpreter. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../oracle/lw_oracle.h"

static uint64_t rng_state = 88172645463325252ULL;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(void) {
    /* NTT: Stark252 64 coefficients, blow-up 2, coset 3; round trip */
    enum { N = 64 };
    uint64_t *a = malloc(N * 32), *ev = malloc(2 * N * 32), *back = malloc(2 * N * 32);
    for (int i = 0; i < N * 4; i++) a[i] = rnd();
    for (int i = 0; i < N; i++) a[4 * i] &= (1ULL << 59) - 1;
    uint64_t three[4] = {0, 0, 0, 3}, off[4];
    if (orc_fe_op(ORC_F_STARK252, ORC_OP_TO_MONT, three, NULL, off)) return 1;
    size_t len = 0;
    if (orc_evaluate_fft(ORC_F_STARK252, a, N, 2, N, off, ev, &len) || len != 2 * N) return 2;
    size_t clen = 0;
    if (orc_interpolate_fft(ORC_F_STARK252, ev, 2 * N, off, back, &clen)) return 3;
    if (memcmp(back, a, N * 32)) return 4;
    /* BabyBear u32 */
    uint32_t *b = malloc(N * 4), *bev = malloc(N * 4);
    for (int i = 0; i < N; i++) b[i] = (uint32_t)(rnd() % 2013265921u);
    if (orc_evaluate_fft(ORC_F_BABYBEAR_U32, b, N, 1, 0, NULL, bev, &len)) return 5;
    /* MSM: BN254 G1, 24 points from the generator (1,2) */
    uint64_t one[4] = {0, 0, 0, 1}, two[4] = {0, 0, 0, 2}, gen[12];
    orc_fe_op(ORC_F_FP254, ORC_OP_TO_MONT, one, NULL, gen);
    orc_fe_op(ORC_F_FP254, ORC_OP_TO_MONT, two, NULL, gen + 4);
    memcpy(gen + 8, gen, 32);
    enum { M = 24 };
    uint64_t *pts = malloc(M * 96), *sc = malloc(M * 32), s0[4] = {0, 0, 0, 77}, d[4] = {0, 0, 0, 5}, r1[12], r2[12];
    if (orc_gen_points(ORC_C_BN254_G1, gen, s0, d, 4, M, pts)) return 6;
    for (int i = 0; i < M * 4; i++) sc[i] = rnd();
    if (orc_msm(ORC_C_BN254_G1, sc, M, 4, pts, M, r1)) return 7;
    if (orc_msm_naive(ORC_C_BN254_G1, sc, 4, pts, M, r2)) return 8;
    if (orc_ec_op(ORC_C_BN254_G1, ORC_EC_EQ, r1, r2, NULL) != 1) return 9;
    if (orc_parallel_msm_with(ORC_C_BN254_G1, sc, 4, pts, M, 3, 2, r2)) return 10;
    if (orc_ec_op(ORC_C_BN254_G1, ORC_EC_EQ, r1, r2, NULL) != 1) return 11;
    if (orc_msm(ORC_C_BN254_G1, sc, M, 4, pts, M - 1, r1) != ORC_ERR_LENGTH_MISMATCH) return 12;
    free(a); free(ev); free(back); free(b); free(bev); free(pts); free(sc);
    printf("ok\n");
    return 0;
}
