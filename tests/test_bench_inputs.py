"""bench.py synthesises its MSM inputs without the oracle; the oracle checks here that they are what they claim."""
import numpy as np

import bench_msm
from oracle import oracle as O


def test_synthetic_bls12381_points_are_distinct_curve_points_with_z_not_one():
    pts = bench_msm.synth_points_bls12381_g1(256, 7)
    assert pts.shape == (256, 18) and pts.dtype == np.uint64
    one = O.int_to_limbs(O.field_params(O.F_FP381)["one"], 6)       # Montgomery form of 1
    p = bench_msm.BLS_P
    seen = set()
    for row in pts:
        assert not np.array_equal(row[12:18], one)                 # Z != 1: not normalised
        x, y = O.point_to_affine_ints(O.C_BLS12_381_G1, row)[:2]
        assert (y * y - x * x * x - 4) % p == 0                    # on y^2 = x^3 + 4
        seen.add((x, y))
    assert len(seen) == 256


def test_adds_ref_matches_survey_table():
    # 2^20 -> 18.87 M, 2^22 -> 71.30 M, 2^24 -> 249.56 M, 2^26 -> 899.68 M
    assert bench_msm.adds_ref(1 << 20) == (1 << 20) * 16 + 2 * 16 * ((1 << 16) - 1)
    assert round(bench_msm.adds_ref(1 << 24) / 1e6, 2) == 249.56
    assert round(bench_msm.adds_ref(1 << 26) / 1e6, 2) == 899.68


def test_synthetic_scalars_are_uniform_256_bit_reduced_mod_r():
    rng_n = 5000
    a = bench_msm.synth_scalars_mod_r(rng_n, 42)
    rng = np.random.default_rng(42)   # the same draw, reduced with Python integers
    raw = rng.integers(0, 1 << 63, size=(rng_n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(rng_n, 4), dtype=np.uint64)
    want = [v % bench_msm.BLS_R for v in O.array_to_ints(raw)]
    assert O.array_to_ints(a) == want
    assert sum(1 for v, w in zip(O.array_to_ints(raw), want) if v != w) > rng_n // 3    # the reduction is exercised


def test_synthetic_run_starting_at_zero_begins_with_the_identity():
    cols = bench_msm.synth_run_bls12381_g1(4, 0, 5, 3)
    oid = O.C_BLS12_381_G1
    assert O.point_to_affine_ints(oid, cols[0]) is None
    g5 = O.ec_mul(oid, __import__("tests.util", fromlist=["generator"]).generator(oid), 5, 1)
    assert O.ec_eq(oid, cols[1], g5) and O.ec_eq(oid, cols[2], O.ec_add(oid, g5, g5))


def test_cfg5_inputs_and_closed_form_match_the_oracle():
    """bench_cfg.py synthesises BN254 G1 / G2 runs P_i = [s0 + i d]G and checks the timed 2^26 MSM against the closed form
    [sum k_i (s0 + i d) mod r]G, both with its own Python big-integer arithmetic: pin that arithmetic to the oracle."""
    import bench_cfg as B
    from oracle import oracle as O
    from tests import util
    sc = B.scalars_mod(1000, 1, B.BN_R)
    ks = O.array_to_ints(sc)
    assert all(k < B.BN_R for k in ks)
    s0, s1 = B.weighted_scalar_sums(sc, 12345)
    assert s0 == sum(ks) and s1 == sum((12345 + i) * k for i, k in enumerate(ks))
    for name, oid in (("bn254_g1", O.C_BN254_G1), ("bn254_g2", O.C_BN254_G2)):
        pts = B.synth_run(name, 6, 5, 3, 9)
        g = util.generator(oid)
        for i in range(6):
            assert O.ec_eq(oid, pts[i], O.ec_mul(oid, g, 5 + 3 * i, 4)), (name, i)
        sc6 = B.scalars_mod(6, 3, B.BN_R)
        a, b = B.weighted_scalar_sums(sc6, 0)
        assert B.point_to_affine(name, O.msm(oid, sc6, pts)) == B.closed_form_msm(name, 5, 3, a, b, B.BN_R)
