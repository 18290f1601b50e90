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
