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
