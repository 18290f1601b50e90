"""Pins the oracle's group law + Pippenger restatement: the reference's curve KATs, and affine big-int
definitions.  Mirrors bls12_381/curve.rs:154-232, bn_254/curve.rs:99-153, short_weierstrass/point.rs tests and
msm/pippenger.rs:204-233 (Pippenger == naive for windows 1..8, <=30 points, 384-bit scalars; parallel == seq)."""
import random

import numpy as np
import pytest

from oracle import bigint_def as D
from oracle import oracle as O

H = lambda s: int(s, 16)
CURVES = [O.C_BLS12_381_G1, O.C_BN254_G1, O.C_BN254_G2, O.C_BLS12_381_G2]


def gen_point(curve):
    c = D.CURVES[curve]
    return O.point_from_affine_ints(curve, c.gen[0].tup(), c.gen[1].tup())


def aff(curve, p):
    return O.point_to_affine_ints(curve, p)


def test_bls12_381_point1_times_5(kats):
    k = kats["bls12_381_g1"]
    c = O.C_BLS12_381_G1
    p1 = O.point_from_affine_ints(c, H(k["point_1"][0]), H(k["point_1"][1]))
    assert D.BLS12_381_G1.on_curve(D.BLS12_381_G1.pt(H(k["point_1"][0]), H(k["point_1"][1])))
    p5 = O.ec_mul(c, p1, 5, 1)
    assert aff(c, p5) == (H(k["point_1_times_5"][0]), H(k["point_1_times_5"][1]))
    # equality by cross-multiplication (elliptic_curve/point.rs:57-63)
    assert O.ec_eq(c, p5, O.point_from_affine_ints(c, H(k["point_1_times_5"][0]), H(k["point_1_times_5"][1])))
    g = gen_point(c)
    assert aff(c, g) == (H(k["generator"][0]), H(k["generator"][1]))
    assert not O.ec_eq(c, g, O.ec_add(c, g, g))            # curve.rs:186-191
    assert O.ec_eq(c, O.ec_add(c, O.ec_add(c, g, g), g), O.ec_mul(c, g, 3, 1))  # :211-217
    # subgroup order kills the generator (:219-222 generator_g1_is_in_subgroup)
    assert aff(c, O.ec_mul(c, g, H(k["subgroup_order"]))) is None


def test_bn254_point_times_5(kats):
    k = kats["bn254_g1"]
    c = O.C_BN254_G1
    p = O.point_from_affine_ints(c, H(k["point"][0]), H(k["point"][1]))
    assert aff(c, O.ec_mul(c, p, 5, 1)) == (H(k["point_times_5"][0]), H(k["point_times_5"][1]))
    assert aff(c, O.ec_mul(c, gen_point(c), H(k["subgroup_order"]))) is None


@pytest.mark.parametrize("curve", CURVES)
def test_group_law_vs_affine_definition(curve):
    cd = D.CURVES[curve]
    assert cd.on_curve(cd.gen)
    rng = random.Random(curve)
    g = gen_point(curve)
    ks = [rng.randrange(1, 1 << 64) for _ in range(4)]
    pts = [O.ec_mul(curve, g, k, 1) for k in ks]
    dpts = [cd.mul(k, cd.gen) for k in ks]
    for p, dp in zip(pts, dpts):
        assert aff(curve, p) == cd.tup(dp)
    # general add, with non-normalised (Z != 1) operands
    s = O.ec_add(curve, pts[0], pts[1])
    assert aff(curve, s) == cd.tup(cd.add(dpts[0], dpts[1]))
    # exceptional cases (point.rs:171-189): identity, P+P -> double, P+(-P) -> neutral
    n = O.ec_neutral(curve)
    assert aff(curve, n) is None
    assert np.array_equal(O.ec_add(curve, pts[0], n), pts[0]) and np.array_equal(O.ec_add(curve, n, pts[0]), pts[0])
    assert aff(curve, O.ec_add(curve, pts[2], pts[2])) == cd.tup(cd.add(dpts[2], dpts[2]))
    assert aff(curve, O.ec_double(curve, pts[2])) == cd.tup(cd.add(dpts[2], dpts[2]))
    assert aff(curve, O.ec_add(curve, pts[3], O.ec_neg(curve, pts[3]))) is None
    # same point, different projective representative -> still doubles
    two = O.ec_add(curve, pts[2], pts[2])           # some non-normalised point
    same = O.ec_add(curve, O.ec_to_affine(curve, two), n)
    assert aff(curve, O.ec_add(curve, two, same)) == cd.tup(cd.mul(4, dpts[2]))
    assert np.array_equal(O.ec_double(curve, n), n)   # double(neutral) = neutral (:55-57)
    # to_affine of neutral is (0,1,0) (elliptic_curve/point.rs:43-51)
    assert np.array_equal(O.ec_to_affine(curve, n), n)


def _rand_msm(curve, n, k_limbs, rng):
    g = gen_point(curve)
    pts = np.stack([O.ec_mul(curve, g, rng.randrange(1, 1 << 62), 1) for _ in range(n)]) if n else np.zeros((0, g.size), np.uint64)
    ks = [rng.getrandbits(64 * k_limbs) for _ in range(n)]
    return ks, O.ints_to_array(ks, k_limbs), pts


def test_pippenger_equals_naive_bls12_381_384bit_scalars():
    # pippenger.rs:204-219 restated with a fixed seed: window 1..8, <= 30 points, UnsignedInteger<6> scalars
    rng = random.Random(42)
    c = O.C_BLS12_381_G1
    for case in range(6):
        n = rng.randrange(1, 31)
        ks, sc, pts = _rand_msm(c, n, 6, rng)
        naive = O.msm_naive(c, sc, pts, 6)
        for window in (1, 2, 5, 8):
            got = O.msm_with(c, sc, pts, window, 6)
            assert O.ec_eq(c, got, naive)
            par = O.parallel_msm_with(c, sc, pts, window, 4, 6)   # :221-232
            assert O.ec_eq(c, par, got)


@pytest.mark.parametrize("curve", CURVES)
def test_msm_vs_affine_definition(curve):
    cd = D.CURVES[curve]
    rng = random.Random(10 + curve)
    n = 12 if curve in (O.C_BLS12_381_G1, O.C_BN254_G1) else 6
    g = gen_point(curve)
    mult = [rng.randrange(1, 1 << 30) for _ in range(n)]
    pts = np.stack([O.ec_mul(curve, g, m, 1) for m in mult])
    ks = [rng.getrandbits(256) for _ in range(n)]
    ks[0] = 0; ks[1] = 1                                  # edge scalars
    sc = O.ints_to_array(ks, 4)
    got = O.msm(curve, sc, pts)
    total = sum(k * m for k, m in zip(ks, mult))
    assert aff(curve, got) == cd.tup(cd.mul(total, cd.gen))


def test_msm_edge_cases():
    c = O.C_BLS12_381_G1
    g = gen_point(c)
    # empty -> neutral (pippenger.rs:102)
    out = O.msm(c, np.zeros((0, 4), np.uint64), np.zeros((0, 18), np.uint64))
    assert aff(c, out) is None
    # length mismatch -> MSMError::LengthMismatch (:25-27)
    with pytest.raises(O.OracleError) as e:
        O.msm(c, np.zeros((2, 4), np.uint64), np.stack([g]))
    assert e.value.code == O.ERR_LENGTH_MISMATCH
    # duplicates, P and -P in one bucket, neutral inputs, scalar = r-1
    r = D.P_FR381
    p = O.ec_mul(c, g, 12345, 1)
    pts = np.stack([p, p, O.ec_neg(c, p), O.ec_neutral(c), g])
    ks = [7, 7, 7, 99, r - 1]
    got = O.msm(c, O.ints_to_array(ks, 4), pts)
    exp = D.BLS12_381_G1.mul((7 * 12345 + r - 1) % r, D.BLS12_381_G1.gen)
    assert aff(c, got) == D.BLS12_381_G1.tup(exp)
    # window rule pippenger.rs:34-40
    assert [O.optimum_window_size(n) for n in (0, 1, 2, 1 << 10, 1 << 20, 1 << 24)] == [0, 0, 0, 8, 16, 19]
    assert D.adds_ref(1 << 20) == 18874336 and D.adds_ref(1 << 24) == 249561060


def test_gen_points_are_non_normalised_and_correct():
    c = O.C_BLS12_381_G1
    g = gen_point(c)
    pts = O.gen_points(c, g, 5, 3, 6)
    one = O.field_params(O.F_FP381)["one"]
    for i in range(6):
        assert O.point_to_affine_ints(c, pts[i]) == D.BLS12_381_G1.tup(D.BLS12_381_G1.mul(5 + 3 * i, D.BLS12_381_G1.gen))
    assert O.limbs_to_int(pts[3][12:18]) != one


def test_oracle_reproduces_the_reference_held_plonk_round_1_commitments(kats):
    # provers/plonk/src/prover.rs:760-785: interpolate_fft over BLS12-381 Fr, then the KZG commitment = msm against the
    # test SRS; the expected points are hard-coded in the reference's test
    from tests import util
    kat = kats["plonk_round_1_commitments"]
    fr, oid = O.F_FR381, O.C_BLS12_381_G1
    srs = util.plonk_test_srs(oid, kat["srs_len"], kat["srs_secret"])
    for name, col in kat["columns"].items():
        coeffs = O.interpolate_fft(fr, O.elems_to_mont(fr, col), strip=True)
        ks = O.elems_from_mont(fr, coeffs)                      # .representative()
        got = O.msm(oid, O.ints_to_array(ks, 4), srs[:len(ks)])
        assert O.point_to_affine_ints(oid, got) == tuple(int(v, 16) for v in kat["expected"][name]), name


class _OraclePlonkOps:
    """tests/plonk_kat.py operations on the CPU oracle"""

    def __init__(self, srs):
        self.fr, self.oid, self.srs = O.F_FR381, O.C_BLS12_381_G1, srs

    def _m(self, v):
        return O.elems_to_mont(self.fr, v)

    def interp(self, evals):
        return O.elems_from_mont(self.fr, O.interpolate_fft(self.fr, self._m(evals)))

    def eval_offset(self, coeffs, domain_size, offset):
        import numpy as np
        c = self._m(coeffs) if coeffs else np.zeros((0, 4), np.uint64)
        return O.elems_from_mont(self.fr, O.evaluate_fft(self.fr, c, 1, domain_size, self._m([offset])[0]))

    def interp_offset(self, evals, offset):
        return O.elems_from_mont(self.fr, O.interpolate_fft(self.fr, self._m(evals), self._m([offset])[0]))

    def commit(self, coeffs):
        import numpy as np
        ks = O.ints_to_array(coeffs, 4) if coeffs else np.zeros((0, 4), np.uint64)
        return O.point_to_affine_ints(self.oid, O.msm(self.oid, ks, self.srs[:len(coeffs)]))


def test_oracle_reproduces_the_reference_held_plonk_round_2_and_3_commitments(kats):
    # provers/plonk/src/prover.rs:787-836: z_1, t_lo_1, t_mid_1 hard-coded, t_hi_1 the neutral element — pins
    # interpolate_fft, evaluate_offset_fft on a 4x larger coset, interpolate_offset_fft over BLS12-381 Fr and the KZG MSM
    from oracle import bigint_def as D
    from tests import plonk_kat, util
    fr, oid = O.F_FR381, O.C_BLS12_381_G1
    srs = util.plonk_test_srs(oid, 7, 2)
    omega = O.elems_from_mont(fr, [O.get_primitive_root_of_unity(fr, 2)])[0]
    got = plonk_kat.rounds_1_to_3(_OraclePlonkOps(srs), omega)
    want = dict(kats["plonk_round_1_commitments"]["expected"])
    want = {k + "_1": v for k, v in want.items()}
    want.update(kats["plonk_round_2_3_commitments"]["expected"])
    for name, v in want.items():
        assert got[name] == (tuple(int(t, 16) for t in v) if v else None), name
