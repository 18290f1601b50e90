"""The HIP path against the committed golden vectors (tests/golden/reference_kats.json — values the reference's own
tests hold, plus the SURVEY anchors derived from its constants), with no oracle in between."""
import numpy as np
import pytest

from oracle import oracle as O      # only for integer <-> memory-layout conversion helpers
from tests import util

pytestmark = pytest.mark.gpu
H = lambda s: int(s, 16)


def test_msm_reproduces_reference_curve_kats(kats):
    # 5 * point_1 (bls12_381/curve.rs:154-171) and 5 * P (bn_254/curve.rs:99-153) as one-term MSMs
    from lambda_elliptic_curves_amd import msm
    for name, key, pk, p5k in (("bls12_381_g1", "bls12_381_g1", "point_1", "point_1_times_5"), ("bn254_g1", "bn254_g1", "point", "point_times_5")):
        crv, oid = util.curve_pairs()[name]
        k = kats[key]
        p = O.point_from_affine_ints(oid, H(k[pk][0]), H(k[pk][1]))
        got = msm.msm(crv, O.ints_to_array([5], 4), np.stack([p]))
        assert O.point_to_affine_ints(oid, got) == (H(k[p5k][0]), H(k[p5k][1]))
        # 2*P + 3*P through two buckets, and the subgroup order annihilates the generator
        got = msm.msm(crv, O.ints_to_array([2, 3], 4), np.stack([p, p]))
        assert O.point_to_affine_ints(oid, got) == (H(k[p5k][0]), H(k[p5k][1]))
        g = O.point_from_affine_ints(oid, H(k["generator"][0]), H(k["generator"][1]))
        assert O.point_to_affine_ints(oid, msm.msm(crv, O.ints_to_array([H(k["subgroup_order"])], 4), np.stack([g]))) is None


def test_ntt_reproduces_survey_anchor_vectors(kats):
    from lambda_elliptic_curves_amd import fft
    a = kats["survey_anchors"]
    f = O.F_STARK252
    ev = fft.evaluate_fft(fft.Stark252PrimeField, O.elems_to_mont(f, [1, 2, 3, 4]))
    assert O.elems_from_mont(f, ev) == [H(x) for x in a["stark252_ntt4_1234"]]
    # twiddle generation: w_4 and w_{2^16} (natural config, entry 1 = w)
    assert O.elems_from_mont(f, fft.get_twiddles(fft.Stark252PrimeField, 2, fft.ROOTS_NATURAL))[1] == H(a["stark252_w4"])
    assert O.elems_from_mont(f, fft.get_twiddles(fft.Stark252PrimeField, 16, fft.ROOTS_NATURAL)[:2])[1] == H(a["stark252_w_2_16"])
    assert O.elems_from_mont(O.F_FR381, fft.get_twiddles(fft.FrField, 20, fft.ROOTS_NATURAL)[:2])[1] == H(a["fr381_w_2_20"])
    assert O.elems_from_mont(O.F_BABYBEAR_U32, fft.get_twiddles(fft.Babybear31PrimeFieldU32, 20, fft.ROOTS_NATURAL)[:2])[1] == H(a["babybear_w_2_20"])
    # interpolating the constant-one evaluations over 2^20 points gives the polynomial 1; N^-1 is the anchor value
    one = O.elems_to_mont(f, [1])
    n_inv = fft.interpolate_fft(fft.Stark252PrimeField, np.concatenate([one, np.zeros((3, 4), np.uint64)]))   # [1,0,0,0] -> all 1/4
    assert O.elems_from_mont(f, n_inv) == [pow(4, -1, 0x800000000000011000000000000000000000000000000000000000000000001)] * 4


@pytest.mark.parametrize("name", ["stark252", "fr381", "babybear_u64", "babybear_u32"])
def test_compose_fft_fixed_case(name):
    """composition_fft_works (fft/polynomial.rs:346-354): compose_fft(2x, x^3) == 2x^3 through the HIP evaluate_fft /
    interpolate_fft (the reference's own fixed case for this API; field-independent identity)."""
    fld, oid = util.field_pairs()[name]
    p = util.field_modulus(name)
    ev = O.elems_from_mont(oid, fft_mod().evaluate_fft(fld, O.elems_to_mont(oid, [0, 0, 0, 1])))
    vals = O.elems_to_mont(oid, [2 * v % p for v in ev])
    assert O.elems_from_mont(oid, fft_mod().interpolate_fft(fld, vals)) == [0, 0, 0, 2]


def fft_mod():
    from lambda_elliptic_curves_amd import fft
    return fft


def test_bit_reverse_table_16(kats):
    from lambda_elliptic_curves_amd import fft
    arr = O.elems_to_mont(O.F_BABYBEAR_U32, list(range(16)))
    got = O.elems_from_mont(O.F_BABYBEAR_U32, fft.bitrev_permutation(fft.Babybear31PrimeFieldU32, arr))
    assert got == kats["bit_reverse_16"]["expected"]      # bit_reversing.rs:32-35


def test_hip_path_reproduces_the_reference_held_stone_compat_trace_commitments(kats):
    # reference-held roots (provers/stark/src/prover.rs:1273-1281,1659-1667) through the HIP path end to end:
    # lw_polynomial_interpolate_fft -> lw_polynomial_evaluate_fft (blow-up, coset offset) -> lw_stark_commit_columns, and
    # the device-resident chain lw_hip_ntt_device(inverse) -> lw_hip_ntt_lde_device -> lw_stark_commit_columns_device
    import torch
    from lambda_elliptic_curves_amd import fft, merkle
    from oracle import oracle as O
    from tests import util
    fld, oid = util.field_pairs()["stark252"]
    for case in kats["stone_compat_trace_commitments"]["cases"]:
        n, blow = case["trace_length"], case["blowup_factor"]
        off = util.offset_elem("stark252", case["coset_offset"])
        cols_mont = [O.elems_to_mont(oid, col) for col in util.stone_compat_trace_columns(case["initial"], n)]
        lde = [fft.evaluate_offset_fft(fld, fft.interpolate_fft(fld, c), blow, n, off) for c in cols_mont]
        root, nodes = merkle.commit_columns(fld, np.stack(lde), bit_reverse=True, return_nodes=True)
        assert bytes(root).hex() == case["root"], case["name"]
        have = {bytes(x).hex() for x in np.asarray(nodes).reshape(-1, 32)}
        for h in case["auth_path_nodes"]:
            assert h in have, (case["name"], h)
        # device-resident: trace columns -> coefficients -> LDE -> tree without leaving HBM
        log_n, log_lde = n.bit_length() - 1, (n * blow).bit_length() - 1
        t_tr = torch.from_numpy(np.concatenate(cols_mont).view(np.int64)).cuda()
        t_co = torch.empty_like(t_tr)
        fft.ntt_device(fld, t_tr, t_co, log_n, inverse=True, batch=2)
        t_lde = torch.empty((2 << log_lde, 4), dtype=torch.int64, device="cuda")
        fft.lde_device(fld, t_co, log_n, t_lde, log_lde, batch=2, offset=off)
        t_nodes = torch.empty(((2 << log_lde) - 1) * 4, dtype=torch.int64, device="cuda")
        root_d = merkle.commit_columns_device(fld, t_lde, 2, log_lde, t_nodes)
        assert bytes(root_d).hex() == case["root"], case["name"] + " (device-resident)"


def test_hip_path_reproduces_the_reference_held_plonk_round_1_commitments(kats):
    # provers/plonk/src/prover.rs:760-785 through the HIP path: lw_polynomial_interpolate_fft over BLS12-381 Fr, then the
    # KZG commitment in its real call shape — Montgomery-form coefficients (representative() on the device) against a
    # device-cached SRS prefix (lw_hip_srs_create + lw_hip_msm_srs_fr), and through plain lw_hip_msm as well
    from lambda_elliptic_curves_amd import fft, msm
    from oracle import oracle as O
    from tests import util
    kat = kats["plonk_round_1_commitments"]
    fr, oid = O.F_FR381, O.C_BLS12_381_G1
    crv = msm.BLS12381Curve
    srs_pts = util.plonk_test_srs(oid, kat["srs_len"], kat["srs_secret"])
    srs = msm.Srs(crv, srs_pts)
    try:
        for name, col in kat["columns"].items():
            want = tuple(int(v, 16) for v in kat["expected"][name])
            coeffs = fft.interpolate_fft(fft.FrField, O.elems_to_mont(fr, col), strip=True)   # Polynomial::new strips
            assert O.point_to_affine_ints(oid, srs.msm_fr(coeffs)) == want, name
            canon = O.ints_to_array(O.elems_from_mont(fr, coeffs), 4)
            assert O.point_to_affine_ints(oid, msm.msm(crv, canon, srs_pts[:len(canon)])) == want, name
    finally:
        srs.close()


class _HipPlonkOps:
    """tests/plonk_kat.py operations on the HIP path (host-buffer C ABI; KZG commit through a device-cached SRS)"""

    def __init__(self, srs):
        from lambda_elliptic_curves_amd import fft
        from oracle import oracle as O
        self.fft, self.O, self.fld, self.fr, self.oid, self.srs = fft, O, fft.FrField, O.F_FR381, O.C_BLS12_381_G1, srs

    def _m(self, v):
        return self.O.elems_to_mont(self.fr, v)

    def interp(self, evals):
        return self.O.elems_from_mont(self.fr, self.fft.interpolate_fft(self.fld, self._m(evals)))

    def eval_offset(self, coeffs, domain_size, offset):
        c = self._m(coeffs) if coeffs else np.zeros((0, 4), np.uint64)
        return self.O.elems_from_mont(self.fr, self.fft.evaluate_offset_fft(self.fld, c, 1, domain_size, self._m([offset])[0]))

    def interp_offset(self, evals, offset):
        return self.O.elems_from_mont(self.fr, self.fft.interpolate_offset_fft(self.fld, self._m(evals), self._m([offset])[0]))

    def commit(self, coeffs):   # kzg.rs:159-163: msm(coefficients.representative(), srs[..len]); Montgomery scalars in, prefix call
        c = self._m(coeffs) if coeffs else np.zeros((0, 4), np.uint64)
        return self.O.point_to_affine_ints(self.oid, self.srs.msm_fr(c))


def test_hip_path_reproduces_the_reference_held_plonk_round_2_and_3_commitments(kats):
    # provers/plonk/src/prover.rs:787-836 through the HIP path: interpolate_fft, sixteen evaluate_offset_fft calls on the
    # 4n coset, interpolate_offset_fft (all BLS12-381 Fr) and seven KZG commitments, one of them of the zero polynomial
    from lambda_elliptic_curves_amd import msm
    from oracle import oracle as O
    from tests import plonk_kat, util
    fr, oid = O.F_FR381, O.C_BLS12_381_G1
    srs = msm.Srs(msm.BLS12381Curve, util.plonk_test_srs(oid, 7, 2))
    try:
        omega = O.elems_from_mont(fr, [O.get_primitive_root_of_unity(fr, 2)])[0]
        got = plonk_kat.rounds_1_to_3(_HipPlonkOps(srs), omega)
    finally:
        srs.close()
    want = {k + "_1": v for k, v in kats["plonk_round_1_commitments"]["expected"].items()}
    want.update(kats["plonk_round_2_3_commitments"]["expected"])
    for name, v in want.items():
        assert got[name] == (tuple(int(t, 16) for t in v) if v else None), name
