"""Pins the oracle's NTT restatement: reference fixed vectors + the defining sum / Horner evaluation.
Mirrors math/src/fft/cpu/{fft,bit_reversing,roots_of_unity}.rs tests and math/src/fft/polynomial.rs:302-457."""
import random

import numpy as np
import pytest

from oracle import bigint_def as D
from oracle import oracle as O

H = lambda s: int(s, 16)
NTT_FIELDS = [(O.F_STARK252, D.P_STARK252), (O.F_FR381, D.P_FR381), (O.F_BABYBEAR_U64, D.P_BABYBEAR),
              (O.F_BABYBEAR_U32, D.P_BABYBEAR)]


def test_bit_reverse_16(kats):
    # bit_reversing.rs:32-35
    arr = O.elems_to_mont(O.F_BABYBEAR_U32, list(range(16)))
    out = O.elems_from_mont(O.F_BABYBEAR_U32, O.bit_reverse_permute(O.F_BABYBEAR_U32, arr))
    assert out == kats["bit_reverse_16"]["expected"]


def test_bit_reverse_small_and_involution():
    f = O.F_STARK252
    for n in (1, 2, 4, 64):
        arr = O.elems_to_mont(f, list(range(1, n + 1)))
        rev = O.bit_reverse_permute(f, arr)
        bits = n.bit_length() - 1
        exp = [D.bit_reverse(i, bits) + 1 for i in range(n)]
        assert O.elems_from_mont(f, rev) == exp
        assert np.array_equal(O.bit_reverse_permute(f, rev), arr)


def test_roots_of_unity_anchors(kats):
    a = kats["survey_anchors"]
    assert O.elems_from_mont(O.F_STARK252, O.get_primitive_root_of_unity(O.F_STARK252, 2)[None])[0] == H(a["stark252_w4"])
    assert O.elems_from_mont(O.F_STARK252, O.get_primitive_root_of_unity(O.F_STARK252, 16)[None])[0] == H(a["stark252_w_2_16"])
    assert O.elems_from_mont(O.F_FR381, O.get_primitive_root_of_unity(O.F_FR381, 20)[None])[0] == H(a["fr381_w_2_20"])
    assert O.elems_from_mont(O.F_BABYBEAR_U64, O.get_primitive_root_of_unity(O.F_BABYBEAR_U64, 20)[None])[0] == H(a["babybear_w_2_20"])
    assert O.elems_from_mont(O.F_BABYBEAR_U32, np.array([O.get_primitive_root_of_unity(O.F_BABYBEAR_U32, 20)]))[0] == H(a["babybear_w_2_20"])


@pytest.mark.parametrize("field,p", NTT_FIELDS)
def test_root_of_unity_rules(field, p):
    ta, _ = D.FFT_PARAMS[p]
    # order 0 -> one; order > TWO_ADICITY -> RootOfUnityError (traits.rs:86-90)
    one = O.elems_from_mont(field, np.atleast_1d(O.get_primitive_root_of_unity(field, 0))[None] if field != O.F_BABYBEAR_U32
                            else np.array([O.get_primitive_root_of_unity(field, 0)]))
    assert one[0] == 1
    with pytest.raises(O.OracleError) as e:
        O.get_primitive_root_of_unity(field, ta + 1)
    assert e.value.code == O.ERR_ROOT_OF_UNITY
    for order in (1, 2, 5, min(ta, 24)):
        w = O.elems_from_mont(field, np.atleast_2d(O.get_primitive_root_of_unity(field, order)) if field != O.F_BABYBEAR_U32
                              else np.array([O.get_primitive_root_of_unity(field, order)]))[0]
        assert w == D.primitive_root_of_unity(p, order)
        assert pow(w, 1 << order, p) == 1 and pow(w, 1 << (order - 1), p) != 1


def test_get_twiddles_order_error():
    # roots_of_unity.rs:70-72 / test :99-106
    with pytest.raises(O.OracleError) as e:
        O.get_twiddles(O.F_STARK252, 64, O.ROOTS_NATURAL)
    assert e.value.code == O.ERR_ORDER


@pytest.mark.parametrize("field,p", NTT_FIELDS)
def test_twiddle_configs(field, p):
    order = 5
    n = 1 << order
    w = D.primitive_root_of_unity(p, order)
    nat = O.elems_from_mont(field, O.get_twiddles(field, order, O.ROOTS_NATURAL))
    assert nat == [pow(w, i, p) for i in range(n // 2)]
    inv = O.elems_from_mont(field, O.get_twiddles(field, order, O.ROOTS_NATURAL_INV))
    assert inv == [pow(w, -i, p) for i in range(n // 2)]
    br = O.elems_from_mont(field, O.get_twiddles(field, order, O.ROOTS_BITREV))
    assert br == [pow(w, D.bit_reverse(i, order - 1), p) for i in range(n // 2)]
    bri = O.elems_from_mont(field, O.get_twiddles(field, order, O.ROOTS_BITREV_INV))
    assert bri == [pow(w, -D.bit_reverse(i, order - 1), p) for i in range(n // 2)]
    # order 0 -> empty table (roots_of_unity.rs:18-20)
    assert len(O.get_twiddles(field, 0, O.ROOTS_BITREV)) == 0


def test_stark252_ntt4_anchor(kats):
    out = O.evaluate_fft(O.F_STARK252, O.elems_to_mont(O.F_STARK252, [1, 2, 3, 4]))
    assert O.elems_from_mont(O.F_STARK252, out) == [H(x) for x in kats["survey_anchors"]["stark252_ntt4_1234"]]


@pytest.mark.parametrize("field,p", NTT_FIELDS)
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 6, 8])
def test_fft_matches_defining_sum(field, p, log_n):
    # fft.rs:215-260: NR radix-2 + bit-reverse == naive DFT
    rng = random.Random(100 * field + log_n)
    n = 1 << log_n
    c = [rng.randrange(p) for _ in range(n)]
    arr = O.elems_to_mont(field, c)
    tw = O.get_twiddles(field, log_n, O.ROOTS_BITREV)
    got = O.elems_from_mont(field, O.fft(field, arr, tw))
    w = D.primitive_root_of_unity(p, log_n)
    assert got == D.ntt_by_definition(c, p, w)
    # NR output before the permutation is the bit-reversed spectrum
    nr = O.in_place_nr_2radix_fft(field, arr, tw)
    assert O.elems_from_mont(field, O.bit_reverse_permute(field, nr)) == got


def test_fft_rejects_non_power_of_two():
    arr = O.elems_to_mont(O.F_STARK252, [1, 2, 3])
    with pytest.raises(O.OracleError) as e:
        O.fft(O.F_STARK252, arr, O.get_twiddles(O.F_STARK252, 2, O.ROOTS_BITREV))
    assert e.value.code == O.ERR_INPUT_NOT_POW2


@pytest.mark.parametrize("field,p", NTT_FIELDS)
def test_evaluate_fft_length_rule_and_values(field, p):
    # fft/polynomial.rs:30-38 and tests :302-344,:399-439
    rng = random.Random(field)
    for ncoef, blowup, ds in [(5, 1, None), (8, 2, None), (3, 4, 16), (1, 1, None), (7, 1, 4), (6, 8, None)]:
        c = [rng.randrange(1, p) for _ in range(ncoef)]
        got = O.elems_from_mont(field, O.evaluate_fft(field, O.elems_to_mont(field, c), blowup, ds))
        assert got == D.evaluate_fft_def(c, p, blowup, ds)
    # trailing zeros are stripped by Polynomial::new before the length rule: [1,2,0,0] has coeff_len 2
    got = O.evaluate_fft(field, O.elems_to_mont(field, [1, 2, 0, 0]))
    assert got.shape[0] == 2
    # empty / all-zero polynomial -> len zeros, no transform
    z = O.evaluate_fft(field, O.elems_to_mont(field, [0, 0, 0]), 2, 8)
    assert z.shape[0] == 16 and not z.any()


@pytest.mark.parametrize("field,p", NTT_FIELDS)
def test_offset_and_interpolate(field, p):
    rng = random.Random(field + 50)
    for log_n, h in [(3, 3), (5, 7), (6, 2)]:
        n = 1 << log_n
        c = [rng.randrange(p) for _ in range(n - 1)] + [rng.randrange(1, p)]
        arr = O.elems_to_mont(field, c)
        off = O.elems_to_mont(field, [h])[0]
        ev = O.evaluate_fft(field, arr, 1, None, off)
        assert O.elems_from_mont(field, ev) == D.evaluate_fft_def(c, p, 1, None, h)
        back = O.interpolate_fft(field, ev, off)
        assert O.elems_from_mont(field, back) == c
        assert O.elems_from_mont(field, O.interpolate_fft(field, ev)) == D.interpolate_fft_def(
            D.evaluate_fft_def(c, p, 1, None, h), p)
        plain = O.evaluate_fft(field, arr)
        assert np.array_equal(O.interpolate_fft(field, plain), arr)
    # interpolate strips trailing zero coefficients (Polynomial::new): constant polynomial
    ev = O.evaluate_fft(field, O.elems_to_mont(field, [5]), 1, 8)
    assert O.interpolate_fft(field, ev, strip=True).shape[0] == 1
    with pytest.raises(O.OracleError):
        O.interpolate_fft(field, O.elems_to_mont(field, [1, 2, 3]))


def test_ext4_values_over_base_domain():
    # fft/polynomial.rs:442-457 + quartic_babybear.rs:155-166: four interleaved base-field transforms
    p = D.P_BABYBEAR
    rng = random.Random(9)
    n = 16
    cols = [[rng.randrange(p) for _ in range(n)] for _ in range(4)]
    flat = [cols[k][i] for i in range(n) for k in range(4)]
    arr = O.elems_to_mont(O.F_BABYBEAR_EXT4, flat)
    assert arr.shape == (n, 4)
    off = O.elems_to_mont(O.F_BABYBEAR_U64, [2])[0]
    got = O.evaluate_fft(O.F_BABYBEAR_EXT4, arr, 8, 4, off)
    exp_cols = [D.evaluate_fft_def(cols[k], p, 8, 4, 2) for k in range(4)]
    g = O.elems_from_mont(O.F_BABYBEAR_EXT4, got)
    assert g == [exp_cols[k][i] for i in range(len(exp_cols[0])) for k in range(4)]
    ev = O.evaluate_fft(O.F_BABYBEAR_EXT4, arr)
    assert np.array_equal(O.interpolate_fft(O.F_BABYBEAR_EXT4, ev), arr)


@pytest.mark.parametrize("field,p", NTT_FIELDS)
def test_compose_fft_fixed_case(field, p):
    """composition_fft_works (fft/polynomial.rs:346-354): compose_fft(p = 2x, q = x^3) == 2x^3, i.e. evaluate_fft(q), p applied
    pointwise, interpolate_fft (compose_fft, :130-146).  The reference runs it over its u64 test field; the identity holds in
    every field, so it pins evaluate -> interpolate here the same way."""
    q = O.elems_to_mont(field, [0, 0, 0, 1])
    ev = O.elems_from_mont(field, O.evaluate_fft(field, q))
    vals = O.elems_to_mont(field, [2 * v % p for v in ev])            # p(x) = 2x at q's evaluations
    assert O.elems_from_mont(field, O.interpolate_fft(field, vals)) == [0, 0, 0, 2]

