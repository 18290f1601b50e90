"""QuadraticArithmeticProgram::calculate_h_coefficients as one device pipeline, bit-exact against the
same composition of oracle calls (provers/groth16/src/qap.rs:15-39)."""
import numpy as np
import pytest

from oracle import bigint_def as D
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


oracle_h = util.groth16_h_by_composition


@pytest.mark.parametrize("gates", [1, 2, 8, 64, 512])
def test_h_coefficients_match_reference_composition(gates):
    from lambda_elliptic_curves_amd import groth16
    l = util.rand_elems("fr381", gates, 70 + gates)
    r = util.rand_elems("fr381", gates, 71 + gates)
    o = util.rand_elems("fr381", gates, 72 + gates)
    got = groth16.calculate_h_coefficients(l, r, o, gates)
    exp = oracle_h(l, r, o, gates)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    # fewer coefficients than gates, and non power of two gate count is rejected
    got2 = groth16.calculate_h_coefficients(l[: max(1, gates // 2)], r[: max(1, gates // 2)], o[: max(1, gates // 2)], gates)
    assert np.array_equal(got2, oracle_h(l[: max(1, gates // 2)], r[: max(1, gates // 2)], o[: max(1, gates // 2)], gates))


def test_h_is_a_polynomial_quotient():
    # if o = l*r on the gate domain (a satisfied R1CS), (l*r - o) is divisible by t = x^g - 1 and deg h < g
    from lambda_elliptic_curves_amd import errors, fft, groth16
    f = O.F_FR381
    p = D.P_FR381
    g = 32
    rng = np.random.default_rng(4)
    lv = [int.from_bytes(rng.bytes(31), "big") for _ in range(g)]
    rv = [int.from_bytes(rng.bytes(31), "big") for _ in range(g)]
    ov = [a * b % p for a, b in zip(lv, rv)]
    coeffs = [O.interpolate_fft(f, O.elems_to_mont(f, v)) for v in (lv, rv, ov)]   # interpolate over the gate domain
    h = groth16.calculate_h_coefficients(*coeffs, g)
    assert h.shape[0] <= g - 1
    with pytest.raises(errors.InputError):
        groth16.calculate_h_coefficients(coeffs[0][:3], coeffs[1][:3], coeffs[2][:3], 3)
