"""Pins the oracle's limb/Montgomery arithmetic against (1) every fixed vector the reference's tests
hold for it and (2) Python big-int `% p`.  Mirrors math/src/unsigned_integer/montgomery.rs:216-268 and
math/src/field/fields/montgomery_backed_prime_fields.rs tests."""
import random

import pytest

from oracle import bigint_def as D
from oracle import oracle as O

H = lambda s: int(s, 16)

FIELDS = [(O.F_STARK252, D.P_STARK252, 4), (O.F_FR381, D.P_FR381, 4), (O.F_BABYBEAR_U64, D.P_BABYBEAR, 1),
          (O.F_FP381, D.P_FP381, 6), (O.F_FP254, D.P_FP254, 4), (O.F_FR254, D.P_FR254, 4)]


def test_cios_reference_vectors(kats):
    for v in kats["cios_u384"]:
        got = O.mont_cios(H(v["x"]), H(v["y"]), H(v["m"]), int(v["mu_dec"]), 6)
        assert got == H(v["c"]), v["cite"]


def test_cios_vs_cios_spare_vs_bigint():
    # montgomery.rs:216-239 proptest, restated with a fixed seed
    rng = random.Random(1)
    m = H("cdb061954fdd36e5176f50dbdcfd349570a29ce1")
    mu = 16085280245840369887
    rinv = pow(1 << 384, -1, m)
    for _ in range(200):
        a, b = rng.getrandbits(384), rng.getrandbits(384)
        r1 = O.mont_cios(a, b, m, mu, 6)
        r2 = O.mont_cios(a, b, m, mu, 6, spare=True)
        assert r1 == r2
        assert r1 % m == a * b * rinv % m


def test_params_p23(kats):
    v = kats["montgomery_params_p23_u384"]
    p = O.derive_params(H(v["modulus"]), v["limbs"])
    assert p["r2"] == H(v["r2"])
    assert p["mu"] == int(v["mu_dec"])
    q = H(v["modulus"])
    x = O.fe_op_mod(q, 6, O.OP_TO_MONT, 770)
    assert x == H(v["from_u64_770"])
    assert O.fe_op_mod(q, 6, O.OP_FROM_MONT, x) == H(v["representative_of_from_u64_770"])


def test_u256_field_vectors(kats):
    for v in kats["u256_field_ops"]:
        q = 0
        for limb in v["modulus_limbs_dec"]:
            q = (q << 64) | int(limb)
        mx = O.fe_op_mod(q, 4, O.OP_TO_MONT, H(v["x"]))
        my = O.fe_op_mod(q, 4, O.OP_TO_MONT, H(v["y"]))
        assert O.fe_op_mod(q, 4, O.OP_ADD, mx, my) == O.fe_op_mod(q, 4, O.OP_TO_MONT, H(v["sum"])), v["cite"]
        assert O.fe_op_mod(q, 4, O.OP_MUL, mx, my) == O.fe_op_mod(q, 4, O.OP_TO_MONT, H(v["product"])), v["cite"]


def test_field_params_match_survey_anchors(kats):
    a = kats["survey_anchors"]
    p = O.field_params(O.F_STARK252)
    assert p["q"] == D.P_STARK252 and p["one"] == H(a["stark252_one"]) and p["r2"] == H(a["stark252_r2"])
    assert p["mu"] == H(a["stark252_mu"])
    assert D.P_STARK252.bit_length() == kats["stark252_bit_size"]["bits"]
    assert O.field_params(O.F_FR381)["mu"] == H(a["fr381_mu"])
    assert O.field_params(O.F_FP381)["mu"] == H(a["fp381_mu"])
    assert O.field_params(O.F_FP254)["mu"] == H(a["fp254_mu"])
    assert O.field_params(O.F_FR254)["mu"] == H(a["fr254_mu"])
    b = O.field_params(O.F_BABYBEAR_U64)
    assert (b["mu"], b["one"], b["r2"]) == (H(a["babybear64_mu"]), H(a["babybear64_one"]), H(a["babybear64_r2"]))
    c = O.field_params(O.F_BABYBEAR_U32)
    assert (c["mu"], c["one"], c["r2"]) == (H(a["babybear32_mu"]), H(a["babybear32_one"]), H(a["babybear32_r2"]))


@pytest.mark.parametrize("field,p,words", FIELDS)
def test_field_ops_vs_bigint(field, p, words):
    rng = random.Random(field + 7)
    R = 1 << (64 * words)
    params = O.field_params(field)
    assert params["one"] == R % p and params["r2"] == R * R % p
    assert (params["mu"] * p) % (1 << 64) == (1 << 64) - 1
    edge = [0, 1, p - 1, p - 2, 2]
    vals = edge + [rng.randrange(p) for _ in range(40)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        ma, mb = O.to_mont(field, a), O.to_mont(field, b)
        assert ma == a * R % p
        assert O.from_mont(field, ma) == a
        assert O.from_mont(field, O.fe_op(field, O.OP_ADD, ma, mb)) == (a + b) % p
        assert O.from_mont(field, O.fe_op(field, O.OP_SUB, ma, mb)) == (a - b) % p
        assert O.from_mont(field, O.fe_op(field, O.OP_MUL, ma, mb)) == a * b % p
        assert O.from_mont(field, O.fe_op(field, O.OP_NEG, ma)) == (-a) % p
        if a:
            assert O.from_mont(field, O.fe_op(field, O.OP_INV, ma)) == pow(a, -1, p)
    with pytest.raises(O.OracleError):
        O.fe_op(field, O.OP_INV, 0)


def test_babybear_u32_ops_vs_bigint():
    p = D.P_BABYBEAR
    f = O.F_BABYBEAR_U32
    rng = random.Random(3)
    R = 1 << 32
    vals = [0, 1, p - 1, 2] + [rng.randrange(p) for _ in range(60)]
    for i, a in enumerate(vals):
        b = vals[(i * 5 + 1) % len(vals)]
        ma, mb = O.to_mont(f, a), O.to_mont(f, b)
        assert ma == a * R % p and O.from_mont(f, ma) == a
        assert O.from_mont(f, O.fe_op(f, O.OP_ADD, ma, mb)) == (a + b) % p
        assert O.from_mont(f, O.fe_op(f, O.OP_SUB, ma, mb)) == (a - b) % p
        assert O.from_mont(f, O.fe_op(f, O.OP_MUL, ma, mb)) == a * b % p
        if a:
            assert O.from_mont(f, O.fe_op(f, O.OP_INV, ma)) == pow(a, -1, p)
