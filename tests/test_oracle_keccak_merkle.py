"""Pins the oracle's Keccak-256 (the reference's third-party `sha3` 0.10 dependency, restated from the published
algorithm) and its restatement of the reference's Merkle tree layout."""
import hashlib

import numpy as np

from oracle import oracle as O
from tests import util


def test_keccak256_known_answers():
    # Keccak team known answers (original padding, not SHA-3's)
    assert O.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert O.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # multi-block and rate-boundary lengths: differs from SHA3-256 only by the domain byte, so the permutation and the
    # absorb loop are cross-checked against hashlib by flipping that byte on a message that fills the block exactly
    for ln in (1, 55, 135, 136, 137, 271, 272, 1000):
        msg = bytes((7 * i + 3) & 0xFF for i in range(ln))
        assert O.keccak256(msg) != hashlib.sha3_256(msg).digest()
        assert len(O.keccak256(msg)) == 32
    # Ethereum's well-known empty-list RLP hash: keccak256(0xc0)
    assert O.keccak256(b"\xc0").hex() == "1dcc4de8dec75d7aab85b567b6ccd41ad312451b948a7413f0a142fd40d49347"
    # keccak256 of the ASCII function signature used all over Ethereum: "transfer(address,uint256)" -> a9059cbb...
    assert O.keccak256(b"transfer(address,uint256)").hex().startswith("a9059cbb")


def test_merkle_layout_matches_reference_rules():
    # merkle.rs:31-56 + utils.rs:44-72: nodes = inner (root first) then leaves; parent(i) = H(nodes[2i+1] || nodes[2i+2])
    n_cols, n = 3, 16
    cols = np.stack([util.rand_elems("stark252", n, 40 + c) for c in range(n_cols)])
    nodes = O.merkle_commit_columns(cols, bit_reverse=True)
    assert nodes.shape == (2 * n - 1, 32)
    for i in range(n - 1):
        assert O.keccak256(nodes[2 * i + 1].tobytes() + nodes[2 * i + 2].tobytes()) == nodes[i].tobytes()
    # leaf i hashes the big-endian raw Montgomery limbs of row bitrev(i) (field_element_vector.rs:41-49)
    for i in (0, 1, 5, 15):
        src = int(format(i, "04b")[::-1], 2)
        row = b"".join(cols[c, src].astype(">u8").tobytes() for c in range(n_cols))
        assert O.keccak256(row) == nodes[n - 1 + i].tobytes()
    # without the permutation the leaves follow natural order; a single leaf is its own root
    nat = O.merkle_commit_columns(cols, bit_reverse=False)
    assert nat[n - 1 + 3].tobytes() == O.keccak256(b"".join(cols[c, 3].astype(">u8").tobytes() for c in range(n_cols)))
    one = O.merkle_commit_columns(cols[:, :1], bit_reverse=True)
    assert one.shape == (1, 32) and one[0].tobytes() == O.keccak256(b"".join(cols[c, 0].astype(">u8").tobytes() for c in range(n_cols)))


def test_oracle_reproduces_the_reference_held_stone_compat_trace_commitments(kats):
    # The only vectors in the reference that pin NTT output THROUGH the Keccak Merkle tree (prover.rs:1273-1281,1659-1667):
    # interpolate_fft -> evaluate_offset_fft (blow-up, coset 3) -> bit-reverse -> rows -> BatchedMerkleTree root.
    import numpy as np
    from oracle import oracle as O
    from tests import util
    oid = O.F_STARK252
    for case in kats["stone_compat_trace_commitments"]["cases"]:
        n, blow = case["trace_length"], case["blowup_factor"]
        off = O.elems_to_mont(oid, [case["coset_offset"]])[0]
        cols = []
        for col in util.stone_compat_trace_columns(case["initial"], n):
            poly = O.interpolate_fft(oid, O.elems_to_mont(oid, col))
            cols.append(O.evaluate_fft(oid, poly, blow, n, off))
        nodes = O.merkle_commit_columns(np.stack(cols), bit_reverse=True)
        assert bytes(nodes[0]).hex() == case["root"], case["name"]
        have = {bytes(x).hex() for x in nodes}
        for h in case["auth_path_nodes"]:
            assert h in have, (case["name"], h)


def test_threaded_merkle_equals_sequential():
    cols = np.stack([util.rand_elems("stark252", 1 << 13, 77 + c) for c in range(3)])
    assert np.array_equal(O.merkle_commit_columns(cols, True, threads=5), O.merkle_commit_columns(cols, True))
    assert np.array_equal(O.merkle_commit_columns(cols[:1], False, threads=3), O.merkle_commit_columns(cols[:1], False))


def test_fri_fold_restatement_matches_bigint_definition():
    # 2 * fold_polynomial(p, zeta) (fri/mod.rs:49, fri_functions.rs:7-30): p'_i = 2 (c_2i + zeta c_2i+1), stripped
    from oracle import bigint_def as D
    f, p = O.F_STARK252, D.P_STARK252
    for n in (1, 2, 7, 8, 33):
        a = util.rand_elems("stark252", n, 300 + n)
        zc = 0x1234567890abcdef1234567 + n
        c = O.elems_from_mont(f, a)
        exp = [(2 * (c[2 * i] + (zc * c[2 * i + 1] if 2 * i + 1 < n else 0))) % p for i in range((n + 1) // 2)]
        while exp and exp[-1] == 0:
            exp.pop()
        got = O.fri_fold_twice(f, a, O.elems_to_mont(f, [zc])[0])
        assert O.elems_from_mont(f, got) == exp
    # cancellation: c1 = -c0 / zeta makes the folded coefficient zero -> stripped
    zc = 5
    c0 = 1234567
    c1 = (-c0 * pow(zc, -1, p)) % p
    a = O.elems_to_mont(f, [9, 1, c0, c1])
    assert O.fri_fold_twice(f, a, O.elems_to_mont(f, [zc])[0]).shape[0] == 1


def test_groth16_h_restatement_matches_composition():
    for gates in (1, 2, 16, 128):
        l, r, o = (util.rand_elems("fr381", gates, s + gates) for s in (70, 71, 72))
        assert np.array_equal(O.groth16_h_coefficients(l, r, o, gates), util.groth16_h_by_composition(l, r, o, gates))
    k = 5
    l, r, o = (util.rand_elems("fr381", k, s) for s in (1, 2, 3))
    assert np.array_equal(O.groth16_h_coefficients(l, r, o, 16), util.groth16_h_by_composition(l, r, o, 16))


def test_babybear_leaves_hash_the_raw_words_big_endian():
    # U32MontgomeryBackendPrimeField::as_bytes = value().to_be_bytes() (u32_montgomery_backend_prime_field.rs:258-262); the
    # u64-limb BabyBear hashes its one limb big-endian (montgomery_backed_prime_fields.rs:367-373): leaves by definition
    rng = np.random.default_rng(1)
    n = 8
    rev = lambda i: int("{:03b}".format(i)[::-1], 2)
    for dt in (np.uint32, np.uint64):
        for n_cols in (1, 3, 4, 35):          # 35 u32 columns = 140 bytes: two Keccak blocks, odd word count
            cols = rng.integers(0, util.P_BABYBEAR, size=(n_cols, n), dtype=dt)
            nodes = O.merkle_commit_columns_babybear(cols, True, threads=1)
            for i in range(n):
                row = b"".join(int(cols[c, rev(i)]).to_bytes(cols.dtype.itemsize, "big") for c in range(n_cols))
                assert nodes[n - 1 + i].tobytes() == O.keccak256(row), (dt, n_cols, i)
            for k in range(n - 1):
                assert nodes[k].tobytes() == O.keccak256(nodes[2 * k + 1].tobytes() + nodes[2 * k + 2].tobytes())
            assert np.array_equal(nodes, O.merkle_commit_columns_babybear(cols, True, threads=4))
