"""GPU parity for the Keccak-256 Merkle commitment of LDE columns against the oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


# 2^10 .. 2^16: levels fused into the leaf kernel and four per launch (1 - 4 of them); 2^17: the tiled bit-reversed gather
@pytest.mark.parametrize("n_cols,log_n", [(1, 0), (1, 1), (1, 6), (2, 3), (4, 10), (5, 7), (17, 5), (34, 4), (3, 14), (1, 9), (2, 11), (1, 12),
                                          (1, 13), (2, 16), (2, 17), (5, 18)])
def test_commit_matches_oracle(n_cols, log_n):
    from lambda_elliptic_curves_amd import fft, merkle
    n = 1 << log_n
    cols = np.stack([util.rand_elems("stark252", n, 900 + 13 * c + log_n) for c in range(n_cols)])
    exp = O.merkle_commit_columns(cols, bit_reverse=True)
    root, nodes = merkle.commit_columns(fft.Stark252PrimeField, cols, return_nodes=True)
    assert np.array_equal(nodes, exp)
    assert root == exp[0].tobytes()
    assert merkle.commit_columns(fft.Stark252PrimeField, cols, bit_reverse=False) == O.merkle_commit_columns(cols, False)[0].tobytes()


def test_lde_then_commit_stays_on_device():
    # the prover's round-1 shape: LDE of the trace columns on a coset, then commit (prover.rs:208-244), all device-resident
    import torch
    from lambda_elliptic_curves_amd import fft, merkle
    fld, oid = util.field_pairs()["stark252"]
    n_cols, log_m, blow = 3, 10, 2
    n, N = 1 << log_m, 1 << (log_m + blow)
    coeffs = np.concatenate([util.rand_elems("stark252", n, 5 + c) for c in range(n_cols)])
    off = util.offset_elem("stark252", 3)
    t_c = torch.from_numpy(coeffs.view(np.int64)).cuda()
    t_lde = torch.empty((n_cols * N, 4), dtype=torch.int64, device="cuda")
    t_nodes = torch.empty(((2 * N - 1) * 4,), dtype=torch.int64, device="cuda")
    fft.lde_device(fld, t_c, log_m, t_lde, log_m + blow, batch=n_cols, offset=off)
    root = merkle.commit_columns_device(fld, t_lde, n_cols, log_m + blow, t_nodes)
    exp_cols = np.stack([O.evaluate_fft(oid, coeffs[c * n:(c + 1) * n], 1 << blow, n, off) for c in range(n_cols)])
    assert root == O.merkle_commit_columns(exp_cols, True)[0].tobytes()


@pytest.mark.parametrize("n_coeffs,domain", [(8, 16), (7, 8), (64, 256), (33, 64), (2, 2), (1024, 4096)])
def test_fri_layer_matches_reference_composition(n_coeffs, domain):
    # commit_phase loop body (provers/stark/src/fri/mod.rs:44-58) = 2*fold_polynomial + new_fri_layer (:115-141)
    from lambda_elliptic_curves_amd import fft, merkle
    from oracle import bigint_def as D
    f, p = O.F_STARK252, D.P_STARK252
    a = util.rand_elems("stark252", n_coeffs, 50 + n_coeffs)
    a[-1, -1] |= np.uint64(1)
    zeta_c, off_c = 0x1234567890abcdef1234567, 9      # canonical challenge and (already squared) coset offset
    zeta, off = O.elems_to_mont(f, [zeta_c])[0], O.elems_to_mont(f, [off_c])[0]
    c = O.elems_from_mont(f, a)
    folded = [(2 * (c[2 * i] + (zeta_c * c[2 * i + 1] if 2 * i + 1 < n_coeffs else 0))) % p for i in range((n_coeffs + 1) // 2)]
    while folded and folded[-1] == 0:
        folded.pop()
    exp_poly = O.elems_to_mont(f, folded)
    exp_ev = O.bit_reverse_permute(f, O.evaluate_fft(f, exp_poly, 1, domain, off))
    leaves = exp_ev.reshape(domain // 2, 2, 4)            # chunks(2) of the permuted evaluation
    exp_nodes = O.merkle_commit_columns(np.ascontiguousarray(leaves.transpose(1, 0, 2)), bit_reverse=False)
    poly, ev, root, nodes = merkle.fri_layer(fft.Stark252PrimeField, a, zeta, off, domain, return_nodes=True)
    assert np.array_equal(poly, exp_poly)
    assert np.array_equal(ev, exp_ev)
    assert np.array_equal(nodes, exp_nodes) and root == exp_nodes[0].tobytes()


@pytest.mark.parametrize("name", ["babybear_u32", "babybear_u64"])
@pytest.mark.parametrize("n_cols,log_n", [(1, 0), (1, 3), (3, 5), (4, 10), (33, 6), (35, 4), (17, 9), (3, 13), (3, 17), (4, 18)])
def test_babybear_commit_matches_oracle(name, n_cols, log_n):
    """The commitment over BabyBear columns (BASELINE config 4's field): leaves hash the raw words big-endian, 4 or 8 bytes per
    element; odd u32 column counts leave half a Keccak lane before the padding, 35 columns span two blocks."""
    import torch
    from lambda_elliptic_curves_amd import merkle
    fld, _ = util.field_pairs()[name]
    n = 1 << log_n
    cols = np.stack([util.rand_elems(name, n, 1900 + 7 * c + log_n) for c in range(n_cols)])
    t_cols = torch.from_numpy(cols.view(np.int32 if cols.dtype == np.uint32 else np.int64)).cuda()
    t_nodes = torch.empty(((2 * n - 1) * 4,), dtype=torch.int64, device="cuda")
    for br in (True, False):
        root = merkle.commit_columns_layout_device(fld, t_cols, n_cols, log_n, t_nodes, bit_reverse=br)
        exp = O.merkle_commit_columns_babybear(cols, br)
        assert np.array_equal(t_nodes.cpu().numpy().view(np.uint8).reshape(2 * n - 1, 32), exp)
        assert root == exp[0].tobytes()


def test_commit_layout_entry_rejects_the_extension_layout_and_serves_256_bit_fields():
    import torch
    from lambda_elliptic_curves_amd import errors, fft, merkle
    cols = np.stack([util.rand_elems("stark252", 64, 5 + c) for c in range(2)])
    t_cols = torch.from_numpy(cols.view(np.int64)).cuda()
    t_nodes = torch.empty((127 * 4,), dtype=torch.int64, device="cuda")
    assert merkle.commit_columns_layout_device(fft.Stark252PrimeField, t_cols, 2, 6, t_nodes) == O.merkle_commit_columns(cols, True)[0].tobytes()
    with pytest.raises(errors.HipError):
        merkle.commit_columns_layout_device(fft.Degree4BabyBearExtensionField, t_cols, 1, 6, t_nodes)
