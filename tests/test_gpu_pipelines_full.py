"""The SURVEY 8(f) pipelines at the sizes their timings are quoted on, device-resident, against the oracle
(the compiled-in tile shapes, merkle_top_kernel and the three-pass plans only run at scale):
Merkle commit 4 x 2^22 and 1 x 2^24, Groth16 h coefficients for 2^20 gates feeding the MSM without leaving HBM, and a
whole FRI commit phase from 2^20 coefficients down to a constant."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _nodes_sample(n_nodes, k=64, seed=9):
    rng = np.random.default_rng(seed)
    idx = set(int(x) for x in rng.integers(0, n_nodes, size=k))
    idx.update([0, 1, 2, n_nodes - 1, n_nodes // 2, n_nodes // 2 - 1, 255, 256, 510, 511, 512])   # root, top-of-tree kernel seam, first/last leaf
    return sorted(i for i in idx if i < n_nodes)


@pytest.mark.parametrize("n_cols,log_n", [(4, 22), (1, 24), (16, 20)])
def test_merkle_commit_at_bench_sizes(n_cols, log_n):
    import torch
    from lambda_elliptic_curves_amd import fft, merkle
    n = 1 << log_n
    cols = np.stack([util.rand_elems("stark252", n, 7000 + 31 * c + log_n) for c in range(n_cols)])
    t_cols = torch.from_numpy(cols.view(np.int64)).cuda()
    t_nodes = torch.empty(((2 * n - 1) * 4,), dtype=torch.int64, device="cuda")
    root = merkle.commit_columns_device(fft.Stark252PrimeField, t_cols, n_cols, log_n, t_nodes)
    exp = O.merkle_commit_columns(cols, True, threads=util.host_threads())
    assert root == exp[0].tobytes()
    got = t_nodes.cpu().numpy().view(np.uint8).reshape(2 * n - 1, 32)
    for i in _nodes_sample(2 * n - 1):
        assert np.array_equal(got[i], exp[i]), f"node {i}"
    # every level once more through a checksum of the whole node array (a checksum of checksums, cheap at any size)
    assert hashlib.sha256(got.tobytes()).digest() == hashlib.sha256(exp.tobytes()).digest()


def test_config4_lde_then_commit_stays_on_device():
    """BASELINE config 4 end to end on one GPU: the STARK LDE of four BabyBear columns (2^22 coefficients -> 2^24 evaluations on
    the coset 3<w>, lw_hip_ntt_lde_device) and the Merkle commitment of the result (prover.rs:208-244), nothing leaving HBM
    but the root; against the oracle's padded transforms and its tree."""
    import torch
    from lambda_elliptic_curves_amd import fft, merkle
    fld, oid = util.field_pairs()["babybear_u32"]
    B, log_m, blow = 4, 22, 2
    n, N = 1 << log_m, 1 << (log_m + blow)
    coeffs = [util.rand_elems("babybear_u32", n, 6100 + c) for c in range(B)]
    off = util.offset_elem("babybear_u32", 3)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        futs = [ex.submit(O.evaluate_fft, oid, c, 1 << blow, n, off) for c in coeffs]
        t_c = torch.from_numpy(np.concatenate(coeffs).view(np.int32)).cuda()
        t_lde = torch.empty(B * N, dtype=torch.int32, device="cuda")
        t_nodes = torch.empty(((2 * N - 1) * 4,), dtype=torch.int64, device="cuda")
        fft.lde_device(fld, t_c, log_m, t_lde, log_m + blow, batch=B, offset=off)
        root = merkle.commit_columns_layout_device(fld, t_lde, B, log_m + blow, t_nodes)
        exp_cols = np.stack([np.asarray(f.result()).reshape(-1) for f in futs])
    assert np.array_equal(t_lde.cpu().numpy().view(np.uint32).reshape(B, N), exp_cols)
    exp = O.merkle_commit_columns_babybear(exp_cols, True, threads=util.host_threads())
    assert root == exp[0].tobytes()
    got = t_nodes.cpu().numpy().view(np.uint8).reshape(2 * N - 1, 32)
    for i in _nodes_sample(2 * N - 1):
        assert np.array_equal(got[i], exp[i]), f"node {i}"
    assert hashlib.sha256(got.tobytes()).digest() == hashlib.sha256(exp.tobytes()).digest()


def test_groth16_h_stays_on_device_and_feeds_the_msm():
    """Prover::prove (provers/groth16/src/prover.rs:68-72,97-101): h = calculate_h_coefficients(w), then
    msm(h.representative(), z_powers_of_tau_g1[..h.len()]).  2^20 gates; h never leaves HBM between the two."""
    import torch
    from lambda_elliptic_curves_amd import groth16, msm
    gates = 1 << 20
    l, r, o = (util.rand_elems("fr381", gates, s) for s in (8101, 8102, 8103))
    t_l, t_r, t_o = (torch.from_numpy(x.view(np.int64)).cuda() for x in (l, r, o))
    t_h, clen = groth16.calculate_h_coefficients_device(t_l, t_r, t_o, gates, gates, want_len=True)
    exp = O.groth16_h_coefficients(l, r, o, gates, strip=False)
    exp_len = gates * 2
    while exp_len and not exp[exp_len - 1].any():
        exp_len -= 1
    assert clen == exp_len
    assert np.array_equal(t_h.cpu().numpy().view(np.uint64), exp)
    # the MSM over the device-resident h (scalars in Montgomery form, as stored) against the oracle on a prefix the CPU
    # finishes in seconds; the SRS is a run of distinct points like a powers-of-tau vector
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    m = 1 << 16
    _, points = util.msm_case(oid, m, 8200, threads=util.host_threads())
    srs = msm.Srs(crv, points)
    got = srs.msm_fr_device(t_h, m)
    from oracle import bigint_def as D
    reps = O.ints_to_array([v for v in O.elems_from_mont(O.F_FR381, exp[:m])], 4)
    want = O.parallel_msm_with(oid, reps, points, 13, util.host_threads())
    assert O.point_to_affine_ints(oid, got) == O.point_to_affine_ints(oid, want)
    srs.close()
    # shorter coefficient vectors than gates are zero padded on the device
    k = gates // 2 + 3
    t_h2 = groth16.calculate_h_coefficients_device(t_l, t_r, t_o, k, gates)
    assert np.array_equal(t_h2.cpu().numpy().view(np.uint64), O.groth16_h_coefficients(l[:k], r[:k], o[:k], gates, strip=False))


def _transcript_challenge(state):
    """A stand-in for the prover's transcript (the transcript is the caller's, fri/mod.rs:45,56): a hash chain giving
    canonical challenges below 2^248."""
    return int.from_bytes(hashlib.sha256(state).digest()[:31], "big")


@pytest.mark.parametrize("log_coeffs,blowup_log", [(20, 1), (12, 3)])
def test_fri_commit_phase_stays_on_device(log_coeffs, blowup_log):
    """commit_phase (provers/stark/src/fri/mod.rs:22-75) from 2^log_coeffs coefficients down to the constant: every layer's
    polynomial, evaluation vector and tree stay in HBM; per layer only zeta goes in and the root comes out.  Each layer is
    checked against the oracle's 2*fold_polynomial + new_fri_layer composition, the last value against the final fold."""
    import torch
    from lambda_elliptic_curves_amd import fft, merkle
    from oracle import bigint_def as D
    f, p = O.F_STARK252, D.P_STARK252
    n = 1 << log_coeffs
    domain = n << blowup_log
    number_layers = log_coeffs + 1            # fold until a constant is left (one fold per layer)
    a = util.rand_elems("stark252", n, 9100 + log_coeffs)
    a[-1, -1] |= np.uint64(1)
    t_p0 = torch.from_numpy(a.view(np.int64)).cuda()
    h = 3
    state = {"s": b"fri-test", "zetas": []}

    def sample_zeta():
        z = _transcript_challenge(state["s"])
        state["zetas"].append(z)
        state["s"] = hashlib.sha256(state["s"] + b"z").digest()
        return O.elems_to_mont(f, [z])[0]

    def append_root(root):
        state["s"] = hashlib.sha256(state["s"] + root).digest()

    def offset_sq(k):
        return O.elems_to_mont(f, [pow(h, 1 << k, p)])[0]

    t_last, layers = merkle.fri_commit_phase_device(fft.Stark252PrimeField, number_layers, t_p0, n, sample_zeta, append_root,
                                                    offset_sq, domain)
    assert len(layers) == number_layers - 1
    # the oracle walks the same chain with the challenges the device run drew (they depend on the device's roots: any
    # wrong root changes every later challenge and the comparison fails at that layer)
    poly = a
    dom = domain
    check_full = set(range(3)) | set(range(len(layers) - 3, len(layers)))   # big layers and the small-tile tail in full
    for k, (t_ev, t_nodes, root, dsize) in enumerate(layers, start=1):
        zeta = O.elems_to_mont(f, [state["zetas"][k - 1]])[0]
        poly = O.fri_fold_twice(f, poly, zeta)
        dom //= 2
        assert dsize == dom
        ev = O.bit_reverse_permute(f, O.evaluate_fft(f, poly, 1, dom, offset_sq(k)))
        leaves = ev.reshape(dom // 2, 2, 4)
        exp_nodes = O.merkle_commit_columns(np.ascontiguousarray(leaves.transpose(1, 0, 2)), bit_reverse=False,
                                            threads=util.host_threads())
        assert root == exp_nodes[0].tobytes(), f"layer {k} root"
        if (k - 1) in check_full or k % 4 == 0:
            assert np.array_equal(t_ev.cpu().numpy().view(np.uint64), ev), f"layer {k} evaluation"
            assert np.array_equal(t_nodes.cpu().numpy().view(np.uint8).reshape(dom - 1, 32), exp_nodes), f"layer {k} tree"
    last = O.fri_fold_twice(f, poly, O.elems_to_mont(f, [state["zetas"][-1]])[0], strip=False)
    assert last.shape[0] == 1
    assert np.array_equal(t_last.cpu().numpy().view(np.uint64)[0], last[0])
