"""GPU == CPU on the same input at the sizes BASELINE.json quotes (configs 2, 3 and the per-GPU sizes of 4, 5):
the HIP path through the C ABI against the oracle, directly — not HIP against HIP.  Mirrors the reference's own
pattern for its GPU backend, assert_eq!(gpu(x), cpu(x)) (math/src/fft/gpu/cuda/ops.rs:109-136) and
Pippenger == reference sum (math/src/msm/pippenger.rs:204-233).

MSM inputs are n DISTINCT points (util.msm_case: P_i = [s0 + i*delta]G, Z != 1), nothing is tiled.  The oracle side
runs on the box's host cores: parallel_msm_with (window-parallel restatement of pippenger.rs:109-161) for the MSM,
single-threaded evaluate_fft / interpolate_fft per transform for the NTT, several transforms at a time."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _aff(oid, p):
    return O.point_to_affine_ints(oid, p)


def _msm_vs_oracle(name, L, seed):
    import torch
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()[name]
    n = 1 << L
    thr = util.host_threads()
    scalars, points = util.msm_case(oid, n, seed, threads=thr)
    ts = torch.from_numpy(scalars.view(np.int64)).cuda()
    tp = torch.from_numpy(points.view(np.int64)).cuda()
    got = msm.msm_device(crv, ts, tp, n)
    del ts, tp
    torch.cuda.empty_cache()
    exp = O.parallel_msm_with(oid, scalars, points, max(2, O.optimum_window_size(n)), thr)
    assert _aff(oid, got) == _aff(oid, exp)
    return scalars, points, got


@pytest.mark.parametrize("L", [20, 22, 24])
def test_bls12_381_g1_msm_matches_oracle_on_distinct_points(L):
    # BASELINE config 3 (2^20 - 2^24).  2^22 and up take the normalise + mixed/batched-affine path; 2^24 sits on the
    # 32-bit packed sort-item boundary.
    _msm_vs_oracle("bls12_381_g1", L, 2400 + L)


@pytest.mark.parametrize("name,L", [("bn254_g1", 20), ("bn254_g1", 22), ("bn254_g1", 23), ("bn254_g2", 18), ("bn254_g2", 22),
                                    ("bn254_g2", 23), ("bls12_381_g2", 19), ("bls12_381_g2", 20)])
def test_other_groups_msm_matches_oracle_at_sharded_per_gpu_size(name, L):
    # BASELINE config 5: BN254 G1 + G2 2^26 over 8 GPUs = 2^23 points per GPU
    _msm_vs_oracle(name, L, 2500 + L)


def test_bls12_381_g1_srs_path_matches_oracle_2_22():
    # the cached affine SRS (lw_hip_srs_*) against the oracle on distinct points, prefix call shape
    import torch
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    n = 1 << 22
    thr = util.host_threads()
    scalars, points = util.msm_case(oid, n, 77, threads=thr)
    srs = msm.Srs(crv, t_points=torch.from_numpy(points.view(np.int64)).cuda(), n=n)
    m = n - 12345
    got = srs.msm_device(torch.from_numpy(scalars[:m].view(np.int64)).cuda(), m)
    srs.close()
    exp = O.parallel_msm_with(oid, scalars[:m], points[:m], max(2, O.optimum_window_size(m)), thr)
    assert _aff(oid, got) == _aff(oid, exp)


@pytest.mark.parametrize("L", [24])
def test_256_bit_ntt_and_intt_2_24_match_oracle(L):
    # BASELINE config 2 / the bench's own size: Stark252 and BLS12-381 Fr, forward and inverse, byte equality
    import torch
    from lambda_elliptic_curves_amd import fft
    n = 1 << L
    cases = {}
    for name in ("stark252", "fr381"):
        fld, oid = util.field_pairs()[name]
        a = util.rand_elems(name, n, 0x5EED0000 + L)
        cases[name] = (fld, oid, a)
    with ThreadPoolExecutor(4) as ex:   # the oracle is single-threaded per transform, as the reference is
        futs = {}
        for name, (fld, oid, a) in cases.items():
            futs[name, "fwd"] = ex.submit(O.evaluate_fft, oid, a)
            futs[name, "inv"] = ex.submit(O.interpolate_fft, oid, a)
        got = {}
        for name, (fld, oid, a) in cases.items():
            t_in = torch.from_numpy(a.view(np.int64)).cuda()
            t_out = torch.empty_like(t_in)
            fft.ntt_device(fld, t_in, t_out, L)
            torch.cuda.synchronize()
            got[name, "fwd"] = t_out.cpu().numpy().view(np.uint64)
            fft.ntt_device(fld, t_in, t_out, L, inverse=True)
            torch.cuda.synchronize()
            got[name, "inv"] = t_out.cpu().numpy().view(np.uint64)
            del t_in, t_out
        for key, f in futs.items():
            assert np.array_equal(got[key], f.result()), key


def test_stark252_ntt_and_intt_2_26_match_oracle():
    """The upper end of BASELINE config 2 (2 GiB per buffer, the four-pass plan 6 + 6 + 6 + 8): forward and inverse byte for byte
    against the oracle's single-threaded evaluate_fft / interpolate_fft (about 40 s each, run side by side on two host
    threads while the GPU results are produced)."""
    import torch
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()["stark252"]
    L = 26
    a = util.rand_elems("stark252", 1 << L, 0x5EED0000 + L)
    with ThreadPoolExecutor(2) as ex:
        f_fwd = ex.submit(O.evaluate_fft, oid, a)
        f_inv = ex.submit(O.interpolate_fft, oid, a)
        t_in = torch.from_numpy(a.view(np.int64)).cuda()
        t_out = torch.empty_like(t_in)
        fft.ntt_device(fld, t_in, t_out, L)
        torch.cuda.synchronize()
        fwd = t_out.cpu().numpy().view(np.uint64)
        fft.ntt_device(fld, t_in, t_out, L, inverse=True)
        torch.cuda.synchronize()
        inv = t_out.cpu().numpy().view(np.uint64)
        del t_in, t_out
        assert np.array_equal(fwd, f_fwd.result()), "forward 2^26"
        assert np.array_equal(inv, f_inv.result()), "inverse 2^26"


@pytest.mark.parametrize("name", ["babybear_u32", "babybear_u64"])
def test_babybear_4_columns_2_24_match_oracle(name):
    # BASELINE config 4's workload on one GPU: 4 columns x 2^24, one batched call; every column against the oracle
    import torch
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()[name]
    L, B = 24, 4
    n = 1 << L
    cols = [util.rand_elems(name, n, 4000 + c) for c in range(B)]
    with ThreadPoolExecutor(4) as ex:
        futs = [ex.submit(O.evaluate_fft, oid, c) for c in cols]
        host = np.concatenate(cols)
        t_in = torch.from_numpy(host.view(np.int32 if host.dtype == np.uint32 else np.int64)).cuda()
        t_out = torch.empty_like(t_in)
        fft.ntt_device(fld, t_in, t_out, L, batch=B)
        torch.cuda.synchronize()
        got = t_out.cpu().numpy().view(host.dtype)
        for c in range(B):
            assert np.array_equal(got[c * n:(c + 1) * n].reshape(-1), np.asarray(futs[c].result()).reshape(-1)), c
        fft.ntt_device(fld, t_out, t_out, L, inverse=True, batch=B)
        torch.cuda.synchronize()
        assert torch.equal(t_out, t_in)


def test_babybear_ext4_2_24_matches_oracle():
    """Degree4BabyBearExtensionField values over a base-field domain at 2^24 (the size the ext4 timing is quoted on): an E-valued
    transform with F twiddles is four interleaved base transforms (quartic_babybear.rs:155-166), so every component column
    must equal the oracle's base-field (u64-limb) transform of that component; the 2^24 run goes through the three-pass
    plan with lgV = 2, which the small sizes never reach."""
    import torch
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()["babybear_ext4"]
    _, base_oid = util.field_pairs()["babybear_u64"]
    L = 24
    a = util.rand_elems("babybear_ext4", 1 << L, 4242)
    with ThreadPoolExecutor(4) as ex:
        futs = [ex.submit(O.evaluate_fft, base_oid, np.ascontiguousarray(a[:, k])) for k in range(4)]
        t_in = torch.from_numpy(a.view(np.int64)).cuda()
        t_out = torch.empty_like(t_in)
        fft.ntt_device(fld, t_in, t_out, L)
        torch.cuda.synchronize()
        got = t_out.cpu().numpy().view(np.uint64)
        for k in range(4):
            assert np.array_equal(got[:, k], np.asarray(futs[k].result()).reshape(-1)), f"component {k}"
    fft.ntt_device(fld, t_out, t_out, L, inverse=True)
    torch.cuda.synchronize()
    assert torch.equal(t_out, t_in)


@pytest.mark.parametrize("L", [20, 22])
def test_msm_skewed_scalar_distribution_matches_oracle(L):
    # A prover's witness is not uniform: mostly 0 / 1 / small values, some repeated constants.  Window 0 then has keys
    # holding a large share of all items and the upper windows are nearly empty: the sort must not serialise on them
    # (csrc/msm.hip level B cuts coarse bins into sub-blocks) and the result must still equal the oracle's.
    import torch
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    n = 1 << L
    thr = util.host_threads()
    scalars, points = util.msm_case(oid, n, 2600 + L, threads=thr)
    rng = np.random.default_rng(L)
    kind = rng.random(n)
    small = np.zeros((n, 4), np.uint64)
    small[:, 3] = rng.integers(0, 2, size=n, dtype=np.uint64)
    mid = np.zeros((n, 4), np.uint64)
    mid[:, 3] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
    const = np.tile(O.int_to_limbs(0x0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef, 4), (n, 1))
    scalars = np.where((kind < 0.6)[:, None], small, np.where((kind < 0.8)[:, None], mid, np.where((kind < 0.9)[:, None], const, scalars)))
    scalars = np.ascontiguousarray(scalars)
    got = msm.msm_device(crv, torch.from_numpy(scalars.view(np.int64)).cuda(), torch.from_numpy(points.view(np.int64)).cuda(), n)
    exp = O.parallel_msm_with(oid, scalars, points, max(2, O.optimum_window_size(n)), thr)
    assert _aff(oid, got) == _aff(oid, exp)
