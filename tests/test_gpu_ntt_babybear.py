"""GPU parity for the BabyBear NTT in the reference's three memory shapes (u32 R=2^32, u64-limb R=2^64,
quartic-extension values over a base-field domain), bit-exact vs the CPU oracle.
Mirrors babybear.rs / babybear_u32.rs / quartic_babybear.rs:397-569 FFT tests and math/src/fft/polynomial.rs:442-457."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu
BB = ["babybear_u32", "babybear_u64", "babybear_ext4"]


def same(a, b):
    """bit equality irrespective of the (n,) vs (n,1) view of one-word elements"""
    a, b = np.asarray(a), np.asarray(b)
    return a.size == b.size and np.array_equal(a.reshape(-1), b.reshape(-1))


def nz_last(name, a):
    if a.ndim == 1:
        a[-1] |= 1
    else:
        a[-1, -1] |= np.uint64(1)
    return a


@pytest.mark.parametrize("name", BB)
@pytest.mark.parametrize("log_n", list(range(0, 15)) + [16, 18])
def test_evaluate_and_interpolate_match_oracle(name, log_n):
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()[name]
    a = nz_last(name, util.rand_elems(name, 1 << log_n, 500 + log_n))
    ev = fft.evaluate_fft(fld, a)
    assert ev.shape == a.shape and same(ev, O.evaluate_fft(oid, a))
    back = fft.interpolate_fft(fld, ev)
    assert same(back, O.interpolate_fft(oid, ev)) and same(back, a)


@pytest.mark.parametrize("name", BB)
@pytest.mark.parametrize("log_n,h,blowup", [(3, 3, 1), (6, 7, 2), (10, 2, 8), (13, 3, 4)])
def test_offset_blowup_domain(name, log_n, h, blowup):
    # prover-style LDE: evaluate_offset_fft(poly, blowup, Some(n), offset) (provers/stark/src/prover.rs:150-167)
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()[name]
    a = nz_last(name, util.rand_elems(name, 1 << log_n, 900 + log_n))
    off = util.offset_elem(name, h)
    ev = fft.evaluate_offset_fft(fld, a, blowup, 1 << log_n, off)
    assert same(ev, O.evaluate_fft(oid, a, blowup, 1 << log_n, off))
    ev1 = fft.evaluate_offset_fft(fld, a, 1, None, off)
    assert same(fft.interpolate_offset_fft(fld, ev1, off), a)


@pytest.mark.parametrize("name", BB)
@pytest.mark.parametrize("log_m,blow", [(0, 3), (1, 1), (3, 2), (6, 3), (9, 3), (12, 1), (13, 4), (16, 3), (17, 1), (20, 2)])
def test_low_degree_extension_path(name, log_m, blow):
    """BASELINE config 4 is a STARK LDE: evaluate_offset_fft(poly, blowup, Some(n), offset) (provers/stark/src/prover.rs:150-167,
    math/src/fft/polynomial.rs:30-38,74-82).  The device skips the log2(blowup) stages that only replicate the zero-padded
    block and never materialises the padding — the bytes must equal the oracle's padded transform, for every layout, through
    the host API (length rule) and through lw_hip_ntt_lde_device with a batch of 2."""
    import torch
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()[name]
    n = 1 << log_m
    a = nz_last(name, util.rand_elems(name, n, 600 + log_m))
    off = util.offset_elem(name, 3)
    exp = O.evaluate_fft(oid, a, 1 << blow, n, off)
    assert same(fft.evaluate_offset_fft(fld, a, 1 << blow, n, off), exp)
    if log_m <= 16:
        assert same(fft.evaluate_fft(fld, a, 1 << blow, n), O.evaluate_fft(oid, a, 1 << blow, n))
        short = nz_last(name, a[: max(1, n - n // 3)].copy())
        assert same(fft.evaluate_offset_fft(fld, short, 1 << blow, n, off), O.evaluate_fft(oid, short, 1 << blow, n, off))
    b = util.rand_elems(name, n, 5)
    two = np.concatenate([a, b])
    words = 4 if name == "babybear_ext4" else 1
    tdt = torch.int32 if name == "babybear_u32" else torch.int64
    ndt = np.int32 if name == "babybear_u32" else np.int64
    t_in = torch.from_numpy(two.view(ndt)).cuda()
    N = n << blow
    t_out = torch.empty((2 * N * words,), dtype=tdt, device="cuda")
    fft.lde_device(fld, t_in, log_m, t_out, log_m + blow, batch=2, offset=off)
    torch.cuda.synchronize()
    got = t_out.cpu().numpy().view(two.dtype).reshape((2 * N,) + two.shape[1:])
    assert same(got[:N], exp)
    pad = np.zeros((N,) + two.shape[1:], two.dtype)
    pad[:n] = b
    assert same(got[N:], O.evaluate_fft(oid, pad, 1, N, off))


@pytest.mark.parametrize("name", BB)
def test_two_adicity_limit_and_errors(name):
    # TWO_ADICITY is declared 24 (babybear.rs:29, babybear_u32.rs:17): 2^25 has no root of unity in the reference
    from lambda_elliptic_curves_amd import errors, fft
    fld, _ = util.field_pairs()[name]
    with pytest.raises(errors.RootOfUnityError):
        fft.ntt(fld, util.rand_elems(name, 2, 1), log2n=25)
    with pytest.raises(errors.InputError):
        fft.interpolate_fft(fld, util.rand_elems(name, 6, 1))


def test_ext4_equals_four_interleaved_base_transforms():
    from lambda_elliptic_curves_amd import fft
    n = 1 << 9
    a = util.rand_elems("babybear_ext4", n, 77)
    ev = fft.ntt(fft.Degree4BabyBearExtensionField, a)
    for k in range(4):
        col = np.ascontiguousarray(a[:, k])
        assert np.array_equal(ev[:, k], fft.ntt(fft.Babybear31PrimeField, col))


def test_batched_columns_2_20_u32():
    # 4 columns x 2^20 (column-major batch), the shape of BASELINE config 4 scaled to one GPU
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()["babybear_u32"]
    n, batch = 1 << 20, 4
    a = util.rand_elems("babybear_u32", n * batch, 5)
    got = fft.ntt(fld, a, log2n=20, batch=batch)
    tw = O.get_twiddles(oid, 20, O.ROOTS_BITREV)
    for b in range(batch):
        assert np.array_equal(got[b * n:(b + 1) * n], O.fft(oid, a[b * n:(b + 1) * n], tw))
    assert np.array_equal(fft.ntt(fld, got, inverse=True, log2n=20, batch=batch), a)


def test_max_size_2_24_roundtrip_u32():
    # largest legal BabyBear transform (TWO_ADICITY 24): NTT then INTT restores the input; two outputs checked
    # against the defining sum (ev[0] = sum c_j, ev[N/2] = sum (-1)^j c_j)
    from lambda_elliptic_curves_amd import fft
    from oracle import bigint_def as D
    fld, oid = util.field_pairs()["babybear_u32"]
    a = util.rand_elems("babybear_u32", 1 << 24, 9)
    ev = fft.ntt(fld, a)
    assert np.array_equal(fft.ntt(fld, ev, inverse=True), a)
    p = D.P_BABYBEAR
    canon = (a.astype(np.uint64) * np.uint64(pow(1 << 32, -1, p))) % np.uint64(p)   # out of Montgomery form, vectorised
    s0 = int(canon.sum() % p)
    s1 = int((int(canon[0::2].sum()) - int(canon[1::2].sum())) % p)
    assert O.elems_from_mont(oid, ev[:1])[0] == s0
    assert O.elems_from_mont(oid, ev[1 << 23:(1 << 23) + 1])[0] == s1
