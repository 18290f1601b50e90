"""Seeded differential tests in the spirit of the reference's GPU fuzz targets (fuzz/cuda_fuzz/src/cuda_fft_fuzzer.rs,
polynomial_fft_diff.rs, twiddles_generation_diff.rs): random shapes and parameters, HIP result == CPU result, and where
the CPU path reports an error the HIP path must report the same one."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import util

import os

pytestmark = pytest.mark.gpu
FIELDS = ["stark252", "fr381", "babybear_u32", "babybear_u64", "babybear_ext4"]
EXTRA = int(os.environ.get("LW_DIFF_EXTRA_SEEDS", "0"))   # a longer campaign: LW_DIFF_EXTRA_SEEDS=40 pytest tests/test_gpu_differential.py


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.size == b.size and np.array_equal(a.reshape(-1), b.reshape(-1))


@pytest.mark.parametrize("seed", range(6 + EXTRA))
def test_polynomial_fft_diff_random_shapes(seed):
    # polynomial_fft_diff: evaluate_fft / evaluate_offset_fft / interpolate_fft over random coefficient counts (not powers
    # of two), blow-up factors, domain sizes and offsets, all five element shapes
    from lambda_elliptic_curves_amd import fft
    rng = np.random.default_rng(1000 + seed)
    for _ in range(25):
        name = FIELDS[int(rng.integers(0, len(FIELDS)))]
        fld, oid = util.field_pairs()[name]
        n_coeffs = int(rng.integers(0, 700))
        blow = int(2 ** rng.integers(0, 4))
        dom = None if rng.random() < 0.4 else int(2 ** rng.integers(0, 11))
        a = util.rand_elems(name, n_coeffs, int(rng.integers(0, 1 << 30)))
        if n_coeffs and rng.random() < 0.3:       # trailing zero coefficients are stripped by Polynomial::new
            a[-int(rng.integers(1, min(n_coeffs, 5) + 1)):] = 0
        off = util.offset_elem(name, int(rng.integers(2, 50))) if rng.random() < 0.5 else None
        exp = O.evaluate_fft(oid, a, blow, dom, off)
        got = fft.evaluate_fft(fld, a, blow, dom, off)
        assert _eq(got, exp), (name, n_coeffs, blow, dom, off is not None)
        if exp.shape[0] and not (exp.shape[0] & (exp.shape[0] - 1)):
            assert _eq(fft.interpolate_fft(fld, exp, off), O.interpolate_fft(oid, exp, off))


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_fft_seam_diff_random_batches(seed):
    # cuda_fft_fuzzer: the backend seam on already padded slices — random log2 size, batch, stride, direction, in place
    from lambda_elliptic_curves_amd import errors, fft
    rng = np.random.default_rng(2000 + seed)
    for _ in range(25):
        name = FIELDS[int(rng.integers(0, len(FIELDS)))]
        fld, oid = util.field_pairs()[name]
        lg = int(rng.integers(0, 13))
        n = 1 << lg
        batch = int(rng.integers(1, 6))
        stride = n + int(rng.integers(0, 3)) * 8
        inverse = bool(rng.integers(0, 2))
        buf = util.rand_elems(name, batch * stride, int(rng.integers(0, 1 << 30)))
        got = fft.ntt(fld, buf, inverse=inverse, log2n=lg, batch=batch, batch_stride=stride)
        tw = O.get_twiddles(oid, lg, O.ROOTS_BITREV_INV if inverse else O.ROOTS_BITREV)
        for b in range(batch):
            seg = buf[b * stride:b * stride + n]
            exp = O.interpolate_fft(oid, seg) if inverse else O.fft(oid, seg, tw)
            assert _eq(got[b * stride:b * stride + n], exp), (name, lg, batch, stride, inverse, b)
    fld, _ = util.field_pairs()["stark252"]
    with pytest.raises(errors.InputError):            # not a power of two: CPU and GPU both refuse (ops.rs:17-19)
        fft.ntt(fld, util.rand_elems("stark252", 12, 1))


@pytest.mark.parametrize("seed", range(4 + EXTRA))
def test_msm_diff_random_inputs(seed):
    # Pippenger == reference sum on random lengths, scalar widths and degenerate points, all four groups
    from lambda_elliptic_curves_amd import msm
    rng = np.random.default_rng(3000 + seed)
    names = list(util.curve_pairs())
    for _ in range(10):
        name = names[int(rng.integers(0, len(names)))]
        crv, oid = util.curve_pairs()[name]
        n = int(rng.integers(0, 600))
        scalars, points = util.msm_case(oid, max(n, 1), int(rng.integers(0, 1 << 30)))
        scalars, points = scalars[:n].copy(), points[:n].copy()
        if n:
            bits = int(rng.integers(1, 257))              # scalars of a random width
            ks = [int.from_bytes(rng.bytes(32), "big") >> (256 - bits) for _ in range(n)]
            scalars = O.ints_to_array(ks, 4)
            for _ in range(int(rng.integers(0, 4))):      # a few identity rows, duplicates and negated duplicates
                i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
                r = rng.random()
                points[i] = O.ec_neutral(oid) if r < 0.34 else (points[j] if r < 0.67 else O.ec_neg(oid, points[j]))
        got = msm.msm(crv, scalars, points)
        exp = O.msm(oid, scalars, points) if n else O.ec_neutral(oid)
        assert O.point_to_affine_ints(oid, got) == O.point_to_affine_ints(oid, exp), (name, n)
        if n and rng.random() < 0.5:                      # the same call over a pre-normalised handle (affine rows), a random prefix
            srs = msm.Srs(crv, points)
            m = int(rng.integers(1, n + 1))
            want = O.msm(oid, scalars[:m], points[:m])
            assert O.point_to_affine_ints(oid, srs.msm(scalars[:m])) == O.point_to_affine_ints(oid, want), (name, n, m, "srs")
            srs.close()
