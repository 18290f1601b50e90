"""Sharded NTT / MSM schedule with the real HIP kernels: G virtual ranks (threads) share the one GPU of the test
box through SimComm; results must equal the single-device oracle transform of the whole vector."""
import threading

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def _run_ranks(G, fn):
    outs, errs = [None] * G, []

    def wrap(r):
        try:
            outs[r] = fn(r)
        except Exception as e:   # pragma: no cover
            import traceback
            errs.append(traceback.format_exc())
    th = [threading.Thread(target=wrap, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[0]
    return outs


@pytest.mark.parametrize("name", ["stark252", "fr381", "babybear_u32", "babybear_u64", "babybear_ext4"])
@pytest.mark.parametrize("G,L", [(2, 10), (4, 13), (8, 16)])
def test_sharded_ntt_matches_single_device_oracle(name, G, L):
    from lambda_elliptic_curves_amd import distributed as D
    fld, oid = util.field_pairs()[name]
    n = 1 << L
    M = n // G
    full = util.rand_elems(name, n, 40 + L)
    exp = O.fft(oid, full, O.get_twiddles(oid, L, O.ROOTS_BITREV))
    comms = D.SimComm.make(G)
    as_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()

    def rank(r):
        t = as_t(full[r * M:(r + 1) * M])
        out = D.ntt_sharded(fld, t, L, comms[r])
        cyc = D.ntt_sharded(fld, t, L, comms[r], natural_output=False)
        back = D.ntt_sharded(fld, out, L, comms[r], inverse=True)
        torch.cuda.synchronize()
        return [x.cpu().numpy().view(full.dtype).reshape(full[:M].shape) for x in (out, cyc, back)]

    outs = _run_ranks(G, rank)
    for r in range(G):
        assert np.array_equal(outs[r][0], exp[r * M:(r + 1) * M])
        assert np.array_equal(outs[r][1], exp[r::G])
        assert np.array_equal(outs[r][2], full[r * M:(r + 1) * M])


def test_sharded_msm_four_virtual_ranks():
    from lambda_elliptic_curves_amd import distributed as D
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    G, n = 4, 4096
    scalars, points = util.msm_case(oid, n, 21)
    h = n // G
    comms = D.SimComm.make(G)

    def rank(r):
        ts = torch.from_numpy(np.ascontiguousarray(scalars[r * h:(r + 1) * h]).view(np.int64)).cuda()
        tp = torch.from_numpy(np.ascontiguousarray(points[r * h:(r + 1) * h]).view(np.int64)).cuda()
        return D.msm_sharded(crv, ts, tp, h, comms[r])

    outs = _run_ranks(G, rank)
    exp = O.point_to_affine_ints(oid, O.parallel_msm_with(oid, scalars, points, 9, 8))
    for r in range(G):
        assert O.point_to_affine_ints(oid, outs[r]) == exp
