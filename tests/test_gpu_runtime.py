"""Runtime behaviour of the C ABI on a real device: concurrent callers (the reference calls the NTT from rayon worker
threads, provers/stark/src/trace.rs:186-190), context shutdown / re-init, timing counters, large batches."""
import ctypes as C
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu


def test_concurrent_callers_get_correct_results():
    from lambda_elliptic_curves_amd import fft, msm
    fld, oid = util.field_pairs()["stark252"]
    crv, coid = util.curve_pairs()["bn254_g1"]
    cases = [(util.rand_elems("stark252", 1 << (8 + t % 5), t)) for t in range(8)]
    sc, pts = util.msm_case(coid, 300, 4)
    exp_msm = O.point_to_affine_ints(coid, O.msm(coid, sc, pts))
    errs = []

    def worker(t):
        try:
            for _ in range(3):
                a = cases[t]
                if not np.array_equal(fft.ntt(fld, a), O.fft(oid, a, O.get_twiddles(oid, a.shape[0].bit_length() - 1, O.ROOTS_BITREV))):
                    errs.append(f"ntt {t}")
                if t % 2 == 0 and O.point_to_affine_ints(coid, msm.msm(crv, sc, pts)) != exp_msm:
                    errs.append(f"msm {t}")
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs


def test_shutdown_reinit_and_timings():
    from lambda_elliptic_curves_amd import _lib, fft
    L = _lib.lib()
    fld, oid = util.field_pairs()["fr381"]
    a = util.rand_elems("fr381", 1 << 10, 3)
    exp = O.evaluate_fft(oid, a | np.uint64(0))
    assert np.array_equal(fft.ntt(fld, a), exp)
    t = _lib.Timings()
    assert L.lw_hip_get_timings(C.byref(t)) == 0
    assert t.ntt_calls >= 1 and t.twiddle_bytes > 0 and t.last_ntt_ms > 0
    L.lw_hip_shutdown()                       # drops twiddle caches, scratch, workspaces
    assert np.array_equal(fft.ntt(fld, a), exp)   # lazily re-initialised, tables rebuilt
    assert L.lw_hip_device_count() >= 1
    dev = C.c_int(99)
    assert L.lw_hip_init(C.byref(dev), 1) == _lib.ERR_NO_DEVICE      # out-of-range device id
    dev = C.c_int(0)
    assert L.lw_hip_init(C.byref(dev), 1) == 0


def test_many_small_columns_in_one_call():
    # 1000 columns x 2^8 (a wide trace): one batched call instead of the reference's per-column par_iter
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()["babybear_u32"]
    n, batch = 256, 1000
    a = util.rand_elems("babybear_u32", n * batch, 8)
    got = fft.ntt(fld, a, log2n=8, batch=batch)
    tw = O.get_twiddles(oid, 8, O.ROOTS_BITREV)
    for b in (0, 1, 499, 999):
        assert np.array_equal(got[b * n:(b + 1) * n], O.fft(oid, a[b * n:(b + 1) * n], tw))
    assert np.array_equal(fft.ntt(fld, got, inverse=True, log2n=8, batch=batch), a)
