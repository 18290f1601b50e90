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


def test_calls_on_different_streams_share_the_context_buffers_safely():
    # scratch / tables are context-owned; a multi-pass in-place NTT queued on a side stream must not be
    # overtaken by the next call on another stream (here: the host-buffer path on the null stream, and a second side
    # stream).  The library orders consecutive calls with an event (include/lw_hip.h, stream contract).
    import torch
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()["stark252"]
    L = 20
    a = util.rand_elems("stark252", 1 << L, 71)
    b = util.rand_elems("stark252", 1 << 18, 72)
    exp_a = O.evaluate_fft(oid, a)
    exp_b = O.evaluate_fft(oid, b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ta = torch.from_numpy(a.view(np.int64)).cuda()
    tb = torch.from_numpy(b.view(np.int64)).cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        ta.copy_(torch.from_numpy(a.view(np.int64)))
        tb.copy_(torch.from_numpy(b.view(np.int64)))
        torch.cuda.synchronize()
        fft.ntt_device(fld, ta, ta, L, stream=s1.cuda_stream)          # in place: 3 passes through c.scratch
        got_host = fft.ntt(fld, b)                                     # null stream, same scratch and tables
        fft.ntt_device(fld, tb, tb, 18, stream=s2.cuda_stream)         # a third stream
        torch.cuda.synchronize()
        assert np.array_equal(got_host, exp_b)
        assert np.array_equal(ta.cpu().numpy().view(np.uint64), exp_a)
        assert np.array_equal(tb.cpu().numpy().view(np.uint64), exp_b)


def test_init_takes_one_device_per_process_and_cross_step_keeps_tables_small():
    import torch
    from lambda_elliptic_curves_amd import _lib
    from lambda_elliptic_curves_amd import distributed as D
    L = _lib.lib()
    ids = (C.c_int * 2)(0, 1)
    assert L.lw_hip_init(ids, 2) == _lib.ERR_BAD_ARG          # one process per GPU; sharding goes through lw_hip_comm_init
    L.lw_hip_shutdown()
    # a G = 8 sharded transform of 2^21 elements needs the local 2^18 table, not a 2^21 one (ntt_cross.hip)
    fld, oid = util.field_pairs()["stark252"]
    a = util.rand_elems("stark252", 1 << 21, 73)
    got = D.ntt_sharded_selftest(fld, torch.from_numpy(a.view(np.int64)).cuda(), 21, 3)
    torch.cuda.synchronize()
    t = _lib.Timings()
    assert L.lw_hip_get_timings(C.byref(t)) == 0
    assert t.twiddle_bytes <= (1 << 17) * 32, t.twiddle_bytes
    assert np.array_equal(got.cpu().numpy().view(np.uint64), O.evaluate_fft(oid, a))


def test_pooled_result_buffers():
    """lw_hip_result_acquire / release: pinned result memory from the library's pool; results are the same bytes as through
    a caller-owned buffer, buffers are recycled, foreign pointers are rejected."""
    import ctypes as C
    import numpy as np
    from lambda_elliptic_curves_amd import _lib as L_
    from lambda_elliptic_curves_amd import fft
    from oracle import oracle as O
    from tests import util
    fld, oid = util.field_pairs()["stark252"]
    a = util.rand_elems("stark252", 1 << 21, 11)            # 64 MiB: above the populate threshold
    exp = O.evaluate_fft(oid, a)
    fresh = fft.ntt(fld, a)                                  # fresh numpy buffer: huge-page advice + chunked copy-back path
    assert np.array_equal(fresh, exp)
    with fft.ResultBuffer(fld, 1 << 21) as rb:
        first = rb._p.value
        got = fft.ntt(fld, a, out=rb.array)
        assert got is rb.array and np.array_equal(got, exp)
    with fft.ResultBuffer(fld, 1 << 21) as rb2:              # same size again: the pool hands the buffer back
        assert rb2._p.value == first
        rb2.array[:] = 0
        assert np.array_equal(fft.ntt(fld, a, out=rb2.array), exp)
    junk = np.zeros(16, np.uint8)
    assert L_.lib().lw_hip_result_release(junk.ctypes.data_as(C.c_void_p)) == L_.ERR_BAD_ARG
    p = C.c_void_p()
    assert L_.lib().lw_hip_result_acquire(0, C.byref(p)) == L_.ERR_BAD_ARG


def test_concurrent_callers_run_on_lanes_and_share_growing_tables():
    """Calls from different host threads run concurrently on different lanes (csrc/context.h); the twiddle tables are shared
    and a call that needs a bigger one rebuilds it with every other call out of the library.  Eight threads, each walking
    sizes that force rebuilds of three tables (forward, inverse, BabyBear) while the others are mid-call, host and device
    entry points mixed; every result against the oracle."""
    import threading
    import torch
    from lambda_elliptic_curves_amd import _lib, fft
    from oracle import oracle as O
    from tests import util
    _lib.lib().lw_hip_shutdown()              # start from empty tables
    fs, os_ = util.field_pairs()["stark252"]
    fb, ob = util.field_pairs()["babybear_u32"]
    sizes = [6, 17, 12, 18, 9, 19, 16]       # tables are built for >= 2^16: 17, 18, 19 each force a rebuild
    cases = {}
    for L in set(sizes):
        a = util.rand_elems("stark252", 1 << L, 100 + L)
        b = util.rand_elems("babybear_u32", 1 << L, 200 + L)
        cases[L] = (a, O.evaluate_fft(os_, a), b, np.asarray(O.evaluate_fft(ob, b)).reshape(-1))
    errs = []

    def worker(t):
        try:
            for k in range(len(sizes)):
                L = sizes[(k + t) % len(sizes)]
                a, ea, b, eb = cases[L]
                if not np.array_equal(fft.ntt(fs, a), ea):
                    errs.append(f"stark fwd 2^{L} thread {t}")
                if not np.array_equal(fft.ntt(fs, ea, inverse=True), a):
                    errs.append(f"stark inv 2^{L} thread {t}")
                if not np.array_equal(fft.ntt(fb, b), eb):
                    errs.append(f"babybear 2^{L} thread {t}")
                if t % 2:
                    s = torch.cuda.Stream()
                    ta = torch.from_numpy(a.view(np.int64)).cuda()
                    to = torch.empty_like(ta)
                    fft.ntt_device(fs, ta, to, L, stream=s.cuda_stream)
                    s.synchronize()
                    if not np.array_equal(to.cpu().numpy().view(np.uint64), ea):
                        errs.append(f"stark device 2^{L} thread {t}")
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs[:5]
    t = _lib.Timings()
    assert _lib.lib().lw_hip_get_timings(C.byref(t)) == 0 and t.ntt_calls >= 8 * len(sizes) * 3
