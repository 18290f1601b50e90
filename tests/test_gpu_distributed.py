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


# ---------------------------------------------------------------- the C++ schedule behind the C ABI (csrc/comm.hip)
def _as_t(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()


@pytest.mark.parametrize("name", ["stark252", "fr381", "babybear_u32", "babybear_u64", "babybear_ext4"])
@pytest.mark.parametrize("lg,L", [(1, 10), (2, 13), (3, 16), (3, 6)])
def test_library_sharded_schedule_selftest_matches_oracle(name, lg, L):
    # lw_hip_ntt_sharded_selftest_device: the exchange schedule RCCL runs, with 2^lg virtual ranks on this one GPU
    from lambda_elliptic_curves_amd import distributed as D
    fld, oid = util.field_pairs()[name]
    n, G = 1 << L, 1 << lg
    M = n // G
    full = util.rand_elems(name, n, 140 + L)
    exp = O.fft(oid, full, O.get_twiddles(oid, L, O.ROOTS_BITREV))
    t = _as_t(full)
    nat = D.ntt_sharded_selftest(fld, t, L, lg)
    cyc = D.ntt_sharded_selftest(fld, t, L, lg, natural_output=False)
    back = D.ntt_sharded_selftest(fld, nat, L, lg, inverse=True)
    torch.cuda.synchronize()
    view = lambda x: x.cpu().numpy().view(full.dtype).reshape(full.shape)
    assert np.array_equal(view(nat), exp)
    cyc = view(cyc)
    for g in range(G):
        assert np.array_equal(cyc[g * M:(g + 1) * M], exp[g::G])
    assert np.array_equal(view(back), full)


@pytest.mark.parametrize("name,lg,L,batch", [("stark252", 1, 8, 1), ("babybear_u32", 2, 10, 3), ("fr381", 3, 9, 2), ("babybear_ext4", 2, 8, 2)])
@pytest.mark.parametrize("natural", [True, False])
def test_python_schedule_is_the_cpp_schedule_step_by_step(name, lg, L, batch, natural):
    """One schedule, two spellings: csrc/comm.hip ntt_sharded_run (what RCCL runs) and distributed.ntt_sharded (what the
    world-size-2 gloo test drives).  After EVERY step (exchange A, cross step, exchange C, local NTT, exchange E,
    interleave) every virtual rank's buffer must hold the same bytes in both, for every batch column — chunk order of the
    exchanges and the (j2_begin, slice_len) arguments included."""
    from lambda_elliptic_curves_amd import distributed as D
    fld, oid = util.field_pairs()[name]
    n, G = 1 << L, 1 << lg
    M = n // G
    full = util.rand_elems(name, batch * n, 940 + L)
    t_full = _as_t(full)
    comms = D.SimComm.make(G)
    for step in range(1, (6 if natural else 4) + 1):
        cpp = D.ntt_sharded_selftest_steps(fld, t_full, L, lg, step, natural_output=natural, batch=batch)
        torch.cuda.synchronize()
        cpp = cpp.cpu().numpy().view(full.dtype).reshape((batch, G, M) + full.shape[1:])

        def rank(r):
            mine = np.concatenate([full[bi * n + r * M: bi * n + (r + 1) * M] for bi in range(batch)])
            out = D.ntt_sharded(fld, _as_t(mine), L, comms[r], natural_output=natural, batch=batch, stop_after=step)
            torch.cuda.synchronize()
            return out.cpu().numpy().view(full.dtype).reshape((batch, M) + full.shape[1:])

        outs = _run_ranks(G, rank)
        for r in range(G):
            assert np.array_equal(outs[r], cpp[:, r]), f"step {step}, rank {r}"


@pytest.mark.parametrize("natural", [True, False])
def test_library_sharded_schedule_batched_and_cyclic(natural):
    # batch > 1 with the cyclic output: step D must not read one batch entry with the other's stride
    from lambda_elliptic_curves_amd import distributed as D
    fld, oid = util.field_pairs()["babybear_u64"]
    L, lg, B = 12, 2, 4
    n, G = 1 << L, 1 << lg
    M = n // G
    full = util.rand_elems("babybear_u64", B * n, 77)
    got = D.ntt_sharded_selftest(fld, _as_t(full), L, lg, natural_output=natural, batch=B)
    torch.cuda.synchronize()
    got = got.cpu().numpy().view(np.uint64).reshape(B, n)
    for bi in range(B):
        exp = np.asarray(O.evaluate_fft(oid, full[bi * n:(bi + 1) * n])).reshape(-1)
        if natural:
            assert np.array_equal(got[bi], exp)
        else:
            for g in range(G):
                assert np.array_equal(got[bi, g * M:(g + 1) * M], exp[g::G])


def test_library_sharded_schedule_selftest_config4_shape():
    # BASELINE config 4: 4 BabyBear columns x 2^24 over 8 ranks (2^21 elements per rank and column), one batched call;
    # every column equals the single-device transform, which test_gpu_parity_full pins to the oracle at this size
    from lambda_elliptic_curves_amd import distributed as D
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()["babybear_u32"]
    L, B = 24, 4
    full = util.rand_elems("babybear_u32", B << L, 99)
    t = _as_t(full)
    got = D.ntt_sharded_selftest(fld, t, L, 3, batch=B)
    ref = torch.empty_like(t)
    fft.ntt_device(fld, t, ref, L, batch=B)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    col = O.evaluate_fft(oid, full[: 1 << L])
    assert np.array_equal(got[: 1 << L].cpu().numpy().view(np.uint32), np.asarray(col).reshape(-1))


def test_library_sharded_large_stark252_2_22_over_8():
    from lambda_elliptic_curves_amd import distributed as D
    from lambda_elliptic_curves_amd import fft
    fld, oid = util.field_pairs()["stark252"]
    L = 22
    full = util.rand_elems("stark252", 1 << L, 98)
    t = _as_t(full)
    got = D.ntt_sharded_selftest(fld, t, L, 3)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), O.evaluate_fft(oid, full))


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_g1", "bn254_g2", "bls12_381_g2"])
@pytest.mark.parametrize("lg,n", [(1, 257), (2, 3001), (3, 5000), (3, 5)])
def test_sharded_msm_bucket_slice_exchange_selftest(name, lg, n):
    """lw_hip_msm_sharded_selftest_device: the schedule RCCL runs for the sharded MSM (local accumulation -> all-to-all of
    bucket slices -> per-slice sums and running sums -> all-gather of the (S, A) pairs -> fold), with 2^lg virtual ranks on
    this one GPU, against the oracle's msm over all pairs.  Uneven shards, shards without pairs (n = 5 over 8 ranks),
    adversarial scalars (all windows at the signed-digit boundary, 2^256 - 1, r - 1, zero)."""
    from lambda_elliptic_curves_amd import distributed as D
    from oracle import bigint_def as DD
    crv, oid = util.curve_pairs()[name]
    scalars, points = util.msm_case(oid, n, 3100 + n + lg)
    r = DD.P_FR381 if name.startswith("bls") else DD.P_FR254
    ks = [int(O.limbs_to_int(row)) for row in scalars]
    for i, v in enumerate([(1 << 256) - 1, r - 1, 0, sum((1 << 7) << (8 * w) for w in range(32)), sum(((1 << 7) + 1) << (8 * w) for w in range(32))]):
        ks[(i * 7) % n] = v
    scalars = O.ints_to_array(ks, 4)
    points = points.copy()
    points[n // 2] = O.ec_neutral(oid)
    got = D.msm_sharded_selftest(crv, _as_t(scalars), _as_t(points), n, lg)
    exp = O.parallel_msm_with(oid, scalars, points, 8, 8)
    assert O.point_to_affine_ints(oid, got) == O.point_to_affine_ints(oid, exp)


def test_sharded_msm_selftest_wide_windows_2_18():
    # 2^18 BN254 G1 pairs over 8 virtual ranks: 2^15 per rank, c = 16 (the wide-item sort), 2^12 buckets per slice
    from lambda_elliptic_curves_amd import distributed as D
    crv, oid = util.curve_pairs()["bn254_g1"]
    n = 1 << 18
    scalars, points = util.msm_case(oid, n, 3218, threads=util.host_threads())
    got = D.msm_sharded_selftest(crv, _as_t(scalars), _as_t(points), n, 3)
    exp = O.parallel_msm_with(oid, scalars, points, 14, util.host_threads())
    assert O.point_to_affine_ints(oid, got) == O.point_to_affine_ints(oid, exp)


def test_rccl_communicator_one_rank_through_the_c_abi():
    # The library-owned RCCL communicator with nranks = 1: librccl is loaded, ncclCommInitRank runs, and the sharded
    # entry points go through ncclSend/ncclRecv (to self) and ncclAllGather on the real transport.
    from lambda_elliptic_curves_amd import distributed as D
    from lambda_elliptic_curves_amd import errors, msm
    from lambda_elliptic_curves_amd import _lib as L_
    import ctypes as C
    L_.lib().lw_hip_comm_shutdown()
    fld, oid = util.field_pairs()["stark252"]
    a = util.rand_elems("stark252", 1 << 14, 5)
    t = _as_t(a)
    out = torch.empty_like(t)
    rc = L_.lib().lw_hip_ntt_sharded_device(fld.field, fld.layout, 0, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), 14, 1, 1, None)
    assert rc == L_.ERR_COMM                        # no communicator yet
    with pytest.raises(errors.CommError):
        errors.check(rc)
    comm = D.HipComm(D.HipComm.unique_id(), 0, 1)
    try:
        r, n = C.c_int(-1), C.c_int(-1)
        assert L_.lib().lw_hip_comm_info(C.byref(r), C.byref(n)) == 0 and (r.value, n.value) == (0, 1)
        got = D.ntt_sharded(fld, t, 14, comm)
        back = D.ntt_sharded(fld, got, 14, comm, inverse=True)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint64), O.evaluate_fft(oid, a))
        assert np.array_equal(back.cpu().numpy().view(np.uint64), a)
        bb, boid = util.field_pairs()["babybear_u32"]
        cols = util.rand_elems("babybear_u32", 4 << 12, 6)
        gb = D.ntt_sharded(bb, _as_t(cols), 12, comm, batch=4)
        torch.cuda.synchronize()
        for c in range(4):
            assert np.array_equal(gb[c << 12:(c + 1) << 12].cpu().numpy().view(np.uint32),
                                  np.asarray(O.evaluate_fft(boid, cols[c << 12:(c + 1) << 12])).reshape(-1))
        crv, coid = util.curve_pairs()["bn254_g1"]
        sc, pts = util.msm_case(coid, 3000, 8)
        p = D.msm_sharded(crv, _as_t(sc), _as_t(pts), 3000, comm)
        assert O.point_to_affine_ints(coid, p) == O.point_to_affine_ints(coid, O.msm(coid, sc, pts))
    finally:
        comm.close()
