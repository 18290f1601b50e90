"""N > 1 path on CPU: world_size-2 gloo process group runs the same SPMD exchange schedule the GPUs run
(lambda_elliptic_curves_amd/distributed.py), with the CPU oracle standing in for the local kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from tests import util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from lambda_elliptic_curves_amd import distributed as D
        from lambda_elliptic_curves_amd import fft, msm
        from tests.dist_helpers import OracleBackend
        comm = D.TorchDistComm()
        ok = True
        for name, fld in (("babybear_u32", fft.Babybear31PrimeFieldU32), ("stark252", fft.Stark252PrimeField)):
            oid = util.field_pairs()[name][1]
            L = 6
            n = 1 << L
            full = util.rand_elems(name, n, 99)               # same seed on every rank
            M = n // world
            mine = np.ascontiguousarray(full[rank * M:(rank + 1) * M])
            t = torch.from_numpy(mine.view(np.int32 if mine.dtype == np.uint32 else np.int64))
            be = OracleBackend(oid)
            exp = O.fft(oid, full, O.get_twiddles(oid, L, O.ROOTS_BITREV))
            out = D.ntt_sharded(fld, t, L, comm, backend=be)
            got = out.numpy().view(mine.dtype).reshape(mine.shape)
            ok &= np.array_equal(got, exp[rank * M:(rank + 1) * M])
            cyc = D.ntt_sharded(fld, t, L, comm, backend=be, natural_output=False)
            ok &= np.array_equal(cyc.numpy().view(mine.dtype).reshape(mine.shape), exp[rank::world])
            back = D.ntt_sharded(fld, out, L, comm, inverse=True, backend=be)
            ok &= np.array_equal(back.numpy().view(mine.dtype).reshape(mine.shape), mine)
        # a batch of two columns through the same per-column steps (BASELINE config 4 has four)
        name, fld = "babybear_u32", fft.Babybear31PrimeFieldU32
        oid = util.field_pairs()[name][1]
        L, B = 5, 2
        n, M = 1 << L, (1 << L) // world
        cols = [util.rand_elems(name, n, 200 + c) for c in range(B)]
        mine = np.concatenate([c[rank * M:(rank + 1) * M] for c in cols])
        got = D.ntt_sharded(fld, torch.from_numpy(mine.view(np.int32)), L, comm, backend=OracleBackend(oid), batch=B).numpy().view(np.uint32)
        for c in range(B):
            exp = np.asarray(O.fft(oid, cols[c], O.get_twiddles(oid, L, O.ROOTS_BITREV))).reshape(-1)
            ok &= np.array_equal(got[c * M:(c + 1) * M], exp[rank * M:(rank + 1) * M])
        # MSM: shard by points, all-gather the partial sums
        oid = O.C_BN254_G1
        scalars, points = util.msm_case(oid, 12, 5)
        h = 12 // world
        ts = torch.from_numpy(np.ascontiguousarray(scalars[rank * h:(rank + 1) * h]).view(np.int64))
        tp = torch.from_numpy(np.ascontiguousarray(points[rank * h:(rank + 1) * h]).view(np.int64))
        got = D.msm_sharded(msm.BN254Curve, ts, tp, h, comm, backend=OracleBackend(None, oid))
        ok &= O.point_to_affine_ints(oid, got) == O.point_to_affine_ints(oid, O.msm(oid, scalars, points))
        q.put((rank, bool(ok), ""))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


def test_sharded_ntt_and_msm_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in results:
        assert ok, f"rank {rank}: {msg}"


def test_sim_comm_four_virtual_ranks_cpu():
    # the in-process simulator used on the single-GPU box, checked here with the oracle backend and G = 4
    import threading
    from lambda_elliptic_curves_amd import distributed as D
    from lambda_elliptic_curves_amd import fft
    from tests.dist_helpers import OracleBackend
    name, fld = "babybear_u32", fft.Babybear31PrimeFieldU32
    oid = util.field_pairs()[name][1]
    G, L = 4, 6
    n = 1 << L
    full = util.rand_elems(name, n, 3)
    exp = O.fft(oid, full, O.get_twiddles(oid, L, O.ROOTS_BITREV))
    comms = D.SimComm.make(G)
    outs = [None] * G

    def run(r):
        M = n // G
        t = torch.from_numpy(np.ascontiguousarray(full[r * M:(r + 1) * M]).view(np.int32))
        outs[r] = D.ntt_sharded(fld, t, L, comms[r], backend=OracleBackend(oid)).numpy().view(np.uint32)

    th = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert np.array_equal(np.concatenate(outs), exp)
