"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no multi-device code at all; these entry points produce the same results as the
single-device API on the concatenated input.

* ntt_sharded   one large transform block-distributed over G = 2/4/8 ranks (Bailey four-step with N1 = G):
                all-to-all -> cross-shard step (lw_hip_ntt_cross_device) -> all-to-all -> local M-point NTT
                (lw_hip_ntt_device) [-> all-to-all + local interleave for natural block order].
                xGMI is a full mesh, so every all-to-all keeps all 7 links busy; payload per rank and exchange
                is (G-1)/G of the local shard.
* msm_sharded   points (and scalars) are sharded.  HipComm (the library, csrc/comm.hip msm_sharded_run): every rank
                accumulates its pairs into the full bucket array, an all-to-all hands rank g bucket range g of every
                window, rank g sums the G contributions and runs the running sums over its slice, an all-gather of the
                per-slice sums lets every rank fold the result — the north star's "bucket all-reduce" as
                reduce-scatter + all-gather (RCCL has no user-defined reduction).  The TorchDistComm / SimComm path
                below is the simpler first form (all-gather of one partial sum per rank), kept for the CPU gloo test.
* batches with batch >= G need no collective: give each rank whole columns (fft.ntt_device per rank).

The production path is `HipComm`: the communicator and the whole exchange schedule live INSIDE the C library
(csrc/comm.hip: lw_hip_comm_init, lw_hip_ntt_sharded_device, lw_hip_msm_sharded_device — RCCL send/recv groups and
all-gather issued from C++), so a Rust caller reaches the multi-GPU path through the same extern "C" boundary and this
module is only a thin caller.  `TorchDistComm` / `SimComm` run a literal transliteration of that schedule in Python over
torch.distributed (gloo on CPU for tests) or G in-process virtual ranks, checked against the C++ one step by step on
the GPU (`ntt_sharded(..., stop_after=k)` vs `ntt_sharded_selftest_steps`); `backend` abstracts their local compute
(default: the HIP library; there is no CPU fallback).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .errors import check


# ---------------------------------------------------------------- communicators
class HipComm:
    """Library-owned RCCL communicator (include/lw_hip.h "Multi-GPU").  `unique_id` is the 128-byte id rank 0 got from
    HipComm.unique_id() and handed to the other processes out of band (ncclGetUniqueId's contract);
    HipComm.from_torch_dist() does that hand-over with torch.distributed's broadcast."""

    def __init__(self, unique_id, rank, size):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        check(L.lib().lw_hip_comm_init(buf, rank, size))
        self.rank, self.size = rank, size

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(L.lib().lw_hip_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch_dist(cls, group=None):
        import torch.distributed as dist
        rank, size = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], rank, size)

    def close(self):
        check(L.lib().lw_hip_comm_shutdown())


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ntt_sharded_selftest(field, x_full, log2n_total, log2_shards, inverse=False, natural_output=True, batch=1):
    """lw_hip_ntt_sharded_selftest_device: the C++ exchange schedule with 2^log2_shards virtual ranks on one device."""
    import torch
    out = torch.empty_like(x_full)
    check(L.lib().lw_hip_ntt_sharded_selftest_device(field.field, field.layout, L.DIR_INVERSE if inverse else L.DIR_FORWARD,
                                                     C.c_void_p(x_full.data_ptr()), C.c_void_p(out.data_ptr()), log2n_total,
                                                     log2_shards, batch, 1 if natural_output else 0, _stream_ptr()))
    return out


class TorchDistComm:
    """torch.distributed process group (backend 'nccl' is RCCL on ROCm; 'gloo' for CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def all_to_all(self, send):
        """send: tensor [G, ...] — chunk h goes to rank h; returns [G, ...] with chunk j from rank j."""
        import torch
        send = send.contiguous()
        recv = torch.empty_like(send)
        self.dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def all_gather(self, t):
        import torch
        out = [torch.empty_like(t) for _ in range(self.size)]
        self.dist.all_gather(out, t.contiguous(), group=self.group)
        return out


class SimComm:
    """G virtual ranks inside one process (one Python thread each) — for exercising the SPMD schedule on one GPU."""

    class _Shared:
        def __init__(self, size):
            import threading
            self.size = size
            self.barrier = threading.Barrier(size)
            self.slots = [None] * size

    def __init__(self, shared, rank):
        self.shared, self.rank, self.size = shared, rank, shared.size

    @classmethod
    def make(cls, size):
        sh = cls._Shared(size)
        return [cls(sh, r) for r in range(size)]

    def _exchange(self, payload):
        sh = self.shared
        sh.slots[self.rank] = payload
        sh.barrier.wait()
        got = list(sh.slots)
        sh.barrier.wait()
        return got

    def all_to_all(self, send):
        import torch
        everyone = self._exchange(send.contiguous())
        return torch.stack([everyone[j][self.rank] for j in range(self.size)])

    def all_gather(self, t):
        return [x.clone() for x in self._exchange(t.contiguous())]


# ---------------------------------------------------------------- local compute
class HipBackend:
    """Local steps on the GPU through the C ABI (device-resident torch tensors, torch's current stream)."""

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def cross(self, field, t_in, log2n_total, log2g, j2_begin, slice_len, inverse):
        import torch
        t_out = torch.empty_like(t_in)
        check(L.lib().lw_hip_ntt_cross_device(field.field, field.layout, L.DIR_INVERSE if inverse else L.DIR_FORWARD,
                                              C.c_void_p(t_in.data_ptr()), C.c_void_p(t_out.data_ptr()), log2n_total, log2g,
                                              j2_begin, slice_len, slice_len, 1, 0, self._stream()))
        return t_out

    def local_ntt(self, field, t_in, log2m, inverse):
        import torch
        from . import fft
        t_out = torch.empty_like(t_in)
        fft.ntt_device(field, t_in, t_out, log2m, inverse=inverse)
        return t_out

    def local_msm(self, curve, t_scalars, t_points, n):
        from . import msm
        return msm.msm_device(curve, t_scalars, t_points, n)

    def sum_points(self, curve, pts):
        """Sum of a handful of projective points = MSM with unit scalars (runs on the same HIP path)."""
        from . import msm
        ones = np.zeros((len(pts), 4), np.uint64)
        ones[:, 3] = 1
        return msm.msm(curve, ones, np.stack(pts))


def _log2(n):
    if n <= 0 or n & (n - 1):
        raise ValueError(f"{n} is not a power of two")
    return n.bit_length() - 1


# ---------------------------------------------------------------- sharded NTT
def ntt_sharded(field, x_local, log2n_total, comm, inverse=False, backend=None, natural_output=True, batch=1, stop_after=0):
    """x_local: this rank's contiguous block of the natural-order vector, tensor [M, words] (M = N / G); `batch` > 1:
    `batch` such blocks back to back.  Returns this rank's block of the natural-order result (natural_output=True), or the
    cyclic shard X[rank + G*k2] (one exchange fewer; convenient when a bit-reverse + commit follows).

    With a HipComm this is one call into the library (csrc/comm.hip ntt_sharded_run, RCCL).  With TorchDistComm / SimComm it
    is the LITERAL TRANSLITERATION of that C++ schedule — same six steps per batch column, same chunk order in every
    exchange, same (j2_begin, slice_len) arguments to the cross step — kept so that the world-size-2 gloo test
    (tests/test_distributed_cpu.py) exercises the schedule across real processes on CPU.  `stop_after` = k returns the
    buffer step k wrote, [batch, M, ...]; tests/test_gpu_distributed.py compares it with the C++ schedule's
    (lw_hip_ntt_sharded_selftest_steps_device) after every step, so the two cannot drift apart."""
    if isinstance(comm, HipComm):   # production path: the schedule runs in C++ over the library's RCCL communicator
        import torch
        out = torch.empty_like(x_local)
        check(L.lib().lw_hip_ntt_sharded_device(field.field, field.layout, L.DIR_INVERSE if inverse else L.DIR_FORWARD,
                                                C.c_void_p(x_local.data_ptr()), C.c_void_p(out.data_ptr()), log2n_total, batch,
                                                1 if natural_output else 0, _stream_ptr()))
        return out
    import torch
    backend = backend or HipBackend()
    G, g = comm.size, comm.rank
    if G == 1:
        return backend.local_ntt(field, x_local, log2n_total, inverse)
    lg = _log2(G)
    if lg > 3:
        raise ValueError("ntt_sharded supports 2, 4 or 8 ranks")
    M = x_local.shape[0] // batch
    if (M << lg) != (1 << log2n_total) or M % G or M * batch != x_local.shape[0]:
        raise ValueError(f"local shard of {M} elements does not match 2^{log2n_total} over {G} ranks")
    sl = M // G
    tail = tuple(x_local.shape[1:])
    last_step = stop_after if stop_after else (6 if natural_output else 4)
    outs = []
    for bi in range(batch):                 # comm.hip walks the columns diagonal by diagonal on two streams; per column the
        col = x_local[bi * M:(bi + 1) * M]  # steps and their data are exactly these
        # 1 (A). all-to-all: slice h of my block goes to rank h; chunk j of the receive buffer came from rank j
        cur = comm.all_to_all(col.reshape((G, sl) + tail)).reshape((M,) + tail)
        if last_step >= 2:   # 2 (B). cross-shard step on this rank's j2 slice [g*sl, (g+1)*sl), chunks sl apart
            cur = backend.cross(field, cur, log2n_total, lg, g * sl, sl, inverse)
        if last_step >= 3:   # 3 (C). all-to-all: row k1 goes to rank k1  ->  Y[g][all j2]
            cur = comm.all_to_all(cur.reshape((G, sl) + tail)).reshape((M,) + tail)
        if last_step >= 4:   # 4 (D). local M-point transform: z[k2] = X[g + G*k2]
            cur = backend.local_ntt(field, cur, log2n_total - lg, inverse)
        if last_step >= 5:   # 5 (E). all-to-all: slice h of my cyclic shard goes to rank h  ->  [k1, k2', ...]
            cur = comm.all_to_all(cur.reshape((G, sl) + tail)).reshape((M,) + tail)
        if last_step >= 6:   # 6 (F). local interleave: natural index g*M + (k1 + G*k2')
            cur = cur.reshape((G, sl) + tail).transpose(0, 1).contiguous().reshape((M,) + tail)
        outs.append(cur)
    if stop_after:
        return torch.stack(outs)
    return outs[0] if batch == 1 else torch.cat(outs)


def ntt_sharded_selftest_steps(field, x_full, log2n_total, log2_shards, stop_after, inverse=False, natural_output=True, batch=1):
    """lw_hip_ntt_sharded_selftest_steps_device: the C++ schedule cut after step `stop_after`; block g of every batch entry
    of the result is what virtual rank g holds at that point."""
    import torch
    out = torch.empty_like(x_full)
    check(L.lib().lw_hip_ntt_sharded_selftest_steps_device(field.field, field.layout, L.DIR_INVERSE if inverse else L.DIR_FORWARD,
                                                           C.c_void_p(x_full.data_ptr()), C.c_void_p(out.data_ptr()), log2n_total,
                                                           log2_shards, batch, 1 if natural_output else 0, stop_after, _stream_ptr()))
    return out


def msm_sharded_selftest(curve, t_scalars, t_points, n_total, log2_shards):
    """lw_hip_msm_sharded_selftest_device: the bucket-slice exchange schedule of the sharded MSM (csrc/comm.hip
    msm_sharded_run) with 2^log2_shards virtual ranks on one device."""
    out = np.zeros(curve.point_words, dtype=np.uint64)
    check(L.lib().lw_hip_msm_sharded_selftest_device(curve.curve, C.c_void_p(t_scalars.data_ptr()), C.c_void_p(t_points.data_ptr()), n_total,
                                                     log2_shards, out.ctypes.data_as(C.c_void_p), _stream_ptr()))
    return out


# ---------------------------------------------------------------- sharded MSM
def msm_sharded(curve, t_scalars, t_points, n_local, comm, backend=None):
    """Every rank holds n_local (scalar, point) pairs; returns sum over all ranks' pairs on every rank."""
    import torch
    if isinstance(comm, HipComm):   # production path: local Pippenger + RCCL all-gather + final adds inside the library
        out = np.zeros(curve.point_words, dtype=np.uint64)
        check(L.lib().lw_hip_msm_sharded_device(curve.curve, C.c_void_p(t_scalars.data_ptr()), C.c_void_p(t_points.data_ptr()),
                                                n_local, out.ctypes.data_as(C.c_void_p), _stream_ptr()))
        return out
    backend = backend or HipBackend()
    part = backend.local_msm(curve, t_scalars, t_points, n_local)          # numpy projective point
    if comm.size == 1:
        return part
    t = torch.from_numpy(np.ascontiguousarray(part).view(np.int64)).to(t_scalars.device)
    parts = [p.cpu().numpy().view(np.uint64) for p in comm.all_gather(t)]
    return backend.sum_points(curve, parts)
