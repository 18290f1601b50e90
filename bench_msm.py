"""MSM leg of bench.py: BLS12-381 G1 Pippenger MSM on device-resident inputs (Groth16 commit path)."""
import os
import time

import numpy as np


def adds_ref(n, num_limbs=4):
    """Point additions the reference's msm() performs for n points (math/src/msm/pippenger.rs:34-40,51-57,66-98):
    c = floor(log2 n)*4/5 clamped to [2,32], W = (64*NUM_LIMBS-1)/c + 1, adds = n*W + 2*W*(2^c - 1)."""
    lg = n.bit_length() - 1 if n else 0
    c = min(max((lg * 4) // 5, 2), 32)
    W = (64 * num_limbs - 1) // c + 1
    return n * W + 2 * W * ((1 << c) - 1)


def run_msm_leg(args, world, rank, barrier, max_over_ranks):
    import torch
    import torch.distributed as dist
    from lambda_elliptic_curves_amd import _lib, msm
    from oracle import oracle as O          # input generation (point set) and the cross-rank check only
    from tests import util

    L = args.msm_log2n
    n = 1 << L
    oid = O.C_BLS12_381_G1
    crv = msm.BLS12381Curve
    # points: a host-generated SRS-like run of 2^16 distinct non-normalised points, tiled to n (scalars differ per slot)
    base_n = min(n, 1 << 16)
    _, base_pts = util.msm_case(oid, base_n, 0x5EED + rank)
    rng = np.random.default_rng(42 + 1000 * rank)
    scalars = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    # reduce mod r is not required by the API (any 256-bit integer is legal, pippenger.rs doc); keep them < 2^255
    scalars[:, 0] &= np.uint64((1 << 63) - 1)
    t_pts = torch.from_numpy(base_pts.view(np.int64)).cuda().repeat(n // base_n, 1)
    t_sc = torch.from_numpy(scalars.view(np.int64)).cuda()
    steps = max(1, min(args.steps, 5))
    warm = max(1, min(args.warmup, 2))
    from lambda_elliptic_curves_amd import distributed as D
    comm = D.TorchDistComm() if world > 1 else None

    def step():
        if world > 1:   # per-rank Pippenger + all_gather of the partial sums + final adds
            return D.msm_sharded(crv, t_sc, t_pts, n, comm)
        return msm.msm_device(crv, t_sc, t_pts, n)

    for _ in range(warm):
        out = step()
    barrier()
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = _lib.profile_end()
    dt = max_over_ranks(dt)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle's restatement of the reference's sequential msm() (pippenger.rs:18-103), one thread
        Lc = 18
        sc_c, pts_c = util.msm_case(oid, 1 << Lc, 0x5EED)
        tc0 = time.perf_counter()
        O.msm(oid, sc_c, pts_c)
        dtc = time.perf_counter() - tc0
        cpu = {"value": adds_ref(1 << Lc) / dtc, "unit": "point-adds/s", "cores": 1, "kind": "port",
               "sample": "one BLS12-381 G1 msm() of 2^%d points, oracle/lw_oracle.c (sequential Pippenger, reference window "
                         "rule), %.1f s" % (Lc, dtc), "points_per_s": (1 << Lc) / dtc}
    acc_l = sum(v[0] for k, v in prof.items() if k.startswith("msm_accumulate_kernel"))
    acc_ms = sum(v[1] for k, v in prof.items() if k.startswith("msm_accumulate_kernel"))
    a_launch = prof.get("msm_accumulate_kernel", prof.get("msm_accumulate_kernel<final>", (0, 0.0)))
    alg_bytes = n * (32 + 144)
    return {
        "metric": "MSM G1 point-adds/sec (BLS12-381, 2^%d points, reference add count adds_ref(N))" % L,
        "value": world * adds_ref(n) * steps / dt, "unit": "point-adds/s",
        "points_per_s": world * n * steps / dt, "ms_per_step": dt * 1e3 / steps, "steps": steps,
        "adds_ref": adds_ref(n), "n_gpus": world,
        "config": {"workload": "BLS12-381 G1 Pippenger MSM, 2^%d points per GPU, inputs resident in HBM" % L,
                   "curve": "BLS12-381 G1", "log2n": L},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (dt / steps) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg_bytes / (dt / steps) / 1e9 / 8000.0, "traffic": None,
                     "kernel": "msm_accumulate_kernel", "avg_launch_ms": (a_launch[1] / a_launch[0]) if a_launch[0] else None,
                     "note": "whole-MSM algorithmic bytes N*(32+144) over the step time; the MSM is integer-VALU bound by ~2 orders of magnitude"},
        "kernel_times_ms": {k: {"launches": v[0], "avg_ms": v[1] / max(v[0], 1)} for k, v in prof.items()},
        "cpu_baseline": cpu,
    }
