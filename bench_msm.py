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


# BLS12-381 G1 (y^2 = x^3 + 4 over Fp): public curve constants, used only to synthesise benchmark inputs
BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_GX = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
BLS_GY = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1


def synth_points_bls12381_g1(n, seed):
    """n distinct, non-normalised (Z != 1) projective BLS12-381 G1 points in the reference memory layout
    (3 x 6 u64 limbs, most significant first, Montgomery form R = 2^384): P_0 = [s0]G, P_i = P_{i-1} + [d]G, each
    triple re-randomised (X,Y,Z) -> (lX, lY, lZ) as SURVEY 8(d) prescribes.  Pure Python big-int affine arithmetic
    (one modular inverse per point), so the benchmark's inputs do not come from the oracle."""
    import random
    p = BLS_P
    rnd = random.Random(seed)

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    def mul(k, P):
        acc = None
        while k:
            if k & 1:
                acc = add(acc, P)
            P = add(P, P)
            k >>= 1
        return acc

    g = (BLS_GX, BLS_GY)
    cur = mul(rnd.getrandbits(200) | 1, g)
    step = mul(rnd.getrandbits(200) | 1, g)
    R = 1 << 384
    out = np.empty((n, 18), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i in range(n):
        lam = rnd.getrandbits(380) | 1
        for k, v in enumerate((cur[0] * lam % p, cur[1] * lam % p, lam % p)):
            m = v * R % p
            for j in range(6):
                out[i, 6 * k + j] = (m >> (64 * (5 - j))) & mask
        cur = add(cur, step)
    return out


def run_msm_leg(args, world, rank, barrier, max_over_ranks):
    import torch
    import torch.distributed as dist
    from lambda_elliptic_curves_amd import _lib, msm

    L = args.msm_log2n
    n = 1 << L
    crv = msm.BLS12381Curve
    # points: a host-generated SRS-like run of 2^16 distinct non-normalised points, tiled to n (scalars differ per slot)
    base_n = min(n, 1 << 16)
    base_pts = synth_points_bls12381_g1(base_n, 0x5EED + rank)
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

    mode = {"sharded": world > 1}

    def step():
        if mode["sharded"]:   # per-rank Pippenger + all_gather of the partial sums + final adds
            return D.msm_sharded(crv, t_sc, t_pts, n, comm)
        return msm.msm_device(crv, t_sc, t_pts, n)

    try:
        for _ in range(warm):
            out = step()
    except Exception as e:   # keep the run measurable if the collective is unavailable: independent per-rank MSMs
        if not mode["sharded"]:
            raise
        mode["sharded"] = False
        mode["note"] = "all_gather combine failed (%s); ranks ran independent MSMs" % str(e)[:120]
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
        # CPU baseline: the oracle's restatement of the reference's sequential msm() (pippenger.rs:18-103), one thread.
        # This leg is the only place the benchmark touches oracle/.
        from oracle import oracle as O
        from tests import util
        oid = O.C_BLS12_381_G1
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
    # additions the device's accumulate kernels perform: one per (point, window) item minus one per non-empty bucket
    dev_c = min(max(L - 4, 4), 16)
    dev_w = (256 + dev_c - 1) // dev_c
    dev_adds = max(n * dev_w - dev_w * ((1 << dev_c) - 1), 0)
    mixed = L >= 22 and os.environ.get("LW_HIP_MSM_NORMALIZE", "1") != "0"
    mac_pairs = 2736 if mixed else 3024
    return {
        "metric": "MSM G1 point-adds/sec (BLS12-381, 2^%d points, reference add count adds_ref(N))" % L,
        "value": world * adds_ref(n) * steps / dt, "unit": "point-adds/s",
        "points_per_s": world * n * steps / dt, "ms_per_step": dt * 1e3 / steps, "steps": steps,
        "adds_ref": adds_ref(n), "n_gpus": world,
        "config": {"workload": "BLS12-381 G1 Pippenger MSM, 2^%d points per GPU, inputs resident in HBM" % L,
                   "curve": "BLS12-381 G1", "log2n": L,
                   "parallelism": ("single" if world == 1 else ("points sharded, partial sums all-gathered" if mode["sharded"]
                                                                   else mode.get("note", "independent")))},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (dt / steps) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg_bytes / (dt / steps) / 1e9 / 8000.0, "traffic": None,
                     "kernel": "msm_accumulate_kernel", "avg_launch_ms": (a_launch[1] / a_launch[0]) if a_launch[0] else None,
                     "note": "whole-MSM algorithmic bytes N*(32+144) over the step time; the MSM is integer-VALU bound by ~2 orders of magnitude"},
        # the bound that applies: complete additions/s of the accumulate kernels against the MAC-pair rate (17.0 T
        # pairs/s measured, profiles/r01_microbench.txt).  From 2^22 points the library normalises the inputs first and
        # accumulates with the mixed addition (19 N^2 = 2736 MAC pairs); below that the projective one (21 N^2 = 3024).
        "valu": {"accumulate_adds_per_s": dev_adds * steps / (acc_ms * 1e-3) if acc_ms else None,
                 "peak_adds_per_s": 17.0e12 / mac_pairs, "unit": "complete point additions/s",
                 "frac": (dev_adds * steps / (acc_ms * 1e-3)) / (17.0e12 / mac_pairs) if acc_ms else None,
                 "device_adds_per_msm": dev_adds, "device_window_bits": dev_c, "mac_pairs_per_addition": mac_pairs},
        "kernel_times_ms": {k: {"launches": v[0], "avg_ms": v[1] / max(v[0], 1)} for k, v in prof.items()},
        "cpu_baseline": cpu,
    }
