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


# BLS12-381 G1 (y^2 = x^3 + 4 over Fp) and its scalar field: public curve constants, used only to synthesise inputs
BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_GX = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
BLS_GY = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
BLS_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def _aff_add(P, Q, p=BLS_P):
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


def _aff_mul(k, P):
    acc = None
    while k:
        if k & 1:
            acc = _aff_add(acc, P)
        P = _aff_add(P, P)
        k >>= 1
    return acc


def synth_run_bls12381_g1(n, start, step, seed):
    """n projective BLS12-381 G1 points [start + i*step]G, i < n, in the reference memory layout (3 x 6 u64 limbs, most
    significant first, Montgomery form R = 2^384), each triple re-randomised (X,Y,Z) -> (lX, lY, lZ) so Z != 1
    (SURVEY 8(d)); [0]G is the identity (0 : l : 0).  Pure Python big-int affine arithmetic, so the benchmark's inputs
    do not come from the oracle."""
    import random
    p = BLS_P
    rnd = random.Random(seed)
    g = (BLS_GX, BLS_GY)
    cur = _aff_mul(start, g) if start else None
    stp = _aff_mul(step, g)
    R = 1 << 384
    out = np.empty((n, 18), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i in range(n):
        lam = rnd.getrandbits(380) | 1
        trip = (cur[0] * lam % p, cur[1] * lam % p, lam % p) if cur is not None else (0, lam % p, 0)
        for k, v in enumerate(trip):
            m = v * R % p
            for j in range(6):
                out[i, 6 * k + j] = (m >> (64 * (5 - j))) & mask
        cur = _aff_add(cur, stp)
    return out


def synth_points_bls12381_g1(n, seed):
    """n distinct, non-normalised BLS12-381 G1 points P_i = [s0 + i*d]G (small n; see synth_points_device for 2^24)."""
    import random
    rnd = random.Random(seed)
    return synth_run_bls12381_g1(n, rnd.getrandbits(200) | 1, rnd.getrandbits(200) | 1, seed + 1)


def synth_points_device(n, seed):
    """n DISTINCT BLS12-381 G1 points on the device: P_{j*m + i} = [s0 + i*d]G + [j*m*d]G = [s0 + (j*m + i)*d]G, the
    run SURVEY 8(d) prescribes (P_i = P_{i-1} + [d]G), built from two short host-made runs (m and n/m points) with
    one batched group addition of the library (lw_hip_ec_add_outer_device).  The sums come out of the complete
    projective addition, so Z != 1.  Returns (tensor [n, 18] int64, rows, cols)."""
    import ctypes as C
    import random
    import torch
    from lambda_elliptic_curves_amd import _lib
    from lambda_elliptic_curves_amd.errors import check
    rnd = random.Random(seed)
    s0, d = rnd.getrandbits(200) | 1, rnd.getrandbits(200) | 1
    m = 1
    while m * m < n:
        m <<= 1
    m = min(m, n)
    k = n // m
    rows = synth_run_bls12381_g1(m, s0, d, seed + 1)
    cols = synth_run_bls12381_g1(k, 0, m * d, seed + 2)
    t_rows = torch.from_numpy(rows.view(np.int64)).cuda()
    t_cols = torch.from_numpy(cols.view(np.int64)).cuda()
    out = torch.empty((n, 18), dtype=torch.int64, device="cuda")
    check(_lib.lib().lw_hip_ec_add_outer_device(_lib.CURVE_BLS12_381_G1, C.c_void_p(t_rows.data_ptr()), m, C.c_void_p(t_cols.data_ptr()), k,
                                                C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out, rows, cols


def synth_scalars_mod_r(n, seed):
    """n uniform 256-bit integers reduced mod r (canonical U256, MS limb first), as math/benches/criterion_msm.rs:17-32
    draws them.  2^256 / r < 2.21, so at most two subtractions of r."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    r = np.array([(BLS_R >> (64 * (3 - k))) & ((1 << 64) - 1) for k in range(4)], dtype=np.uint64)
    for _ in range(2):
        ge = np.ones(n, dtype=bool)            # a >= r, decided from the most significant limb down
        decided = np.zeros(n, dtype=bool)
        for k in range(4):
            gt, lt = a[:, k] > r[k], a[:, k] < r[k]
            ge = np.where(~decided & lt, False, ge)
            decided |= gt | lt
        borrow = np.zeros(n, dtype=np.uint64)
        for k in (3, 2, 1, 0):
            x = a[:, k]
            y = x - r[k]
            b1 = (x < r[k]).astype(np.uint64)
            z = y - borrow
            b2 = (y < borrow).astype(np.uint64)
            a[:, k] = np.where(ge, z, x)
            borrow = b1 | b2
    return a


def run_msm_leg(args, world, rank, barrier, max_over_ranks, all_ranks_ok, comm, comm_error):
    import torch
    from lambda_elliptic_curves_amd import _lib, msm
    from lambda_elliptic_curves_amd import distributed as D

    L = args.msm_log2n
    n = 1 << L
    crv = msm.BLS12381Curve
    t_pts, rows, cols = synth_points_device(n, 0x5EED + rank)
    scalars = synth_scalars_mod_r(n, 42 + 1000 * rank)
    t_sc = torch.from_numpy(scalars.view(np.int64)).cuda()
    steps = max(1, min(args.steps, 5))
    warm = max(1, min(args.warmup, 2))
    sharded = world > 1 and comm is not None
    note = comm_error

    def step():
        if sharded:   # per-rank Pippenger + RCCL all-gather of the partial sums + final adds, inside the library
            return D.msm_sharded(crv, t_sc, t_pts, n, comm)
        return msm.msm_device(crv, t_sc, t_pts, n)

    ok, err = True, None
    try:
        for _ in range(warm):
            out = step()
    except Exception as e:
        ok, err = False, "%s: %s" % (type(e).__name__, str(e)[:160])
    if not all_ranks_ok(ok):
        if not sharded:
            raise RuntimeError("MSM warm-up failed: %s" % err)
        sharded, note = False, (err or "another rank failed in the sharded warm-up")   # all ranks switch together
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

    cpu = cpu_all = bit_exact = host_path = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The only place this leg touches oracle/: CPU baselines on the bench's own inputs, compared with the GPU results.
        from oracle import oracle as O
        oid = O.C_BLS12_381_G1
        aff = lambda p: O.point_to_affine_ints(oid, p)
        pts_host = t_pts.cpu().numpy().view(np.uint64)
        # the synthesised points are what they claim: spot-check the batched addition against the oracle's group law
        m = rows.shape[0]
        spots_ok = all(O.ec_eq(oid, pts_host[i], O.ec_add(oid, rows[i % m], cols[i // m])) for i in (0, 1, m, n // 2 + 3, n - 1))
        # (i) the reference's sequential msm(), one thread, on the first 2^18 pairs; the GPU runs the same prefix
        Lc = min(18, L)
        nc = 1 << Lc
        tc0 = time.perf_counter()
        ref_c = O.msm(oid, scalars[:nc], pts_host[:nc])
        dtc = time.perf_counter() - tc0
        got_c = msm.msm_device(crv, t_sc[:nc], t_pts[:nc], nc)
        prefix_ok = aff(got_c) == aff(ref_c)
        cpu = {"value": adds_ref(nc) / dtc, "unit": "point-adds/s", "cores": 1, "kind": "port",
               "sample": "one BLS12-381 G1 msm() of the first 2^%d bench pairs, oracle/lw_oracle.c (sequential Pippenger, reference "
                         "window rule), %.1f s" % (Lc, dtc), "points_per_s": nc / dtc}
        # (ii) context row C4: window-parallel msm (pippenger.rs:109-161) on all host cores over the WHOLE input when the
        # box has the cores for it (<= W threads are useful), else over a 2^20 prefix
        cores = os.cpu_count() or 1
        Lv = L if cores >= 12 else min(L, 20)
        nv = 1 << Lv
        win = max(2, O.optimum_window_size(nv))
        thr = max(1, min(cores, (255 // win) + 1))
        tv0 = time.perf_counter()
        ref_v = O.parallel_msm_with(oid, scalars[:nv], pts_host[:nv], win, thr)
        dtv = time.perf_counter() - tv0
        got_v = out if nv == n else msm.msm_device(crv, t_sc[:nv], t_pts[:nv], nv)
        full_ok = aff(got_v) == aff(ref_v)
        cpu_all = {"value": adds_ref(nv) / dtv, "unit": "point-adds/s", "cores": thr, "kind": "port",
                   "sample": "parallel_msm_with over %s 2^%d bench pairs, window %d, %d threads, %.1f s"
                             % ("all" if nv == n else "the first", Lv, win, thr, dtv), "points_per_s": nv / dtv}
        bit_exact = {"msm": bool(full_ok and spots_ok), "msm_prefix": bool(prefix_ok),
                     "msm_check": "affine image of the %s GPU result == oracle parallel_msm_with on the same 2^%d pairs; "
                                  "prefix 2^%d == oracle msm(); 5 synthesised points == oracle ec_add of their two summands"
                                  % ("timed" if nv == n else "prefix", Lv, Lc)}
        del pts_host
    if rank == 0 and world == 1 and not args.no_host_path:
        Lh = min(L, 22)
        nh = 1 << Lh
        ph = t_pts[:nh].cpu().numpy().view(np.uint64)
        sh = scalars[:nh]
        msm.msm(crv, sh, ph)
        th0 = time.perf_counter()
        msm.msm(crv, sh, ph)
        dth = time.perf_counter() - th0
        host_path = {"ms": dth * 1e3, "points_per_s": nh / dth,
                     "what": "lw_hip_msm on host buffers, BLS12-381 G1 2^%d: H2D of %d MiB + MSM" % (Lh, nh * 176 >> 20)}

    srs_path = None
    if rank == 0 and world == 1 and not args.no_host_path:
        # the shape the reference's callers have (a fixed SRS / proving key, kzg.rs:159-163, groth16 prover.rs:69-85): points
        # resident and pre-normalised in an lw_hip_srs handle (window-shifted copies from 2^19 points), scalars on the device
        ts0 = time.perf_counter()
        srs = msm.Srs(crv, t_points=t_pts, n=n)
        torch.cuda.synchronize()
        t_create = time.perf_counter() - ts0
        got_s = srs.msm_device(t_sc, n)
        ts0 = time.perf_counter()
        for _ in range(3):
            got_s = srs.msm_device(t_sc, n)
        dts = (time.perf_counter() - ts0) / 3
        srs.close()
        same = bool(np.array_equal(np.asarray(got_s), np.asarray(out)))
        srs_path = {"ms": dts * 1e3, "points_per_s": n / dts, "srs_create_ms": t_create * 1e3, "equals_timed_result": same,
                    "what": "lw_hip_msm_srs_device over the same 2^%d pairs: lw_hip_srs handle built once, MSM per call" % L}
        if not same:
            bit_exact = dict(bit_exact or {}, msm=False, srs_mismatch=True)

    acc_names = [k for k in prof if k.startswith("msm_accumulate") or k.startswith("msm_batch")]
    acc_ms = sum(prof[k][1] for k in acc_names)
    dom = max(prof.items(), key=lambda kv: kv[1][1]) if prof else ("", (0, 0.0))
    alg_bytes = n * (32 + 144)
    from bench import traffic_entry, pmc_counter
    traffic, traffic_src = traffic_entry("msm", L)
    # additions the device's accumulation performs: one per (point, window) item minus one per non-empty bucket
    # (signed digits, csrc/msm.hip: W = ceil(257/c) windows of 2^(c-1) buckets; csrc/msm_core.cuh pick_window)
    dev_c = int(os.environ.get("LW_HIP_MSM_C", 0)) or (8 if L < 15 else 16 if L < 23 else 20)
    dev_w = (256 + dev_c) // dev_c
    dev_items = n * (dev_w - 1) if 256 % dev_c == 0 else n * dev_w      # the window at bit 256 is empty for scalars below 2^255
    dev_adds = max(dev_items - dev_w * (1 << (dev_c - 1)), 0)
    return {
        "metric": "MSM G1 point-adds/sec (BLS12-381, 2^%d distinct points, reference add count adds_ref(N))" % L,
        "value": world * adds_ref(n) * steps / dt, "unit": "point-adds/s",
        "points_per_s": world * n * steps / dt, "ms_per_step": dt * 1e3 / steps, "steps": steps,
        "adds_ref": adds_ref(n), "n_gpus": world,
        "config": {"workload": "BLS12-381 G1 Pippenger MSM, 2^%d points per GPU, inputs resident in HBM" % L,
                   "curve": "BLS12-381 G1", "log2n": L,
                   "inputs": "P_i = [s0 + i*d]G, all distinct, projective with Z != 1; scalars uniform 256-bit reduced mod r",
                   "parallelism": ("single" if world == 1 else ("points sharded, partial sums all-gathered over the library's RCCL "
                                                                   "communicator" if sharded else "independent")),
                   "sharded_error": note},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (dt / steps) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg_bytes / (dt / steps) / 1e9 / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": dom[0], "avg_launch_ms": (dom[1][1] / dom[1][0]) if dom[1][0] else None,
                     "note": "whole-MSM algorithmic bytes N*(32+144) over the step time; traffic = HBM bytes of all MSM kernels of one "
                             "step (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE); the MSM is integer-VALU bound by ~2 orders of magnitude"},
        "valu": {"accumulate_adds_per_s": dev_adds * steps / (acc_ms * 1e-3) if acc_ms else None,
                 "unit": "bucket additions/s", "device_adds_per_msm": dev_adds, "device_window_bits": dev_c,
                 "accumulate_ms_per_msm": acc_ms / steps if acc_ms else None,
                 # the first-round kernel over the normalised points (template arguments <curve, waves, AFFINE = true>)
                 "valu_busy_pmc": pmc_counter("msm_accumulate_kernel<lw::Bls12381G1Iso, 2, true>", "VALUBusy"),
                 "valu_busy_kernel": "msm_accumulate_kernel<lw::Bls12381G1Iso, 2, true> (first round, normalised points on the "
                                     "isomorphic curve) in the newest profiles/rNN_pmc_summary.csv"},
        "kernel_times_ms": {k: {"launches": v[0], "avg_ms": v[1] / max(v[0], 1)} for k, v in prof.items()},
        "cpu_baseline": cpu, "cpu_all_cores": cpu_all, "bit_exact": bit_exact, "host_path": host_path, "srs_path": srs_path,
    }
