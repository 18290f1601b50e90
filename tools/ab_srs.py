#!/usr/bin/env python3
"""MSM over a device-resident SRS (lw_hip_msm_srs_device) with per-kernel times: usage ab_srs.py CURVE L [L ...]
(LW_HIP_SRS_FOLD=0: single affine copy; default: 13 window-shifted copies from 2^19 points)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import msm, _lib
from tools.synth import distinct_points
crv = {"bls12_381_g1": msm.BLS12381Curve, "bn254_g1": msm.BN254Curve, "bn254_g2": msm.BN254TwistCurve, "bls12_381_g2": msm.BLS12381TwistCurve}[sys.argv[1]]
rng = np.random.default_rng(5)
for L in map(int, sys.argv[2:]):
    n = 1 << L
    tp = distinct_points(crv, n)
    ts = torch.from_numpy(rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64).view(np.int64)).cuda()
    t0 = time.perf_counter()
    srs = msm.Srs(crv, t_points=tp, n=n)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    del tp
    srs.msm_device(ts, n)
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(3):
        srs.msm_device(ts, n)
    dt = (time.perf_counter() - t0) / 3
    prof = _lib.profile_end()
    print("fold=%s %s 2^%d: %.2f ms (SRS build %.0f ms)" % (os.environ.get("LW_HIP_SRS_FOLD", "1"), sys.argv[1], L, dt * 1e3, t_build * 1e3),
          {k: round(v[1] / max(v[0], 1), 3) for k, v in prof.items()}, flush=True)
    srs.close()
