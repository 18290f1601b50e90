#!/usr/bin/env python3
"""MSM wall time and kernel breakdown over sizes for the G2 groups (the slowest MSM of a Groth16 proof): usage msm_sizes_g2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import _lib, msm
from tools.synth import distinct_points
rng = np.random.default_rng(8)
for crv, name in ((msm.BN254TwistCurve, "bn254 g2"), (msm.BLS12381TwistCurve, "bls12-381 g2")):
    for L in (12, 14, 16, 18, 20):
        n = 1 << L
        pts = distinct_points(crv, n)
        sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        t = torch.from_numpy(sc.view(np.int64)).cuda()
        for _ in range(2): msm.msm_device(crv, t, pts, n)
        torch.cuda.synchronize()
        _lib.profile_begin()
        t0 = time.perf_counter()
        R = 5
        for _ in range(R): msm.msm_device(crv, t, pts, n)
        torch.cuda.synchronize()
        d = (time.perf_counter() - t0) / R
        prof = _lib.profile_end()
        print("%s 2^%d: %.3f ms" % (name, L, d * 1e3), {k.replace("msm_", "").replace("_kernel", ""): round(v[1] / R, 3) for k, v in prof.items() if v[1] / R > 0.02}, flush=True)
        del pts
