#!/usr/bin/env python3
"""MSM wall time with scalars below 2^255 (what callers pass) and uniform over 2^256 (the API allows it): usage ab_msm_unreduced.py L [L ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import msm
from tools.synth import distinct_points
crv = msm.BLS12381Curve
rng = np.random.default_rng(5)
for L in map(int, sys.argv[1:]):
    n = 1 << L
    tp = distinct_points(crv, n)
    lo = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    full = lo * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    for name, sc in (("< 2^255", lo), ("uniform 2^256", full)):
        ts = torch.from_numpy(sc.view(np.int64)).cuda()
        msm.msm_device(crv, ts, tp, n)
        t0 = time.perf_counter()
        for _ in range(5):
            msm.msm_device(crv, ts, tp, n)
        print("2^%d scalars %s: %.3f ms" % (L, name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
