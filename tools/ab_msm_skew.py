#!/usr/bin/env python3
"""MSM wall time on a skewed (witness-like) scalar distribution vs uniform, BLS12-381 G1: usage ab_msm_skew.py L"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import msm
from tools.synth import distinct_points
crv = msm.BLS12381Curve
L = int(sys.argv[1]); n = 1 << L
rng = np.random.default_rng(5)
tp = distinct_points(crv, n)
uni = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
kind = rng.random(n)
small = np.zeros((n, 4), np.uint64); small[:, 3] = rng.integers(0, 2, size=n, dtype=np.uint64)
skew = np.ascontiguousarray(np.where((kind < 0.7)[:, None], small, uni))
same = np.ascontiguousarray(np.tile(uni[:1], (n, 1)))
for name, sc in (("uniform", uni), ("70% of scalars in {0,1}", skew), ("all scalars equal", same)):
    ts = torch.from_numpy(sc.view(np.int64)).cuda()
    msm.msm_device(crv, ts, tp, n)
    t0 = time.perf_counter()
    for _ in range(3):
        msm.msm_device(crv, ts, tp, n)
    print("2^%d %s: %.2f ms" % (L, name, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
