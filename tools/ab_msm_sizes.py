#!/usr/bin/env python3
"""MSM wall time per size for the current environment (LW_HIP_MSM_C etc.): usage ab_msm_sizes.py L [L ...]"""
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
    ts = torch.from_numpy(rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64).view(np.int64)).cuda()
    msm.msm_device(crv, ts, tp, n)
    t0 = time.perf_counter()
    for _ in range(5):
        msm.msm_device(crv, ts, tp, n)
    print("c=%s 2^%d: %.3f ms" % (os.environ.get("LW_HIP_MSM_C", "auto"), L, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
