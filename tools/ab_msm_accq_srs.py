#!/usr/bin/env python3
"""MSMs over an lw_hip_srs handle (affine rows) at small sizes with the accumulation on one / four lanes per piece
(LW_HIP_MSM_ACCQ, read per call).  Run with LW_HIP_TUNING=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import msm
from tools.synth import distinct_points
rng = np.random.default_rng(8)
for crv, name in ((msm.BLS12381Curve, "bls12-381 g1"), (msm.BLS12381TwistCurve, "bls12-381 g2")):
    for L in (8, 10, 12, 14, 16):
        n = 1 << L
        srs = msm.Srs(crv, t_points=distinct_points(crv, n), n=n)
        sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        t = torch.from_numpy(sc.view(np.int64)).cuda()
        out = []
        for q in (0, 19, 0, 19):
            os.environ["LW_HIP_MSM_ACCQ"] = str(q)
            for _ in range(2): srs.msm_device(t, n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8): srs.msm_device(t, n)
            torch.cuda.synchronize()
            out.append("q=%d %.3f" % (q, (time.perf_counter() - t0) / 8 * 1e3))
        os.environ.pop("LW_HIP_MSM_ACCQ")
        print("%s srs 2^%d: %s" % (name, L, "  ".join(out)), flush=True)
        srs.close()
