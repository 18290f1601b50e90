#!/usr/bin/env python3
"""Window width at mid sizes, per group (LW_HIP_MSM_C is read per call): c = 8 / 13 / 16 (the widths whose top window is empty
or well filled) for 2^12 .. 2^20 points.  Run with LW_HIP_TUNING=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import msm
from tools.synth import distinct_points
rng = np.random.default_rng(8)
for crv, name, r in ((msm.BLS12381Curve, "bls12-381 g1", 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001),
                     (msm.BN254TwistCurve, "bn254 g2", 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001),
                     (msm.BLS12381TwistCurve, "bls12-381 g2", 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001)):
    for L in (12, 14, 16, 18, 20):
        n = 1 << L
        pts = distinct_points(crv, n)
        # uniform scalars below r (what the reference's callers pass): top limb below r's
        sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
        sc[:, 0] %= np.uint64(r >> 192)
        t = torch.from_numpy(sc.view(np.int64)).cuda()
        out = []
        for c in (8, 13, 16):
            os.environ["LW_HIP_MSM_C"] = str(c)
            for _ in range(2): msm.msm_device(crv, t, pts, n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): msm.msm_device(crv, t, pts, n)
            torch.cuda.synchronize()
            out.append("c=%d %.3f ms" % (c, (time.perf_counter() - t0) / 5 * 1e3))
        os.environ.pop("LW_HIP_MSM_C")
        print("%s 2^%d: %s" % (name, L, "   ".join(out)), flush=True)
        del pts
