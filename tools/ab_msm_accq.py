#!/usr/bin/env python3
"""Accumulation of small MSMs on the quad kernel (LW_HIP_MSM_ACCQ = log2 of the widest launch in lanes, 0 = none; read per
call).  Run with LW_HIP_TUNING=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import _lib, msm
from tools.synth import distinct_points
rng = np.random.default_rng(8)
for crv, name in ((msm.BLS12381Curve, "bls12-381 g1"), (msm.BN254Curve, "bn254 g1"), (msm.BN254TwistCurve, "bn254 g2"), (msm.BLS12381TwistCurve, "bls12-381 g2")):
    for L in (8, 10, 12, 14, 16, 18):
        n = 1 << L
        pts = distinct_points(crv, n)
        sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        t = torch.from_numpy(sc.view(np.int64)).cuda()
        out = []
        for q in (0, 17, 19, 21, 0, 19):
            os.environ["LW_HIP_MSM_ACCQ"] = str(q)
            for _ in range(2): msm.msm_device(crv, t, pts, n)
            torch.cuda.synchronize()
            _lib.profile_begin()
            t0 = time.perf_counter()
            for _ in range(8): msm.msm_device(crv, t, pts, n)
            torch.cuda.synchronize()
            d = (time.perf_counter() - t0) / 8 * 1e3
            prof = _lib.profile_end()
            out.append("q=%d %.3f (acc %.3f)" % (q, d, sum(v[1] for k, v in prof.items() if "accumulate" in k) / 8))
        os.environ.pop("LW_HIP_MSM_ACCQ")
        print("%s 2^%d: %s" % (name, L, "  ".join(out)), flush=True)
        del pts
