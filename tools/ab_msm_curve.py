#!/usr/bin/env python3
"""MSM wall time for one curve/size: usage ab_msm_curve.py {bls12_381_g1|bn254_g1|bn254_g2|bls12_381_g2} L [L ...]"""
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
    msm.msm_device(crv, ts, tp, n)
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(3):
        msm.msm_device(crv, ts, tp, n)
    dt = (time.perf_counter() - t0) / 3
    prof = _lib.profile_end()
    print("%s 2^%d: %.2f ms" % (sys.argv[1], L, dt * 1e3), {k: round(v[1] / max(v[0], 1), 3) for k, v in prof.items()}, flush=True)
