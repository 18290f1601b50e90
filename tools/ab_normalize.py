#!/usr/bin/env python3
"""Standalone time of the projective -> affine normalisation (lw_hip_srs_create_device) for the current environment
(LW_HIP_MSM_CHK = points per work-item): usage ab_normalize.py CURVE L [L ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lambda_elliptic_curves_amd import msm, _lib
from tools.synth import distinct_points
crv = {"bls12_381_g1": msm.BLS12381Curve, "bn254_g1": msm.BN254Curve, "bn254_g2": msm.BN254TwistCurve, "bls12_381_g2": msm.BLS12381TwistCurve}[sys.argv[1]]
for L in map(int, sys.argv[2:]):
    n = 1 << L
    tp = distinct_points(crv, n)
    msm.Srs(crv, t_points=tp, n=n).close()
    _lib.profile_begin()
    for _ in range(5):
        msm.Srs(crv, t_points=tp, n=n).close()
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    print("chk=%s %s 2^%d:" % (os.environ.get("LW_HIP_MSM_CHK", "auto"), sys.argv[1], L), {k: round(v[1] / max(v[0], 1), 3) for k, v in prof.items()}, flush=True)
