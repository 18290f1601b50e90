#!/usr/bin/env python3
"""The workloads bench.py does NOT run, a few launches each, so that `rocprofv3 --kernel-trace --stats` / `--pmc`
over this script gives kernel rows for them: BabyBear (three layouts), BLS12-381 Fr, BN254 G1/G2, BLS12-381 G2,
the Merkle commit, the FRI layer and the Groth16 h pipeline.  No checks here (tests/ does that); inputs are random
canonical residues / tiled valid points, sizes = the per-GPU sizes of BASELINE.json's configs.
usage: profile_workloads.py [ntt|msm|aux|all]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    import torch
    from lambda_elliptic_curves_amd import fft, merkle, groth16, msm
    import bench_msm
    P_BB = 2013265921
    rng = np.random.default_rng(1)
    reps = 3

    def t(a):
        return torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()

    def elems256(n, top_bits):
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        a[:, 0] &= np.uint64((1 << top_bits) - 1)
        return a

    if which in ("ntt", "all"):
        L, B = 24, 4
        for fld, a in ((fft.Babybear31PrimeFieldU32, rng.integers(0, P_BB, size=B << L, dtype=np.uint32)),
                       (fft.Babybear31PrimeField, rng.integers(0, P_BB, size=B << L, dtype=np.uint64))):
            x = t(a); y = torch.empty_like(x)
            for _ in range(reps):
                fft.ntt_device(fld, x, y, L, batch=B)
                fft.ntt_device(fld, y, y, L, inverse=True, batch=B)
            torch.cuda.synchronize()
        # BASELINE config 4 as the LDE it is: 2^22 coefficients per column -> 2^24 evaluations on a coset (lw_hip_ntt_lde_device)
        x = t(rng.integers(0, P_BB, size=B << (L - 2), dtype=np.uint32)); y = torch.empty(B << L, dtype=torch.int32, device="cuda")
        off = np.array([268435454 * 3 % P_BB], dtype=np.uint32)
        for _ in range(reps):
            fft.lde_device(fft.Babybear31PrimeFieldU32, x, L - 2, y, L, batch=B, offset=off)
        torch.cuda.synchronize()
        a = rng.integers(0, P_BB, size=(1 << L, 4), dtype=np.uint64)
        x = t(a); y = torch.empty_like(x)
        for _ in range(reps):
            fft.ntt_device(fft.Degree4BabyBearExtensionField, x, y, L)
        torch.cuda.synchronize()
        x = t(elems256(1 << L, 62)); y = torch.empty_like(x)
        for _ in range(reps):
            fft.ntt_device(fft.FrField, x, y, L)
            fft.ntt_device(fft.FrField, y, y, L, inverse=True)
        torch.cuda.synchronize()
        x = t(elems256(1 << L, 59)); y = torch.empty_like(x)
        for _ in range(reps):
            fft.ntt_device(fft.Stark252PrimeField, x, y, L, inverse=True)
        torch.cuda.synchronize()
        del x, y
        torch.cuda.empty_cache()

    if which in ("msm", "all"):
        from tools.synth import distinct_points
        for crv, L in ((msm.BN254Curve, 23), (msm.BN254TwistCurve, 22), (msm.BLS12381TwistCurve, 20), (msm.BLS12381Curve, 20)):
            n = 1 << L
            pts = distinct_points(crv, n)
            sc = t(rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64))
            for _ in range(2):
                msm.msm_device(crv, sc, pts, n)
            del pts, sc
            torch.cuda.empty_cache()

    if which in ("aux", "all"):
        L = 22
        cols = t(elems256(4 << L, 59))
        nodes = torch.empty(((2 << L) - 1) * 4, dtype=torch.int64, device="cuda")
        for _ in range(reps):
            merkle.commit_columns_device(fft.Stark252PrimeField, cols, 4, L, nodes)
        torch.cuda.synchronize()
        del cols, nodes
        coeffs = elems256(1 << 20, 59)
        one = np.zeros(4, np.uint64); one[3] = 5
        for _ in range(reps):
            merkle.fri_layer(fft.Stark252PrimeField, coeffs, one, one, 1 << 21)
        g = 1 << 18
        l, r, o = elems256(g, 62), elems256(g, 62), elems256(g, 62)
        for _ in range(reps):
            groth16.calculate_h_coefficients(l, r, o, g)
        # the device-resident forms: one FRI layer with everything in HBM, the Groth16 quotient at 2^20 gates
        t_co = t(coeffs)
        for _ in range(reps):
            merkle.fri_layer_device(fft.Stark252PrimeField, t_co, 1 << 20, one, one, 1 << 21)
        g = 1 << 20
        tl, tr, to = (t(elems256(g, 62)) for _ in range(3))
        for _ in range(reps):
            groth16.calculate_h_coefficients_device(tl, tr, to, g, g)
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
