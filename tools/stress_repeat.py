#!/usr/bin/env python3
"""Repeat the same MSM / NTT many times and require bit-identical results (scheduling-dependent races in the sort, the
piece order or the partial-sum rounds would show up as sporadic differences): usage stress_repeat.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import msm, fft
from tools.synth import distinct_points
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(11)
bad = 0
for crv, L in ((msm.BLS12381Curve, 24), (msm.BLS12381Curve, 20), (msm.BLS12381Curve, 16), (msm.BN254Curve, 23), (msm.BN254TwistCurve, 21)):
    n = 1 << L
    tp = distinct_points(crv, n)
    for kind in ("uniform", "skewed"):
        sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
        if kind == "skewed":
            small = rng.random(n) < 0.7
            sc[small, :3] = 0
            sc[small, 3] = rng.integers(0, 3, size=int(small.sum()), dtype=np.uint64)
        ts = torch.from_numpy(sc.view(np.int64)).cuda()
        ref = msm.msm_device(crv, ts, tp, n)
        diff = sum(0 if np.array_equal(msm.msm_device(crv, ts, tp, n), ref) else 1 for _ in range(reps))
        print("%s 2^%d %s: %d/%d repeats differ" % (crv.name if hasattr(crv, "name") else crv, L, kind, diff, reps), flush=True)
        bad += diff
    del tp
def rand_elems(name, n):
    if name == "babybear_u32":
        return rng.integers(0, 2013265921, size=n, dtype=np.uint32)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 0] &= np.uint64((1 << (59 if name == "stark252" else 62)) - 1)
    return a


fp = {"stark252": fft.Stark252PrimeField, "babybear_u32": fft.Babybear31PrimeFieldU32, "fr381": fft.FrField}
for name, L, batch in (("stark252", 24, 1), ("babybear_u32", 24, 4), ("fr381", 22, 1)):
    fld = fp[name]
    a = rand_elems(name, (1 << L) * batch)
    t_in = torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()
    t_out = torch.empty_like(t_in)
    fft.ntt_device(fld, t_in, t_out, L, batch=batch)
    ref = t_out.clone()
    diff = 0
    for _ in range(reps):
        fft.ntt_device(fld, t_in, t_out, L, batch=batch)
        diff += 0 if torch.equal(t_out, ref) else 1
    print("%s 2^%d x %d: %d/%d repeats differ" % (name, L, batch, diff, reps), flush=True)
    bad += diff
sys.exit(1 if bad else 0)
