#!/usr/bin/env python3
"""Wall time per call of the device entry points on small inputs (launch- and host-bound): NTT 2^8 .. 2^16, coset NTT,
Merkle commit, MSM 2^8 .. 2^12.  usage: small_calls.py [reps=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import fft, merkle, msm
from tools.synth import distinct_points
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(4)
def e256(n, bits=59):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); a[:, 0] &= np.uint64((1 << bits) - 1); return a
def timed(fn, r=reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(r): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / r * 1e6
off = e256(1)[0]
for L in (8, 10, 12, 14, 16, 18):
    x = torch.from_numpy(e256(1 << L).view(np.int64)).cuda(); y = torch.empty_like(x)
    a = timed(lambda: fft.ntt_device(fft.Stark252PrimeField, x, y, L))
    b = timed(lambda: fft.ntt_device(fft.Stark252PrimeField, x, y, L, inverse=True))
    c = timed(lambda: fft.ntt_device(fft.Stark252PrimeField, x, y, L, offset=off))
    print("stark252 2^%d: ntt %.1f us  intt %.1f us  coset ntt %.1f us" % (L, a, b, c), flush=True)
for L in (8, 12, 16):
    x = torch.from_numpy(rng.integers(0, 2013265921, size=4 << L, dtype=np.uint32).view(np.int32)).cuda(); y = torch.empty_like(x)
    a = timed(lambda: fft.ntt_device(fft.Babybear31PrimeFieldU32, x, y, L, batch=4))
    print("babybear 4 x 2^%d: ntt %.1f us" % (L, a), flush=True)
for L in (8, 12, 16):
    n = 1 << L
    x = torch.from_numpy(e256(2 * n).view(np.int64)).cuda(); nodes = torch.empty(((2 * n - 1) * 4,), dtype=torch.int64, device="cuda")
    a = timed(lambda: merkle.commit_columns_device(fft.Stark252PrimeField, x, 2, L, nodes))
    print("merkle 2 cols x 2^%d: %.1f us" % (L, a), flush=True)
for L in (8, 10, 12):
    n = 1 << L
    pts = distinct_points(msm.BLS12381Curve, n)
    sc = torch.from_numpy(e256(n, 62).view(np.int64)).cuda()
    a = timed(lambda: msm.msm_device(msm.BLS12381Curve, sc, pts, n), 50)
    print("bls12-381 g1 msm 2^%d: %.1f us" % (L, a), flush=True)
