#!/usr/bin/env python3
"""Stark252 NTT wall time for one size under the current LW_HIP_NTT_PLAN: usage ab_ntt_plan.py L"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import fft
L = int(sys.argv[1]); n = 1 << L
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); a[:, 0] &= np.uint64((1 << 59) - 1)
x = torch.from_numpy(a.view(np.int64)).cuda(); y = torch.empty_like(x)
for _ in range(5): fft.ntt_device(fft.Stark252PrimeField, x, y, L)
torch.cuda.synchronize(); reps = 200 if L <= 22 else 30
t0 = time.perf_counter()
for _ in range(reps): fft.ntt_device(fft.Stark252PrimeField, x, y, L)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("plan=%s 2^%d: %.4f ms  %.2f G elem/s" % (os.environ.get("LW_HIP_NTT_PLAN", "default"), L, dt * 1e3, n / dt / 1e9), flush=True)
