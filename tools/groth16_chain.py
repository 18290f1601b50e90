#!/usr/bin/env python3
"""calculate_h_coefficients on device-resident vectors alone, a few times: a target for rocprofv3 --kernel-trace
(tools/kernel_timeline.py) and a wall-clock figure.  usage: groth16_chain.py [reps=3] [log2gates=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import groth16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = 1 << L
rng = np.random.default_rng(3)
def e(n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); a[:, 0] &= np.uint64((1 << 62) - 1); return a
tl, tr, to = (torch.from_numpy(e(g).view(np.int64)).cuda() for _ in range(3))
th = torch.empty((2 * g, 4), dtype=torch.int64, device="cuda")
groth16.calculate_h_coefficients_device(tl, tr, to, g, g, t_out=th); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): groth16.calculate_h_coefficients_device(tl, tr, to, g, g, t_out=th)
torch.cuda.synchronize()
print("groth16 h, 2^%d gates: %.3f ms per run" % (L, (time.perf_counter() - t0) / reps * 1e3))
