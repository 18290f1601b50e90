#!/usr/bin/env python3
"""The device-resident FRI commit phase alone (2^20 coefficients, blow-up 2, 20 layers), a few times: a target for
rocprofv3 --kernel-trace (tools/kernel_timeline.py) and a wall-clock figure.  usage: fri_chain.py [reps=3] [log2n=20]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import fft, merkle
P = 0x800000000000011000000000000000000000000000000000000000000000001
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << L
def mont(v):
    m = v * (1 << 256) % P
    return np.array([(m >> (64 * (3 - k))) & ((1 << 64) - 1) for k in range(4)], dtype=np.uint64)
rng = np.random.default_rng(5)
co = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); co[:, 0] &= np.uint64((1 << 59) - 1)
offs = [mont(pow(3, 1 << k, P)) for k in range(L + 1)]
t_co = torch.from_numpy(co.view(np.int64)).cuda()
def run():
    st = {"s": b"t"}
    def sample(): return mont(int.from_bytes(hashlib.sha256(st["s"]).digest()[:31], "big"))
    def absorb(root): st["s"] = hashlib.sha256(st["s"] + root).digest()
    return merkle.fri_commit_phase_device(fft.Stark252PrimeField, L + 1, t_co, n, sample, absorb, lambda k: offs[k], 2 * n)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize()
print("FRI commit phase 2^%d, %d layers: %.3f ms per run" % (L, L, (time.perf_counter() - t0) / reps * 1e3))
