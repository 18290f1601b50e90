#!/usr/bin/env python3
"""Hammer the library from many host threads at once (lanes, csrc/context.h): every thread loops over a mix of host and
device entry points — NTTs of several fields and sizes, MSMs, an SRS handle, Merkle commits, FRI layers, Groth16 — and every
result must be bit-identical to the one the same call gave single-threaded.  usage: stress_lanes.py [seconds=60] [threads=12]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lambda_elliptic_curves_amd import fft, groth16, merkle, msm
from tools.synth import distinct_points

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(5)
P_BB = 2013265921


def e256(n, bits):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 0] &= np.uint64((1 << bits) - 1)
    return a


jobs = []   # (name, callable returning bytes)
for L in (8, 13, 17, 20):
    a = e256(1 << L, 59)
    jobs.append(("stark ntt 2^%d" % L, lambda a=a: fft.ntt(fft.Stark252PrimeField, a).tobytes()))
    jobs.append(("stark intt 2^%d" % L, lambda a=a: fft.ntt(fft.Stark252PrimeField, a, inverse=True).tobytes()))
b = rng.integers(0, P_BB, size=4 << 16, dtype=np.uint32)
jobs.append(("babybear 4 x 2^16", lambda: fft.ntt(fft.Babybear31PrimeFieldU32, b, log2n=16, batch=4).tobytes()))
f = e256(1 << 15, 62)
jobs.append(("fr381 lde", lambda: fft.evaluate_fft(fft.FrField, f, 4, 1 << 15).tobytes()))
pts_t = distinct_points(msm.BN254Curve, 1 << 14)
pts = pts_t.cpu().numpy().view(np.uint64)
sc = rng.integers(0, 1 << 62, size=(1 << 14, 4), dtype=np.uint64)
jobs.append(("bn254 msm 2^14", lambda: msm.msm(msm.BN254Curve, sc, pts).tobytes()))
srs = msm.Srs(msm.BN254Curve, pts)
jobs.append(("srs msm", lambda: srs.msm(sc[:9000]).tobytes()))
cols = np.stack([e256(1 << 12, 59) for _ in range(3)])
jobs.append(("merkle", lambda: merkle.commit_columns(fft.Stark252PrimeField, cols)))
co = e256(1 << 12, 59)
z = e256(1, 59)[0]
jobs.append(("fri layer", lambda: b"".join(x.tobytes() if hasattr(x, "tobytes") else x for x in merkle.fri_layer(fft.Stark252PrimeField, co, z, z, 1 << 13))))
l, r, o = e256(1 << 10, 62), e256(1 << 10, 62), e256(1 << 10, 62)
jobs.append(("groth16 h", lambda: groth16.calculate_h_coefficients(l, r, o, 1 << 10, strip=False).tobytes()))
td = torch.from_numpy(e256(1 << 18, 59).view(np.int64)).cuda()


def dev_ntt():
    s = torch.cuda.Stream()
    out = torch.empty_like(td)
    fft.ntt_device(fft.Stark252PrimeField, td, out, 18, stream=s.cuda_stream)
    s.synchronize()
    return out.cpu().numpy().tobytes()


jobs.append(("device ntt 2^18", dev_ntt))
ref = [fn() for _, fn in jobs]
errs, counts = [], [0] * T
stop = time.time() + secs


def worker(t):
    k = t
    try:
        while time.time() < stop and not errs:
            name, fn = jobs[k % len(jobs)]
            if fn() != ref[k % len(jobs)]:
                errs.append("%s differs (thread %d)" % (name, t))
            k += 1 + t % 3
            counts[t] += 1
    except Exception as e:   # noqa: BLE001
        errs.append("%r (thread %d)" % (e, t))


th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[x.start() for x in th]
[x.join() for x in th]
srs.close()
print("%d threads, %.0f s: %d calls, %d errors %s" % (T, secs, sum(counts), len(errs), errs[:3]), flush=True)
sys.exit(1 if errs else 0)
