#!/usr/bin/env python3
"""Host-buffer MSMs large enough for the upload-under-sort path (>= 2^19 points: the points go up in chunks on a side stream
and are normalised on another, csrc/msm.hip) from several threads at once, beside host NTTs; every result must equal the one
the same call gave single-threaded.  usage: stress_host_msm.py [seconds=30] [threads=6]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import fft, msm
from tools.synth import distinct_points
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(9)
jobs = []
for crv, L in ((msm.BLS12381Curve, 20), (msm.BN254Curve, 19), (msm.BN254TwistCurve, 19), (msm.BLS12381Curve, 19)):
    n = (1 << L) + 12345 * (L == 19)
    pts = distinct_points(crv, n).cpu().numpy().view(np.uint64)
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    jobs.append(("msm %s %d" % (getattr(crv, "name", "curve"), n), lambda crv=crv, sc=sc, pts=pts: msm.msm(crv, sc, pts).tobytes()))
a = rng.integers(0, 1 << 63, size=(1 << 18, 4), dtype=np.uint64); a[:, 0] &= np.uint64((1 << 59) - 1)
jobs.append(("ntt 2^18", lambda: fft.ntt(fft.Stark252PrimeField, a).tobytes()))
ref = [fn() for _, fn in jobs]
errs, counts = [], [0] * T
stop = time.time() + secs
def worker(t):
    k = t
    try:
        while time.time() < stop and not errs:
            name, fn = jobs[k % len(jobs)]
            if fn() != ref[k % len(jobs)]: errs.append("%s differs (thread %d)" % (name, t))
            k += 1; counts[t] += 1
    except Exception as e:   # noqa: BLE001
        errs.append("%r (thread %d)" % (e, t))
th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[x.start() for x in th]; [x.join() for x in th]
print("%d threads, %.0f s: %d calls, %d errors %s" % (T, secs, sum(counts), len(errs), errs[:3]))
sys.exit(1 if errs else 0)
