#!/usr/bin/env python3
"""Host-buffer calls from T concurrent host threads (the reference's rayon loop over columns, trace.rs:186-190): aggregate
throughput of lw_hip_ntt on Stark252 2^L columns with caller buffers reused, T = 1, 2, 4, 8.  One lane per running call
(csrc/context.h): one caller's download overlaps another's upload and kernels.  usage: ab_lanes.py [L=22]"""
import ctypes as C
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lambda_elliptic_curves_amd import _lib, fft

L = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << L
fld = fft.Stark252PrimeField
rng = np.random.default_rng(3)
calls = 8
for T in (1, 2, 4, 8):
    ins = [rng.integers(0, 1 << 59, size=(n, 4), dtype=np.uint64) for _ in range(T)]
    outs = [np.empty_like(a) for a in ins]
    lib = _lib.lib()

    def worker(t):
        a, o = ins[t], outs[t]
        for _ in range(calls):
            rc = lib.lw_hip_ntt(fld.field, fld.layout, _lib.DIR_FORWARD, a.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), L, 1, 0, None)
            assert rc == 0, _lib.last_error()
    worker(0)   # warm: tables, staging
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.perf_counter() - t0
    print("T=%d threads x %d calls of lw_hip_ntt 2^%d (host buffers, %d MiB each way): %.1f ms per call and thread, %.2f G elements/s aggregate"
          % (T, calls, L, n * 32 >> 20, dt / calls * 1e3, T * calls * n / dt / 1e9), flush=True)
