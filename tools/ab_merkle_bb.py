#!/usr/bin/env python3
"""Merkle commitment of BabyBear columns (BASELINE config 4's field): 4 columns x 2^24, u32 and u64 words"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import _lib, merkle, fft
rng = np.random.default_rng(2)
for fld, dt in ((fft.Babybear31PrimeFieldU32, np.uint32), (fft.Babybear31PrimeField, np.uint64)):
    for ncols, L in ((4, 24), (16, 22)):
        n = 1 << L
        a = rng.integers(0, 2013265921, size=n * ncols, dtype=dt)
        t = torch.from_numpy(a.view(np.int32 if dt == np.uint32 else np.int64)).cuda()
        nodes = torch.empty(((2 * n - 1) * 4,), dtype=torch.int64, device="cuda")
        for _ in range(2):
            merkle.commit_columns_layout_device(fld, t, ncols, L, nodes)
        torch.cuda.synchronize()
        _lib.profile_begin()
        t0 = time.perf_counter()
        for _ in range(5):
            merkle.commit_columns_layout_device(fld, t, ncols, L, nodes)
        torch.cuda.synchronize()
        d = (time.perf_counter() - t0) / 5
        prof = _lib.profile_end()
        print("%s %d cols x 2^%d: %.3f ms" % (dt.__name__, ncols, L, d * 1e3), {k: round(v[1] / max(v[0], 1), 4) for k, v in prof.items()}, flush=True)
