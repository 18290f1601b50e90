#!/usr/bin/env python3
"""Merkle commitment timing (lw_stark_commit_columns_device): 4 columns x 2^22 (LDE commit) and 1 column x 2^24"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import _lib, merkle, fft
from tools import inputs as util
fld = util.field_pairs()["stark252"][0]
for ncols, L in ((4, 22), (1, 24), (16, 20)):
    n = 1 << L
    a = util.rand_elems("stark252", n * ncols, 1)
    t = torch.from_numpy(a.view(np.int64)).cuda()
    nodes = torch.empty((2 * n - 1, 4), dtype=torch.int64, device="cuda")
    for _ in range(2):
        merkle.commit_columns_device(fld, t, ncols, L, nodes)
    torch.cuda.synchronize()
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(10):
        merkle.commit_columns_device(fld, t, ncols, L, nodes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    prof = _lib.profile_end()
    print("%d cols x 2^%d: %.3f ms" % (ncols, L, dt * 1e3), {k: round(v[1] / max(v[0], 1), 4) for k, v in prof.items()}, flush=True)
