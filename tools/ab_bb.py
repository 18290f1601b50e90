#!/usr/bin/env python3
"""BabyBear NTT timing (4 x 2^24 u32 / u64, ext4 2^24, other sizes) with per-kernel times: usage ab_bb.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import _lib, fft
from tools import inputs as util
fp = util.field_pairs()
for tag, name, L, batch in (("u32 4x2^24", "babybear_u32", 24, 4), ("u64 4x2^24", "babybear_u64", 24, 4), ("ext4 2^24", "babybear_ext4", 24, 1),
                            ("u32 4x2^20", "babybear_u32", 20, 4), ("u32 1x2^22", "babybear_u32", 22, 1)):
    fld = fp[name][0]
    n = (1 << L) * batch
    a = util.rand_elems(name, n, 1)
    t_in = torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()
    t_out = torch.empty_like(t_in)
    for _ in range(3):
        fft.ntt_device(fld, t_in, t_out, L, batch=batch)
    torch.cuda.synchronize()
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(20):
        fft.ntt_device(fld, t_in, t_out, L, batch=batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    prof = _lib.profile_end()
    print("%-12s %.4f ms" % (tag, dt * 1e3), {k: round(v[1] / max(v[0], 1), 4) for k, v in prof.items()}, flush=True)
