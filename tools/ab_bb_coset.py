#!/usr/bin/env python3
"""BabyBear coset transforms (evaluate_offset_fft shape: c_i * h^i then NTT) against plain ones, per-kernel times"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lambda_elliptic_curves_amd import _lib, fft
from tools import inputs as util
fp = util.field_pairs()
for tag, name, L, batch in (("u32 4x2^24", "babybear_u32", 24, 4), ("u64 4x2^24", "babybear_u64", 24, 4), ("u32 4x2^20", "babybear_u32", 20, 4)):
    fld = fp[name][0]
    n = (1 << L) * batch
    a = util.rand_elems(name, n, 1)
    t_in = torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()
    t_out = torch.empty_like(t_in)
    off = util.offset_elem(name, 3)
    for inverse in (False, True):
        for o in (None, off):
            for _ in range(3):
                fft.ntt_device(fld, t_in, t_out, L, inverse=inverse, batch=batch, offset=o)
            torch.cuda.synchronize()
            _lib.profile_begin()
            t0 = time.perf_counter()
            for _ in range(20):
                fft.ntt_device(fld, t_in, t_out, L, inverse=inverse, batch=batch, offset=o)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            prof = _lib.profile_end()
            print("%-11s %s %-6s %.4f ms" % (tag, "inv" if inverse else "fwd", "coset" if o is not None else "plain", dt * 1e3),
                  {k: round(v[1] / max(v[0], 1), 4) for k, v in prof.items()}, flush=True)
