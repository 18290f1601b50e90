#!/usr/bin/env python3
"""Sharded MSM (bucket-slice exchange, csrc/comm.hip msm_sharded_run): per-rank kernel time by phase as the number of ranks
grows at a FIXED number of pairs per rank — what the running sums (msm_group_sum + msm_combine + msm_slice_sum) cost a rank
with G = 1 (single-GPU msm_device), 2, 4, 8.  The ranks are virtual (one device, lw_hip_msm_sharded_selftest_device), so
the exchange itself is device-to-device copies; the kernel times are what one rank of a real run would spend.
usage: ab_msm_sharded_reduce.py [LOG2_PAIRS_PER_RANK=23] [CURVE=BN254Curve]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lambda_elliptic_curves_amd import _lib, msm
from lambda_elliptic_curves_amd import distributed as D
from tools.synth import distinct_points

L = int(sys.argv[1]) if len(sys.argv) > 1 else 23
crv = getattr(msm, sys.argv[2] if len(sys.argv) > 2 else "BN254Curve")
rng = np.random.default_rng(7)
for lg in (0, 1, 2, 3):
    G = 1 << lg
    n = G << L
    tp = distinct_points(crv, n)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    ts = torch.from_numpy(sc.view(np.int64)).cuda()
    run = (lambda: msm.msm_device(crv, ts, tp, n)) if lg == 0 else (lambda: D.msm_sharded_selftest(crv, ts, tp, n, lg))
    run()
    _lib.profile_begin()
    run()
    prof = _lib.profile_end()
    per_rank = {k: v[1] / G for k, v in prof.items()}
    red = sum(v for k, v in per_rank.items() if k.startswith(("msm_group_sum", "msm_combine", "msm_slice_sum")))
    acc = sum(v for k, v in per_rank.items() if k.startswith("msm_accumulate"))
    rest = sum(per_rank.values()) - red - acc
    print("G=%d  2^%d pairs per rank: per-rank kernel ms: accumulate %.2f  running sums (+ slice sum) %.3f  sort + normalise %.2f   %s"
          % (G, L, acc, red, rest, {k: round(v, 3) for k, v in per_rank.items() if k.startswith(("msm_group", "msm_combine", "msm_slice"))}), flush=True)
    del tp, ts
    torch.cuda.empty_cache()
