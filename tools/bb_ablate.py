"""Diagnostic: per-pass time of the BabyBear NTT (4 x 2^24, u32) under the LW_HIP_NTT_DBG ablation bits."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lambda_elliptic_curves_amd import fft, _lib
from tools import inputs as util
fld = fft.Babybear31PrimeFieldU32
L, batch = 24, 4
a = util.rand_elems("babybear_u32", (1 << L) * batch, 1)
t_in = torch.from_numpy(a.view(np.int32)).cuda(); t_out = torch.empty_like(t_in)
for _ in range(2): fft.ntt_device(fld, t_in, t_out, L, batch=batch)
torch.cuda.synchronize(); _lib.profile_begin()
for _ in range(10): fft.ntt_device(fld, t_in, t_out, L, batch=batch)
print(os.environ.get("LW_HIP_NTT_DBG"), {k: round(v[1] / v[0], 4) for k, v in _lib.profile_end().items()})
