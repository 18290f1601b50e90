"""Test-only local-compute backend built on the CPU oracle, so the SPMD exchange schedule of
lambda_elliptic_curves_amd/distributed.py can run under gloo on CPU (no GPU in the build container)."""
import numpy as np
import torch

from oracle import bigint_def as D
from oracle import oracle as O

P_OF = {O.F_STARK252: D.P_STARK252, O.F_FR381: D.P_FR381, O.F_BABYBEAR_U32: D.P_BABYBEAR, O.F_BABYBEAR_U64: D.P_BABYBEAR}


class OracleBackend:
    def __init__(self, oid, curve_oid=None):
        self.oid, self.curve_oid = oid, curve_oid

    def _np(self, t):
        a = t.cpu().numpy()
        return a.view(np.uint32) if a.dtype == np.int32 else a.view(np.uint64)

    def _t(self, a, like):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else np.int64)).to(like.device)

    def cross(self, field, t_in, log2n_total, log2g, j2_begin, slice_len, inverse):
        """y[k1][j2] = w_N^(j2 k1) * sum_j1 w_G^(j1 k1) x[j1][j2]  by definition, in canonical integers."""
        p = P_OF[self.oid]
        G = 1 << log2g
        x = O.elems_from_mont(self.oid, self._np(t_in))
        wN = D.primitive_root_of_unity(p, log2n_total)
        wG = D.primitive_root_of_unity(p, log2g)
        if inverse:
            wN, wG = pow(wN, -1, p), pow(wG, -1, p)
        ginv = pow(G, -1, p) if inverse else 1
        y = [0] * (G * slice_len)
        for t in range(slice_len):
            j2 = j2_begin + t
            for k1 in range(G):
                acc = sum(x[j1 * slice_len + t] * pow(wG, j1 * k1, p) for j1 in range(G)) % p
                y[k1 * slice_len + t] = acc * pow(wN, j2 * k1, p) * ginv % p
        return self._t(O.elems_to_mont(self.oid, y), t_in)

    def local_ntt(self, field, t_in, log2m, inverse):
        a = self._np(t_in)
        out = O.interpolate_fft(self.oid, a) if inverse else O.fft(self.oid, a, O.get_twiddles(self.oid, log2m, O.ROOTS_BITREV))
        return self._t(out, t_in)

    def local_msm(self, curve, t_scalars, t_points, n):
        return O.msm(self.curve_oid, self._np(t_scalars).reshape(n, 4), self._np(t_points).reshape(n, -1))

    def sum_points(self, curve, pts):
        acc = O.ec_neutral(self.curve_oid)
        for p in pts:
            acc = O.ec_add(self.curve_oid, acc, p)
        return acc
