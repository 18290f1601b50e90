"""Host-side mirror of provers/groth16/src/qap.rs `calculate_h_coefficients` (after the variable polynomials have been
accumulated) for the HIP backend: three coset LDEs, the pointwise quotient and the coset INTT run as one device pipeline."""
import ctypes as C

import numpy as np

from . import _lib as L
from .errors import check


def calculate_h_coefficients(l, r, o, num_gates, strip=True):
    """l, r, o: (n, 4) uint64 FrElement arrays (n <= num_gates, BLS12-381 Fr, reference layout).
    Returns the coefficients of h = interpolate_offset_fft((l*r - o) / t) over the coset 7*<w>, |<w>| = 2*num_gates."""
    arrs = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (l, r, o)]
    n = arrs[0].shape[0]
    if any(a.shape[0] != n for a in arrs):
        raise ValueError("l, r, o must have the same number of coefficients")
    out = np.zeros((2 * num_gates, 4), dtype=np.uint64)
    clen = C.c_size_t(0)
    check(L.lib().lw_groth16_h_coefficients(*[a.ctypes.data_as(C.c_void_p) for a in arrs], n, num_gates,
                                            out.ctypes.data_as(C.c_void_p), C.byref(clen)))
    return out[:clen.value] if strip else out


def calculate_h_coefficients_device(t_l, t_r, t_o, n_coeffs, num_gates, t_out=None, want_len=False, stream=None):
    """Device-resident form: t_l / t_r / t_o are torch tensors holding n_coeffs FrElements each; the 2*num_gates coefficients
    of h are written to t_out (allocated when None) and stay in HBM — feed them to msm.Srs.msm_fr_device, as
    Prover::prove feeds h into msm (provers/groth16/src/prover.rs:68-72,97-101).  want_len: also return the stripped length
    (synchronises)."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    if t_out is None:
        t_out = torch.empty((2 * num_gates, 4), dtype=torch.int64, device=t_l.device)
    clen = C.c_size_t(0)
    check(L.lib().lw_groth16_h_coefficients_device(C.c_void_p(t_l.data_ptr()), C.c_void_p(t_r.data_ptr()), C.c_void_p(t_o.data_ptr()),
                                                   n_coeffs, num_gates, C.c_void_p(t_out.data_ptr()),
                                                   C.byref(clen) if want_len else None, C.c_void_p(stream)))
    return (t_out, clen.value) if want_len else t_out
