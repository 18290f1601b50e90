"""Host-side mirror of math/src/msm/pippenger.rs `msm` for the HIP backend."""
import ctypes as C

import numpy as np

from . import _lib as L
from .errors import check


class Curve:
    def __init__(self, name, curve, coord_words):
        self.name, self.curve, self.coord_words = name, curve, coord_words
        self.point_words = 3 * coord_words

    def __repr__(self):
        return f"Curve({self.name})"


BLS12381Curve = Curve("BLS12381Curve", L.CURVE_BLS12_381_G1, 6)
BN254Curve = Curve("BN254Curve", L.CURVE_BN254_G1, 4)
BN254TwistCurve = Curve("BN254TwistCurve", L.CURVE_BN254_G2, 8)
BLS12381TwistCurve = Curve("BLS12381TwistCurve", L.CURVE_BLS12_381_G2, 12)


def msm_fr(curve, fr_elements, points):
    """msm over FrElements as stored (Montgomery form): representative() + msm in one device call."""
    s = np.ascontiguousarray(fr_elements, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, curve.point_words)
    out = np.zeros(curve.point_words, dtype=np.uint64)
    check(L.lib().lw_hip_msm_fr(curve.curve, s.ctypes.data_as(C.c_void_p), s.shape[0], p.ctypes.data_as(C.c_void_p),
                                p.shape[0], out.ctypes.data_as(C.c_void_p)))
    return out


def msm(curve, cs, points):
    """pippenger::msm(cs, points): cs = (n,4) uint64 canonical scalars (MS limb first), points = (n, 3*coord_words)
    uint64 projective points.  Returns one projective point; only its affine image is canonical.
    Different lengths -> LengthMismatch (pippenger.rs:25-27); empty input -> neutral element."""
    s = np.ascontiguousarray(cs, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, curve.point_words)
    out = np.zeros(curve.point_words, dtype=np.uint64)
    check(L.lib().lw_hip_msm(curve.curve, s.ctypes.data_as(C.c_void_p), s.shape[0], p.ctypes.data_as(C.c_void_p),
                             p.shape[0], out.ctypes.data_as(C.c_void_p)))
    return out


def msm_with(curve, cs, points, window_size):
    """pippenger::msm_with (math/src/msm/pippenger.rs:42-103).  The window only shapes the reference's schedule;
    the group element is the same for every window, so this is msm() (the device picks its own window)."""
    if len(np.asarray(cs).reshape(-1, 4)) != len(np.asarray(points).reshape(-1, curve.point_words)):
        from .errors import LengthMismatch
        raise LengthMismatch("scalars and points have different lengths")
    return msm(curve, cs, points)


def msm_device(curve, t_scalars, t_points, n, stream=None):
    """Device-resident MSM on torch tensors; returns the projective result as a numpy array."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    out = np.zeros(curve.point_words, dtype=np.uint64)
    check(L.lib().lw_hip_msm_device(curve.curve, C.c_void_p(t_scalars.data_ptr()), C.c_void_p(t_points.data_ptr()), n,
                                    out.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))
    return out


class Srs:
    """A fixed point set kept on the device in affine form (lw_hip_srs_*): what the reference's KZG
    `StructuredReferenceString.powers_main_group` (crypto/src/commitments/kzg.rs:159-163) or a Groth16 proving-key
    vector (provers/groth16/src/prover.rs:69-85) is to repeated msm() calls.  `srs.msm(cs)` equals
    msm(cs, points[:len(cs)]) — fewer scalars than points is the KZG call shape; more is LengthMismatch.
    Sets of 2^19 points and more keep 13 window-shifted copies on the device (include/lw_hip.h: 13 x the affine bytes,
    LW_HIP_SRS_FOLD=0 to keep one); results do not depend on it."""

    def __init__(self, curve, points=None, t_points=None, n=None, stream=None):
        self.curve = curve
        self._h = C.c_void_p()
        if t_points is not None:        # device-resident projective points (torch tensor)
            import torch
            if stream is None:
                stream = torch.cuda.current_stream().cuda_stream
            self.n = int(n if n is not None else t_points.shape[0])
            check(L.lib().lw_hip_srs_create_device(curve.curve, C.c_void_p(t_points.data_ptr()), self.n, C.c_void_p(stream),
                                                   C.byref(self._h)))
        else:
            p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, curve.point_words)
            self.n = p.shape[0]
            check(L.lib().lw_hip_srs_create(curve.curve, p.ctypes.data_as(C.c_void_p), self.n, C.byref(self._h)))

    def msm(self, cs):
        s = np.ascontiguousarray(cs, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(self.curve.point_words, dtype=np.uint64)
        check(L.lib().lw_hip_msm_srs(self._h, s.ctypes.data_as(C.c_void_p), s.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def msm_fr(self, fr_elements):
        s = np.ascontiguousarray(fr_elements, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(self.curve.point_words, dtype=np.uint64)
        check(L.lib().lw_hip_msm_srs_fr(self._h, s.ctypes.data_as(C.c_void_p), s.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def msm_device(self, t_scalars, n, stream=None):
        import torch
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        out = np.zeros(self.curve.point_words, dtype=np.uint64)
        check(L.lib().lw_hip_msm_srs_device(self._h, C.c_void_p(t_scalars.data_ptr()), n, out.ctypes.data_as(C.c_void_p),
                                            C.c_void_p(stream)))
        return out

    def msm_fr_device(self, t_fr_elements, n, stream=None):
        """Scalars as stored FrElements (Montgomery form) already on the device — e.g. the h coefficients
        groth16.calculate_h_coefficients_device leaves in HBM (provers/groth16/src/prover.rs:68-72,97-101)."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        out = np.zeros(self.curve.point_words, dtype=np.uint64)
        check(L.lib().lw_hip_msm_srs_fr_device(self._h, C.c_void_p(t_fr_elements.data_ptr()), n, out.ctypes.data_as(C.c_void_p),
                                               C.c_void_p(stream)))
        return out

    def close(self):
        if self._h:
            L.lib().lw_hip_srs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
