"""lambda_elliptic_curves_amd — MI355X (gfx950) backend for lambdaworks' NTT + MSM prover hot path.

Hand-written HIP kernels behind a C ABI (include/lw_hip.h); this package is the thin host-side mirror of
the reference's operator interface for that path (Polynomial::evaluate_fft / interpolate_fft and
msm::pippenger::msm).  No CPU fallback.
"""
from . import _lib, errors, fft, groth16, merkle, msm  # noqa: F401

__all__ = ["_lib", "errors", "fft", "groth16", "merkle", "msm"]
