"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/lw_hip.h declares; without a GPU every compute entry point fails loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lw_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lw_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lambda_elliptic_curves_amd import _lib
    L = _lib.lib()
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in include/lw_hip.h but not exported"
    assert set(_lib.EXPORTS) == set(declared)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lambda_elliptic_curves_amd import errors, fft, msm
    with pytest.raises(errors.HipError):
        fft.evaluate_fft(fft.Stark252PrimeField, np.ones((4, 4), np.uint64))
    with pytest.raises(errors.HipError):
        msm.msm(msm.BLS12381Curve, np.ones((1, 4), np.uint64), np.ones((1, 18), np.uint64))


def test_host_side_argument_checks_do_not_need_a_device():
    from lambda_elliptic_curves_amd import errors, fft, msm
    with pytest.raises(errors.InputError):
        fft.interpolate_fft(fft.Stark252PrimeField, np.ones((3, 4), np.uint64))
    with pytest.raises(errors.LengthMismatch):
        msm.msm(msm.BLS12381Curve, np.ones((2, 4), np.uint64), np.ones((1, 18), np.uint64))
    # zero polynomial: len zeros, no transform, no device needed (fft/polynomial.rs:33-35)
    z = fft.evaluate_fft(fft.Stark252PrimeField, np.zeros((3, 4), np.uint64), 2, 8)
    assert z.shape == (16, 4) and not z.any()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "lambda_elliptic_curves_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), f"{f} mentions the oracle"


def test_tools_do_not_use_the_checker():
    # oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import it
    # (tests/util.py does, so the tools take their inputs from tools/inputs.py instead)
    for f in sorted(os.listdir(os.path.join(ROOT, "tools"))):
        if f.endswith(".py"):
            text = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", text, re.M), f"tools/{f} imports the checker"


def test_rust_shim_declares_every_symbol_and_status():
    # rust-shim/src/ffi.rs cannot be compiled here (no Rust toolchain); at least keep it in step with the header
    ffi = open(os.path.join(ROOT, "rust-shim", "src", "ffi.rs")).read()
    declared = set(_declared_symbols())
    assert set(re.findall(r"pub fn (lw_[a-z0-9_]+)\s*\(", ffi)) == declared
    header = open(os.path.join(ROOT, "include", "lw_hip.h")).read()
    for name, val in re.findall(r"\b(LW_(?:OK|ERR_[A-Z0-9_]+))\s*=\s*(-?\d+)", header):
        assert re.search(r"pub const %s: c_int = %s;" % (name, val), ffi), name
    for enum, variants in (("lw_field_t", 3), ("lw_layout_t", 4), ("lw_curve_t", 4)):
        body = re.search(r"typedef enum \{([^}]*)\} %s;" % enum, header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        assert len(re.findall(r"=\s*\d+", body)) == variants
