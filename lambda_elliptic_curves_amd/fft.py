"""Host-side mirror of the reference's Polynomial FFT API for the HIP backend
(math/src/fft/polynomial.rs:25-127): same names, argument meaning and error behaviour.

Arrays are numpy, bit-for-bit the reference's memory: a field element is `limbs` uint64 words, most
significant first, Montgomery form (BabyBear u32 layout: one uint32).  Device entry points take torch
tensors already resident in HBM and run on torch's current stream.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .errors import check


class Field:
    """(lw_field_t, lw_layout_t) pair = one reference field type."""

    def __init__(self, name, field, layout, words, dtype, two_adicity, field_name):
        self.name, self.field, self.layout, self.words, self.dtype = name, field, layout, words, dtype
        self.two_adicity, self.field_name = two_adicity, field_name   # IsFFTField::{TWO_ADICITY, field_name()}
        self.elem_bytes = words * np.dtype(dtype).itemsize

    def __repr__(self):
        return f"Field({self.name})"


Stark252PrimeField = Field("Stark252PrimeField", L.FIELD_STARK252, L.LAYOUT_U64_LIMBS_MS_FIRST, 4, np.uint64, 192, "stark256")
FrField = Field("BLS12-381 FrField", L.FIELD_BLS12_381_FR, L.LAYOUT_U64_LIMBS_MS_FIRST, 4, np.uint64, 32, "")
Babybear31PrimeField = Field("Babybear31PrimeField(u64 limb)", L.FIELD_BABYBEAR, L.LAYOUT_BABYBEAR_U64_R64, 1, np.uint64, 24, "babybear31")
Babybear31PrimeFieldU32 = Field("Babybear31PrimeField(u32)", L.FIELD_BABYBEAR, L.LAYOUT_BABYBEAR_U32_R32, 1, np.uint32, 24, "babybear31")
Degree4BabyBearExtensionField = Field("Degree4BabyBearExtensionField", L.FIELD_BABYBEAR, L.LAYOUT_EXT4_INTERLEAVED, 4, np.uint64, 24, "babybear31")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _as_elems(field, a):
    a = np.ascontiguousarray(a, dtype=field.dtype)
    return a.reshape(-1) if field.words == 1 else a.reshape(-1, field.words)


def _offset_arg(field, offset):
    """The coset offset is one element of the domain field F (the base field for the extension layout)."""
    if offset is None:
        return None
    words = field.words if field.layout == L.LAYOUT_U64_LIMBS_MS_FIRST else 1
    return np.ascontiguousarray(offset, dtype=field.dtype).reshape(-1)[:words].copy()


class ResultBuffer:
    """A pinned, resident result buffer from the library's pool (lw_hip_result_acquire / lw_hip_result_release): the
    device-to-host copy into it runs at the PCIe rate, where a fresh numpy / Vec result pays one page fault per 4 KiB first.
    `array` is a numpy view of the memory, valid until release() (or the end of the `with` block)."""

    def __init__(self, field, n_elems):
        self.field = field
        self._p = C.c_void_p()
        nbytes = n_elems * field.elem_bytes
        check(L.lib().lw_hip_result_acquire(nbytes, C.byref(self._p)))
        raw = (C.c_uint8 * nbytes).from_address(self._p.value)
        shape = (n_elems,) if field.words == 1 else (n_elems, field.words)
        self.array = np.frombuffer(raw, dtype=field.dtype).reshape(shape)

    def release(self):
        if self._p:
            self.array = None
            check(L.lib().lw_hip_result_release(self._p))
            self._p = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def evaluate_fft(field, coefficients, blowup_factor=1, domain_size=None, offset=None):
    """Polynomial::evaluate_fft (offset=None) / evaluate_offset_fft."""
    a = _as_elems(field, coefficients)
    n = a.shape[0]
    ds = 0 if domain_size is None else int(domain_size)
    off = _offset_arg(field, offset)
    out_len = C.c_size_t(0)
    lib = L.lib()
    check(lib.lw_polynomial_evaluate_fft(field.field, field.layout, _ptr(a), n, blowup_factor, ds, _ptr(off), None, 0,
                                          C.byref(out_len)))
    shape = (out_len.value,) if field.words == 1 else (out_len.value, field.words)
    out = np.empty(shape, dtype=field.dtype)
    check(lib.lw_polynomial_evaluate_fft(field.field, field.layout, _ptr(a), n, blowup_factor, ds, _ptr(off), _ptr(out),
                                          out_len.value, C.byref(out_len)))
    return out


def evaluate_offset_fft(field, coefficients, blowup_factor, domain_size, offset):
    return evaluate_fft(field, coefficients, blowup_factor, domain_size, offset)


def interpolate_fft(field, fft_evals, offset=None, strip=False):
    """Polynomial::interpolate_fft / interpolate_offset_fft. Returns all N coefficients; strip=True applies
    Polynomial::new's removal of trailing zero coefficients."""
    a = _as_elems(field, fft_evals)
    out = np.empty_like(a)
    clen = C.c_size_t(0)
    off = _offset_arg(field, offset)
    check(L.lib().lw_polynomial_interpolate_fft(field.field, field.layout, _ptr(a), a.shape[0], _ptr(off), _ptr(out),
                                                C.byref(clen)))
    return out[:clen.value] if strip else out


def interpolate_offset_fft(field, fft_evals, offset):
    return interpolate_fft(field, fft_evals, offset)


# RootsConfig (math/src/field/traits.rs)
ROOTS_NATURAL, ROOTS_NATURAL_INVERSED, ROOTS_BIT_REVERSE, ROOTS_BIT_REVERSE_INVERSED = 0, 1, 2, 3


def get_twiddles(field, order, config):
    """roots_of_unity::get_twiddles / the CUDA seam's gen_twiddles: 2^order / 2 powers of the primitive 2^order-th
    root (or its inverse), natural or bit-reversed, as domain-field elements."""
    count = (1 << order) // 2 if order <= 63 else 0
    base_words = field.words if field.layout == L.LAYOUT_U64_LIMBS_MS_FIRST else 1
    out = np.empty((count,) if base_words == 1 else (count, base_words), dtype=field.dtype)
    check(L.lib().lw_hip_gen_twiddles(field.field, field.layout, order, config, _ptr(out) if count else None))
    return out


def get_powers_of_primitive_root(field, n, count, config):
    """roots_of_unity::get_powers_of_primitive_root: `count` powers of the primitive 2^n-th root (or its inverse); the
    bit-reversed configurations return next_power_of_two(count) entries, bit-reverse permuted."""
    return _gen_powers(field, n, count, config, None)


def get_powers_of_primitive_root_coset(field, n, count, offset):
    """roots_of_unity::get_powers_of_primitive_root_coset: offset * w^i, natural order."""
    return _gen_powers(field, n, count, ROOTS_NATURAL, offset)


def _gen_powers(field, n, count, config, offset):
    base_words = field.words if field.layout == L.LAYOUT_U64_LIMBS_MS_FIRST else 1
    off = _offset_arg(field, offset)
    olen = C.c_size_t(0)
    check(L.lib().lw_hip_gen_powers(field.field, field.layout, n, count, config, _ptr(off), None, C.byref(olen)))
    out = np.empty((olen.value,) if base_words == 1 else (olen.value, base_words), dtype=field.dtype)
    if olen.value:
        check(L.lib().lw_hip_gen_powers(field.field, field.layout, n, count, config, _ptr(off), _ptr(out), C.byref(olen)))
    return out


def bitrev_permutation(field, data):
    """in_place_bit_reverse_permute / the CUDA seam's bitrev_permutation (returns a new array)."""
    a = _as_elems(field, data)
    out = np.empty_like(a)
    check(L.lib().lw_hip_bitrev_permutation(field.field, field.layout, _ptr(a), _ptr(out), a.shape[0]))
    return out


def ntt(field, data, inverse=False, log2n=None, batch=1, batch_stride=0, offset=None, out=None):
    """Backend seam on host buffers (evaluate_fft_cuda / interpolate_fft_cuda equivalents,
    math/src/fft/gpu/cuda/polynomial.rs:16-49): the slice is already power-of-two sized.  `out`: write the result into this
    array (e.g. a ResultBuffer's) instead of a new one."""
    a = _as_elems(field, data)
    if log2n is None:
        n = a.shape[0] // batch
        if n == 0 or n & (n - 1):
            from .errors import InputError
            raise InputError(f"Input length is {n}, which is not a power of two")
        log2n = n.bit_length() - 1
    if out is None:
        out = np.empty_like(a)
    elif out.nbytes != a.nbytes or out.dtype != a.dtype or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous array of the input's size and type")
    off = _offset_arg(field, offset)
    check(L.lib().lw_hip_ntt(field.field, field.layout, L.DIR_INVERSE if inverse else L.DIR_FORWARD, _ptr(a), _ptr(out),
                             log2n, batch, batch_stride, _ptr(off)))
    return out


def lde_device(field, t_coeffs, log2_coeffs, t_out, log2n, batch=1, offset=None, stream=None):
    """Device-resident low-degree extension: evaluate_offset_fft(poly, blowup, Some(domain), offset) for `batch`
    blocks of 2^log2_coeffs coefficients -> 2^log2n evaluations each, without materialising the zero padding."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    off = _offset_arg(field, offset)
    check(L.lib().lw_hip_ntt_lde_device(field.field, field.layout, C.c_void_p(t_coeffs.data_ptr()), log2_coeffs,
                                        C.c_void_p(t_out.data_ptr()), log2n, batch, _ptr(off), C.c_void_p(stream)))


def ntt_device(field, t_in, t_out, log2n, inverse=False, batch=1, batch_stride=0, offset=None, stream=None):
    """Device-resident transform on torch tensors (any dtype, bytes = reference layout); asynchronous on
    `stream` (default: torch's current stream)."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    off = _offset_arg(field, offset)
    check(L.lib().lw_hip_ntt_device(field.field, field.layout, L.DIR_INVERSE if inverse else L.DIR_FORWARD,
                                    C.c_void_p(t_in.data_ptr()), C.c_void_p(t_out.data_ptr()), log2n, batch,
                                    batch_stride, _ptr(off), C.c_void_p(stream)))
