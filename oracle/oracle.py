"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes loader for oracle/_build/liblw_oracle.so (the plain-C CPU restatement of the reference's
NTT + MSM path; see oracle/lw_oracle.c for the reference file:line each function follows).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (lambda_elliptic_curves_amd) never does.

Arrays are numpy, laid out exactly like the reference's memory: a field element is `limbs` uint64
words, MOST significant first, Montgomery form; BabyBear-u32 elements are single uint32 words.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liblw_oracle.so")

F_STARK252, F_FR381, F_BABYBEAR_U64, F_BABYBEAR_U32, F_BABYBEAR_EXT4, F_FP381, F_FP254, F_FR254 = range(8)
C_BLS12_381_G1, C_BN254_G1, C_BN254_G2, C_BLS12_381_G2 = range(4)
OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_INV, OP_TO_MONT, OP_FROM_MONT = range(7)
EC_ADD, EC_DOUBLE, EC_NEG, EC_TO_AFFINE, EC_EQ, EC_NEUTRAL = range(6)
ROOTS_NATURAL, ROOTS_NATURAL_INV, ROOTS_BITREV, ROOTS_BITREV_INV = range(4)

ERR_INPUT_NOT_POW2, ERR_ORDER, ERR_ROOT_OF_UNITY, ERR_LENGTH_MISMATCH, ERR_INV_ZERO, ERR_ALLOC, ERR_BAD_ARG = (
    -1, -2, -3, -4, -5, -6, -7)

FIELD_WORDS = {F_STARK252: 4, F_FR381: 4, F_BABYBEAR_U64: 1, F_BABYBEAR_U32: 1, F_BABYBEAR_EXT4: 4,
               F_FP381: 6, F_FP254: 4, F_FR254: 4}
FIELD_DTYPE = {f: (np.uint32 if f == F_BABYBEAR_U32 else np.uint64) for f in FIELD_WORDS}
# words per coordinate, coordinates are base-field (or Fp2) elements
CURVE_COORD_WORDS = {C_BLS12_381_G1: 6, C_BN254_G1: 4, C_BN254_G2: 8, C_BLS12_381_G2: 12}
CURVE_BASE_FIELD = {C_BLS12_381_G1: F_FP381, C_BN254_G1: F_FP254, C_BN254_G2: F_FP254, C_BLS12_381_G2: F_FP381}


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("lw_oracle.c", "lw_oracle.h", "orc_field.h", "orc_ntt_tmpl.h", "orc_ec_tmpl.h", "orc_keccak.h")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_field_elem_bytes.restype = C.c_size_t
        _lib.orc_curve_point_bytes.restype = C.c_size_t
        _lib.orc_optimum_window_size.restype = C.c_size_t
        _lib.orc_optimum_window_size.argtypes = [C.c_size_t]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _chk(rc):
    if rc < 0:
        raise OracleError(rc)
    return rc


# ---------------------------------------------------------------- integer <-> limb helpers
def int_to_limbs(v, words):
    return np.array([(v >> (64 * (words - 1 - i))) & 0xFFFFFFFFFFFFFFFF for i in range(words)], dtype=np.uint64)


def limbs_to_int(a):
    v = 0
    for w in np.asarray(a).reshape(-1):
        v = (v << 64) | int(w)
    return v


def ints_to_array(vals, words):
    out = np.zeros((len(vals), words), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = int_to_limbs(v, words)
    return out


def array_to_ints(a):
    a = np.asarray(a)
    if a.dtype == np.uint32:
        return [int(x) for x in a.reshape(-1)]
    return [limbs_to_int(row) for row in a.reshape(a.shape[0], -1)]


# ---------------------------------------------------------------- fields
def field_params(field):
    w = FIELD_WORDS[field] if field != F_BABYBEAR_EXT4 else 1
    q = np.zeros(6, np.uint64); r2 = np.zeros(6, np.uint64); one = np.zeros(6, np.uint64)
    mu = C.c_uint64(0)
    _chk(lib().orc_field_params(field, _p(q), C.byref(mu), _p(r2), _p(one)))
    return dict(q=limbs_to_int(q[:w]), mu=mu.value, r2=limbs_to_int(r2[:w]), one=limbs_to_int(one[:w]))


def derive_params(q_int, words):
    q = int_to_limbs(q_int, words)
    r2 = np.zeros(words, np.uint64); one = np.zeros(words, np.uint64)
    mu = C.c_uint64(0); spare = C.c_int(0)
    _chk(lib().orc_derive_params(words, _p(q), C.byref(mu), _p(r2), _p(one), C.byref(spare)))
    return dict(mu=mu.value, r2=limbs_to_int(r2), one=limbs_to_int(one), spare_bit=bool(spare.value))


def mont_cios(a, b, q, mu, words, spare=False):
    A, B, Q = (int_to_limbs(x, words) for x in (a, b, q))
    R = np.zeros(words, np.uint64)
    fn = lib().orc_mont_cios_spare if spare else lib().orc_mont_cios
    _chk(fn(words, _p(A), _p(B), _p(Q), C.c_uint64(mu), _p(R)))
    return limbs_to_int(R)


def fe_op_mod(q, words, op, a, b=0):
    """Raw (Montgomery-domain) field op on integers for an arbitrary modulus."""
    Q, A, B = (int_to_limbs(x, words) for x in (q, a, b))
    R = np.zeros(words, np.uint64)
    _chk(lib().orc_fe_op_mod(words, _p(Q), op, _p(A), _p(B), _p(R)))
    return limbs_to_int(R)


def fe_op(field, op, a, b=0):
    """Raw field op on integers holding the in-memory (Montgomery) words."""
    if field == F_BABYBEAR_U32:
        A = np.array([a], np.uint32); B = np.array([b], np.uint32); R = np.zeros(1, np.uint32)
        _chk(lib().orc_fe_op(field, op, _p(A), _p(B), _p(R)))
        return int(R[0])
    w = FIELD_WORDS[field]
    A, B = int_to_limbs(a, w), int_to_limbs(b, w)
    R = np.zeros(w, np.uint64)
    _chk(lib().orc_fe_op(field, op, _p(A), _p(B), _p(R)))
    return limbs_to_int(R)


def to_mont(field, v):
    return fe_op(field, OP_TO_MONT, v)


def from_mont(field, v):
    return fe_op(field, OP_FROM_MONT, v)


def elems_to_mont(field, vals):
    """Canonical integers -> numpy array in the reference's in-memory (Montgomery) form."""
    if field == F_BABYBEAR_U32:
        return np.array([to_mont(field, v) for v in vals], np.uint32)
    w = FIELD_WORDS[field]
    if field == F_BABYBEAR_EXT4:
        return ints_to_array([to_mont(F_BABYBEAR_U64, v) for v in vals], 1).reshape(-1, 4)
    return ints_to_array([to_mont(field, v) for v in vals], w)


def elems_from_mont(field, arr):
    bf = F_BABYBEAR_U64 if field == F_BABYBEAR_EXT4 else field
    flat = np.asarray(arr).reshape(-1) if bf in (F_BABYBEAR_U64, F_BABYBEAR_U32) else np.asarray(arr)
    if bf in (F_BABYBEAR_U64, F_BABYBEAR_U32):
        return [from_mont(bf, int(x)) for x in flat]
    return [from_mont(bf, v) for v in array_to_ints(flat)]


# ---------------------------------------------------------------- NTT
def _tw_field(field):
    return F_BABYBEAR_U64 if field == F_BABYBEAR_EXT4 else field


def _empty(field, n):
    if field == F_BABYBEAR_U32:
        return np.zeros(n, np.uint32)
    return np.zeros((n, FIELD_WORDS[field]), np.uint64)


def get_primitive_root_of_unity(field, order):
    out = _empty(_tw_field(field), 1)
    _chk(lib().orc_get_primitive_root_of_unity(_tw_field(field), C.c_uint64(order), _p(out)))
    return out[0]


def get_twiddles(field, order, config):
    if order > 63:
        raise OracleError(ERR_ORDER)
    n = (1 << order) // 2
    out = _empty(_tw_field(field), max(n, 1))
    _chk(lib().orc_get_twiddles(_tw_field(field), C.c_uint64(order), config, _p(out)))
    return out[:n]


def get_powers_of_primitive_root(field, n, count, config):
    cap = count
    if config in (ROOTS_BITREV, ROOTS_BITREV_INV):
        cap = 1
        while cap < count:
            cap <<= 1
    out = _empty(_tw_field(field), max(cap, 1))
    _chk(lib().orc_get_powers_of_primitive_root(_tw_field(field), C.c_uint64(n), C.c_size_t(count), config, _p(out)))
    return out[:cap] if count else out[:0]


def bit_reverse_permute(field, arr):
    a = np.ascontiguousarray(arr).copy()
    _chk(lib().orc_bit_reverse_permute(field, _p(a), C.c_size_t(a.shape[0])))
    return a


def in_place_nr_2radix_fft(field, arr, twiddles):
    a = np.ascontiguousarray(arr).copy()
    tw = np.ascontiguousarray(twiddles)
    _chk(lib().orc_in_place_nr_2radix_fft(field, _p(a), C.c_size_t(a.shape[0]), _p(tw)))
    return a


def fft(field, arr, twiddles):
    a = np.ascontiguousarray(arr)
    tw = np.ascontiguousarray(twiddles)
    out = np.empty_like(a)
    _chk(lib().orc_fft(field, _p(a), C.c_size_t(a.shape[0]), _p(tw), _p(out)))
    return out


def evaluate_fft(field, coeffs, blowup_factor=1, domain_size=None, offset=None):
    """Polynomial::evaluate_fft / evaluate_offset_fft on raw in-memory arrays."""
    a = np.ascontiguousarray(coeffs)
    n = a.shape[0]
    ds = 0 if domain_size is None else domain_size
    out_len = C.c_size_t(0)
    off = np.ascontiguousarray(offset) if offset is not None else None
    _chk(lib().orc_evaluate_fft(field, _p(a), C.c_size_t(n), C.c_size_t(blowup_factor), C.c_size_t(ds), _p(off),
                                None, C.byref(out_len)))
    out = _empty(field, out_len.value)
    _chk(lib().orc_evaluate_fft(field, _p(a), C.c_size_t(n), C.c_size_t(blowup_factor), C.c_size_t(ds), _p(off),
                                _p(out), C.byref(out_len)))
    return out


def interpolate_fft(field, evals, offset=None, strip=False):
    """Polynomial::interpolate_fft / interpolate_offset_fft. Returns all N coefficients unless strip."""
    a = np.ascontiguousarray(evals)
    out = np.empty_like(a)
    clen = C.c_size_t(0)
    off = np.ascontiguousarray(offset) if offset is not None else None
    _chk(lib().orc_interpolate_fft(field, _p(a), C.c_size_t(a.shape[0]), _p(off), _p(out), C.byref(clen)))
    return out[:clen.value] if strip else out


# ---------------------------------------------------------------- curves
def _pt(curve):
    return np.zeros(3 * CURVE_COORD_WORDS[curve], np.uint64)


def ec_neutral(curve):
    r = _pt(curve)
    _chk(lib().orc_ec_op(curve, EC_NEUTRAL, None, None, _p(r)))
    return r


def ec_add(curve, p, q):
    r = _pt(curve)
    _chk(lib().orc_ec_op(curve, EC_ADD, _p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(q)), _p(r)))
    return r


def ec_double(curve, p):
    r = _pt(curve)
    _chk(lib().orc_ec_op(curve, EC_DOUBLE, _p(np.ascontiguousarray(p)), None, _p(r)))
    return r


def ec_neg(curve, p):
    r = _pt(curve)
    _chk(lib().orc_ec_op(curve, EC_NEG, _p(np.ascontiguousarray(p)), None, _p(r)))
    return r


def ec_to_affine(curve, p):
    r = _pt(curve)
    _chk(lib().orc_ec_op(curve, EC_TO_AFFINE, _p(np.ascontiguousarray(p)), None, _p(r)))
    return r


def ec_eq(curve, p, q):
    return bool(_chk(lib().orc_ec_op(curve, EC_EQ, _p(np.ascontiguousarray(p)), _p(np.ascontiguousarray(q)), None)))


def ec_mul(curve, p, k, k_limbs=4):
    r = _pt(curve)
    K = int_to_limbs(k, k_limbs)
    _chk(lib().orc_ec_mul(curve, _p(np.ascontiguousarray(p)), _p(K), k_limbs, _p(r)))
    return r


def point_from_affine_ints(curve, x, y):
    """x, y canonical ints (or (c0,c1) tuples for G2) -> projective point array, Z = 1 (Montgomery)."""
    bf = CURVE_BASE_FIELD[curve]
    w = FIELD_WORDS[bf]
    one = field_params(bf)["one"]
    if curve in (C_BN254_G2, C_BLS12_381_G2):
        words = [to_mont(bf, x[0]), to_mont(bf, x[1]), to_mont(bf, y[0]), to_mont(bf, y[1]), one, 0]
    else:
        words = [to_mont(bf, x), to_mont(bf, y), one]
    return np.concatenate([int_to_limbs(v, w) for v in words])


def point_to_affine_ints(curve, p):
    """-> None for the neutral element, else canonical (x, y) ints (tuples of (c0,c1) for G2)."""
    bf = CURVE_BASE_FIELD[curve]
    w = FIELD_WORDS[bf]
    cw = CURVE_COORD_WORDS[curve]
    p = np.asarray(p).reshape(-1)
    if not p[2 * cw:].any():
        return None
    a = ec_to_affine(curve, p)
    comps = [from_mont(bf, limbs_to_int(a[i * w:(i + 1) * w])) for i in range(2 * cw // w)]
    if cw == w:
        return (comps[0], comps[1])
    return ((comps[0], comps[1]), (comps[2], comps[3]))


def msm(curve, scalars, points, k_limbs=4):
    """pippenger::msm. scalars: (n, k_limbs) uint64 canonical MS-first; points: (m, 3*coord_words)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, k_limbs)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 3 * CURVE_COORD_WORDS[curve])
    r = _pt(curve)
    _chk(lib().orc_msm(curve, _p(s), C.c_size_t(s.shape[0]), k_limbs, _p(pts), C.c_size_t(pts.shape[0]), _p(r)))
    return r


def msm_with(curve, scalars, points, window, k_limbs=4):
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, k_limbs)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 3 * CURVE_COORD_WORDS[curve])
    assert s.shape[0] == pts.shape[0]
    r = _pt(curve)
    _chk(lib().orc_msm_with(curve, _p(s), k_limbs, _p(pts), C.c_size_t(pts.shape[0]), C.c_size_t(window), _p(r)))
    return r


def parallel_msm_with(curve, scalars, points, window, threads, k_limbs=4):
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, k_limbs)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 3 * CURVE_COORD_WORDS[curve])
    assert s.shape[0] == pts.shape[0]
    r = _pt(curve)
    _chk(lib().orc_parallel_msm_with(curve, _p(s), k_limbs, _p(pts), C.c_size_t(pts.shape[0]), C.c_size_t(window),
                                     threads, _p(r)))
    return r


def msm_naive(curve, scalars, points, k_limbs=4):
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, k_limbs)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 3 * CURVE_COORD_WORDS[curve])
    r = _pt(curve)
    _chk(lib().orc_msm_naive(curve, _p(s), k_limbs, _p(pts), C.c_size_t(pts.shape[0]), _p(r)))
    return r


def optimum_window_size(n):
    return lib().orc_optimum_window_size(n)


def gen_points(curve, gen, s0, delta, n, k_limbs=4):
    out = np.zeros((n, 3 * CURVE_COORD_WORDS[curve]), np.uint64)
    _chk(lib().orc_gen_points(curve, _p(np.ascontiguousarray(gen)), _p(int_to_limbs(s0, k_limbs)),
                              _p(int_to_limbs(delta, k_limbs)), k_limbs, C.c_size_t(n), _p(out)))
    return out


# ---------------------------------------------------------------- Keccak / Merkle commitment
def keccak256(data: bytes) -> bytes:
    out = np.zeros(32, np.uint8)
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    _chk(lib().orc_keccak256_bytes(_p(np.ascontiguousarray(buf)), C.c_size_t(len(data)), _p(out)))
    return out.tobytes()


def merkle_commit_columns(columns, bit_reverse=True, threads=1):
    """columns: (n_cols, N, 4) uint64, natural-order LDE columns -> nodes array ((2N-1), 32) uint8, root = nodes[0]
    (BatchedMerkleTree over the rows of the bit-reverse-permuted columns, provers/stark/src/prover.rs:229-244).
    threads > 1 splits the (independent) hashes of the leaves and of every level over worker threads."""
    cols = np.ascontiguousarray(columns, dtype=np.uint64)
    n_cols, n = cols.shape[0], cols.shape[1]
    log2n = n.bit_length() - 1
    assert 1 << log2n == n
    nodes = np.zeros((2 * n - 1, 32), np.uint8)
    if threads > 1:
        _chk(lib().orc_merkle_commit_columns_mt(_p(cols), n_cols, log2n, 1 if bit_reverse else 0, _p(nodes), threads))
    else:
        _chk(lib().orc_merkle_commit_columns(_p(cols), n_cols, log2n, 1 if bit_reverse else 0, _p(nodes)))
    return nodes


def merkle_commit_columns_babybear(columns, bit_reverse=True, threads=1):
    """The same commitment over BabyBear columns: (n_cols, N) uint32 (U32MontgomeryBackendPrimeField: value().to_be_bytes(),
    u32_montgomery_backend_prime_field.rs:258-262) or uint64 (the u64-limb backend: its one limb big-endian)."""
    cols = np.ascontiguousarray(columns)
    assert cols.dtype in (np.uint32, np.uint64) and cols.ndim == 2
    n_cols, n = cols.shape
    log2n = n.bit_length() - 1
    assert 1 << log2n == n
    nodes = np.zeros((2 * n - 1, 32), np.uint8)
    _chk(lib().orc_merkle_commit_columns_bytes(_p(cols), cols.dtype.itemsize, n_cols, log2n, 1 if bit_reverse else 0, _p(nodes), threads))
    return nodes


def fri_fold_twice(field, coeffs, zeta, strip=True):
    """2 * fold_polynomial(p, zeta) (provers/stark/src/fri/mod.rs:49, fri/fri_functions.rs:7-30) on 4-limb elements."""
    a = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    z = np.ascontiguousarray(zeta, dtype=np.uint64).reshape(4)
    out = np.zeros(((a.shape[0] + 1) // 2, 4), np.uint64)
    olen = C.c_size_t(0)
    _chk(lib().orc_fri_fold_twice(field, _p(a), C.c_size_t(a.shape[0]), _p(z), _p(out), C.byref(olen)))
    return out[:olen.value] if strip else out


def groth16_h_coefficients(l, r, o, gates, strip=True):
    """QuadraticArithmeticProgram::calculate_h_coefficients (provers/groth16/src/qap.rs:15-39) on the accumulated L, R, O."""
    arrs = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (l, r, o)]
    out = np.zeros((2 * gates, 4), np.uint64)
    clen = C.c_size_t(0)
    _chk(lib().orc_groth16_h_coefficients(_p(arrs[0]), _p(arrs[1]), _p(arrs[2]), C.c_size_t(arrs[0].shape[0]), C.c_size_t(gates), _p(out),
                                          C.byref(clen)))
    return out[:clen.value] if strip else out
