"""ctypes binding of include/lw_hip.h (liblw_hip.so, gfx950 code objects only).

There is no CPU fallback anywhere in this package: if the HIP library is missing or no MI355X is
visible, every compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblw_hip.so")

# lw_field_t / lw_layout_t / lw_dir_t / lw_curve_t / lw_status_t (include/lw_hip.h)
FIELD_STARK252, FIELD_BLS12_381_FR, FIELD_BABYBEAR = 0, 1, 2
LAYOUT_U64_LIMBS_MS_FIRST, LAYOUT_BABYBEAR_U32_R32, LAYOUT_BABYBEAR_U64_R64, LAYOUT_EXT4_INTERLEAVED = 0, 1, 2, 3
DIR_FORWARD, DIR_INVERSE = 0, 1
CURVE_BLS12_381_G1, CURVE_BN254_G1, CURVE_BN254_G2, CURVE_BLS12_381_G2 = 0, 1, 2, 3
(OK, ERR_INPUT_NOT_POW2, ERR_ORDER_TOO_LARGE, ERR_ROOT_OF_UNITY, ERR_LENGTH_MISMATCH, ERR_NO_DEVICE, ERR_ALLOC,
 ERR_LAUNCH, ERR_COMM, ERR_BAD_ARG, ERR_INV_ZERO) = (0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10)

EXPORTS = [
    "lw_hip_init", "lw_hip_shutdown", "lw_hip_device_count", "lw_hip_last_error", "lw_hip_get_timings",
    "lw_hip_profile_begin", "lw_hip_profile_end", "lw_hip_result_acquire", "lw_hip_result_release",
    "lw_hip_field_elem_bytes", "lw_hip_curve_point_bytes", "lw_hip_ntt", "lw_hip_ntt_device", "lw_hip_ntt_cross_device",
    "lw_hip_gen_twiddles", "lw_hip_gen_powers", "lw_hip_bitrev_permutation", "lw_hip_ntt_lde_device",
    "lw_polynomial_evaluate_fft", "lw_polynomial_interpolate_fft", "lw_hip_msm", "lw_hip_msm_device",
    "lw_hip_msm_fr", "lw_hip_msm_fr_device", "lw_groth16_h_coefficients",
    "lw_stark_commit_columns", "lw_stark_commit_columns_device", "lw_stark_commit_columns_layout_device", "lw_stark_fri_layer",
    "lw_hip_srs_create", "lw_hip_srs_create_device", "lw_hip_srs_destroy", "lw_hip_msm_srs", "lw_hip_msm_srs_device",
    "lw_hip_msm_srs_fr", "lw_hip_msm_srs_fr_device", "lw_stark_fri_layer_device", "lw_groth16_h_coefficients_device",
    "lw_hip_ec_add_outer_device",
    "lw_hip_comm_unique_id", "lw_hip_comm_init", "lw_hip_comm_shutdown", "lw_hip_comm_info",
    "lw_hip_ntt_sharded_device", "lw_hip_ntt_sharded_selftest_device", "lw_hip_ntt_sharded_selftest_steps_device",
    "lw_hip_msm_sharded_device", "lw_hip_msm_sharded_selftest_device",
]


class Timings(C.Structure):
    _fields_ = [("last_ntt_ms", C.c_double), ("last_msm_ms", C.c_double), ("ntt_calls", C.c_uint64),
                ("msm_calls", C.c_uint64), ("twiddle_bytes", C.c_uint64), ("scratch_bytes", C.c_uint64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class Profile(C.Structure):
    _fields_ = [("n", C.c_int), ("k", KernelTime * 32)]


_lib = None


def lib():
    """Load liblw_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). lambda_elliptic_curves_amd has no CPU fallback.")
    # torch ships its own libamdhip64.so.7; two HIP runtimes in one process cannot both own the GPU.
    # Importing torch first makes the loader bind liblw_hip.so to the runtime torch already loaded (same
    # SONAME), so device pointers and streams are shared with torch (memory/stream plumbing only).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64p, i = C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint64), C.c_int
    L.lw_hip_init.argtypes = [C.POINTER(C.c_int), i]
    L.lw_hip_init.restype = i
    L.lw_hip_shutdown.restype = None
    L.lw_hip_device_count.restype = i
    L.lw_hip_last_error.restype = C.c_char_p
    L.lw_hip_get_timings.argtypes = [C.POINTER(Timings)]
    L.lw_hip_profile_begin.restype = i
    L.lw_hip_profile_end.argtypes = [C.POINTER(Profile)]
    L.lw_hip_profile_end.restype = i
    L.lw_hip_result_acquire.argtypes = [sz, C.POINTER(vp)]
    L.lw_hip_result_acquire.restype = i
    L.lw_hip_result_release.argtypes = [vp]
    L.lw_hip_result_release.restype = i
    L.lw_hip_field_elem_bytes.argtypes = [i, i]
    L.lw_hip_field_elem_bytes.restype = sz
    L.lw_hip_curve_point_bytes.argtypes = [i]
    L.lw_hip_curve_point_bytes.restype = sz
    L.lw_hip_ntt.argtypes = [i, i, i, vp, vp, u32, u32, sz, vp]
    L.lw_hip_ntt.restype = i
    L.lw_hip_ntt_device.argtypes = [i, i, i, vp, vp, u32, u32, sz, vp, vp]
    L.lw_hip_ntt_device.restype = i
    L.lw_hip_ntt_cross_device.argtypes = [i, i, i, vp, vp, u32, u32, C.c_uint64, C.c_uint64, C.c_uint64, u32, C.c_uint64, vp]
    L.lw_hip_ntt_cross_device.restype = i
    L.lw_hip_ntt_lde_device.argtypes = [i, i, vp, u32, vp, u32, u32, vp, vp]
    L.lw_hip_ntt_lde_device.restype = i
    L.lw_hip_gen_twiddles.argtypes = [i, i, C.c_uint64, i, vp]
    L.lw_hip_gen_twiddles.restype = i
    L.lw_hip_gen_powers.argtypes = [i, i, C.c_uint64, sz, i, vp, vp, C.POINTER(sz)]
    L.lw_hip_gen_powers.restype = i
    L.lw_hip_bitrev_permutation.argtypes = [i, i, vp, vp, sz]
    L.lw_hip_bitrev_permutation.restype = i
    L.lw_polynomial_evaluate_fft.argtypes = [i, i, vp, sz, sz, sz, vp, vp, sz, C.POINTER(sz)]
    L.lw_polynomial_evaluate_fft.restype = i
    L.lw_polynomial_interpolate_fft.argtypes = [i, i, vp, sz, vp, vp, C.POINTER(sz)]
    L.lw_polynomial_interpolate_fft.restype = i
    L.lw_hip_msm.argtypes = [i, vp, sz, vp, sz, vp]
    L.lw_hip_msm.restype = i
    L.lw_hip_msm_device.argtypes = [i, vp, vp, sz, vp, vp]
    L.lw_hip_msm_device.restype = i
    L.lw_stark_commit_columns.argtypes = [i, vp, u32, u32, i, vp, vp]
    L.lw_stark_commit_columns.restype = i
    L.lw_stark_commit_columns_device.argtypes = [i, vp, u32, C.c_uint64, u32, i, vp, vp, vp]
    L.lw_stark_commit_columns_device.restype = i
    L.lw_stark_commit_columns_layout_device.argtypes = [i, i, vp, u32, C.c_uint64, u32, i, vp, vp, vp]
    L.lw_stark_commit_columns_layout_device.restype = i
    L.lw_stark_fri_layer.argtypes = [i, vp, sz, vp, vp, sz, vp, C.POINTER(sz), vp, vp, vp]
    L.lw_stark_fri_layer.restype = i
    L.lw_groth16_h_coefficients.argtypes = [vp, vp, vp, sz, sz, vp, C.POINTER(sz)]
    L.lw_groth16_h_coefficients.restype = i
    L.lw_hip_msm_fr.argtypes = [i, vp, sz, vp, sz, vp]
    L.lw_hip_msm_fr.restype = i
    L.lw_hip_msm_fr_device.argtypes = [i, vp, vp, sz, vp, vp]
    L.lw_hip_msm_fr_device.restype = i
    L.lw_hip_srs_create.argtypes = [i, vp, sz, C.POINTER(vp)]
    L.lw_hip_srs_create.restype = i
    L.lw_hip_srs_create_device.argtypes = [i, vp, sz, vp, C.POINTER(vp)]
    L.lw_hip_srs_create_device.restype = i
    L.lw_hip_srs_destroy.argtypes = [vp]
    L.lw_hip_srs_destroy.restype = i
    L.lw_hip_msm_srs.argtypes = [vp, vp, sz, vp]
    L.lw_hip_msm_srs.restype = i
    L.lw_hip_msm_srs_fr.argtypes = [vp, vp, sz, vp]
    L.lw_hip_msm_srs_fr.restype = i
    L.lw_hip_msm_srs_device.argtypes = [vp, vp, sz, vp, vp]
    L.lw_hip_msm_srs_device.restype = i
    L.lw_hip_msm_srs_fr_device.argtypes = [vp, vp, sz, vp, vp]
    L.lw_hip_msm_srs_fr_device.restype = i
    L.lw_stark_fri_layer_device.argtypes = [i, vp, sz, vp, vp, sz, vp, vp, vp, vp, vp]
    L.lw_stark_fri_layer_device.restype = i
    L.lw_groth16_h_coefficients_device.argtypes = [vp, vp, vp, sz, sz, vp, C.POINTER(sz), vp]
    L.lw_groth16_h_coefficients_device.restype = i
    L.lw_hip_ec_add_outer_device.argtypes = [i, vp, sz, vp, sz, vp, vp]
    L.lw_hip_ec_add_outer_device.restype = i
    L.lw_hip_comm_unique_id.argtypes = [vp]
    L.lw_hip_comm_unique_id.restype = i
    L.lw_hip_comm_init.argtypes = [vp, i, i]
    L.lw_hip_comm_init.restype = i
    L.lw_hip_comm_shutdown.restype = i
    L.lw_hip_comm_info.argtypes = [C.POINTER(i), C.POINTER(i)]
    L.lw_hip_comm_info.restype = i
    L.lw_hip_ntt_sharded_device.argtypes = [i, i, i, vp, vp, u32, u32, i, vp]
    L.lw_hip_ntt_sharded_device.restype = i
    L.lw_hip_ntt_sharded_selftest_device.argtypes = [i, i, i, vp, vp, u32, u32, u32, i, vp]
    L.lw_hip_ntt_sharded_selftest_device.restype = i
    L.lw_hip_ntt_sharded_selftest_steps_device.argtypes = [i, i, i, vp, vp, u32, u32, u32, i, i, vp]
    L.lw_hip_ntt_sharded_selftest_steps_device.restype = i
    L.lw_hip_msm_sharded_device.argtypes = [i, vp, vp, sz, vp, vp]
    L.lw_hip_msm_sharded_device.restype = i
    L.lw_hip_msm_sharded_selftest_device.argtypes = [i, vp, vp, sz, u32, vp, vp]
    L.lw_hip_msm_sharded_selftest_device.restype = i
    _lib = L
    return L


def profile_begin():
    rc = lib().lw_hip_profile_begin()
    if rc:
        raise RuntimeError(f"lw_hip_profile_begin: [{rc}] {last_error()}")


def profile_end():
    """-> {kernel name: (launches, total_ms)} measured with HIP events on the launch stream."""
    p = Profile()
    rc = lib().lw_hip_profile_end(C.byref(p))
    if rc:
        raise RuntimeError(f"lw_hip_profile_end: [{rc}] {last_error()}")
    return {p.k[j].name.decode(): (int(p.k[j].launches), float(p.k[j].total_ms)) for j in range(p.n)}


def last_error():
    return lib().lw_hip_last_error().decode("utf-8", "replace")
