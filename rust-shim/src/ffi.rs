//! `extern "C"` declarations of include/lw_hip.h, one to one.  UNVERIFIED: written without a Rust toolchain.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

/// lw_field_t
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Field {
    Stark252 = 0,
    Bls12381Fr = 1,
    BabyBear = 2,
}

/// lw_layout_t
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Layout {
    /// MontgomeryBackendPrimeField<_, 4>: 4 x u64, most significant limb first, R = 2^256
    U64LimbsMsFirst = 0,
    /// U32MontgomeryBackendPrimeField: one u32, R = 2^32
    BabyBearU32R32 = 1,
    /// MontgomeryBackendPrimeField<_, 1>: one u64, R = 2^64
    BabyBearU64R64 = 2,
    /// Degree4BabyBearExtensionField values: 4 x u64 per element, domain in the base field
    Ext4Interleaved = 3,
}

/// lw_dir_t
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Dir {
    Forward = 0,
    /// scaled by N^-1
    Inverse = 1,
}

/// lw_curve_t
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Curve {
    Bls12381G1 = 0,
    Bn254G1 = 1,
    Bn254G2 = 2,
    Bls12381G2 = 3,
}

// lw_status_t
pub const LW_OK: c_int = 0;
pub const LW_ERR_INPUT_NOT_POW2: c_int = -1;
pub const LW_ERR_ORDER_TOO_LARGE: c_int = -2;
pub const LW_ERR_ROOT_OF_UNITY: c_int = -3;
pub const LW_ERR_LENGTH_MISMATCH: c_int = -4;
pub const LW_ERR_NO_DEVICE: c_int = -5;
pub const LW_ERR_ALLOC: c_int = -6;
pub const LW_ERR_LAUNCH: c_int = -7;
pub const LW_ERR_COMM: c_int = -8;
pub const LW_ERR_BAD_ARG: c_int = -9;
pub const LW_ERR_INV_ZERO: c_int = -10;

pub const LW_HIP_COMM_ID_BYTES: usize = 128;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct lw_timings_t {
    pub last_ntt_ms: f64,
    pub last_msm_ms: f64,
    pub ntt_calls: u64,
    pub msm_calls: u64,
    pub twiddle_bytes: u64,
    pub scratch_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct lw_kernel_time_t {
    pub name: [c_char; 48],
    pub launches: u64,
    pub total_ms: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct lw_profile_t {
    pub n: c_int,
    pub k: [lw_kernel_time_t; 32],
}

/// opaque `lw_srs_t`
#[repr(C)]
pub struct lw_srs_t {
    _private: [u8; 0],
}

extern "C" {
    // ---- context
    pub fn lw_hip_init(device_ids: *const c_int, n_devices: c_int) -> c_int;
    pub fn lw_hip_shutdown();
    pub fn lw_hip_device_count() -> c_int;
    pub fn lw_hip_last_error() -> *const c_char;
    pub fn lw_hip_get_timings(out: *mut lw_timings_t) -> c_int;
    pub fn lw_hip_profile_begin() -> c_int;
    pub fn lw_hip_profile_end(out: *mut lw_profile_t) -> c_int;
    pub fn lw_hip_field_elem_bytes(field: Field, layout: Layout) -> usize;
    pub fn lw_hip_curve_point_bytes(curve: Curve) -> usize;

    // ---- NTT backend seam
    pub fn lw_hip_result_acquire(bytes: usize, out_ptr: *mut *mut c_void) -> c_int;
    pub fn lw_hip_result_release(ptr: *mut c_void) -> c_int;
    pub fn lw_hip_ntt(field: Field, layout: Layout, dir: Dir, input: *const c_void, output: *mut c_void, log2n: u32,
                      batch: u32, batch_stride_elems: usize, coset_offset_or_null: *const c_void) -> c_int;
    pub fn lw_hip_ntt_device(field: Field, layout: Layout, dir: Dir, d_in: *const c_void, d_out: *mut c_void, log2n: u32,
                             batch: u32, batch_stride_elems: usize, coset_offset_or_null: *const c_void,
                             hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_ntt_lde_device(field: Field, layout: Layout, d_coeffs: *const c_void, log2_coeffs: u32, d_out: *mut c_void,
                                 log2n: u32, batch: u32, coset_offset_or_null: *const c_void, hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_gen_twiddles(field: Field, layout: Layout, order: u64, config: c_int, out: *mut c_void) -> c_int;
    pub fn lw_hip_gen_powers(field: Field, layout: Layout, order: u64, count: usize, config: c_int, offset_or_null: *const c_void,
                             out: *mut c_void, out_len: *mut usize) -> c_int;
    pub fn lw_hip_bitrev_permutation(field: Field, layout: Layout, input: *const c_void, output: *mut c_void, n: usize) -> c_int;
    pub fn lw_hip_ntt_cross_device(field: Field, layout: Layout, dir: Dir, d_in: *const c_void, d_out: *mut c_void,
                                   log2n_total: u32, log2_shards: u32, j2_begin: u64, slice_len: u64, chunk_stride_elems: u64,
                                   batch: u32, batch_stride_elems: u64, hip_stream: *mut c_void) -> c_int;

    // ---- multi-GPU (library-owned RCCL communicator)
    pub fn lw_hip_comm_unique_id(out_id: *mut u8) -> c_int;
    pub fn lw_hip_comm_init(unique_id: *const u8, rank: c_int, nranks: c_int) -> c_int;
    pub fn lw_hip_comm_shutdown() -> c_int;
    pub fn lw_hip_comm_info(rank: *mut c_int, nranks: *mut c_int) -> c_int;
    pub fn lw_hip_ntt_sharded_device(field: Field, layout: Layout, dir: Dir, d_in_local: *const c_void, d_out_local: *mut c_void,
                                     log2n_total: u32, batch: u32, natural_output: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_ntt_sharded_selftest_device(field: Field, layout: Layout, dir: Dir, d_in_full: *const c_void,
                                              d_out_full: *mut c_void, log2n_total: u32, log2_shards: u32, batch: u32,
                                              natural_output: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_ntt_sharded_selftest_steps_device(field: Field, layout: Layout, dir: Dir, d_in_full: *const c_void,
                                                    d_out_full: *mut c_void, log2n_total: u32, log2_shards: u32, batch: u32,
                                                    natural_output: c_int, stop_after: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_msm_sharded_device(curve: Curve, d_scalars: *const u64, d_points: *const c_void, n_local: usize,
                                     out_point_host: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_msm_sharded_selftest_device(curve: Curve, d_scalars: *const u64, d_points: *const c_void, n_total: usize, log2_shards: u32,
                                              out_point_host: *mut c_void, hip_stream: *mut c_void) -> c_int;

    // ---- Polynomial FFT API (host buffers, reference semantics)
    pub fn lw_polynomial_evaluate_fft(field: Field, layout: Layout, coeffs: *const c_void, n_coeffs: usize, blowup_factor: usize,
                                      domain_size: usize, offset_or_null: *const c_void, out: *mut c_void,
                                      out_capacity_elems: usize, out_len: *mut usize) -> c_int;
    pub fn lw_polynomial_interpolate_fft(field: Field, layout: Layout, evals: *const c_void, n: usize,
                                         offset_or_null: *const c_void, out_coeffs: *mut c_void, coeff_len: *mut usize) -> c_int;

    // ---- STARK commitment / FRI layer / Groth16 quotient
    pub fn lw_stark_commit_columns(field: Field, columns: *const c_void, n_cols: u32, log2n: u32, bit_reverse: c_int,
                                   out_root: *mut u8, out_nodes_or_null: *mut u8) -> c_int;
    pub fn lw_stark_commit_columns_device(field: Field, d_columns: *const c_void, n_cols: u32, col_stride_elems: u64, log2n: u32,
                                          bit_reverse: c_int, d_nodes: *mut c_void, out_root_or_null: *mut u8,
                                          hip_stream: *mut c_void) -> c_int;
    pub fn lw_stark_commit_columns_layout_device(field: Field, layout: Layout, d_columns: *const c_void, n_cols: u32, col_stride_elems: u64,
                                                 log2n: u32, bit_reverse: c_int, d_nodes: *mut c_void, out_root_or_null: *mut u8,
                                                 hip_stream: *mut c_void) -> c_int;
    pub fn lw_stark_fri_layer(field: Field, coeffs: *const c_void, n_coeffs: usize, zeta: *const c_void, coset_offset: *const c_void,
                              domain_size: usize, out_poly: *mut c_void, out_poly_len: *mut usize, out_evaluation: *mut c_void,
                              out_root: *mut u8, out_nodes_or_null: *mut u8) -> c_int;
    pub fn lw_stark_fri_layer_device(field: Field, d_coeffs: *const c_void, n_coeffs: usize, zeta: *const c_void,
                                     coset_offset: *const c_void, domain_size: usize, d_out_poly: *mut c_void,
                                     d_out_evaluation_or_null: *mut c_void, d_nodes_or_null: *mut c_void, out_root_or_null: *mut u8,
                                     hip_stream: *mut c_void) -> c_int;
    pub fn lw_groth16_h_coefficients_device(d_l: *const c_void, d_r: *const c_void, d_o: *const c_void, n_coeffs: usize,
                                            num_gates: usize, d_out_h: *mut c_void, coeff_len_or_null: *mut usize,
                                            hip_stream: *mut c_void) -> c_int;
    pub fn lw_groth16_h_coefficients(l_coeffs: *const c_void, r_coeffs: *const c_void, o_coeffs: *const c_void, n_coeffs: usize,
                                     num_gates: usize, out_h: *mut c_void, coeff_len: *mut usize) -> c_int;

    // ---- MSM
    pub fn lw_hip_msm(curve: Curve, scalars: *const u64, n_scalars: usize, points: *const c_void, n_points: usize,
                      out_point: *mut c_void) -> c_int;
    pub fn lw_hip_msm_device(curve: Curve, d_scalars: *const u64, d_points: *const c_void, n: usize, out_point_host: *mut c_void,
                             hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_msm_fr(curve: Curve, fr_elements: *const u64, n_scalars: usize, points: *const c_void, n_points: usize,
                         out_point: *mut c_void) -> c_int;
    pub fn lw_hip_msm_fr_device(curve: Curve, d_fr_elements: *const u64, d_points: *const c_void, n: usize,
                                out_point_host: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_ec_add_outer_device(curve: Curve, d_rows: *const c_void, m: usize, d_cols: *const c_void, k: usize,
                                      d_out: *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_srs_create(curve: Curve, points: *const c_void, n_points: usize, out_srs: *mut *mut lw_srs_t) -> c_int;
    pub fn lw_hip_srs_create_device(curve: Curve, d_points: *const c_void, n_points: usize, hip_stream: *mut c_void,
                                    out_srs: *mut *mut lw_srs_t) -> c_int;
    pub fn lw_hip_srs_destroy(srs: *mut lw_srs_t) -> c_int;
    pub fn lw_hip_msm_srs(srs: *const lw_srs_t, scalars: *const u64, n_scalars: usize, out_point: *mut c_void) -> c_int;
    pub fn lw_hip_msm_srs_device(srs: *const lw_srs_t, d_scalars: *const u64, n_scalars: usize, out_point_host: *mut c_void,
                                 hip_stream: *mut c_void) -> c_int;
    pub fn lw_hip_msm_srs_fr(srs: *const lw_srs_t, fr_elements: *const u64, n_scalars: usize, out_point: *mut c_void) -> c_int;
    pub fn lw_hip_msm_srs_fr_device(srs: *const lw_srs_t, d_fr_elements: *const u64, n_scalars: usize, out_point_host: *mut c_void,
                                    hip_stream: *mut c_void) -> c_int;
}

// The C structs above must keep the sizes the header gives them.
const _: () = assert!(core::mem::size_of::<lw_timings_t>() == 48);
const _: () = assert!(core::mem::size_of::<lw_kernel_time_t>() == 64);
const _: () = assert!(core::mem::size_of::<lw_profile_t>() == 8 + 32 * 64);
