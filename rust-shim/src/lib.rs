//! lambdaworks-hip — safe wrappers over liblw_hip.so (include/lw_hip.h): the MI355X (gfx950) NTT + MSM backend.
//!
//! UNVERIFIED: written without a Rust toolchain (see README.md).
//!
//! This crate knows nothing about lambdaworks' types: every function takes raw element slices whose bytes are the
//! reference's in-memory representation (Montgomery form, `u64` limbs most significant first — exactly what
//! `FieldElement::value()` holds, math/src/gpu/cuda/field/element.rs:30-42 passes the same bytes to CUDA).  The typed
//! entry points (`evaluate_fft_hip`, `interpolate_fft_hip`, `msm_hip`) live in the lambdaworks tree
//! (rust-shim/lambdaworks/…), the way the CUDA ones live in math/src/fft/gpu/cuda/polynomial.rs.
pub mod error;
pub mod ffi;

pub use error::{check, HipError};
pub use ffi::{Curve, Dir, Field, Layout};

use core::ffi::{c_int, c_void};
use core::mem::size_of;
use core::ptr;

/// One context per process, bound to one device (`lw_hip_init`).  Optional: every entry point initialises lazily on
/// the calling thread's current device.
pub fn init(device: Option<i32>) -> Result<(), HipError> {
    // SAFETY: the pointer is valid for one c_int or NULL with n = 0.
    let rc = unsafe {
        match device {
            Some(d) => ffi::lw_hip_init(&d as *const c_int, 1),
            None => ffi::lw_hip_init(ptr::null(), 0),
        }
    };
    check(rc)
}

pub fn shutdown() {
    // SAFETY: no arguments; idempotent.
    unsafe { ffi::lw_hip_shutdown() }
}

pub fn device_count() -> usize {
    // SAFETY: no arguments.
    unsafe { ffi::lw_hip_device_count().max(0) as usize }
}

pub fn field_elem_bytes(field: Field, layout: Layout) -> usize {
    // SAFETY: pure function of its arguments.
    unsafe { ffi::lw_hip_field_elem_bytes(field, layout) }
}

pub fn curve_point_bytes(curve: Curve) -> usize {
    // SAFETY: pure function of its arguments.
    unsafe { ffi::lw_hip_curve_point_bytes(curve) }
}

/// Marker for element types that are plain data: any bit pattern of `size_of::<Self>()` bytes is a valid value, there is
/// no padding, no `Drop` and no interior pointer.  The library fills such values byte for byte.
///
/// # Safety
/// Implement it only for types for which that holds — `[u64; N]`, and the `#[repr(transparent)]`-style newtype nesting of
/// `FieldElement<MontgomeryBackendPrimeField<_, N>>` (math/src/field/element.rs:40-42, unsigned_integer/element.rs:29-37),
/// which the typed modules under rust-shim/lambdaworks/ assert by size.
pub unsafe trait Pod: Copy + 'static {}
unsafe impl Pod for u32 {}
unsafe impl Pod for u64 {}
unsafe impl<const N: usize> Pod for [u64; N] {}
unsafe impl<const N: usize> Pod for [u32; N] {}

fn check_elems<T>(field: Field, layout: Layout) -> Result<(), HipError> {
    if size_of::<T>() != field_elem_bytes(field, layout) {
        return Err(HipError::BadArgument(format!(
            "element type is {} bytes, the backend expects {} for {:?}/{:?}",
            size_of::<T>(),
            field_elem_bytes(field, layout),
            field,
            layout
        )));
    }
    Ok(())
}

/// The backend seam on host slices — what `evaluate_fft_cuda` / `interpolate_fft_cuda` are to CUDA
/// (math/src/fft/gpu/cuda/polynomial.rs:16-49).  `input.len()` must be a power of two (it is the already padded
/// coefficient / evaluation vector); `T` is the element type as it sits in memory.  `Dir::Inverse` results are already
/// multiplied by N^-1.  `coset_offset`: one domain-field element (same layout's base word) or `None`.
pub fn ntt<T: Pod, O: Pod>(field: Field, layout: Layout, dir: Dir, input: &[T], coset_offset: Option<&O>) -> Result<Vec<T>, HipError> {
    // SAFETY: T and O are Pod.
    unsafe { ntt_unchecked(field, layout, dir, input, coset_offset) }
}

/// [`ntt`] without the `Pod` bounds, for element types the caller vouches for (the typed lambdaworks modules: a
/// `FieldElement<F>` whose `field_name()` and size matched a kernel family is the bare limb array).
///
/// # Safety
/// `T` must be plain data of exactly the backend's element size: every byte pattern the library writes must be a valid
/// `T`, `T` must not implement `Drop`; `O` likewise must be readable as raw bytes of one domain-field element.
pub unsafe fn ntt_unchecked<T, O>(field: Field, layout: Layout, dir: Dir, input: &[T], coset_offset: Option<&O>) -> Result<Vec<T>, HipError> {
    check_elems::<T>(field, layout)?;
    let n = input.len();
    if n == 0 || !n.is_power_of_two() {
        return Err(HipError::InputNotPowerOfTwo(format!("Input length is {n}, which is not a power of two")));
    }
    let mut out: Vec<T> = Vec::with_capacity(n);
    let off = coset_offset.map_or(ptr::null(), |o| o as *const O as *const c_void);
    // SAFETY: `input` holds n elements of the size the library expects (checked above); `out` has capacity for n; the
    // library writes exactly n elements on success and retains no pointer.  (A fresh Vec has never been touched: the
    // library asks for huge pages on it and populates it while the upload and the kernels run, include/lw_hip.h.)
    let rc = unsafe {
        ffi::lw_hip_ntt(field, layout, dir, input.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void, n.trailing_zeros(), 1, 0, off)
    };
    check(rc)?;
    // SAFETY: all n elements were initialised by the call above and T: Pod accepts any bytes.
    unsafe { out.set_len(n) };
    Ok(out)
}

/// The same transform into a caller-provided slice (`out.len() == input.len()`): no allocation, and the place to pass a
/// [`HipBuf`] so that the download lands in pinned memory.
pub fn ntt_into<T: Pod, O: Pod>(field: Field, layout: Layout, dir: Dir, input: &[T], out: &mut [T], coset_offset: Option<&O>) -> Result<(), HipError> {
    check_elems::<T>(field, layout)?;
    let n = input.len();
    if n == 0 || !n.is_power_of_two() {
        return Err(HipError::InputNotPowerOfTwo(format!("Input length is {n}, which is not a power of two")));
    }
    if out.len() != n {
        return Err(HipError::BadArgument(format!("output slice holds {} elements, the transform has {n}", out.len())));
    }
    let off = coset_offset.map_or(ptr::null(), |o| o as *const O as *const c_void);
    // SAFETY: both slices are valid for n elements of the checked size; `in` may alias `out`; nothing is retained.
    let rc = unsafe {
        ffi::lw_hip_ntt(field, layout, dir, input.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void, n.trailing_zeros(), 1, 0, off)
    };
    check(rc)
}

/// A result buffer from the library's pool of pinned, resident host memory (`lw_hip_result_acquire`): derefs to `[T]`,
/// goes back to the pool on drop.  Downloads into it run at the PCIe rate; a fresh `Vec` of the same size first pays one
/// page fault per 4 KiB.  For provers that keep evaluation vectors alive across calls (the LDE columns of a STARK round).
pub struct HipBuf<T: Pod> {
    ptr: ptr::NonNull<T>,
    len: usize,
}
impl<T: Pod> HipBuf<T> {
    pub fn new(len: usize) -> Result<Self, HipError> {
        let mut p: *mut c_void = ptr::null_mut();
        // SAFETY: out pointer valid; the library returns memory aligned for any element type (page aligned).
        check(unsafe { ffi::lw_hip_result_acquire(len.max(1) * size_of::<T>(), &mut p) })?;
        // memory from the pool may hold an earlier result: any bytes are a valid T (Pod)
        Ok(HipBuf { ptr: ptr::NonNull::new(p as *mut T).ok_or_else(|| HipError::AllocateMemory("null result buffer".into()))?, len })
    }
}
impl<T: Pod> core::ops::Deref for HipBuf<T> {
    type Target = [T];
    fn deref(&self) -> &[T] {
        // SAFETY: ptr is valid for len elements until drop.
        unsafe { core::slice::from_raw_parts(self.ptr.as_ptr(), self.len) }
    }
}
impl<T: Pod> core::ops::DerefMut for HipBuf<T> {
    fn deref_mut(&mut self) -> &mut [T] {
        // SAFETY: unique owner.
        unsafe { core::slice::from_raw_parts_mut(self.ptr.as_ptr(), self.len) }
    }
}
impl<T: Pod> Drop for HipBuf<T> {
    fn drop(&mut self) {
        // SAFETY: the pointer came from lw_hip_result_acquire and is released exactly once.
        unsafe { ffi::lw_hip_result_release(self.ptr.as_ptr() as *mut c_void) };
    }
}
// SAFETY: plain memory owned by the value.
unsafe impl<T: Pod> Send for HipBuf<T> {}

/// `batch` transforms of 2^log2n elements, `stride` elements apart, in place of a rayon loop over columns
/// (provers/stark/src/trace.rs:186-190): one call, one upload.
pub fn ntt_batch_in_place<T: Pod>(field: Field, layout: Layout, dir: Dir, data: &mut [T], log2n: u32, batch: u32, stride: usize) -> Result<(), HipError> {
    check_elems::<T>(field, layout)?;
    let n = 1usize << log2n;
    let stride_eff = if stride == 0 { n } else { stride };
    if batch as usize > 0 && (batch as usize - 1) * stride_eff + n > data.len() {
        return Err(HipError::BadArgument("batch does not fit the slice".into()));
    }
    // SAFETY: extent checked above; `in` may alias `out` (include/lw_hip.h).
    let rc = unsafe {
        ffi::lw_hip_ntt(field, layout, dir, data.as_ptr() as *const c_void, data.as_mut_ptr() as *mut c_void, log2n, batch, stride, ptr::null())
    };
    check(rc)
}

/// `msm::pippenger::msm` on host slices: `scalars` are canonical U256 integers (4 x u64, most significant limb first —
/// what `.representative()` returns), `points` projective points as they sit in memory (`P` = the point type).
/// Returns the normalised representative (x/z : y/z : 1) or (0 : 1 : 0).
pub fn msm<P: Copy>(curve: Curve, scalars: &[[u64; 4]], points: &[P]) -> Result<P, HipError> {
    if size_of::<P>() != curve_point_bytes(curve) {
        return Err(HipError::BadArgument(format!("point type is {} bytes, the backend expects {}", size_of::<P>(), curve_point_bytes(curve))));
    }
    let mut out = core::mem::MaybeUninit::<P>::uninit();
    // SAFETY: slices are valid for their lengths; the library writes one point of curve_point_bytes(curve) bytes on
    // success (also for empty input: the neutral element) and nothing is retained.
    let rc = unsafe {
        ffi::lw_hip_msm(curve, scalars.as_ptr() as *const u64, scalars.len(), points.as_ptr() as *const c_void, points.len(), out.as_mut_ptr() as *mut c_void)
    };
    check(rc)?;
    // SAFETY: initialised by the successful call.
    Ok(unsafe { out.assume_init() })
}

/// Same with the scalars given as stored `FrElement`s (Montgomery form): the `.representative()` loop every reference
/// caller runs on the CPU first (provers/groth16/src/prover.rs:69-78) happens on the device.
pub fn msm_fr<P: Copy>(curve: Curve, fr_elements: &[[u64; 4]], points: &[P]) -> Result<P, HipError> {
    if size_of::<P>() != curve_point_bytes(curve) {
        return Err(HipError::BadArgument("point type has the wrong size".into()));
    }
    let mut out = core::mem::MaybeUninit::<P>::uninit();
    // SAFETY: as in `msm`.
    let rc = unsafe {
        ffi::lw_hip_msm_fr(curve, fr_elements.as_ptr() as *const u64, fr_elements.len(), points.as_ptr() as *const c_void, points.len(), out.as_mut_ptr() as *mut c_void)
    };
    check(rc)?;
    // SAFETY: initialised by the successful call.
    Ok(unsafe { out.assume_init() })
}

/// A fixed point set kept on the device in affine form (`lw_hip_srs_*`): KZG's `srs.powers_main_group`
/// (crypto/src/commitments/kzg.rs:159-163) or a Groth16 proving-key vector (provers/groth16/src/prover.rs:69-85).
pub struct Srs {
    handle: *mut ffi::lw_srs_t,
    curve: Curve,
    len: usize,
}

// SAFETY: the handle is only used through the library, which serialises all calls on its context lock.
unsafe impl Send for Srs {}
unsafe impl Sync for Srs {}

impl Srs {
    pub fn new<P: Copy>(curve: Curve, points: &[P]) -> Result<Self, HipError> {
        if size_of::<P>() != curve_point_bytes(curve) {
            return Err(HipError::BadArgument("point type has the wrong size".into()));
        }
        let mut handle: *mut ffi::lw_srs_t = ptr::null_mut();
        // SAFETY: `points` is valid for its length; `handle` receives an owned handle on success.
        let rc = unsafe { ffi::lw_hip_srs_create(curve, points.as_ptr() as *const c_void, points.len(), &mut handle) };
        check(rc)?;
        Ok(Self { handle, curve, len: points.len() })
    }

    pub fn len(&self) -> usize {
        self.len
    }

    pub fn is_empty(&self) -> bool {
        self.len == 0
    }

    /// `msm(scalars, &points[..scalars.len()])` — fewer scalars than points is the KZG call shape.
    pub fn msm<P: Copy>(&self, scalars: &[[u64; 4]]) -> Result<P, HipError> {
        if size_of::<P>() != curve_point_bytes(self.curve) {
            return Err(HipError::BadArgument("point type has the wrong size".into()));
        }
        let mut out = core::mem::MaybeUninit::<P>::uninit();
        // SAFETY: the handle is live until drop; one point is written on success.
        let rc = unsafe { ffi::lw_hip_msm_srs(self.handle, scalars.as_ptr() as *const u64, scalars.len(), out.as_mut_ptr() as *mut c_void) };
        check(rc)?;
        // SAFETY: initialised by the successful call.
        Ok(unsafe { out.assume_init() })
    }
}

impl Drop for Srs {
    fn drop(&mut self) {
        // SAFETY: the handle came from lw_hip_srs_create and is destroyed exactly once.
        unsafe {
            ffi::lw_hip_srs_destroy(self.handle);
        }
    }
}

/// The library-owned RCCL communicator (one process per GPU).  Rank 0 calls `Comm::unique_id()` and hands the bytes to
/// the other processes out of band (a file, MPI, a socket — `ncclGetUniqueId`'s contract); then every process calls
/// `Comm::init` (collective).
pub struct Comm {
    pub rank: i32,
    pub nranks: i32,
}

impl Comm {
    pub fn unique_id() -> Result<[u8; ffi::LW_HIP_COMM_ID_BYTES], HipError> {
        let mut id = [0u8; ffi::LW_HIP_COMM_ID_BYTES];
        // SAFETY: the buffer has the length the header names.
        check(unsafe { ffi::lw_hip_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }

    pub fn init(unique_id: &[u8; ffi::LW_HIP_COMM_ID_BYTES], rank: i32, nranks: i32) -> Result<Self, HipError> {
        // SAFETY: the id has the length the header names.
        check(unsafe { ffi::lw_hip_comm_init(unique_id.as_ptr(), rank, nranks) })?;
        Ok(Self { rank, nranks })
    }

    /// One transform of 2^log2n_total elements (or `batch` of them) block-distributed over the ranks; `d_in_local` /
    /// `d_out_local` are DEVICE pointers to this rank's `batch * 2^log2n_total / nranks` elements.  Collective.
    ///
    /// # Safety
    /// Both pointers must be device allocations of that extent on this process's device; `hip_stream` a valid
    /// `hipStream_t` or null.
    #[allow(clippy::too_many_arguments)]
    pub unsafe fn ntt_sharded_device(&self, field: Field, layout: Layout, dir: Dir, d_in_local: *const c_void, d_out_local: *mut c_void,
                                     log2n_total: u32, batch: u32, natural_output: bool, hip_stream: *mut c_void) -> Result<(), HipError> {
        check(ffi::lw_hip_ntt_sharded_device(field, layout, dir, d_in_local, d_out_local, log2n_total, batch, natural_output as c_int, hip_stream))
    }

    /// `msm` over points sharded across the ranks; every rank receives the total.
    ///
    /// # Safety
    /// `d_scalars` / `d_points` must be device allocations of `n_local` scalars / points.
    pub unsafe fn msm_sharded_device<P: Copy>(&self, curve: Curve, d_scalars: *const u64, d_points: *const c_void, n_local: usize,
                                              hip_stream: *mut c_void) -> Result<P, HipError> {
        if size_of::<P>() != curve_point_bytes(curve) {
            return Err(HipError::BadArgument("point type has the wrong size".into()));
        }
        let mut out = core::mem::MaybeUninit::<P>::uninit();
        check(ffi::lw_hip_msm_sharded_device(curve, d_scalars, d_points, n_local, out.as_mut_ptr() as *mut c_void, hip_stream))?;
        Ok(out.assume_init())
    }
}

impl Drop for Comm {
    fn drop(&mut self) {
        // SAFETY: no arguments; releases the communicator if one exists.
        unsafe {
            ffi::lw_hip_comm_shutdown();
        }
    }
}
