//! `HipError`: the counterpart of `CudaError` (gpu/src/cuda/abstractions/errors.rs:3-21) for the HIP backend.
//! UNVERIFIED: written without a Rust toolchain.
use crate::ffi;
use core::ffi::{c_int, CStr};
use core::fmt;

#[derive(Debug, Clone, PartialEq, Eq)]
pub enum HipError {
    /// LW_ERR_INPUT_NOT_POW2 — FFTError::InputError
    InputNotPowerOfTwo(String),
    /// LW_ERR_ORDER_TOO_LARGE — FFTError::OrderError
    OrderTooLarge(String),
    /// LW_ERR_ROOT_OF_UNITY — FFTError::RootOfUnityError
    RootOfUnity(String),
    /// LW_ERR_LENGTH_MISMATCH — MSMError::LengthMismatch
    LengthMismatch(String),
    /// LW_ERR_NO_DEVICE — CudaError::DeviceNotFound
    DeviceNotFound(String),
    /// LW_ERR_ALLOC — CudaError::AllocateMemory
    AllocateMemory(String),
    /// LW_ERR_LAUNCH — CudaError::Launch / FunctionError
    Launch(String),
    /// LW_ERR_COMM — RCCL unavailable, no communicator, or a failing collective
    Comm(String),
    /// LW_ERR_BAD_ARG
    BadArgument(String),
    /// LW_ERR_INV_ZERO — FieldError::InvZeroError (zero coset offset)
    InvZero(String),
    /// a status this binding does not know
    Unknown(i32, String),
}

impl fmt::Display for HipError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        let (what, msg) = match self {
            HipError::InputNotPowerOfTwo(m) => ("input length is not a power of two", m),
            HipError::OrderTooLarge(m) => ("order is greater than 63", m),
            HipError::RootOfUnity(m) => ("no root of unity of that order", m),
            HipError::LengthMismatch(m) => ("scalars and points differ in length", m),
            HipError::DeviceNotFound(m) => ("no usable gfx950 device", m),
            HipError::AllocateMemory(m) => ("device allocation failed", m),
            HipError::Launch(m) => ("kernel launch or copy failed", m),
            HipError::Comm(m) => ("multi-GPU communication failed", m),
            HipError::BadArgument(m) => ("bad argument", m),
            HipError::InvZero(m) => ("inverse of zero", m),
            HipError::Unknown(_, m) => ("unknown status", m),
        };
        write!(f, "HIP backend: {what}: {msg}")
    }
}

impl std::error::Error for HipError {}

/// thread-local message of the last failing call
pub fn last_error_message() -> String {
    // SAFETY: lw_hip_last_error returns a NUL-terminated string owned by the library, valid until this thread's next call.
    unsafe {
        let p = ffi::lw_hip_last_error();
        if p.is_null() {
            String::new()
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

/// status code -> Result
pub fn check(rc: c_int) -> Result<(), HipError> {
    if rc == ffi::LW_OK {
        return Ok(());
    }
    let m = last_error_message();
    Err(match rc {
        ffi::LW_ERR_INPUT_NOT_POW2 => HipError::InputNotPowerOfTwo(m),
        ffi::LW_ERR_ORDER_TOO_LARGE => HipError::OrderTooLarge(m),
        ffi::LW_ERR_ROOT_OF_UNITY => HipError::RootOfUnity(m),
        ffi::LW_ERR_LENGTH_MISMATCH => HipError::LengthMismatch(m),
        ffi::LW_ERR_NO_DEVICE => HipError::DeviceNotFound(m),
        ffi::LW_ERR_ALLOC => HipError::AllocateMemory(m),
        ffi::LW_ERR_LAUNCH => HipError::Launch(m),
        ffi::LW_ERR_COMM => HipError::Comm(m),
        ffi::LW_ERR_BAD_ARG => HipError::BadArgument(m),
        ffi::LW_ERR_INV_ZERO => HipError::InvZero(m),
        other => HipError::Unknown(other, m),
    })
}
