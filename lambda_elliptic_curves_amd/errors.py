"""Error types mirroring the reference's enums for this path:
FFTError {RootOfUnityError, InputError, OrderError, <backend>Error} (math/src/fft/errors.rs:12-20),
MSMError::LengthMismatch (math/src/msm/naive.rs:7-9), CudaError (gpu/src/cuda/abstractions/errors.rs:3-21)."""
from . import _lib as L


class FFTError(Exception):
    pass


class RootOfUnityError(FFTError):
    """FFTError::RootOfUnityError / FieldError::RootOfUnityError"""


class InputError(FFTError):
    """FFTError::InputError — input length is not a power of two"""


class OrderError(FFTError):
    """FFTError::OrderError — order > 63"""


class HipError(FFTError):
    """FFTError::<backend>Error — device not found / allocation / launch failures"""


class CommError(HipError):
    """LW_ERR_COMM — RCCL unavailable, no communicator, or a failing collective (multi-GPU entry points)"""


class MSMError(Exception):
    pass


class LengthMismatch(MSMError):
    """MSMError::LengthMismatch"""


class FieldError(Exception):
    """FieldError::InvZeroError (zero coset offset)"""


_MAP = {
    L.ERR_INPUT_NOT_POW2: InputError,
    L.ERR_ORDER_TOO_LARGE: OrderError,
    L.ERR_ROOT_OF_UNITY: RootOfUnityError,
    L.ERR_LENGTH_MISMATCH: LengthMismatch,
    L.ERR_INV_ZERO: FieldError,
    L.ERR_COMM: CommError,
}


def check(rc):
    if rc == 0:
        return
    exc = _MAP.get(rc, HipError)
    raise exc(f"[{rc}] {L.last_error()}")
