// math/src/msm/hip.rs — HIP arm of msm::pippenger::msm (math/src/msm/pippenger.rs:18-32).
// UNVERIFIED: written without a Rust toolchain (rust-shim/README.md).
//
// `msm` is generic over `G: IsGroup`; the reference has no backend seam for it.  The patch adds a
// `group_name()` hook to `IsGroup` (default ""), mirroring `IsFFTField::field_name()`, which
// `ShortWeierstrassProjectivePoint<E>` forwards to a new `IsEllipticCurve::curve_name()`; the four groups in scope name
// themselves.  Scalars must be 4-limb canonical integers (what every caller passes: `.representative()`,
// provers/groth16/src/prover.rs:69-78, crypto/src/commitments/kzg.rs:159-163).
use crate::{cyclic_group::IsGroup, msm::naive::MSMError, unsigned_integer::element::UnsignedInteger};
use core::mem::size_of;
use lambdaworks_hip::Curve;

pub fn hip_curve_tag<G: IsGroup>() -> Option<Curve> {
    let curve = match G::group_name() {
        "bls12_381_g1" => Curve::Bls12381G1,
        "bn254_g1" => Curve::Bn254G1,
        "bn254_g2" => Curve::Bn254G2,
        "bls12_381_g2" => Curve::Bls12381G2,
        _ => return None,
    };
    // layout assertion: X, Y, Z consecutive, nothing else (math/src/elliptic_curve/point.rs:8-10)
    (size_of::<G>() == lambdaworks_hip::curve_point_bytes(curve)).then_some(curve)
}

/// `Some(result)` when the HIP backend handled the call, `None` when it does not apply (other group, other scalar
/// width, no device) or failed on the device side — the caller then runs the CPU Pippenger, as the Metal arm of the FFT
/// falls back (math/src/fft/polynomial.rs:45-51).
pub fn msm_hip<const NUM_LIMBS: usize, G: IsGroup>(cs: &[UnsignedInteger<NUM_LIMBS>], points: &[G]) -> Option<Result<G, MSMError>> {
    if NUM_LIMBS != 4 || size_of::<UnsignedInteger<NUM_LIMBS>>() != 32 {
        return None;
    }
    let curve = hip_curve_tag::<G>()?;
    if cs.len() != points.len() {
        return Some(Err(MSMError::LengthMismatch(cs.len(), points.len())));
    }
    // SAFETY: UnsignedInteger<4> is `{ limbs: [u64; 4] }` with limbs[0] most significant (unsigned_integer/element.rs:29-37)
    // and its size was checked to be 32 bytes, so the slice can be viewed as [[u64; 4]].
    let scalars: &[[u64; 4]] = unsafe { core::slice::from_raw_parts(cs.as_ptr() as *const [u64; 4], cs.len()) };
    // G is Clone, not Copy: go through a byte-sized stand-in of the same size and move the result out.
    let pts: &[PointBytes<G>] = unsafe { core::slice::from_raw_parts(points.as_ptr() as *const PointBytes<G>, points.len()) };
    match lambdaworks_hip::msm::<PointBytes<G>>(curve, scalars, pts) {
        // SAFETY: PointBytes<G> has G's size and alignment and was fully written by the library with a valid point.
        Ok(p) => Some(Ok(unsafe { core::mem::transmute_copy::<PointBytes<G>, G>(&p) })),
        Err(e) => {
            #[cfg(feature = "std")]
            std::eprintln!("HIP msm failed ({e}); falling back to the CPU Pippenger");
            let _ = e;
            None
        }
    }
}

/// plain-data stand-in with the size and alignment of `G`
#[repr(transparent)]
struct PointBytes<G>(core::mem::MaybeUninit<G>);
impl<G> Clone for PointBytes<G> {
    fn clone(&self) -> Self {
        // SAFETY: MaybeUninit<G> is Copy-able bytewise.
        unsafe { core::ptr::read(self) }
    }
}
impl<G> Copy for PointBytes<G> {}

#[cfg(test)]
mod tests {
    // Pippenger == naive on the HIP arm, the reference's own property (math/src/msm/pippenger.rs:204-233)
    use super::*;
    use crate::elliptic_curve::short_weierstrass::curves::bls12_381::curve::BLS12381Curve;
    use crate::elliptic_curve::traits::IsEllipticCurve;
    use crate::msm::naive::msm as msm_naive;
    use crate::unsigned_integer::element::U256;

    #[test]
    fn hip_msm_equals_naive() {
        let g = BLS12381Curve::generator();
        let cs: alloc::vec::Vec<U256> = (1..200u64).map(|i| U256::from_u64(i * 0x9E37_79B9_7F4A_7C15)).collect();
        let points: alloc::vec::Vec<_> = (1..200u64).map(|i| g.operate_with_self(i)).collect();
        let hip = msm_hip(&cs, &points).expect("HIP arm applies").unwrap();
        assert_eq!(hip, msm_naive(&cs, &points).unwrap());
    }
}
