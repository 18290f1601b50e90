// math/src/fft/gpu/hip/polynomial.rs — the HIP counterpart of math/src/fft/gpu/cuda/polynomial.rs:16-49.
// UNVERIFIED: written without a Rust toolchain (rust-shim/README.md).
//
// Contract: `evaluate_fft_hip` receives the ALREADY zero-padded power-of-two coefficient slice and returns
// natural-order evaluations; `interpolate_fft_hip` returns coefficients ALREADY scaled by N^-1, wrapped in
// `Polynomial::new` (which strips trailing zeros).  Raw `BaseType` bytes cross the boundary unchanged — Montgomery form
// both ways, as in math/src/gpu/cuda/field/element.rs:30-42.
use crate::{
    fft::errors::FFTError,
    field::{
        element::FieldElement,
        traits::{IsFFTField, IsField, IsSubFieldOf},
    },
    polynomial::Polynomial,
};
use alloc::vec::Vec;
use core::mem::size_of;
use lambdaworks_hip::{Dir, Field, HipError, Layout};

/// (lw_field_t, lw_layout_t) for the pair (domain field `F`, value field `E`), or `None` when the backend has no kernel
/// family for it — then the caller's CPU arm runs, exactly like the reference's `F::field_name()` test
/// (math/src/fft/polynomial.rs:43-62).  The kernel family comes from `IsFFTField::field_name()`
/// (math/src/field/traits.rs:75-79); the memory shape from the element sizes, which are also the layout assertion.
pub fn hip_tag<F, E>() -> Option<(Field, Layout)>
where
    F: IsFFTField + IsSubFieldOf<E>,
    E: IsField,
{
    let (fs, es) = (size_of::<FieldElement<F>>(), size_of::<FieldElement<E>>());
    let tag = match (F::field_name(), fs, es) {
        ("stark256", 32, 32) => (Field::Stark252, Layout::U64LimbsMsFirst),
        // hip-dispatch.patch gives FrField this name (the reference leaves it empty: bls12_381/default_types.rs:25-30)
        ("bls12_381_fr", 32, 32) => (Field::Bls12381Fr, Layout::U64LimbsMsFirst),
        ("babybear31", 4, 4) => (Field::BabyBear, Layout::BabyBearU32R32),   // babybear_u32.rs:6
        ("babybear31", 8, 8) => (Field::BabyBear, Layout::BabyBearU64R64),   // babybear.rs:19-20
        ("babybear31", 8, 32) => (Field::BabyBear, Layout::Ext4Interleaved), // quartic_babybear.rs:16-19
        _ => return None,
    };
    (lambdaworks_hip::field_elem_bytes(tag.0, tag.1) == es).then_some(tag)
}

/// `true` when `evaluate_fft_hip::<F, E>` / `interpolate_fft_hip::<F, E>` can run — ONE predicate for both directions.
pub fn hip_supported<F, E>() -> bool
where
    F: IsFFTField + IsSubFieldOf<E>,
    E: IsField,
{
    hip_tag::<F, E>().is_some() && lambdaworks_hip::device_count() > 0
}

fn to_fft_error(e: HipError, len: usize) -> FFTError {
    match e {
        HipError::InputNotPowerOfTwo(_) => FFTError::InputError(len),
        HipError::OrderTooLarge(_) => FFTError::OrderError(len.trailing_zeros() as u64),
        HipError::RootOfUnity(_) => FFTError::RootOfUnityError(len.trailing_zeros() as u64),
        other => FFTError::HipError(other),
    }
}

/// Evaluations of the (already padded) coefficient vector on the 2^k roots of unity, natural order.
pub fn evaluate_fft_hip<F, E>(coeffs: &[FieldElement<E>]) -> Result<Vec<FieldElement<E>>, FFTError>
where
    F: IsFFTField + IsSubFieldOf<E>,
    E: IsField,
{
    let (field, layout) = hip_tag::<F, E>().ok_or_else(|| {
        FFTError::HipError(HipError::BadArgument(alloc::format!("no HIP kernels for {}", core::any::type_name::<F>())))
    })?;
    // SAFETY: hip_tag matched field_name() and the element sizes, so FieldElement<E> is the bare Montgomery limb array
    // (math/src/field/element.rs:40-42): plain data, no Drop, every byte pattern the backend writes is a canonical residue.
    unsafe { lambdaworks_hip::ntt_unchecked::<FieldElement<E>, FieldElement<F>>(field, layout, Dir::Forward, coeffs, None) }
        .map_err(|e| to_fft_error(e, coeffs.len()))
}

/// The polynomial interpolating `(w^i, fft_evals[i])` — inverse of `evaluate_fft_hip`.
pub fn interpolate_fft_hip<F, E>(fft_evals: &[FieldElement<E>]) -> Result<Polynomial<FieldElement<E>>, FFTError>
where
    F: IsFFTField + IsSubFieldOf<E>,
    E: IsField,
{
    let (field, layout) = hip_tag::<F, E>().ok_or_else(|| {
        FFTError::HipError(HipError::BadArgument(alloc::format!("no HIP kernels for {}", core::any::type_name::<F>())))
    })?;
    // Dir::Inverse already multiplies by N^-1 (the reference does it on the CPU afterwards: cuda/polynomial.rs:46-48)
    // SAFETY: as in evaluate_fft_hip.
    let coeffs = unsafe { lambdaworks_hip::ntt_unchecked::<FieldElement<E>, FieldElement<F>>(field, layout, Dir::Inverse, fft_evals, None) }
        .map_err(|e| to_fft_error(e, fft_evals.len()))?;
    Ok(Polynomial::new(&coeffs))
}

/// `evaluate_offset_fft` without the sequential `poly.scale(offset)` pass (math/src/polynomial/mod.rs:259-271): the
/// coset scaling c_j * h^j is fused into the first NTT pass on the device.
pub fn evaluate_offset_fft_hip<F, E>(coeffs: &[FieldElement<E>], offset: &FieldElement<F>) -> Result<Vec<FieldElement<E>>, FFTError>
where
    F: IsFFTField + IsSubFieldOf<E>,
    E: IsField,
{
    let (field, layout) = hip_tag::<F, E>().ok_or_else(|| {
        FFTError::HipError(HipError::BadArgument(alloc::format!("no HIP kernels for {}", core::any::type_name::<F>())))
    })?;
    // SAFETY: as in evaluate_fft_hip.
    unsafe { lambdaworks_hip::ntt_unchecked(field, layout, Dir::Forward, coeffs, Some(offset)) }.map_err(|e| to_fft_error(e, coeffs.len()))
}

#[cfg(test)]
mod tests {
    // The reference's pattern for its GPU backend: GPU == CPU on the same input (math/src/fft/gpu/cuda/ops.rs:109-136).
    use super::*;
    use crate::fft::cpu::{ops::fft as fft_cpu, roots_of_unity::get_twiddles};
    use crate::field::{fields::fft_friendly::stark_252_prime_field::Stark252PrimeField, traits::RootsConfig};
    type F = Stark252PrimeField;
    type FE = FieldElement<F>;

    #[test]
    fn hip_matches_cpu_on_2_pow_12_and_round_trips() {
        let input: Vec<FE> = (0..1u64 << 12).map(|i| FE::from(i * i + 7)).collect();
        let twiddles = get_twiddles::<F>(12, RootsConfig::BitReverse).unwrap();
        let cpu = fft_cpu(&input, &twiddles).unwrap();
        let gpu = evaluate_fft_hip::<F, F>(&input).unwrap();
        assert_eq!(cpu, gpu);
        let back = interpolate_fft_hip::<F, F>(&gpu).unwrap();
        assert_eq!(back, Polynomial::new(&input));
    }

    #[test]
    fn all_ones_2_pow_20() {
        // math/src/fft/gpu/cuda/ops.rs:124-136
        let input = alloc::vec![FE::one(); 1 << 20];
        let twiddles = get_twiddles::<F>(20, RootsConfig::BitReverse).unwrap();
        assert_eq!(fft_cpu(&input, &twiddles).unwrap(), evaluate_fft_hip::<F, F>(&input).unwrap());
    }
}
