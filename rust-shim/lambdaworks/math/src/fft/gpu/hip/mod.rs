// math/src/fft/gpu/hip/mod.rs — UNVERIFIED (no Rust toolchain in the authoring image), see rust-shim/README.md
pub mod polynomial;
