// UNVERIFIED: written without a Rust toolchain (see README.md).
// Links liblw_hip.so from $LW_HIP_LIB_DIR (default: ../lambda_elliptic_curves_amd/lib, this repository's build output).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("LW_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../lambda_elliptic_curves_amd/lib")
    });
    println!("cargo:rerun-if-env-changed=LW_HIP_LIB_DIR");
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=lw_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
