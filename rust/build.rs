// Links the C-ABI library built by `python -m sparsemat_amd.build` (hipcc, gfx950).
// SPARSEMAT_HIP_LIB_DIR = directory holding libsparsemat_hip.so.
fn main() {
    let dir = std::env::var("SPARSEMAT_HIP_LIB_DIR").unwrap_or_else(|_| "../sparsemat_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=sparsemat_hip");
    println!("cargo:rerun-if-env-changed=SPARSEMAT_HIP_LIB_DIR");
}
