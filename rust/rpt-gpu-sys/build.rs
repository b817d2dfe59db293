// Links against librptgpu.so.  The directory is taken from RPTGPU_LIB_DIR, else from this repository's layout
// (rust/rpt-gpu-sys -> ../../rpt_amd/lib, where `make -C rpt_amd/csrc` puts the library).
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=RPTGPU_LIB_DIR");
    let dir = env::var("RPTGPU_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../rpt_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rptgpu");
    // so that `cargo run` finds the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
