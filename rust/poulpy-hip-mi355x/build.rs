// Links libpoulpy_hip.so (built by `python -c "import __graft_entry__ as g; g.build()"` in the
// poulpy_amd repository).  POULPY_HIP_LIB_DIR points at the directory holding the .so.
fn main() {
    let dir = std::env::var("POULPY_HIP_LIB_DIR").expect("set POULPY_HIP_LIB_DIR to the directory of libpoulpy_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=poulpy_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
