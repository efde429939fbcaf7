// Links tekken-rs_amd/libtekken_hip.so (built by `make -C tekken-rs_amd` or `python -c "import __graft_entry__ as g; g.build()"`).
// TEKKEN_HIP_LIB_DIR overrides the directory; the default is the sibling checkout this crate lives in.
fn main() {
    let dir = std::env::var("TEKKEN_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.parent().unwrap().join("tekken-rs_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=tekken_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=TEKKEN_HIP_LIB_DIR");
}
