// Link against the in-tree libzk_amd.so (set ZK_AMD_LIB_DIR to <repo>/zk_amd).
fn main() {
    let dir = std::env::var("ZK_AMD_LIB_DIR").expect("set ZK_AMD_LIB_DIR to the directory holding libzk_amd.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zk_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
