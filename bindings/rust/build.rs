// Link against libvuke_hip.so (built by vuke_amd/csrc/build.sh).  VUKE_HIP_LIB_DIR = the directory that holds it.
fn main() {
    println!("cargo:rerun-if-env-changed=VUKE_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("VUKE_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=vuke_hip");
}
