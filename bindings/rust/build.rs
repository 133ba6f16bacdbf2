// PLONK_GADGETS_HIP_DIR = directory holding libplonk_gadgets_hip.so (plonk_gadgets_amd/ of this repository)
fn main() {
    let dir = std::env::var("PLONK_GADGETS_HIP_DIR").expect("set PLONK_GADGETS_HIP_DIR");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=plonk_gadgets_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
