fn main() {
    let dir = std::env::var("MGX_LIB_DIR").expect("set MGX_LIB_DIR to the directory holding libmgx.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=mgx");
    println!("cargo:rerun-if-env-changed=MGX_LIB_DIR");
}
