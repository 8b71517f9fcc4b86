// UNCOMPILED SOURCE (no Rust toolchain here).  Links the shim against libmi355rt.so.
fn main() {
    if let Ok(dir) = std::env::var("MI355RT_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=mi355rt");
}
