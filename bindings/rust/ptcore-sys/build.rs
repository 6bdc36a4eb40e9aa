// Links libptcore.so (built by `make -C pathtracer_amd/csrc`).  PTCORE_LIB_DIR = directory holding the library.
fn main() {
    if let Ok(dir) = std::env::var("PTCORE_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=ptcore");
}
