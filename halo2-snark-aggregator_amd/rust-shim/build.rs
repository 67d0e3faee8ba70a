// Link against libh2agg.so (built by `python __graft_entry__.py build` in the h2agg repo).
fn main() {
    let dir = std::env::var("H2AGG_LIB_DIR").expect("set H2AGG_LIB_DIR to the directory holding libh2agg.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=h2agg");
}
