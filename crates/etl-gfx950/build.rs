// Links libetl_gfx950.so (built by `python -m etl_amd.build` in the etl-gfx950 repository).
fn main() {
    if let Ok(dir) = std::env::var("ETLG_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=etl_gfx950");
    println!("cargo:rerun-if-env-changed=ETLG_LIB_DIR");
}
