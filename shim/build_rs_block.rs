// To be added to the reference's build.rs (next to `mod build_wfa`, build.rs:1-56) and called from `main()` behind the
// `hip` feature:  `#[cfg(feature = "hip")] build_hip::build();`
mod build_hip {
    use std::{env, path::PathBuf};

    /// Links `liblocityper_hip.so`. `LOCITYPER_HIP_DIR` = the directory that holds it (`<this repo>/locityper_amd` after
    /// `make -C locityper_amd/csrc`); the ROCm runtime libraries it needs are found through its own RUNPATH (`/opt/rocm/lib`).
    pub fn build() {
        let dir = PathBuf::from(env::var("LOCITYPER_HIP_DIR").expect("set LOCITYPER_HIP_DIR to the directory of liblocityper_hip.so"));
        println!("cargo:rustc-link-search=native={}", dir.display());
        println!("cargo:rustc-link-lib=dylib=locityper_hip");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
        println!("cargo:rerun-if-env-changed=LOCITYPER_HIP_DIR");
        println!("cargo:rerun-if-changed={}/liblocityper_hip.so", dir.display());
    }
}
