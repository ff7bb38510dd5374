// Builds librt355.so with hipcc for gfx950 (or uses RT355_LIB_DIR if it is set)
// and tells cargo to link it.  Needs ROCm >= 7 on the build machine.
use std::{env, path::PathBuf, process::Command};

fn main() {
    if let Ok(dir) = env::var("RT355_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=rt355");
        return;
    }
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../..");
    let csrc = root.join("raytracer_amd/csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let status = Command::new(hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"])
        // part of the numerical contract: multiply-adds round twice
        .args(["-ffp-contract=off", "-fno-fast-math", "-pthread"])
        .arg("-I").arg(root.join("include"))
        .arg("-I").arg(&csrc)
        .args(["-x", "hip"])
        .arg(csrc.join("rt_kernels.hip"))
        .arg(csrc.join("rt_build_gpu.hip"))
        .arg(csrc.join("rt_api.cpp"))
        .arg(csrc.join("rt_bvh.cpp"))
        .arg("-o").arg(out.join("librt355.so"))
        .status()
        .expect("hipcc not found");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt355");
    for f in ["rt_kernels.hip", "rt_build_gpu.hip", "rt_api.cpp", "rt_bvh.cpp", "rt_device.hpp", "rt_kernels.hpp", "rt_bvh.hpp"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rerun-if-changed={}", root.join("include/rt355.h").display());
}
