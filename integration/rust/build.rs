// build.rs — compile the HIP library with hipcc and link it (see raytracer_amd/csrc/Makefile for the
// same recipe).  rt_f32.hip is the production translation unit (contraction on), rt_f64.hip the audit one.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let src = PathBuf::from(env::var("RT_HIP_SRC").unwrap_or_else(|_| "raytracer_amd/csrc".into()));
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"];
    for (file, contract) in [("rt_f32.hip", Some("on")), ("rt_f64.hip", Some("off")), ("rt_api.hip", None)] {
        let mut cmd = Command::new(&hipcc);
        cmd.args(common);
        if let Some(c) = contract {
            cmd.arg(format!("-ffp-contract={c}"));
        }
        let ok = cmd.arg("-c").arg(src.join(file)).arg("-o").arg(out.join(file).with_extension("o"))
            .status().expect("hipcc not found").success();
        assert!(ok, "hipcc failed on {file}");
    }
    let ok = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-shared", "-fPIC", "-o"])
        .arg(out.join("librt_hip.so"))
        .args(["rt_api.o", "rt_f32.o", "rt_f64.o"].map(|o| out.join(o)))
        .status().unwrap().success();
    assert!(ok, "linking librt_hip.so failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    println!("cargo:rerun-if-changed={}", src.display());
}
