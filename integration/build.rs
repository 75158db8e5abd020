//! Replacement for the reference's `build.rs` (row N1).  STATUS: uncompiled (no Rust toolchain in the build image).
//!
//! The reference's build script (build.rs:85-169) does two things: nvcc -> PTX for the two kernel modules, and
//! bindgen on `cuda/includes/bindings.h` -> `src/bindings/cuda.rs`.  Here the kernels ship prebuilt inside
//! `libsdmesh.so` (built by `make -C bevy-signed-distance-mesh-generation_amd/csrc`, hipcc --offload-arch=gfx950),
//! so the script only (1) points the linker at the library and (2) runs bindgen on `include/sdmesh.h`, which
//! includes `sdmesh/bindings.h` -- the same structs and the same three constants as the reference header
//! (pinned by tests/test_reference_pins.py) plus the `sdmesh_*` functions.
//!
//! Environment: SDMESH_DIR = checkout of this repository (default "../sdmesh").

use std::path::PathBuf;

fn main() {
    let root = PathBuf::from(std::env::var("SDMESH_DIR").unwrap_or_else(|_| String::from("../sdmesh")));
    let lib_dir = root.join("bevy-signed-distance-mesh-generation_amd");
    let include = root.join("include");

    println!("cargo:rerun-if-changed={}", include.join("sdmesh.h").display());
    println!("cargo:rerun-if-changed={}", include.join("sdmesh/bindings.h").display());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=sdmesh");
    // libsdmesh.so itself depends on libamdhip64 (and librccl for the multi-GPU entry points)
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,/opt/rocm/lib");

    let bindings = bindgen::Builder::default()
        .header(include.join("sdmesh.h").to_str().unwrap())
        .clang_arg(format!("-I{}", include.display()))
        .parse_callbacks(Box::new(bindgen::CargoCallbacks::new()))
        .allowlist_function("sdmesh_.*")
        .allowlist_type("Point|VoxelField|Vertex|Triangle|IndexTriangle|Mesh|GlobalsBuffer|CameraBuffer|Rgba|RenderTexture|sdmesh_.*")
        .allowlist_var("BLOCK_SIZE|MESH_GENERATION_.*|SDMESH_.*")
        // The reference passes no_copy("*") / no_debug("*") and then adds Default to Point, Vertex and Triangle
        // only (build.rs:147-162), yet its own code copies Point and Vertex by value (src/cuda/mod.rs:165,291).
        // These are plain-old-data structs, so derive what the callers actually use.
        .derive_default(true)
        .derive_copy(true)
        .derive_debug(true)
        .generate()
        .expect("Unable to generate bindings");

    let mut text = String::from("#![allow(warnings)]\n\n");
    text.push_str(bindings.to_string().as_str());
    std::fs::write("src/bindings/cuda.rs", text.as_bytes()).expect("Failed to write bindings");
}
