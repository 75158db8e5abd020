//! Replacement for the reference's `src/bindings/mod.rs`: the generated module keeps its path
//! (`crate::bindings::cuda`, used by src/renderer/mod.rs:254,264), the cudarc `DeviceRepr` impls
//! (reference src/bindings/mod.rs:5-12) go away because no struct is handed to a kernel launch from Rust any more.
pub mod cuda;
