"""Row N1: the replacement src/cuda/mod.rs + build.rs exist as files (integration/).  No Rust toolchain in the image,
so they cannot be compiled here; this test pins the FFI surface they use against include/sdmesh.h instead."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c(txt):
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    return re.sub(r"//[^\n]*", " ", txt)


def _c_prototypes():
    """name -> number of parameters, for every function declared in include/sdmesh.h."""
    txt = _strip_c(open(os.path.join(ROOT, "include/sdmesh.h")).read())
    protos = {}
    for m in re.finditer(r"\b(?:int|void|const char \*)\s*\b(sdmesh_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        args = " ".join(m.group(2).split())
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return protos


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out if a.strip()]


def _rust_calls(txt):
    """[(name, argc)] for every `sdmesh_xxx(...)` call expression in a Rust source text."""
    txt = re.sub(r"//[^\n]*", " ", txt)
    calls = []
    for m in re.finditer(r"\b(sdmesh_\w+)\s*\(", txt):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        calls.append((m.group(1), len(_split_args(txt[m.end(): i - 1]))))
    return calls


def test_integration_files_exist_and_say_they_are_uncompiled():
    for rel in ("integration/src/cuda/mod.rs", "integration/build.rs", "integration/src/bindings/mod.rs", "integration/README.md"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
    for rel in ("integration/src/cuda/mod.rs", "integration/build.rs", "integration/README.md"):
        assert "ncompiled" in open(os.path.join(ROOT, rel)).read(), rel


def test_every_sdmesh_call_matches_the_header():
    protos = _c_prototypes()
    assert len(protos) >= 25
    calls = _rust_calls(open(os.path.join(ROOT, "integration/src/cuda/mod.rs")).read())
    used = {n for n, _ in calls}
    # the reference's five methods + Drop
    assert {"sdmesh_create", "sdmesh_destroy", "sdmesh_create_voxel_field", "sdmesh_refine_voxel_field",
            "sdmesh_voxel_field_to_mesh", "sdmesh_download_mesh", "sdmesh_render", "sdmesh_last_error"} <= used
    for name, argc in calls:
        assert name in protos, f"{name} is not declared in include/sdmesh.h"
        assert argc == protos[name], f"{name}: called with {argc} arguments, header declares {protos[name]}"


def test_rust_imports_exist_in_the_headers():
    rs = open(os.path.join(ROOT, "integration/src/cuda/mod.rs")).read()
    m = re.search(r"use crate::bindings::cuda::\{(.*?)\};", rs, flags=re.S)
    names = [n.strip() for n in m.group(1).split(",") if n.strip()]
    hdr = _strip_c(open(os.path.join(ROOT, "include/sdmesh.h")).read()) + _strip_c(open(os.path.join(ROOT, "include/sdmesh/bindings.h")).read())
    for n in names:
        assert re.search(r"(\bstruct\s+%s\b|#define\s+%s\b|\b%s\s*\()" % (n, n, n), hdr), n


def test_public_surface_is_the_reference_handler_surface():
    """Names and receiver kinds of the reference's CudaHandler methods (src/cuda/mod.rs:49,105,124,204,348)."""
    rs = open(os.path.join(ROOT, "integration/src/cuda/mod.rs")).read()
    for sig in (r"pub fn new\(\) -> CudaHandler",
                r"pub fn create_cuda_voxel_field\(\) -> CudaVoxelField",
                r"pub fn refine_voxel_field\(&mut self, field: &mut CudaVoxelField\)",
                r"pub fn voxel_field_to_mesh\(&mut self, field: &CudaVoxelField\) -> obj::ObjData",
                r"pub fn render\(&mut self, target_image: &mut \[u8\], globals: GlobalsBuffer, camera: CameraBuffer\)",
                r"impl Drop for CudaHandler",
                r"pub struct CudaVoxelField \{\s*pub voxels: Vec<Point>,\s*pub voxel_size: Point,"):
        assert re.search(sig, rs), sig


def test_integration_md_snippets_use_declared_symbols_only():
    protos = _c_prototypes()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in set(re.findall(r"\b(sdmesh_[a-z_0-9]+)\b", md)):
        if name in ("sdmesh_handle", "sdmesh_regen_info", "sdmesh_pkg", "sdmesh_instr", "sdmesh_unique_id", "sdmesh_fast") or name.endswith("_"):
            continue                      # types, the Python loader, library names, and prefixes such as `sdmesh_comm_*`
        assert name in protos, f"INTEGRATION.md mentions {name}, which include/sdmesh.h does not declare"
