"""Host-side sanitizer runs (the reference has none; the GPU pool offers no device-side sanitizer, so the CPU side is
what can be checked): the oracle's C restatement through a whole small regen under AddressSanitizer + UndefinedBehaviorSanitizer,
its result equal to the optimised library's; and the host-only part of the C++ mirror (level-0 field through the C ABI, ObjData
assembly and save, the panic without a device) under the same sanitizers."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2")


@pytest.mark.parametrize("scene_id,scene,levels", [(0, "sdobj", 1), (1, "sphere", 1), (3, "anim", 0)])
def test_oracle_regen_is_clean_under_asan_and_ubsan(scene_id, scene, levels):
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    p = subprocess.run([os.path.join(ROOT, "oracle", "_san", "oracle_san"), str(scene_id), str(levels)], capture_output=True, text=True, env=ENV,
                       timeout=300)
    assert p.returncode == 0 and "ERROR" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    n, nv, nt, fnv, _ = p.stdout.split()
    from oracle import oracle as O
    ref = O.regen(scene, levels, [0.25] if scene == "anim" else None)
    assert (int(nv), int(nt)) == (len(ref["pos"]), len(ref["idx"]))
    assert fnv == "%016x" % O.fnv1a64(np.ascontiguousarray(ref["idx"], dtype=np.uint32))


CPP = r'''
#include <cstdio>
#include "sdmesh/cuda_handler.hpp"
int main(int, char **argv) {
    sdmesh::CudaVoxelField f = sdmesh::CudaHandler::create_cuda_voxel_field();          // host-only entry of the C ABI
    if (f.voxels.size() != 32768 || f.voxels.front().x != -2.5f || f.voxels.back().z != 2.5f - f.voxel_size.z) return 3;
    sdmesh::ObjData obj;
    for (int i = 0; i < 4; i++) { obj.position.push_back({(float) i, 0.5f, -1.25f}); obj.normal.push_back({0.0f, 0.0f, 1.0f}); }
    obj.polys.push_back({sdmesh::IndexTuple{0, 0, 0}, sdmesh::IndexTuple{1, 0, 1}, sdmesh::IndexTuple{2, 0, 2}});
    obj.polys.push_back({sdmesh::IndexTuple{2, 0, 2}, sdmesh::IndexTuple{1, 0, 1}, sdmesh::IndexTuple{3, 0, 3}});
    obj.save(argv[1]);
    try { sdmesh::CudaHandler h; std::printf("device\n"); }
    catch (const sdmesh::Panic &p) { std::printf("panic %d\n", p.code); }
    return 0;
}
'''


def test_cpp_mirror_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    import sdmesh_pkg
    pkg = sdmesh_pkg.load()
    src = tmp_path / "host_side.cpp"
    src.write_text(CPP)
    exe = str(tmp_path / "host_side")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall", "-Wextra", "-Werror",
           "-I", os.path.join(ROOT, "include"), str(src), "-L", PKG, "-lsdmesh", "-Wl,-rpath," + PKG, "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = str(tmp_path / "quad.obj")
    # (the HIP runtime the library links is not instrumented and keeps process-lifetime allocations: leak checking off for this one)
    p = subprocess.run([exe, out], capture_output=True, text=True, env=dict(ENV, ASAN_OPTIONS="detect_leaks=0"), timeout=300)
    assert p.returncode == 0 and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    assert p.stdout.strip() in ("panic 3", "device")
    obj = pkg.ObjData.load(out)
    assert obj.indices.tolist() == [[0, 1, 2], [2, 1, 3]] and obj.position[3].tolist() == [3.0, 0.5, -1.25] and len(obj.normal) == 4
