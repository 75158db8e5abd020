"""Row N2: the Wavefront OBJ the step after the path writes (obj.save, src/renderer/mod.rs:204).  CPU only."""
import os

import numpy as np


def test_obj_roundtrip_and_layout(tmp_path, oracle):
    import sdmesh_pkg
    pkg = sdmesh_pkg.load()
    vox, size = oracle.create_voxel_field()
    pos, nrm, idx = oracle.voxel_field_to_mesh("sdobj", vox, size)
    obj = pkg.ObjData(pos, nrm, idx)
    path = os.path.join(tmp_path, "generated_mesh.obj")
    obj.save(path)
    lines = open(path).read().splitlines()
    assert sum(l.startswith("v ") for l in lines) == len(pos)
    assert sum(l.startswith("vn ") for l in lines) == len(nrm)
    assert lines.count("vt 0 0") == 1                                   # texture: vec![[0.0, 0.0]]  (mod.rs:305)
    faces = [l for l in lines if l.startswith("f ")]
    assert len(faces) == len(idx)
    a, b, c = (int(t) + 1 for t in idx[0])
    assert faces[0] == f"f {a}/1/{a} {b}/1/{b} {c}/1/{c}"                # IndexTuple(i, Some(0), Some(i)), 1-based
    back = pkg.ObjData.load(path)
    assert np.array_equal(back.indices, idx)
    assert np.array_equal(back.position.view(np.uint32), pos.view(np.uint32))
    assert np.array_equal(back.normal.view(np.uint32), nrm.view(np.uint32))


def test_empty_mesh_obj(tmp_path):
    import sdmesh_pkg
    pkg = sdmesh_pkg.load()
    z = np.zeros((0, 3), np.float32)
    obj = pkg.ObjData(z, z.copy(), np.zeros((0, 3), np.uint32))
    path = os.path.join(tmp_path, "empty.obj")
    obj.save(path)
    back = pkg.ObjData.load(path)
    assert len(back.position) == 0 and len(back.indices) == 0
