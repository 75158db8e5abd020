"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package must never do so (see sdmesh_oracle.c header).

The wrapper re-enacts the host half of the reference's CudaHandler
(src/cuda/mod.rs:105-122 create, :124-202 refine + retain, :204-346 mesh + weld) on top
of the C restatement of the kernels.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SCENES = {"sdobj": 0, "sphere": 1, "csg3": 2, "anim": 3, "program": 4, "bulb": 5}
OPS = {"sphere": 1, "box": 2, "box_skeleton": 3, "capsule": 4, "torus_y": 5, "ray": 6, "simple_box": 7, "simple_bounding_box": 8, "wrap": 9}
COMBINES = {"set": 0, "min": 1, "max": 2, "smooth_min": 3, "subtract": 4}
INSTR = np.dtype([("op", "<u4"), ("combine", "<u4"), ("k", "<f4"), ("a", "<f4", (7,))])

POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
VERTEX = np.dtype([("position", POINT), ("normal", POINT)])
TRIANGLE = np.dtype([("vertices", VERTEX, (3,))])
assert POINT.itemsize == 12 and VERTEX.itemsize == 24 and TRIANGLE.itemsize == 72


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "sdmesh_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        vp, u32, u64, sz, fp = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_float
        L.orc_sdf.restype = fp
        L.orc_sdf.argtypes = [ctypes.c_int, vp, fp, fp, fp]
        L.orc_eval_count.restype = u64
        L.orc_eval_count.argtypes = [ctypes.c_int]
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_create_voxel_field.argtypes = [vp, vp]
        L.orc_refine_raw.argtypes = [ctypes.c_int, vp, vp, u32, _PointC, vp]
        L.orc_retain_finite.restype = sz
        L.orc_retain_finite.argtypes = [vp, sz]
        L.orc_triangles_raw.argtypes = [ctypes.c_int, vp, vp, u32, _PointC, vp, vp]
        L.orc_weld.restype = sz
        L.orc_weld.argtypes = [vp, sz, vp, vp, vp, ctypes.POINTER(sz)]
        L.orc_fnv1a64.restype = u64
        L.orc_fnv1a64.argtypes = [vp, sz]
        L.orc_eval_points.argtypes = [ctypes.c_int, vp, vp, ctypes.c_long, vp, vp, vp, vp]
        L.orc_weld_quant.restype = ctypes.c_longlong
        L.orc_weld_quant.argtypes = [fp]
        L.orc_set_program.argtypes = [vp, ctypes.c_int]
        L.orc_dense_mc_count.restype = u64
        L.orc_dense_mc_count.argtypes = [ctypes.c_int, vp, u32]
        _LIB = L
    return _LIB


class _PointC(ctypes.Structure):
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float), ("z", ctypes.c_float)]


def _params(params):
    p = np.zeros(4, dtype=np.float32)
    if params is not None:
        a = np.asarray(params, dtype=np.float32).ravel()
        p[: a.size] = a
    return p


def _pt(a):
    return _PointC(float(a[0]), float(a[1]), float(a[2]))


def set_program(program):
    """[(op, args, combine[, k]), ...] -> the program evaluated by scene "program" (global in the oracle lib)."""
    arr = np.zeros(len(program), INSTR)
    for i, ins in enumerate(program):
        arr[i]["op"] = OPS[ins[0]]
        arr[i]["combine"] = COMBINES[ins[2]]
        arr[i]["k"] = np.float32(ins[3]) if len(ins) > 3 else np.float32(0)
        a = np.asarray(ins[1], np.float32).ravel()
        arr[i]["a"][: a.size] = a
    lib().orc_set_program(arr.ctypes.data, len(arr))
    return arr


def program_at_time(program, motion, t):
    """The program the kernels see at time t: args + t * rates in f32, the product rounded before the sum (sdmesh_set_program_time)."""
    t = np.float32(t)
    out = []
    for ins, rates in zip(program, motion):
        a = np.asarray(ins[1], np.float32)
        r = np.zeros(7, np.float32)
        rr = np.asarray(rates, np.float32).ravel()
        r[: rr.size] = rr
        full = np.zeros(7, np.float32)
        full[: a.size] = a
        eff = (full + (t * r).astype(np.float32)).astype(np.float32)
        out.append((ins[0], eff.tolist()) + tuple(ins[2:]))
    return out


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return lib().orc_max_threads()


def eval_count(reset=False):
    return int(lib().orc_eval_count(1 if reset else 0))


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().orc_fnv1a64(a.ctypes.data, a.nbytes))


def sdf(scene, pts, params=None):
    p = _params(params)
    pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
    L = lib()
    return np.array([L.orc_sdf(SCENES[scene], p.ctypes.data, *map(float, q)) for q in pts], dtype=np.float32)


def eval_points(scene, pts, params=None, normals=True, project=True, flips=True):
    """-> dict(sd [n], nrm [n,3], proj [n,3], flip [n//3]) at arbitrary points (numerics tests)."""
    p = _params(params)
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
    n = pts.shape[0]
    sd = np.empty(n, np.float32)
    nrm = np.empty((n, 3), np.float32) if normals else None
    proj = np.empty((n, 3), np.float32) if project else None
    flip = np.empty(n // 3, np.int32) if flips else None
    lib().orc_eval_points(SCENES[scene], p.ctypes.data, pts.ctypes.data, n, sd.ctypes.data,
                          nrm.ctypes.data if normals else None, proj.ctypes.data if project else None,
                          flip.ctypes.data if flips else None)
    return {"sd": sd, "nrm": nrm, "proj": proj, "flip": flip}


def create_voxel_field():
    """-> (voxels float32 [32768,3], voxel_size float32 [3])   src/cuda/mod.rs:105-122"""
    vox = np.empty((32 ** 3, 3), dtype=np.float32)
    size = np.empty(3, dtype=np.float32)
    lib().orc_create_voxel_field(vox.ctypes.data, size.ctypes.data)
    return vox, size


def refine_raw(scene, voxels, voxel_size, params=None):
    """Kernel only: raw 8n float32 [8n,3] layout, INF for non-border children."""
    p = _params(params)
    vox = np.ascontiguousarray(voxels, dtype=np.float32).reshape(-1, 3)
    out = np.empty((vox.shape[0] * 8, 3), dtype=np.float32)
    lib().orc_refine_raw(SCENES[scene], p.ctypes.data, vox.ctypes.data, vox.shape[0], _pt(voxel_size), out.ctypes.data)
    return out


def refine(scene, voxels, voxel_size, params=None):
    """CudaHandler::refine_voxel_field (src/cuda/mod.rs:124-202): -> (voxels', voxel_size/2)."""
    vox = np.ascontiguousarray(voxels, dtype=np.float32).reshape(-1, 3)
    size = np.asarray(voxel_size, dtype=np.float32)
    if vox.shape[0] == 0:                       # mod.rs:137 -- nothing changes, not even the size
        return vox, size
    raw = refine_raw(scene, vox, size, params)
    n = lib().orc_retain_finite(raw.ctypes.data, raw.shape[0])
    return raw[:n].copy(), (size / np.float32(2.0)).astype(np.float32)


def triangles_raw(scene, voxels, voxel_size, params=None, iter_hist=None):
    """Kernel only: raw 5n Triangle slots (float32 [5n,3,2,3]), NaN/zero padded."""
    p = _params(params)
    vox = np.ascontiguousarray(voxels, dtype=np.float32).reshape(-1, 3)
    out = np.empty((vox.shape[0] * 5, 3, 2, 3), dtype=np.float32)
    hp = iter_hist.ctypes.data if iter_hist is not None else None
    lib().orc_triangles_raw(SCENES[scene], p.ctypes.data, vox.ctypes.data, vox.shape[0], _pt(voxel_size),
                            out.ctypes.data, hp)
    return out


def weld(raw_triangles):
    """Host weld (src/cuda/mod.rs:263-296) -> (pos [V,3] f32, nrm [V,3] f32, idx [T,3] u32)."""
    raw = np.ascontiguousarray(raw_triangles, dtype=np.float32).reshape(-1, 3, 2, 3)
    n = raw.shape[0]
    pos = np.empty((max(n, 1) * 3, 3), dtype=np.float32)
    nrm = np.empty((max(n, 1) * 3, 3), dtype=np.float32)
    idx = np.empty((max(n, 1), 3), dtype=np.uint32)
    nt = ctypes.c_size_t(0)
    nv = lib().orc_weld(raw.ctypes.data, n, pos.ctypes.data, nrm.ctypes.data, idx.ctypes.data, ctypes.byref(nt))
    return pos[:nv].copy(), nrm[:nv].copy(), idx[: nt.value].copy()


def voxel_field_to_mesh(scene, voxels, voxel_size, params=None):
    """CudaHandler::voxel_field_to_mesh (src/cuda/mod.rs:204-346) -> (pos, nrm, idx)."""
    vox = np.ascontiguousarray(voxels, dtype=np.float32).reshape(-1, 3)
    if vox.shape[0] == 0:
        z = np.zeros((0, 3), dtype=np.float32)
        return z, z.copy(), np.zeros((0, 3), dtype=np.uint32)
    return weld(triangles_raw(scene, vox, voxel_size, params))


def regen(scene, levels, params=None):
    """Level-0 field -> `levels` refines -> mesh; the reference's user-visible flow
    (src/renderer/mod.rs:162-225).  Returns dict with voxels, voxel_size, pos, nrm, idx."""
    vox, size = create_voxel_field()
    for _ in range(levels):
        vox, size = refine(scene, vox, size, params)
    pos, nrm, idx = voxel_field_to_mesh(scene, vox, size, params)
    return {"voxels": vox, "voxel_size": size, "pos": pos, "nrm": nrm, "idx": idx}


def dense_mc_count(scene, N, params=None):
    p = _params(params)
    return int(lib().orc_dense_mc_count(SCENES[scene], p.ctypes.data, int(N)))


def render(scene, width, height, position, forward, up, right, fov, screen=None, params=None):
    """Row N4: the ray-march preview (compute_render.cu) as an RGBA8 image [height, width, 4]."""
    import ctypes as C
    p = _params(params)
    L = lib()
    screen = screen or (float(width), float(height))
    g = np.zeros(32, np.uint8)
    g[12:20] = np.array([width, height], np.uint32).view(np.uint8)
    g[20:28] = np.array(screen, np.float32).view(np.uint8)
    cam = np.array(list(position) + list(forward) + list(up) + list(right) + [fov], np.float32)
    L.orc_fov_fac.restype = C.c_float
    L.orc_fov_fac.argtypes = [C.c_float]
    fac = L.orc_fov_fac(C.c_float(fov))
    out = np.zeros((height, width, 4), np.uint8)
    L.orc_render.restype = None
    L.orc_render.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]
    L.orc_render(SCENES[scene], p.ctypes.data, g.ctypes.data, cam.ctypes.data, C.c_float(fac), width, height, out.ctypes.data)
    return out
