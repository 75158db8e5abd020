"""ctypes declarations for libsdmesh.so (include/sdmesh.h).  No torch, no numpy-only fallbacks:
if the library is missing or fails to load this module raises -- there is no CPU path."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDMESH_LIB: development override (A/B timing of two builds of this same library on one box)
LIB_PATH = os.environ.get("SDMESH_LIB") or os.path.join(_HERE, "libsdmesh.so")
FAST_LIB_PATH = os.path.join(_HERE, "libsdmesh_fast.so")     # the opt-in fast numeric build (csrc/Makefile FASTFLAGS)
HSACO_PATH = os.path.join(_HERE, "compute_mesh_generation.hsaco")
RENDER_HSACO_PATH = os.path.join(_HERE, "compute_render.hsaco")

# numpy views of the boundary structs (include/sdmesh/bindings.h)
POINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
VERTEX = np.dtype([("position", POINT), ("normal", POINT)])
TRIANGLE = np.dtype([("vertices", VERTEX, (3,))])
INDEX_TRIANGLE = np.dtype([("indices", "<u4", (3,))])
assert (POINT.itemsize, VERTEX.itemsize, TRIANGLE.itemsize, INDEX_TRIANGLE.itemsize) == (12, 24, 72, 12)

BLOCK_SIZE = 128
MESH_GENERATION_INIT_FACTOR = 32
MESH_GENERATION_BB_SIZE = 5.0
MAX_LEVELS = 15

SCENES = {"sdobj": 0, "sphere": 1, "csg3": 2, "anim": 3, "bulb": 5}         # (4 is the runtime program: set_program)
OPS = {"sphere": 1, "box": 2, "box_skeleton": 3, "capsule": 4, "torus_y": 5, "ray": 6, "simple_box": 7, "simple_bounding_box": 8, "wrap": 9}
COMBINES = {"set": 0, "min": 1, "max": 2, "smooth_min": 3, "subtract": 4}
INSTR = np.dtype([("op", "<u4"), ("combine", "<u4"), ("k", "<f4"), ("a", "<f4", (7,))])
assert INSTR.itemsize == 40


def pack_program(program):
    """[(op, args, combine[, k]), ...] -> structured array in sdmesh_instr layout."""
    out = np.zeros(len(program), INSTR)
    for i, ins in enumerate(program):
        op, args, comb = ins[0], ins[1], ins[2]
        out[i]["op"] = OPS[op]
        out[i]["combine"] = COMBINES[comb]
        out[i]["k"] = np.float32(ins[3]) if len(ins) > 3 else np.float32(0)
        a = np.asarray(args, np.float32).ravel()
        out[i]["a"][: a.size] = a
    return out
PART_NONE, PART_ZSLAB, PART_XSLAB = 0, 1, 2

ERRORS = {1: "invalid argument", 2: "HIP runtime error", 3: "no gfx950 device", 4: "output buffer too small",
          5: "vertex outside the weld key range", 6: "internal device error", 7: "RCCL error",
          8: "another rank of the communicator failed in this collective call"}


class Point(ctypes.Structure):
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float), ("z", ctypes.c_float)]


class VoxelField(ctypes.Structure):
    _fields_ = [("voxel_size", Point), ("voxels", ctypes.c_void_p), ("voxel_count", ctypes.c_uint)]


class Mesh(ctypes.Structure):
    _fields_ = [("vertices", ctypes.c_void_p), ("vertex_count", ctypes.c_uint), ("triangles", ctypes.c_void_p)]


class GlobalsBuffer(ctypes.Structure):
    _fields_ = [("tick", ctypes.c_ulonglong), ("time", ctypes.c_float), ("render_texture_size", ctypes.c_uint * 2),
                ("render_screen_size", ctypes.c_float * 2)]


class CameraBuffer(ctypes.Structure):
    _fields_ = [("position", ctypes.c_float * 3), ("forward", ctypes.c_float * 3), ("up", ctypes.c_float * 3),
                ("right", ctypes.c_float * 3), ("fov", ctypes.c_float)]


class RegenInfo(ctypes.Structure):
    _fields_ = [("levels", ctypes.c_uint32), ("voxel_count", ctypes.c_uint32 * (MAX_LEVELS + 1)),
                ("unique_edges", ctypes.c_uint32), ("triangles", ctypes.c_uint32), ("vertices", ctypes.c_uint32),
                ("voxel_size", Point), ("status", ctypes.c_uint32 * 8), ("used_direct_path", ctypes.c_uint32),
                ("device_ms", ctypes.c_float)]


assert ctypes.sizeof(VoxelField) == 32 and ctypes.sizeof(Mesh) == 24 and ctypes.sizeof(Point) == 12
assert ctypes.sizeof(GlobalsBuffer) == 32 and ctypes.sizeof(CameraBuffer) == 52

# every symbol include/sdmesh.h declares: name -> (restype, argtypes)
_vp, _sz, _i, _u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
_psz = ctypes.POINTER(ctypes.c_size_t)
class Transport(ctypes.Structure):
    """sdmesh_transport (include/sdmesh.h): the caller's collectives, device pointers and the step's stream as plain addresses."""
    AG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
    AR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p)
    SR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p)
    GR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)
    _fields_ = [("user", ctypes.c_void_p), ("all_gather", AG), ("all_reduce_u32", AR), ("send", SR), ("recv", SR),
                ("group_start", GR), ("group_end", GR)]


SYMBOLS = {
    "sdmesh_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "sdmesh_destroy": (None, [_vp]),
    "sdmesh_last_error": (ctypes.c_char_p, [_vp]),
    "sdmesh_set_scene": (_i, [_vp, _i, _vp, _i]),
    "sdmesh_set_program": (_i, [_vp, _vp, _i]),
    "sdmesh_set_stream": (_i, [_vp, _vp]),
    "sdmesh_use_own_stream": (_i, [_vp]),
    "sdmesh_set_option": (_i, [_vp, ctypes.c_char_p, _i]),
    "sdmesh_create_voxel_field": (_i, [_vp, _vp]),
    "sdmesh_refine_voxel_field": (_i, [_vp, _vp, _sz, Point, _vp, _sz, _psz, ctypes.POINTER(Point)]),
    "sdmesh_refine_voxel_field_raw": (_i, [_vp, _vp, _sz, Point, _vp]),
    "sdmesh_voxel_field_to_triangles_raw": (_i, [_vp, _vp, _sz, Point, _vp]),
    "sdmesh_voxel_field_to_mesh": (_i, [_vp, _vp, _sz, Point, _vp, _sz, _psz, _vp, _sz, _psz]),
    "sdmesh_weld_triangles": (_i, [_vp, _vp, _sz, _vp, _sz, _psz, _vp, _sz, _psz]),
    "sdmesh_regen": (_i, [_vp, _i, ctypes.POINTER(RegenInfo)]),
    "sdmesh_regen_part": (_i, [_vp, _i, _i, _i, _i, ctypes.POINTER(RegenInfo)]),
    "sdmesh_regen_begin": (_i, [_vp, _i, _i, _i, _i]),
    "sdmesh_regen_end": (_i, [_vp, ctypes.POINTER(RegenInfo)]),
    "sdmesh_device_mesh": (_i, [_vp, ctypes.POINTER(Mesh), ctypes.POINTER(_u32)]),
    "sdmesh_device_voxels": (_i, [_vp, ctypes.POINTER(VoxelField)]),
    "sdmesh_device_triangles": (_i, [_vp, _vp, _sz, _psz]),
    "sdmesh_weld_device_triangles": (_i, [_vp, _vp, _sz, ctypes.POINTER(RegenInfo)]),
    "sdmesh_column_triangle_counts": (_i, [_vp, _vp]),
    "sdmesh_set_slab_bounds": (_i, [_vp, _i, _i, _vp]),
    "sdmesh_balance_slabs": (_i, [_vp, _i, _i, _i, _vp]),
    "sdmesh_merge_by_column": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _psz]),
    "sdmesh_comm_unique_id": (_i, [_vp]),
    "sdmesh_comm_init": (_i, [_vp, _vp, _i, _i]),
    "sdmesh_comm_destroy": (_i, [_vp]),
    "sdmesh_comm_ranks": (_i, [_vp, ctypes.POINTER(_i)]),
    "sdmesh_comm_init_transport": (_i, [_vp, _i, _i, _vp]),
    "sdmesh_comm_selftest": (_i, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "sdmesh_set_program_motion": (_i, [_vp, _vp, _i]),
    "sdmesh_set_program_time": (_i, [_vp, ctypes.c_float]),
    "sdmesh_sharded_regen": (_i, [_vp, _i, _i, ctypes.POINTER(RegenInfo)]),
    "sdmesh_sharded_route": (_i, [_vp]),
    "sdmesh_zslab_local": (_i, [_vp, _i, _i, _i, _vp, ctypes.POINTER(RegenInfo)]),
    "sdmesh_zslab_merge": (_i, [_vp, _i, _i, _vp, _vp, _vp, ctypes.POINTER(RegenInfo)]),
    "sdmesh_distributed_regen": (_i, [_vp, _i, ctypes.POINTER(RegenInfo), ctypes.POINTER(_u32)]),
    "sdmesh_distributed_route": (_i, [_vp]),
    "sdmesh_distributed_regen_rehearsal": (_i, [_vp, _i, _i, _vp, _vp]),
    "sdmesh_sharded_regen_rehearsal": (_i, [_vp, _i, _i, _i, ctypes.POINTER(RegenInfo)]),
    "sdmesh_download_mesh": (_i, [_vp, _vp, _vp]),
    "sdmesh_download_voxels": (_i, [_vp, _vp]),
    "sdmesh_regen_dense": (_i, [_vp, _i, ctypes.POINTER(RegenInfo)]),
    "sdmesh_sample_dense": (_i, [_vp, _u32, _vp, ctypes.POINTER(ctypes.c_float)]),
    "sdmesh_sample_dense_slab": (_i, [_vp, _u32, _u32, _u32, _vp, _i, ctypes.POINTER(ctypes.c_float)]),
    "sdmesh_time_kernel": (_i, [_vp, _i, _u32, _vp, _i, ctypes.POINTER(ctypes.c_float)]),
    "sdmesh_profile_stages": (_i, [_vp, _vp, _sz, _vp, _i, ctypes.POINTER(_i)]),
    "sdmesh_profile_rounds": (_i, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "sdmesh_selfcheck_arithmetic": (_i, [_vp, ctypes.POINTER(ctypes.c_uint64), _i]),
    "sdmesh_project_diag": (_i, [_vp, _vp, _sz, _psz]),
    "sdmesh_render": (_i, [_vp, _vp, _vp, _vp]),
    "sdmesh_render_device": (_i, [_vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_float)]),
}

_LIB = None
_FAST_LIB = None


def load(fast_math=False):
    """Load libsdmesh.so (or, with fast_math, libsdmesh_fast.so) and bind every declared symbol.  Raises if the
    library or a symbol is missing -- the product has no fallback path."""
    global _LIB, _FAST_LIB
    if fast_math:
        if _FAST_LIB is None:
            if not os.path.exists(FAST_LIB_PATH):
                raise RuntimeError(f"{FAST_LIB_PATH} is missing: build it with __graft_entry__.build()")
            load()                                        # (torch / HIP runtime first, as below)
            lib = ctypes.CDLL(FAST_LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _FAST_LIB = lib
        return _FAST_LIB
    if _LIB is None:
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 with the same
        # SONAME (libamdhip64.so.7) as the system ROCm one libsdmesh.so is linked against.  Importing
        # torch first makes the dynamic linker resolve our NEEDED entry to the copy torch already
        # loaded, so torch tensors, RCCL buffers and our kernels share one runtime and one context.
        # (Loading ours first would bring in a second runtime and torch.cuda would then fail to init.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               "(make -C bevy-signed-distance-mesh-generation_amd/csrc)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB
