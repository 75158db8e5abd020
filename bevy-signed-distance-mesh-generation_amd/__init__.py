"""MI355X-native SDF isosurface mesher: hand-written gfx950 HIP kernels behind the reference's
mesh-generation boundary (bindings.h structs + the CudaHandler call set).  See DESIGN.md.

The directory name contains hyphens, so import it through `sdmesh_pkg.load()` at the repo root
(module name `bevy_signed_distance_mesh_generation_amd`)."""
from . import _abi
from ._abi import (BLOCK_SIZE, HSACO_PATH, LIB_PATH, RENDER_HSACO_PATH, MAX_LEVELS, MESH_GENERATION_BB_SIZE,  # noqa: F401
                   MESH_GENERATION_INIT_FACTOR, PART_NONE, PART_XSLAB, PART_ZSLAB, SCENES)
from .handler import CudaHandler, CudaVoxelField, ObjData, SdMeshError  # noqa: F401

__all__ = ["CudaHandler", "CudaVoxelField", "ObjData", "SdMeshError", "SCENES", "PART_NONE", "PART_ZSLAB", "PART_XSLAB",
           "LIB_PATH", "HSACO_PATH", "RENDER_HSACO_PATH"]
